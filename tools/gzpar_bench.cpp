// Throughput of the parallel gzip decoder (c3poa_amd/csrc/c3_gzpar.hpp) on a .gz file, against the single-thread decoder of c3_inflate.hpp and zlib:
//   g++ -O3 -std=c++17 tools/gzpar_bench.cpp -o /tmp/gzpar_bench -lz -lpthread && /tmp/gzpar_bench file.gz 1 2 4 8 16
#include "../c3poa_amd/csrc/c3_gzpar.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now_() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
  fseek(f, 0, SEEK_END); const size_t n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> d(n); if (fread(d.data(), 1, n, f) != n) return 1; fclose(f);
  size_t total = 0;
  { // zlib
    z_stream z; memset(&z, 0, sizeof(z)); inflateInit2(&z, 31); std::vector<uint8_t> o((size_t)4 << 20);
    z.next_in = d.data(); z.avail_in = (unsigned)std::min<size_t>(n, 0x7fffffff); const double t = now_(); int rc;
    do { z.next_out = o.data(); z.avail_out = (unsigned)o.size(); rc = inflate(&z, Z_NO_FLUSH); total += o.size() - z.avail_out; } while (rc == Z_OK);
    printf("zlib (first member): %zu bytes, %.1f MB/s\n", total, total / (now_() - t) / 1e6); inflateEnd(&z); }
  { // own serial decoder, chunked as the reader drives it
    static c3inf::Inflater inf; const uint8_t* p = c3inf::gz_member_data(d.data(), n); inf.reset(p, d.data() + n);
    const size_t W = 32768, C = (size_t)4 << 20; std::vector<uint8_t> win(W + C + 1024); size_t hist = 0, tot = 0; const double t = now_();
    for (;;) { size_t pos = 0; const int rc = inf.run(win.data() + W, &pos, C, C + 512, hist); if (rc < 0) { printf("serial: error\n"); break; } tot += pos;
      if (pos >= W) { memcpy(win.data(), win.data() + W + pos - W, W); hist = W; } else { memmove(win.data(), win.data() + pos, W); hist = std::min(W, hist + pos); } if (rc == 1) break; }
    printf("own decoder, one thread (first member): %zu bytes, %.1f MB/s\n", tot, tot / (now_() - t) / 1e6); }
  for (int a = 2; a < argc; ++a) {
    c3inf::GzPar par; par.map = d.data(); par.size = n; par.T = atoi(argv[a]); par.chunk = getenv("CHUNK") ? atol(getenv("CHUNK")) : (size_t)1 << 20; par.per_round = getenv("ROUND") ? atoi(getenv("ROUND")) : 0;
    const double t = now_(); size_t tot = 0;
    if (!par.open()) return 2;
    for (;;) { const bool ok = par.next_round(); if (par.bad) { printf("BAD\n"); return 3; } if (ok) for (auto& c : par.chunks) if (c.start != (size_t)-1) tot += c.cb.len; if (!ok || par.done) break; }
    printf("parallel, %2d threads, chunk %zu, %d chunks per round: %zu bytes, %.1f MB/s\n", par.T, par.chunk, par.per_round ? par.per_round : par.T, tot, tot / (now_() - t) / 1e6);
  }
}
