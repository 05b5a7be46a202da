// Sanitizer harness of the parallel gzip decoder (c3poa_amd/csrc/c3_gzpar.hpp; CPU only): intact, bit-flipped, truncated and random gzip images
// through exact-size heap buffers, chunks from a few hundred bytes (every seam inside a block or two) to larger than the file, 1-6 threads.
//   cd tools && g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 gzpar_asan.cpp -o /tmp/gzpar_asan -lz -lpthread && /tmp/gzpar_asan 1 3000
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <zlib.h>
#include "../c3poa_amd/csrc/c3_gzpar.hpp"
static std::vector<unsigned char> gz_of(const std::vector<unsigned char>& data, int level, std::mt19937_64& rng) {
  z_stream z; memset(&z, 0, sizeof(z));
  deflateInit2(&z, level, Z_DEFLATED, 31, 1 + rng() % 9, rng() % 8 == 0 ? Z_FIXED : rng() % 8 == 1 ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY);
  size_t at = 0; const size_t step = rng() % 3 == 0 ? 200 + rng() % 20000 : data.size() + 1;
  std::vector<unsigned char> out(deflateBound(&z, data.size()) + 4096 + (data.size() / step + 2) * 64 + data.size() / 4);
  z.next_out = out.data(); z.avail_out = out.size();
  while (at < data.size()) {
    const size_t k = std::min(step, data.size() - at);
    z.next_in = const_cast<unsigned char*>(data.data()) + at; z.avail_in = k; at += k;
    deflate(&z, at < data.size() ? (rng() & 1 ? Z_SYNC_FLUSH : Z_FULL_FLUSH) : Z_NO_FLUSH);
  }
  z.next_in = nullptr; z.avail_in = 0;
  if (deflate(&z, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "harness: deflate buffer too small\n"); abort(); }
  out.resize(out.size() - z.avail_out);
  deflateEnd(&z);
  return out;
}
int main(int argc, char** argv) {
  std::mt19937_64 rng(argc > 1 ? atoll(argv[1]) : 1);
  const int N = argc > 2 ? atoi(argv[2]) : 3000;
  long errs = 0, oks = 0;
  for (int it = 0; it < N; ++it) {
    std::vector<unsigned char> data, img; std::vector<size_t> bounds;
    const int members = 1 + rng() % 3;
    for (int m = 0; m < members; ++m) {
      size_t n = rng() % 5 == 0 ? 0 : rng() % 200000;
      std::vector<unsigned char> part(n);
      const int kind = rng() % 5;
      for (size_t i = 0; i < n; ++i) part[i] = kind == 0 ? rng() : kind == 1 ? "ACGT"[rng() % 4] : kind == 2 ? (unsigned char)(i / 100) : (unsigned char)('A' + (rng() % 3 == 0));
      if (kind == 4) {          // FASTQ: the headers are copied from one another across the whole chunk (marks that never die out), repeats inside a read
        size_t i = 0; int rec = 0;
        while (i < n) {
          char hd[64]; const int hl = snprintf(hd, sizeof hd, "@read_%08d ch=%d\n", rec++, (int)(rng() % 512));
          const size_t L = 50 + rng() % 3000, unit = 20 + rng() % 400;
          std::vector<unsigned char> u(unit); for (auto& b : u) b = "ACGT"[rng() % 4];
          for (int k = 0; k < hl && i < n; ++k) part[i++] = hd[k];
          for (size_t k = 0; k < L && i < n; ++k) part[i++] = rng() % 8 == 0 ? "ACGT"[rng() % 4] : u[k % unit];
          if (i < n) part[i++] = '\n'; if (i < n) part[i++] = '+'; if (i < n) part[i++] = '\n';
          for (size_t k = 0; k < L && i < n; ++k) part[i++] = (unsigned char)('!' + rng() % 40);
          if (i < n) part[i++] = '\n';
        }
      }
      std::vector<unsigned char> g = gz_of(part, rng() % 10, rng);
      img.insert(img.end(), g.begin(), g.end()); data.insert(data.end(), part.begin(), part.end()); bounds.push_back(data.size());
    }
    const int mode = rng() % 4;
    if (mode == 1 && !img.empty()) img[rng() % img.size()] ^= 1u << (rng() % 8);
    if (mode == 2 && !img.empty()) img.resize(rng() % img.size());
    if (mode == 3) { img.resize(rng() % 3000 + 1); for (auto& b : img) b = rng(); if (rng() & 1) { img[0] = 0x1f; if (img.size() > 3) { img[1] = 0x8b; img[2] = 8; img[3] = 0; } } }
    unsigned char* in = (unsigned char*)malloc(img.size() ? img.size() : 1); memcpy(in, img.data(), img.size());       // exact size: ASan sees any overrun
    c3inf::GzPar par; par.map = in; par.size = img.size(); par.T = 1 + rng() % 6; par.chunk = rng() % 4 == 0 ? (size_t)1 << 22 : 300 + rng() % 60000; par.head = rng() % 3 == 0 ? rng() % 5000 : 0; par.per_round = rng() % 3 == 0 ? 1 + (int)(rng() % 14) : 0;
    std::vector<unsigned char> got; bool bad = !par.open();
    for (int guard = 0; !bad && guard < 100000; ++guard) {
      const bool ok = par.next_round();
      if (par.bad) { bad = true; break; }
      if (ok) for (auto& c : par.chunks) if (c.start != (size_t)-1) got.insert(got.end(), c.cb.out.begin() + (long)par.head, c.cb.out.begin() + (long)(par.head + c.cb.len));
      if (!ok || par.done) break;
    }
    if (mode == 0) { if (bad || got != data) { if (FILE* f = fopen("/tmp/gzpar_fail.gz", "wb")) { fwrite(img.data(), 1, img.size(), f); fclose(f); } printf("MISMATCH it=%d (bad %d, %zu of %zu bytes, T %d chunk %zu)\n", it, (int)bad, got.size(), data.size(), par.T, par.chunk); return 1; } ++oks; }
    else if (bad) ++errs; else if (mode == 1 && got != data && [&] {
      // a flip in the magic / method bytes of a LATER member's header turns the rest of the file into trailing bytes that are no member: the
      // input ends there, as it does for zlib -- the members before it, whole
      for (size_t k = 0; k + 1 < bounds.size(); ++k) if (got.size() == bounds[k] && std::equal(got.begin(), got.end(), data.begin())) return false;
      return true; }()) { /* a flip the CRC cannot see does not exist; one in a header field changes nothing */ printf("SILENT DAMAGE it=%d\n", it); return 1; }
    free(in);
  }
  printf("done: %d images, %ld intact ones decoded exactly, %ld damaged ones reported\n", N, oks, errs);
}
