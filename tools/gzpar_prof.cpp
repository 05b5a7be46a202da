// Per-round phase times of the parallel gzip decoder (find / decode / chain / convert / verify, printed by c3_gzpar.hpp under C3_GZPAR_PROF):
//   g++ -O3 -std=c++17 -DC3_GZPAR_PROF tools/gzpar_prof.cpp -o /tmp/gzpar_prof -lz -lpthread && /tmp/gzpar_prof file.gz THREADS CHUNK_BYTES [CHUNKS_PER_ROUND]
#include <chrono>
#include <cstdio>
static double now_() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#include "../c3poa_amd/csrc/c3_gzpar.hpp"
int main(int argc, char** argv) {
  if (argc < 4) return 1;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
  fseek(f, 0, SEEK_END); const size_t n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> d(n); if (fread(d.data(), 1, n, f) != n) return 1; fclose(f);
  c3inf::GzPar par; par.map = d.data(); par.size = n; par.T = atoi(argv[2]); par.chunk = (size_t)atol(argv[3]); par.head = 4096; par.per_round = argc > 4 ? atoi(argv[4]) : 0;
  const double t0 = now_(); size_t tot = 0;
  if (!par.open()) return 2;
  for (;;) { const bool ok = par.next_round(); if (par.bad) return 3; if (ok) for (auto& c : par.chunks) if (c.start != (size_t)-1) tot += c.cb.len; if (!ok || par.done) break; }
  fprintf(stderr, "total %.3f s, %.1f MB/s\n", now_() - t0, tot / (now_() - t0) / 1e6);
}
