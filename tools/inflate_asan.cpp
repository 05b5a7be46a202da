// Sanitizer harness of the reader's DEFLATE decoder (CPU only): intact, bit-flipped, truncated and random streams through exact-size heap buffers.
//   cd tools && g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 inflate_asan.cpp -o /tmp/inflate_asan -lz && /tmp/inflate_asan 1 6000
// Round 5: 18 000 streams (three seeds), clean.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <zlib.h>
#include "../c3poa_amd/csrc/c3_inflate.hpp"
int main(int argc, char** argv) {
  std::mt19937_64 rng(argc > 1 ? atoll(argv[1]) : 1);
  const int N = argc > 2 ? atoi(argv[2]) : 20000;
  static c3inf::Inflater inf;
  long errs = 0, oks = 0;
  for (int it = 0; it < N; ++it) {
    // payload
    size_t n = rng() % 70000;
    std::vector<unsigned char> data(n);
    const int kind = rng() % 4;
    for (size_t i = 0; i < n; ++i) data[i] = kind == 0 ? rng() : kind == 1 ? "ACGT"[rng() % 4] : kind == 2 ? (unsigned char)(i / 100) : (unsigned char)('A' + (rng() % 3 == 0));
    uLongf clen = compressBound(n) + 64;
    std::vector<unsigned char> comp(clen);
    compress2(comp.data(), &clen, data.data(), n, 1 + rng() % 9);
    // raw deflate = zlib stream without 2-byte header and 4-byte trailer
    std::vector<unsigned char> raw(comp.begin() + 2, comp.begin() + clen - 4);
    const int mode = rng() % 4;
    if (mode == 1 && !raw.empty()) raw[rng() % raw.size()] ^= 1u << (rng() % 8);
    if (mode == 2 && !raw.empty()) raw.resize(rng() % raw.size());
    if (mode == 3) { raw.resize(rng() % 300 + 1); for (auto& b : raw) b = rng(); }
    // exact-size heap buffers so that ASan sees any overrun
    unsigned char* in = (unsigned char*)malloc(raw.size() ? raw.size() : 1); memcpy(in, raw.data(), raw.size());
    const size_t cap = mode == 0 ? n : rng() % 80000;
    const bool chunked = rng() & 1;
    if (!chunked) {
      unsigned char* out = (unsigned char*)malloc(cap ? cap : 1);
      inf.reset(in, in + raw.size()); size_t pos = 0;
      const int rc = inf.run(out, &pos, cap + 1, cap, 0);
      if (mode == 0) { if (rc != 1 || pos != n || memcmp(out, data.data(), n)) { printf("MISMATCH it=%d\n", it); return 1; } ++oks; } else errs += rc < 0;
      free(out);
    } else {
      const size_t W = 32768, chunk = 1 + rng() % 70000;
      unsigned char* win = (unsigned char*)malloc(W + chunk + 512);
      memset(win, 0, W);
      inf.reset(in, in + raw.size()); size_t hist = 0, total = 0; std::vector<unsigned char> got;
      for (int guard = 0; guard < 100000; ++guard) {
        size_t pos = 0;
        const int rc = inf.run(win + W, &pos, chunk, chunk + 512, hist);
        if (rc < 0) { ++errs; break; }
        got.insert(got.end(), win + W, win + W + pos); total += pos;
        if (pos >= W) { memcpy(win, win + W + pos - W, W); hist = W; } else { memmove(win, win + pos, W); hist = hist + pos < W ? hist + pos : W; }
        if (rc == 1) { if (mode == 0 && (got.size() != n || memcmp(got.data(), data.data(), n))) { printf("MISMATCH chunked it=%d\n", it); return 1; } if (mode == 0) ++oks; break; }
        if (total > 400000) break;
      }
      free(win);
    }
    free(in);
  }
  printf("done: %d streams, %ld intact ones decoded exactly, %ld damaged ones reported\n", N, oks, errs);
}
