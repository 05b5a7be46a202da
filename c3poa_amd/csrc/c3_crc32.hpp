// CRC-32 (the gzip one: polynomial 0xEDB88320, reflected) of a byte range with carry-less multiplication, for the gzip reader (c3_io.cpp,
// c3_gzpar.hpp): zlib 1.2.11's crc32() runs at ~1 GB/s per core, a sixth of what a thread of the parallel decoder spends per chunk.
// The method is the published one (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009):
// fold 4 x 128 bits across 512 with x^(512+-32) mod P, then to 128, to 64, and one Barrett reduction.  The constants are derived below
// from the polynomial at start-up (no table of magic numbers to get wrong), and tests/test_inflate.py holds it against zlib over every
// length 0..300, every alignment and long buffers.  Host code only; falls back to zlib where the CPU has no PCLMULQDQ.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <zlib.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

namespace c3crc {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// x^n mod P in the reflected domain: bit 31 of the result is the coefficient of x^0 ... bit 0 that of x^31 (P = 0xEDB88320 | x^32)
static inline uint32_t xpow_mod(unsigned n) {
  uint32_t r = 0x80000000u;                // x^0
  for (unsigned i = 0; i < n; ++i) r = (r >> 1) ^ ((r & 1u) ? 0xEDB88320u : 0u);
  return r;
}
// the folding constants of the method, as 33-bit values in the layout PCLMULQDQ wants for reflected data: (x^n mod P) << 1
static inline uint64_t kconst(unsigned n) { return (uint64_t)xpow_mod(n) << 1; }
// floor(x^64 / P), reflected, 33 bits (the Barrett constant)
static inline uint64_t barrett_mu() {
  // long division of x^64 by P (33 bits, normal bit order), quotient collected MSB first, then reflected over 33 bits
  const uint64_t P = 0x104C11DB7ull;
  uint64_t rem = 1, q = 0;                 // rem starts as x^0, shifted up 64 times
  for (int i = 0; i < 64; ++i) {
    rem <<= 1;
    q <<= 1;
    if (rem & (1ull << 32)) { rem ^= P; q |= 1; }
  }
  // q now holds the low 64 bits of the quotient; the quotient has degree 32: take its 33 low bits and reflect
  uint64_t r = 0;
  for (int i = 0; i < 33; ++i) if (q & (1ull << i)) r |= 1ull << (32 - i);
  return r;
}
struct Consts { uint64_t k1, k2, k3, k4, k5, mu, poly; bool ok; };
static inline const Consts& consts() {
  static const Consts c = [] {
    Consts k;
    k.k1 = kconst(4 * 128 + 32); k.k2 = kconst(4 * 128 - 32);
    k.k3 = kconst(128 + 32);     k.k4 = kconst(128 - 32);
    k.k5 = kconst(64);
    k.mu = barrett_mu(); k.poly = 0x1DB710641ull;
    __builtin_cpu_init();
    k.ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return k;
  }();
  return c;
}

// state in, state out (no inversion), n >= 64 and a multiple of 16
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t fold_le(uint32_t crc, const unsigned char* p, size_t n) {
  const Consts& K = consts();
  __m128i x1 = _mm_loadu_si128((const __m128i*)p), x2 = _mm_loadu_si128((const __m128i*)(p + 16));
  __m128i x3 = _mm_loadu_si128((const __m128i*)(p + 32)), x4 = _mm_loadu_si128((const __m128i*)(p + 48));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  p += 64; n -= 64;
  __m128i k = _mm_set_epi64x((long long)K.k2, (long long)K.k1);
  while (n >= 64) {
    __m128i a1 = _mm_clmulepi64_si128(x1, k, 0x00), b1 = _mm_clmulepi64_si128(x1, k, 0x11);
    __m128i a2 = _mm_clmulepi64_si128(x2, k, 0x00), b2 = _mm_clmulepi64_si128(x2, k, 0x11);
    __m128i a3 = _mm_clmulepi64_si128(x3, k, 0x00), b3 = _mm_clmulepi64_si128(x3, k, 0x11);
    __m128i a4 = _mm_clmulepi64_si128(x4, k, 0x00), b4 = _mm_clmulepi64_si128(x4, k, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(a1, b1), _mm_loadu_si128((const __m128i*)p));
    x2 = _mm_xor_si128(_mm_xor_si128(a2, b2), _mm_loadu_si128((const __m128i*)(p + 16)));
    x3 = _mm_xor_si128(_mm_xor_si128(a3, b3), _mm_loadu_si128((const __m128i*)(p + 32)));
    x4 = _mm_xor_si128(_mm_xor_si128(a4, b4), _mm_loadu_si128((const __m128i*)(p + 48)));
    p += 64; n -= 64;
  }
  k = _mm_set_epi64x((long long)K.k4, (long long)K.k3);
  x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k, 0x00), _mm_clmulepi64_si128(x1, k, 0x11)), x2);
  x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k, 0x00), _mm_clmulepi64_si128(x1, k, 0x11)), x3);
  x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k, 0x00), _mm_clmulepi64_si128(x1, k, 0x11)), x4);
  while (n >= 16) {
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k, 0x00), _mm_clmulepi64_si128(x1, k, 0x11)), _mm_loadu_si128((const __m128i*)p));
    p += 16; n -= 16;
  }
  // 128 -> 64 bits
  const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
  x2 = _mm_clmulepi64_si128(x1, k, 0x10);                 // low half * k4
  x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
  // 64 -> 32 bits
  k = _mm_set_epi64x(0, (long long)K.k5);
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, mask32);
  x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k, 0x00), x2);
  // Barrett
  k = _mm_set_epi64x((long long)K.mu, (long long)K.poly);
  x2 = x1;
  x1 = _mm_and_si128(x1, mask32);
  x1 = _mm_clmulepi64_si128(x1, k, 0x10);
  x1 = _mm_and_si128(x1, mask32);
  x1 = _mm_clmulepi64_si128(x1, k, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// zlib's crc32(), any length
static inline uint32_t crc32_fast(uint32_t crc, const unsigned char* p, size_t n) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (n >= 128 && consts().ok) {
    const size_t body = n & ~(size_t)15;
    crc = ~fold_le(~crc, p, body);
    p += body; n -= body;
  }
#endif
  while (n) { const size_t q = n < ((size_t)1 << 30) ? n : ((size_t)1 << 30); crc = (uint32_t)::crc32(crc, p, (unsigned)q); p += q; n -= q; }
  return crc;
}
}  // namespace c3crc
