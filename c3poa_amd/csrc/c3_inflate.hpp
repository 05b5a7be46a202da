// c3_inflate.hpp -- a DEFLATE (RFC 1951) decoder for the reader of c3_io.cpp (host only, no GPU code).
//
// Why not zlib's: the input side of the command line (C3POa.py:201,239 reads `.gz` through Python's gzip) is one zlib stream per
// plain gzip file, and zlib inflates FASTQ at ~300 MB/s whatever the machine -- 27-31 k reads/s, a fifteenth of one GPU
// (profiles/r04_host_ceiling_gz.txt, profiles/r05_inflate_bench_gpu_box_host.txt: 30.8 -> 49.8 k reads/s with this decoder).  This decoder is written for exactly this job: the WHOLE compressed input is in memory
// (the file is mapped), so there is no input-starvation state; it stops only BETWEEN symbols (when the caller's output chunk is
// full), so the resumable state is the bit buffer, the input position and the current block's tables; literals and matches are
// decoded from an 11-bit first-level table (two literals per entry where both codes fit) with a 64-bit bit buffer that is refilled
// once per three lookups, matches are copied eight bytes at a time.  Every member's CRC-32 and length are checked by the caller (zlib's crc32): a decoding error cannot
// pass silently.  The algorithm is the published one (RFC 1951 3.2; table-driven canonical Huffman decoding as in every fast
// inflater); the code is this repository's own.
#pragma once
#include <stdint.h>
#include <string.h>
#include <stddef.h>

namespace c3inf {

// table entries (32 bits):  bits 0-7 code length (bits to drop for the code itself), bits 8-12 extra-bit count / sub-table bits / 1 = "two
// literals", bits 13-28 literal byte(s) / length base / distance base / sub-table offset, bits 29-30 the kind, bit 31 = the entry exists
enum : uint32_t { K_LIT = 0u << 29, K_LEN = 1u << 29, K_EOB = 2u << 29, K_SUB = 3u << 29, K_MASK = 3u << 29, K_VALID = 1u << 31 };
static inline uint32_t mk(uint32_t kind, uint32_t val, uint32_t extra, uint32_t len) { return K_VALID | kind | (val << 8 << 5) | (extra << 8) | len; }
static inline uint32_t e_len(uint32_t e) { return e & 0xffu; }
static inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 0x1fu; }
static inline uint32_t e_val(uint32_t e) { return (e >> 13) & 0xffffu; }

const int LIT_BITS = 11, DIST_BITS = 8;
// first level + room for the second-level tables.  Enough for every complete code: a prefix whose longest code is sb bits beyond the first
// level holds at least sb + 1 codes, so the second level is at most 57 x 16 = 912 entries for the 286 literal / length codes (sb <= 4) and
// 3 x 128 + 32 = 416 for the 30 distance codes (sb <= 7)
const int LIT_TABLE = (1 << LIT_BITS) + 1024, DIST_TABLE = (1 << DIST_BITS) + 512;

struct Inflater {
  const uint8_t* in = nullptr; const uint8_t* in_end = nullptr;
  uint64_t bitbuf = 0; unsigned bitcnt = 0;
  bool in_block = false, final_block = false, stream_end = false;
  int btype = 0; size_t stored_left = 0;
  uint32_t lit[LIT_TABLE], dist[DIST_TABLE];
  bool fixed_built = false; uint32_t fixed_lit[LIT_TABLE], fixed_dist[DIST_TABLE];

  void reset(const uint8_t* p, const uint8_t* e) { in = p; in_end = e; bitbuf = 0; bitcnt = 0; in_block = final_block = stream_end = false; stored_left = 0; }

  // canonical Huffman table: lens[0..n) code lengths (0 = unused), PB first-level bits.  kind_of(sym) builds the entry for a symbol.
  // Returns false on an over-subscribed set, or an incomplete one (allowed where allow_single: exactly one code of length 1 -- RFC 1951
  // 3.2.7, one distance code; zlib accepts the same for the literal / length set and for no other: the code-length alphabet must be complete)
  template <class F>
  static bool build(const uint8_t* lens, int n, uint32_t* tab, int PB, int cap, F entry_of, bool allow_single = true) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    if (count[0] == n) { for (int i = 0; i < (1 << PB); ++i) tab[i] = 0; return true; }       // no codes at all (a block without distances)
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
    const bool single = allow_single && (n - count[0] == 1) && count[1] == 1;
    if (left > 0 && !single) return false;
    int offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
    uint16_t sorted[320];
    for (int i = 0; i < n; ++i) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
    for (int i = 0; i < (1 << PB); ++i) tab[i] = 0;
    // codes in increasing (length, symbol) order; a code is the bit-reversed canonical code (the stream is LSB first)
    unsigned code = 0; int si = 0;
    // the longest code under every first-level prefix that needs a second level
    uint8_t sub_bits[1 << LIT_BITS];
    memset(sub_bits, 0, (size_t)1 << PB);
    {
      unsigned c2 = 0; int s2 = 0;
      for (int l = 1; l <= 15; ++l) {
        for (int k = 0; k < count[l]; ++k, ++s2, ++c2) {
          if (l > PB) { unsigned rev = 0; for (int b = 0; b < l; ++b) rev |= ((c2 >> b) & 1u) << (l - 1 - b); const unsigned pre = rev & ((1u << PB) - 1); if ((int)sub_bits[pre] < l - PB) sub_bits[pre] = (uint8_t)(l - PB); }
        }
        c2 <<= 1;
      }
    }
    int next_sub = 1 << PB;
    for (int l = 1; l <= 15; ++l) {
      for (int k = 0; k < count[l]; ++k, ++si, ++code) {
        const int sym = sorted[si];
        unsigned rev = 0; for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
        if (l <= PB) {
          const uint32_t e = entry_of(sym, l);
          for (unsigned i = rev; i < (1u << PB); i += 1u << l) tab[i] = e;
        } else {
          const unsigned pre = rev & ((1u << PB) - 1);
          const int sb = sub_bits[pre];
          if (!(tab[pre] & K_VALID)) {
            if (next_sub + (1 << sb) > cap) return false;
            tab[pre] = mk(K_SUB, (uint32_t)next_sub, (uint32_t)sb, (uint32_t)PB);
            for (int i = 0; i < (1 << sb); ++i) tab[next_sub + i] = 0;
            next_sub += 1 << sb;
          }
          const int base = (int)e_val(tab[pre]);
          const uint32_t e = entry_of(sym, l - PB);
          for (unsigned i = rev >> PB; i < (1u << sb); i += 1u << (l - PB)) tab[base + i] = e;
        }
      }
      code <<= 1;
    }
    return true;
  }
  // TWO LITERALS PER LOOKUP: where a first-level index holds a literal of L1 bits and its remaining 11 - L1 bits decide a second
  // literal of L2 <= 11 - L1 bits, the entry becomes both (value = first | second << 8, extra field = 1, length = L1 + L2).  FASTQ
  // deflates to literals for the most part (quality strings: 5-7 bit codes), and Huffman decoding is one dependent lookup per
  // symbol: pairs halve that chain.
  static void pair_literals(uint32_t* tab) {
    uint32_t one[1 << LIT_BITS];
    memcpy(one, tab, sizeof(one));
    for (unsigned i = 0; i < (1u << LIT_BITS); ++i) {
      const uint32_t e = one[i];
      if ((e & (K_MASK | K_VALID)) != (K_LIT | K_VALID)) continue;
      const unsigned l1 = e_len(e);
      if (l1 >= (unsigned)LIT_BITS) continue;
      const uint32_t e2 = one[i >> l1];
      if ((e2 & (K_MASK | K_VALID)) != (K_LIT | K_VALID) || l1 + e_len(e2) > (unsigned)LIT_BITS) continue;
      tab[i] = mk(K_LIT, e_val(e) | (e_val(e2) << 8), 1, l1 + e_len(e2));
    }
  }
  static uint32_t lit_entry(int sym, int len) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (sym < 256) return mk(K_LIT, (uint32_t)sym, 0, (uint32_t)len);
    if (sym == 256) return mk(K_EOB, 0, 0, (uint32_t)len);
    if (sym > 285) return 0;                                        // 286 / 287 take part in the code but never appear (RFC 1951 3.2.6)
    return mk(K_LEN, lbase[sym - 257], lext[sym - 257], (uint32_t)len);
  }
  static uint32_t dist_entry(int sym, int len) {
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (sym > 29) return 0;
    return mk(K_LEN, dbase[sym], dext[sym], (uint32_t)len);
  }

  // ---- bits.  refill: at least 56 valid bits afterwards while 8 bytes of input can be read; near the end of the input byte by byte
  inline void refill_fast() { uint64_t w; memcpy(&w, in, 8); bitbuf |= w << bitcnt; const unsigned add = (63 - bitcnt) >> 3; in += add; bitcnt += add * 8; }
  inline void refill_safe() { while (bitcnt <= 56 && in < in_end) { bitbuf |= (uint64_t)*in++ << bitcnt; bitcnt += 8; } }
  inline bool take(unsigned n, uint32_t* v) { if (bitcnt < n) { refill_safe(); if (bitcnt < n) return false; } *v = (uint32_t)(bitbuf & ((1ull << n) - 1)); bitbuf >>= n; bitcnt -= n; return true; }

  // first byte after the deflate stream (valid once run() has returned 1): the rest of the last byte is padding, whole unread bytes go back
  const uint8_t* byte_pos() const { return in - ((bitcnt - (bitcnt & 7)) >> 3); }

  bool read_block_header() {
    uint32_t v;
    if (!take(3, &v)) return false;
    final_block = v & 1; btype = (int)(v >> 1);
    if (btype == 0) {
      // stored: drop the rest of the byte, give whole unread bytes back to the input
      const unsigned drop = bitcnt & 7; bitbuf >>= drop; bitcnt -= drop;
      in -= bitcnt >> 3; bitbuf = 0; bitcnt = 0;
      if (in_end - in < 4) return false;
      const unsigned len = in[0] | (in[1] << 8), nlen = in[2] | (in[3] << 8);
      if ((len ^ nlen) != 0xffffu) return false;
      in += 4; stored_left = len;
      return true;
    }
    if (btype == 1) {
      if (!fixed_built) {
        uint8_t l[288]; for (int i = 0; i < 144; ++i) l[i] = 8; for (int i = 144; i < 256; ++i) l[i] = 9; for (int i = 256; i < 280; ++i) l[i] = 7; for (int i = 280; i < 288; ++i) l[i] = 8;
        uint8_t d[32]; for (int i = 0; i < 32; ++i) d[i] = 5;
        if (!build(l, 288, fixed_lit, LIT_BITS, LIT_TABLE, lit_entry) || !build(d, 32, fixed_dist, DIST_BITS, DIST_TABLE, dist_entry)) return false;
        pair_literals(fixed_lit);
        fixed_built = true;
      }
      memcpy(lit, fixed_lit, sizeof(lit)); memcpy(dist, fixed_dist, sizeof(dist));
      return true;
    }
    if (btype != 2) return false;
    uint32_t hlit, hdist, hclen;
    if (!take(5, &hlit) || !take(5, &hdist) || !take(4, &hclen)) return false;
    hlit += 257; hdist += 1; hclen += 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (uint32_t i = 0; i < hclen; ++i) { uint32_t v3; if (!take(3, &v3)) return false; cl[order[i]] = (uint8_t)v3; }
    uint32_t cltab[(1 << 7) + 8];
    if (!build(cl, 19, cltab, 7, (1 << 7) + 8, [](int sym, int len) { return mk(K_LIT, (uint32_t)sym, 0, (uint32_t)len); }, false)) return false;
    uint8_t lens[320] = {0};
    for (uint32_t i = 0; i < hlit + hdist;) {
      if (bitcnt < 7 + 7) refill_safe();
      const uint32_t e = cltab[bitbuf & 127];
      if (!(e & K_VALID) || e_len(e) > bitcnt) return false;
      bitbuf >>= e_len(e); bitcnt -= e_len(e);
      const uint32_t sym = e_val(e);
      if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
      uint32_t rep, val = 0, x;
      if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; if (!take(2, &x)) return false; rep = 3 + x; }
      else if (sym == 17) { if (!take(3, &x)) return false; rep = 3 + x; }
      else { if (!take(7, &x)) return false; rep = 11 + x; }
      if (i + rep > hlit + hdist) return false;
      while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) return false;                                // no end-of-block code
    if (!build(lens, (int)hlit, lit, LIT_BITS, LIT_TABLE, lit_entry) || !build(lens + hlit, (int)hdist, dist, DIST_BITS, DIST_TABLE, dist_entry)) return false;
    pair_literals(lit);
    return true;
  }

  // Decode into out[pos ..): stops BETWEEN symbols as soon as pos >= soft (the caller's chunk is full), at the end of the stream, or on an
  // error.  cap = bytes that may be written (>= soft + 258 for a chunked caller; the exact size for a member of known size); hist = bytes
  // before out[0] that are valid history... (the caller keeps the last 32 KiB in front of the chunk).  Returns 1 at the end of the
  // deflate stream, 0 when the chunk is full, -1 on an error.
  int run(uint8_t* out, size_t* pos_io, size_t soft, size_t cap, size_t hist) {
    size_t pos = *pos_io;
    for (;;) {
      if (!in_block) {
        if (stream_end) { *pos_io = pos; return 1; }
        if (!read_block_header()) { *pos_io = pos; return -1; }
        in_block = true;
      }
      if (btype == 0) {
        while (stored_left) {
          if (pos >= soft) { *pos_io = pos; return 0; }
          size_t n = stored_left;
          if (n > cap - pos) n = cap - pos;
          if (n > (size_t)(in_end - in)) { *pos_io = pos; return -1; }
          if (n == 0) { *pos_io = pos; return cap > pos ? -1 : 0; }
          memcpy(out + pos, in, n); in += n; pos += n; stored_left -= n;
        }
        in_block = false; if (final_block) stream_end = true;
        continue;
      }
      // ---- compressed block
      for (;;) {
        // FAST LOOP: while 16 bytes of input and 274 bytes of output room are certain, no per-symbol bounds checks; one refill
        // (>= 56 bits) serves up to three literals (first-level codes are at most 11 bits) and the length code + its extra bits
        // (<= 20), a second one the distance (<= 28).  FASTQ deflates to literals for the most part: this path decides the rate.
        {
          const size_t out_fast = cap >= 274 ? (soft < cap - 274 ? soft : cap - 274) : 0;
          const uint32_t LM = (1u << LIT_BITS) - 1;
          while ((size_t)(in_end - in) >= 16 && pos < out_fast) {      // (not `in <= in_end - 16`: a stream shorter than 16 bytes would form a pointer before its start)
            refill_fast();
            uint32_t e = lit[bitbuf & LM];
            // up to three lookups of literals (one or two bytes each, at most 11 bits each) out of one refill
#define C3INF_LIT() { bitbuf >>= e_len(e); bitcnt -= e_len(e); const uint32_t v_ = e_val(e), two_ = e_extra(e); out[pos] = (uint8_t)v_; out[pos + 1] = (uint8_t)(v_ >> 8); pos += 1 + two_; e = lit[bitbuf & LM]; }
            if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) {
              C3INF_LIT()
              if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) {
                C3INF_LIT()
                if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) C3INF_LIT()
              }
            }
#undef C3INF_LIT
            // e: the next symbol, looked up with >= 23 valid bits left
            if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) continue;                 // (a fourth literal: after the next refill)
            if ((e & (K_MASK | K_VALID)) != (K_LEN | K_VALID)) break;                    // second-level code, end of block, damaged: the careful loop
            bitbuf >>= e_len(e); bitcnt -= e_len(e);
            unsigned len = e_val(e); const unsigned xb = e_extra(e);
            len += (unsigned)(bitbuf & ((1u << xb) - 1)); bitbuf >>= xb; bitcnt -= xb;
            refill_fast();
            uint32_t d = dist[bitbuf & ((1u << DIST_BITS) - 1)];
            if ((d & (K_MASK | K_VALID)) == (K_SUB | K_VALID)) { d = dist[e_val(d) + ((bitbuf >> DIST_BITS) & ((1u << e_extra(d)) - 1))]; bitbuf >>= DIST_BITS; bitcnt -= DIST_BITS; }
            if ((d & (K_MASK | K_VALID)) != (K_LEN | K_VALID)) { *pos_io = pos; return -1; }
            bitbuf >>= e_len(d); bitcnt -= e_len(d);
            unsigned dd = e_val(d); const unsigned db = e_extra(d);
            dd += (unsigned)(bitbuf & ((1u << db) - 1)); bitbuf >>= db; bitcnt -= db;
            if (dd > pos + hist) { *pos_io = pos; return -1; }
            uint8_t* dst = out + pos; const uint8_t* src = dst - dd;
            pos += len;
            uint8_t* const end = dst + len;
            if (dd >= 8) { do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < end); }
            else if (dd == 1) memset(dst, *src, len);
            else { do { *dst++ = *src++; } while (dst < end); }
          }
        }
        if (pos >= soft) { *pos_io = pos; return 0; }
        const bool fast = (in_end - in) >= 16 && cap - pos >= 258 + 16;
        if (fast) refill_fast(); else refill_safe();
        uint32_t e = lit[bitbuf & ((1u << LIT_BITS) - 1)];
        if ((e & (K_MASK | K_VALID)) == (K_SUB | K_VALID)) {             // a code longer than the first level: second-level table
          if (bitcnt < (unsigned)LIT_BITS) { *pos_io = pos; return -1; }
          e = lit[e_val(e) + ((bitbuf >> LIT_BITS) & ((1u << e_extra(e)) - 1))]; bitbuf >>= LIT_BITS; bitcnt -= LIT_BITS;
        }
        if (!(e & K_VALID) || e_len(e) > bitcnt) { *pos_io = pos; return -1; }
        bitbuf >>= e_len(e); bitcnt -= e_len(e);
        const uint32_t kind = e & K_MASK;
        if (kind == K_LIT) {
          const unsigned nl = 1 + e_extra(e);                        // one literal, or a pair
          if (nl > cap - pos) { *pos_io = pos; return -1; }
          out[pos++] = (uint8_t)e_val(e);
          if (nl == 2) out[pos++] = (uint8_t)(e_val(e) >> 8);
          continue;
        }
        if (kind == K_EOB) { in_block = false; if (final_block) stream_end = true; break; }
        // length + distance
        unsigned len = e_val(e), xb = e_extra(e);
        if (xb) { if (bitcnt < xb) { *pos_io = pos; return -1; } len += (unsigned)(bitbuf & ((1u << xb) - 1)); bitbuf >>= xb; bitcnt -= xb; }
        if (!fast) refill_safe();
        uint32_t d = dist[bitbuf & ((1u << DIST_BITS) - 1)];
        if ((d & (K_MASK | K_VALID)) == (K_SUB | K_VALID)) {
          if (bitcnt < (unsigned)DIST_BITS) { *pos_io = pos; return -1; }
          d = dist[e_val(d) + ((bitbuf >> DIST_BITS) & ((1u << e_extra(d)) - 1))]; bitbuf >>= DIST_BITS; bitcnt -= DIST_BITS;
        }
        if (!(d & K_VALID) || e_len(d) > bitcnt) { *pos_io = pos; return -1; }
        bitbuf >>= e_len(d); bitcnt -= e_len(d);
        unsigned dd = e_val(d), db = e_extra(d);
        if (db) { if (bitcnt < db) { refill_safe(); if (bitcnt < db) { *pos_io = pos; return -1; } } dd += (unsigned)(bitbuf & ((1u << db) - 1)); bitbuf >>= db; bitcnt -= db; }
        if (dd > pos + hist || len > cap - pos) { *pos_io = pos; return -1; }
        uint8_t* dst = out + pos; const uint8_t* src = dst - dd;
        pos += len;
        if (fast && dd >= 8) {
          // eight bytes at a time; may write up to 7 bytes beyond the match (room: cap - pos >= 258 + 16 was checked before the symbol)
          uint8_t* const end = dst + len;
          do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < end);
        } else if (dd == 1) {
          memset(dst, *src, len);
        } else {
          for (unsigned i = 0; i < len; ++i) dst[i] = src[i];
        }
      }
    }
  }
};

}  // namespace c3inf
