// c3_gzpar.hpp -- ONE plain gzip stream inflated by several threads (host only, no GPU code; round 6).
//
// Why: the command line reads `.gz` input (C3POa.py:201,239), ONT data ships as plain gzip -- one deflate stream, not BGZF -- and one
// thread inflates FASTQ at ~0.5 GB/s with the decoder of c3_inflate.hpp: ~50 k reads/s, a ninth of ONE GPU (profiles/r05_cli_500k_gz_*).
// A deflate stream cannot be entered in the middle in general: Huffman codes are not self-synchronising at the bit level and a match
// may copy from the 32 KiB before it.  The two-pass scheme used here is the published one (Kerbiriou & Chikhi 2019, "pugz"; rapidgzip):
//
//   1. the compressed file is cut into chunks of `chunk` bytes; in every chunk but the first a thread looks for the first bit offset
//      that holds a DYNAMIC block header whose three Huffman codes are complete and whose block decodes to its end-of-block symbol
//      with another well-formed block header behind it (probe);
//   2. every thread decodes from its offset with an UNKNOWN window: a copy from before the chunk yields a MARK -- "this byte is byte k of
//      the 32 KiB before the chunk" -- kept beside the bytes (a bitmap + a 16-bit index plane: ChunkBuf below), copies of marked bytes copy
//      the mark, and it stops at the block boundary where the next chunk begins;
//   3. the chunks are chained in order: the chunk before must have stopped EXACTLY where this one started (an offset that is not a
//      block boundary of the real stream -- a false positive of step 1 -- is found out here, structurally: its chunk is dropped and
//      the chunk before decodes on through it), its last 32 KiB, resolved, are this chunk's window;
//   4. every thread fills in the marked bytes of its chunk and takes the CRC-32 of every stretch of a gzip member in it; the stretches are
//      combined (crc32_combine) and checked against every member's trailer, as is its length.
// A round holds `per_round` chunks (two per thread in the reader); the threads take chunks off a counter, every numbered step ends at a barrier.
//
// Members may be concatenated (one stream, as for gzread); bytes behind the last member that are no gzip header end the input.
// The code is this repository's own; zlib supplies crc32_combine (and crc32 where the CPU has no carry-less multiply: c3_crc32.hpp).
#pragma once
#include "c3_inflate.hpp"
#include "c3_crc32.hpp"
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace c3inf {

struct MemberEnd { size_t off; uint32_t crc, isize; };      // a gzip member ends after `off` symbols of the chunk; its trailer's fields

// deflate data of the gzip member whose header starts at p (n bytes available); nullptr when p is no gzip header
static inline const uint8_t* gz_member_data(const uint8_t* p, size_t n) {
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return nullptr;
  const int flg = p[3]; size_t q = 10;
  if (flg & 4) { if (q + 2 > n) return nullptr; q += 2 + ((size_t)p[q] | ((size_t)p[q + 1] << 8)); }
  if (flg & 8) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 16) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 2) q += 2;
  if (q + 8 > n) return nullptr;
  return p + q;
}

// What one chunk decodes into when the 32 KiB before it are unknown: BYTES, in the buffer the parser will read them from, plus -- for the
// positions whose value is a byte of that unknown window -- a MARK: a bit in a bitmap (one per output byte) and the index into the window in
// a 16-bit plane beside the bytes (`widx`, never initialised: an entry means something only where the bit is set).  A copy whose source
// range holds no set bit is the ordinary byte copy; the common copy (at most 64 bytes, not overlapping itself) that does hold marks copies
// its stretch of the index plane as well and ORs the source's bits, shifted, into the destination's -- constant work per copy, however
// many marks; only long or self-overlapping copies with marks go byte by byte.  (History: the first version decoded into 16-bit symbols,
// a literal its byte, a window byte 0x8000 | index -- half the copy rate and a pass to narrow the symbols, 0.4x the plain decoder.  The
// second kept the marks in a sorted list: fine at the 0.5 % of marked bytes `gzip -6` FASTQ settles at -- headers are copied from one
// another, so "switch to bytes once the last 32 KiB are clean", the published trick, never fires -- and 2x SLOWER than the first on
// `gzip -1` output of repetitive reads, where 25-45 % of a chunk's bytes stay marked to its end.  This form costs what the first did where
// marks are dense and what the plain decoder does where they are not.)
struct ChunkBuf {
  std::vector<char> out;                   // `head` free bytes, then the chunk's bytes
  size_t head = 0, len = 0, mk_end = 0;    // mk_end: one past the last marked byte
  std::vector<uint64_t> bits;              // bit i: byte i is a placeholder for a byte of the unknown window ...
  uint16_t* widx = nullptr; size_t wcap = 0;      // ... namely window[widx[i]]
  ChunkBuf() = default;
  ChunkBuf(const ChunkBuf&) = delete; ChunkBuf& operator=(const ChunkBuf&) = delete;
  ChunkBuf(ChunkBuf&& o) noexcept { *this = std::move(o); }
  ChunkBuf& operator=(ChunkBuf&& o) noexcept {
    if (this != &o) { free(widx); out = std::move(o.out); bits = std::move(o.bits); head = o.head; len = o.len; mk_end = o.mk_end; widx = o.widx; wcap = o.wcap; o.widx = nullptr; o.wcap = 0; o.len = 0; o.mk_end = 0; }
    return *this;
  }
  ~ChunkBuf() { free(widx); }
  bool grow(size_t need) {                 // room for `need` bytes behind `head`, and bits + index entries for EVERY byte the buffer can hold (the decoder
    const size_t have = out.size() > head ? out.size() - head : 0;      // takes the buffer's size as its room; a buffer that came back from the reader keeps its size)
    size_t nc = std::max<size_t>(have, (size_t)1 << 16);
    while (nc < need) nc *= 2;
    if (out.size() < head + nc) out.resize(head + nc);
    const size_t room = out.size() - head;
    if (bits.size() < room / 64 + 2) bits.resize(room / 64 + 2, 0);
    if (wcap < room) {
      uint16_t* nw = (uint16_t*)realloc(widx, (room + 64) * sizeof(uint16_t));
      if (!nw) return false;
      widx = nw; wcap = room;
    }
    return true;
  }
  void drop_marks() { std::vector<uint64_t>().swap(bits); free(widx); widx = nullptr; wcap = 0; mk_end = 0; }
  uint16_t* release_widx(size_t* cap) { uint16_t* p = widx; *cap = wcap; widx = nullptr; wcap = 0; return p; }      // (the caller keeps the plane for another chunk)
  bool has_marks(size_t src, size_t n) const {          // (conservative: whole words)
    for (size_t w = src >> 6, we = (src + n - 1) >> 6; w <= we; ++w) if (bits[w]) return true;
    return false;
  }
  bool marked(size_t p) const { return (bits[p >> 6] >> (p & 63)) & 1; }
  void mark(size_t p, uint16_t wi) { bits[p >> 6] |= 1ull << (p & 63); widx[p] = wi; mk_end = p + 1; }
  // every mark in [from, to): f(position, window index)
  template <class F> void for_marks(size_t from, size_t to, F f) const {
    if (to > mk_end) to = mk_end;
    for (size_t w = from >> 6; (w << 6) < to; ++w) {
      uint64_t b = bits[w];
      if (w == (from >> 6)) b &= ~0ull << (from & 63);
      while (b) { const size_t p = (w << 6) + (size_t)__builtin_ctzll(b); b &= b - 1; if (p < to) f(p, widx[p]); }
    }
  }
};

struct Dec16 {
  Inflater z;
  const uint8_t* base = nullptr; const uint8_t* fend = nullptr;

  void seek(size_t bit) {
    z.reset(base + (bit >> 3), fend);
    z.refill_safe();
    const unsigned drop = (unsigned)(bit & 7);
    z.bitbuf >>= drop; z.bitcnt -= drop;
  }
  size_t bitpos() const { return (size_t)(z.in - base) * 8 - z.bitcnt; }

  // Decode blocks from the current position into `co` (bytes appended; `known` = the decoder starts at the first block of a member, so
  // nothing may be copied from before the chunk).  Stops at the first block boundary at or beyond stop_bit that lies behind the start
  // (returns 0), at the end of the last member (1), or on an error / more than max_out bytes (-1).  *end_bit = where it stopped.
  int run(ChunkBuf& co, size_t stop_bit, size_t* end_bit, bool known, std::vector<MemberEnd>& ends, size_t max_out) {
    const size_t start_bit = bitpos();
    size_t pos = co.len;
    co.grow(pos + 65536);
    uint8_t* o = (uint8_t*)co.out.data() + co.head; size_t capv = co.out.size() - co.head;
    size_t mstart = 0;                       // first byte of the member being decoded (0: the chunk began inside it, or with it when `known`)
    bool reach = !known;                     // copies may reach up to 32 KiB before the chunk
    const uint32_t LM = (1u << LIT_BITS) - 1;
    auto fail = [&]() { co.len = pos; return -1; };
#define C3GZ_ROOM(n_) if (pos + (n_) + 600 > capv) { if (pos + (n_) + 600 > max_out) return fail(); co.grow(std::max(pos + (n_) + 600, 2 * capv)); o = (uint8_t*)co.out.data() + co.head; capv = co.out.size() - co.head; }
    for (;;) {
      const size_t bp = bitpos();
      if (bp >= stop_bit && bp > start_bit) { *end_bit = bp; co.len = pos; return 0; }
      if (!z.read_block_header()) return fail();
      if (z.btype == 0) {
        const size_t n = z.stored_left;
        if (n > (size_t)(fend - z.in)) return fail();
        C3GZ_ROOM(n)
        memcpy(o + pos, z.in, n);
        z.in += n; pos += n; z.stored_left = 0;
      } else {
        for (;;) {
          C3GZ_ROOM(0)
          if ((size_t)(fend - z.in) >= 16) {
            // FAST LOOP (as Inflater::run's: no per-symbol bounds checks while 16 bytes of input and 600 of room are certain; the bit buffer lives in
            // locals, stores through `o` may alias the decoder's fields).  Leaves at a second-level literal/length code, the end of the block, a
            // copy from before the chunk or anything damaged: the careful code below decodes that symbol again.
            uint64_t bb = z.bitbuf; unsigned bc = z.bitcnt; const uint8_t* in = z.in; const uint8_t* const in_stop = fend - 16;
            const uint32_t* const lit = z.lit; const uint32_t* const dist = z.dist;
            const size_t lim = capv - 600;
            while (in <= in_stop && pos <= lim) {
              { uint64_t w; memcpy(&w, in, 8); bb |= w << bc; const unsigned add = (63 - bc) >> 3; in += add; bc += add * 8; }
              uint32_t e = lit[bb & LM];
#define C3INF_LITB() { bb >>= e_len(e); bc -= e_len(e); const uint32_t v_ = e_val(e), two_ = e_extra(e); o[pos] = (uint8_t)v_; o[pos + 1] = (uint8_t)(v_ >> 8); pos += 1 + two_; e = lit[bb & LM]; }
              if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) {
                C3INF_LITB()
                if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) {
                  C3INF_LITB()
                  if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) C3INF_LITB()
                }
              }
#undef C3INF_LITB
              if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) continue;
              if ((e & (K_MASK | K_VALID)) != (K_LEN | K_VALID)) break;
              const uint64_t bb0 = bb; const unsigned bc0 = bc; const uint8_t* const in0 = in;          // (to hand this symbol to the careful code)
              bb >>= e_len(e); bc -= e_len(e);
              unsigned len = e_val(e); const unsigned xb = e_extra(e);
              len += (unsigned)(bb & ((1u << xb) - 1)); bb >>= xb; bc -= xb;
              { uint64_t w; memcpy(&w, in, 8); bb |= w << bc; const unsigned add = (63 - bc) >> 3; in += add; bc += add * 8; }
              uint32_t d = dist[bb & ((1u << DIST_BITS) - 1)];
              if ((d & (K_MASK | K_VALID)) == (K_SUB | K_VALID)) { d = dist[e_val(d) + ((bb >> DIST_BITS) & ((1u << e_extra(d)) - 1))]; bb >>= DIST_BITS; bc -= DIST_BITS; }
              if ((d & (K_MASK | K_VALID)) != (K_LEN | K_VALID)) { bb = bb0; bc = bc0; in = in0; break; }
              bb >>= e_len(d); bc -= e_len(d);
              unsigned dd = e_val(d); const unsigned db = e_extra(d);
              dd += (unsigned)(bb & ((1u << db) - 1)); bb >>= db; bc -= db;
              if ((size_t)dd > pos - mstart) { bb = bb0; bc = bc0; in = in0; break; }
              const size_t src = pos - dd;
              // the marks of the source: for the common copy (no longer than 64 bytes, not overlapping itself) the exact bits of its range
              uint64_t mm = 0; bool slow = false;
              if (src < co.mk_end) {
                if (len <= 64 && dd >= len) {
                  const unsigned sh = (unsigned)(src & 63); const uint64_t* bw = co.bits.data() + (src >> 6);
                  mm = bw[0] >> sh; if (sh) mm |= bw[1] << (64 - sh);
                  if (len < 64) mm &= (1ull << len) - 1;
                } else slow = co.has_marks(src, std::min<size_t>(len, dd));
              }
              if (!slow) {
                uint8_t* dst = o + pos; const uint8_t* sp = dst - dd; uint8_t* const end = dst + len;
                if (dd >= 8) { do { uint64_t w; memcpy(&w, sp, 8); memcpy(dst, &w, 8); sp += 8; dst += 8; } while (dst < end); }
                else if (dd == 1) memset(dst, *sp, len);
                else { do { *dst++ = *sp++; } while (dst < end); }
                if (mm) {
                  // the index entries of the whole stretch (those of unmarked bytes mean nothing, here as there) and the bits, shifted
                  uint16_t* wd = co.widx + pos; const uint16_t* ws = co.widx + src; unsigned i = 0;
                  do { uint64_t a_, b_; memcpy(&a_, ws + i, 8); memcpy(&b_, ws + i + 4, 8); memcpy(wd + i, &a_, 8); memcpy(wd + i + 4, &b_, 8); i += 8; } while (i < len);   // (dd >= len: no overlap inside the stretch; up to 7 entries beyond it are scratch)
                  const unsigned dh = (unsigned)(pos & 63); uint64_t* bd = co.bits.data() + (pos >> 6);
                  bd[0] |= mm << dh; if (dh) bd[1] |= mm >> (64 - dh);
                  co.mk_end = pos + len;
                }
              } else {
                for (unsigned i = 0; i < len; ++i) {
                  const size_t sp = src + i;
                  o[pos + i] = o[sp];
                  if (co.marked(sp)) co.mark(pos + i, co.widx[sp]);
                }
              }
              pos += len;
            }
            z.bitbuf = bb; z.bitcnt = bc; z.in = in;
            C3GZ_ROOM(0)
          }
          const bool fast = false;
          if (fast) z.refill_fast(); else z.refill_safe();
          uint32_t e = z.lit[z.bitbuf & LM];
          if (fast) {
            // up to three literal lookups (one or two bytes each, at most 11 bits each) out of one refill, as in Inflater::run
#define C3INF_LITB() { z.bitbuf >>= e_len(e); z.bitcnt -= e_len(e); const uint32_t v_ = e_val(e), two_ = e_extra(e); o[pos] = (uint8_t)v_; o[pos + 1] = (uint8_t)(v_ >> 8); pos += 1 + two_; e = z.lit[z.bitbuf & LM]; }
            if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) {
              C3INF_LITB()
              if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) {
                C3INF_LITB()
                if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) C3INF_LITB()
              }
            }
#undef C3INF_LITB
            if ((e & (K_MASK | K_VALID)) == (K_LIT | K_VALID)) continue;
          }
          if ((e & (K_MASK | K_VALID)) == (K_SUB | K_VALID)) {
            if (z.bitcnt < (unsigned)LIT_BITS) return fail();
            e = z.lit[e_val(e) + ((z.bitbuf >> LIT_BITS) & ((1u << e_extra(e)) - 1))]; z.bitbuf >>= LIT_BITS; z.bitcnt -= LIT_BITS;
          }
          if (!(e & K_VALID) || e_len(e) > z.bitcnt) return fail();
          z.bitbuf >>= e_len(e); z.bitcnt -= e_len(e);
          const uint32_t kind = e & K_MASK;
          if (kind == K_LIT) { o[pos++] = (uint8_t)e_val(e); if (e_extra(e)) o[pos++] = (uint8_t)(e_val(e) >> 8); continue; }
          if (kind == K_EOB) break;
          unsigned len = e_val(e), xb = e_extra(e);
          if (xb) { if (z.bitcnt < xb) return fail(); len += (unsigned)(z.bitbuf & ((1u << xb) - 1)); z.bitbuf >>= xb; z.bitcnt -= xb; }
          if (fast) z.refill_fast(); else z.refill_safe();
          uint32_t d = z.dist[z.bitbuf & ((1u << DIST_BITS) - 1)];
          if ((d & (K_MASK | K_VALID)) == (K_SUB | K_VALID)) {
            if (z.bitcnt < (unsigned)DIST_BITS) return fail();
            d = z.dist[e_val(d) + ((z.bitbuf >> DIST_BITS) & ((1u << e_extra(d)) - 1))]; z.bitbuf >>= DIST_BITS; z.bitcnt -= DIST_BITS;
          }
          if ((d & (K_MASK | K_VALID)) != (K_LEN | K_VALID) || e_len(d) > z.bitcnt) return fail();
          z.bitbuf >>= e_len(d); z.bitcnt -= e_len(d);
          unsigned dd = e_val(d), db = e_extra(d);
          if (db) { if (z.bitcnt < db) { z.refill_safe(); if (z.bitcnt < db) return fail(); } dd += (unsigned)(z.bitbuf & ((1u << db) - 1)); z.bitbuf >>= db; z.bitcnt -= db; }
          if ((size_t)dd <= pos - mstart) {
            const size_t src = pos - dd;
            if (src < co.mk_end && co.has_marks(src, std::min<size_t>(len, dd))) {
              // the source may hold marks: byte by byte, the marks travel with their bytes (an overlapping copy reads what it has just written,
              // marks included)
              for (unsigned i = 0; i < len; ++i) {
                const size_t sp = src + i;
                o[pos + i] = o[sp];
                if (co.marked(sp)) co.mark(pos + i, co.widx[sp]);
              }
            } else {
              uint8_t* dst = o + pos; const uint8_t* sp = dst - dd; uint8_t* const end = dst + len;
              if (dd >= 8) { do { uint64_t w; memcpy(&w, sp, 8); memcpy(dst, &w, 8); sp += 8; dst += 8; } while (dst < end); }      // (room: 600 bytes were ensured)
              else if (dd == 1) memset(dst, *sp, len);
              else { do { *dst++ = *sp++; } while (dst < end); }
            }
          } else {
            // from before the chunk: marks for the part that lies there
            if (mstart != 0 || !reach || (size_t)dd - pos > 32768) return fail();
            const long long p0 = (long long)pos - (long long)dd;
            for (unsigned i = 0; i < len; ++i) {
              const long long q = p0 + i;
              if (q < 0) { o[pos + i] = 0; co.mark(pos + i, (uint16_t)(32768 + q)); }
              else { o[pos + i] = o[q]; if (co.marked((size_t)q)) co.mark(pos + i, co.widx[(size_t)q]); }
            }
          }
          pos += len;
        }
      }
      if (z.final_block) {
        // the member's trailer; another member may follow (one stream, as for gzread)
        const unsigned dropb = z.bitcnt & 7; z.bitbuf >>= dropb; z.bitcnt -= dropb;
        const uint8_t* t = z.in - (z.bitcnt >> 3);
        if (t + 8 > fend) return fail();
        const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t isz = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        ends.push_back({pos, crc, isz});
        const uint8_t* nx = t + 8;
        const uint8_t* data = nx < fend ? gz_member_data(nx, (size_t)(fend - nx)) : nullptr;
        if (!data) { *end_bit = (size_t)(nx - base) * 8; co.len = pos; return 1; }      // (trailing bytes that are no member end the input, as zlib does)
        seek((size_t)(data - base) * 8);
        mstart = pos; reach = false;
      }
    }
#undef C3GZ_ROOM
  }

  // first bit offset in [from, to) that opens a dynamic block which decodes to its end with a well-formed block header behind it;
  // (size_t)-1 when there is none.  tmp: scratch
  size_t find_block(size_t from, size_t to, ChunkBuf& tmp, std::vector<MemberEnd>& tends) {
    const size_t nbits = (size_t)(fend - base) * 8;
    if (to > nbits) to = nbits;
    for (size_t bit = from; bit + 64 < nbits && bit < to; ++bit) {
      // BFINAL = 0, BTYPE = 2 (bits 1-2 = 10b, LSB first): the low three bits read 100b; HLIT <= 29; HDIST <= 29
      const size_t by = bit >> 3; const unsigned sh = (unsigned)(bit & 7);
      uint32_t w; memcpy(&w, base + by, 4);
      w >>= sh;
      if ((w & 7u) != 4u || ((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
      // the header proper: the three codes must be complete (most candidates end at the 19 code-length code lengths)
      seek(bit);
      if (!z.read_block_header()) continue;
      // ... the block must decode to its end-of-block symbol, and a well-formed header must follow
      seek(bit);
      if (tmp.len) { for (size_t i = 0; i <= tmp.len / 64 + 1 && i < tmp.bits.size(); ++i) tmp.bits[i] = 0; }
      tmp.len = 0; tmp.mk_end = 0; tends.clear();
      size_t eb = 0;
      const int rc = run(tmp, bit + 1, &eb, false, tends, (size_t)1 << 24);       // exactly one block (stops at the next boundary)
      if (rc < 0) continue;
      if (rc == 0) { seek(eb); if (!z.read_block_header()) continue; }
      return bit;
    }
    return (size_t)-1;
  }
};

// One chunk of a round
struct ParChunk {
  size_t start = (size_t)-1, end = 0;      // bit offsets (start: -1 = no block start found: the chunk before decodes through)
  bool known = false;                      // starts at the first block of a member
  int rc = 0;                              // result of the decode: 0 stopped at a boundary, 1 end of the input, -1 error
  ChunkBuf cb;                             // cb.out: the chunk's cb.len bytes behind `head` free ones (the vector itself is longer)
  std::vector<MemberEnd> ends;
  std::vector<uint8_t> window;             // the 32 KiB before the chunk (resolved), oldest first, 32768 bytes (front padded)
  std::vector<uint32_t> seg_crc; std::vector<size_t> seg_len;     // CRC-32 of every stretch between member ends (ends.size() + 1 stretches)
};

// The whole parallel decoder over a mapped file.  next_round() produces the next stretch of output bytes (in order) into `chunks` --
// false at the end of the input or on an error (bad = true).
struct GzPar {
  const uint8_t* map = nullptr; size_t size = 0;
  int T = 4; size_t chunk = (size_t)1 << 20;
  int per_round = 0;                       // chunks per round (0: one per thread)
  size_t head = 0;                         // bytes left free in front of every chunk's output (the reader moves a partial line there and takes the buffer as it is)
  size_t next_bit = 0; bool next_known = true; bool started = false, done = false, bad = false;
  std::vector<uint8_t> window;             // last 32 KiB of everything produced so far (front padded with zeros)
  uint32_t run_crc = 0; uint64_t run_len = 0;     // the member that is still open: CRC-32 and length so far
  int members = 0;
  std::function<void(std::function<void()>)> with_slot;       // runs a worker body under the process-wide core limiter (c3_io.cpp: CpuSlot)
  // Used output buffers come back in through here (c3_io.cpp: the parser hands every buffer back once it has read it) and the index planes stay
  // with the decoder: their pages are mapped already.  Without this every chunk allocates 4x its size twice over and frees it on another
  // thread -- page faults, page zeroing and munmap for every 2 MiB of input, a fifth of the one parser thread's time.
  std::function<void(std::vector<char>&)> take_buf;
  std::mutex wmu; std::vector<std::pair<uint16_t*, size_t>> wpool;
  ~GzPar() { for (auto& w : wpool) free(w.first); }
  std::vector<ParChunk> chunks;

  bool open() {
    const uint8_t* d = gz_member_data(map, size);
    if (!d) return false;
    next_bit = (size_t)(d - map) * 8; next_known = true; started = true;
    window.assign(32768, 0);
    run_crc = (uint32_t)crc32(0L, Z_NULL, 0); run_len = 0;
    return true;
  }

  // body(0 .. n-1) on min(T, n) threads that take the indices as they get free (a round may hold more chunks than threads: under
  // contention -- the writers and the GPU's host threads share the cores -- the slow thread of a static split held the round up)
  template <class F> void parallel(int n, F body) {
    std::atomic<int> next{0};
    auto worker = [&]() {
      auto loop = [&]() { for (int j; (j = next.fetch_add(1)) < n;) body(j); };
      if (with_slot) with_slot(loop); else loop();
    };
    const int W = std::min(T, n);
    std::vector<std::thread> th;
    for (int k = 1; k < W; ++k) th.emplace_back(worker);
    if (W > 0) worker();
    for (auto& t : th) t.join();
  }

  bool next_round() {
    if (done || bad) return false;
    const size_t nbits = size * 8;
    const size_t b0 = next_bit >> 3;
    int n = per_round > 0 ? per_round : T;
    chunks.clear(); chunks.resize((size_t)n);
    chunks[0].start = next_bit; chunks[0].known = next_known;
#ifdef C3_GZPAR_PROF
    const double T0 = now_();
#endif
    // ---- 1. block starts of the chunks 1 .. n-1, and of the chunk the NEXT round begins with
    std::vector<size_t> starts((size_t)n + 1, (size_t)-1);
    starts[0] = next_bit;
    parallel(n, [&](int j) {
      const size_t from = (b0 + (size_t)(j + 1) * chunk) * 8;
      if (from >= nbits) return;
      Dec16 d; d.base = map; d.fend = map + size;
      ChunkBuf tmp; std::vector<MemberEnd> te;
      // (the last one -- where the next round starts -- searches on until it finds one: a round must end at a block boundary)
      const size_t to = j + 1 < n ? from + chunk * 8 : nbits;
      starts[(size_t)j + 1] = d.find_block(from, to, tmp, te);
    });
    for (int j = 1; j < n; ++j) chunks[(size_t)j].start = starts[(size_t)j];
    const size_t round_end = starts[(size_t)n];            // (size_t)-1: this round runs to the end of the input
#ifdef C3_GZPAR_PROF
    const double T1 = now_();
#endif
    // ---- 2. decode every chunk that has a start up to the start of the next one that has
    auto target_of = [&](int j) { for (int k = j + 1; k < n; ++k) if (chunks[(size_t)k].start != (size_t)-1) return chunks[(size_t)k].start; return round_end; };
    parallel(n, [&](int j) {
      ParChunk& c = chunks[(size_t)j];
      if (c.start == (size_t)-1) return;
      Dec16 d; d.base = map; d.fend = map + size;
      d.seek(c.start);
      c.cb.head = head;
      if (take_buf) take_buf(c.cb.out);
      { std::lock_guard<std::mutex> lk(wmu); if (!wpool.empty()) { c.cb.widx = wpool.back().first; c.cb.wcap = wpool.back().second; wpool.pop_back(); } }
      c.cb.grow(chunk * 4);
      c.rc = d.run(c.cb, target_of(j), &c.end, c.known, c.ends, (size_t)1 << 30);
    });
#ifdef C3_GZPAR_PROF
    const double T2 = now_();
#endif
    // ---- 3. the chain: every chunk must begin where the one before stopped.  One that does not was no block boundary of the real stream
    // (a false positive of step 1): it is dropped and the chunk before decodes on through it (serially: this is the rare path)
    size_t rend = round_end;
    int cur = 0;
    for (;;) {
      ParChunk& p = chunks[(size_t)cur];
      if (p.rc < 0) { bad = true; return false; }
      if (p.rc == 1) { for (int k = cur + 1; k < n; ++k) chunks[(size_t)k].start = (size_t)-1; break; }      // the input ended inside p
      int nx = -1;
      for (int k = cur + 1; k < n; ++k) if (chunks[(size_t)k].start != (size_t)-1) { nx = k; break; }
      const size_t want = nx >= 0 ? chunks[(size_t)nx].start : rend;
      if (want == (size_t)-1 || p.end < want) { bad = true; return false; }          // (cannot happen: p stops at the first boundary at or beyond its target)
      if (p.end == want) { if (nx < 0) break; cur = nx; continue; }
      if (nx < 0) { rend = p.end; break; }                                           // the next round's start was the false positive: it begins where p stopped
      chunks[(size_t)nx].start = (size_t)-1; chunks[(size_t)nx].cb = ChunkBuf(); chunks[(size_t)nx].ends.clear();
      size_t tgt = rend;
      for (int k = nx + 1; k < n; ++k) if (chunks[(size_t)k].start != (size_t)-1) { tgt = chunks[(size_t)k].start; break; }
      if (tgt == (size_t)-1 || p.end < tgt) {
        Dec16 d; d.base = map; d.fend = map + size;
        d.seek(p.end);
        p.rc = d.run(p.cb, tgt, &p.end, p.known, p.ends, (size_t)1 << 30);
      }
    }
    {
      const ParChunk& p = chunks[(size_t)cur];
      if (p.rc == 1) done = true;
      else { next_bit = p.end; next_known = false; }
    }
#ifdef C3_GZPAR_PROF
    const double T3 = now_();
#endif
    // ---- 4. windows, in order (only the last 32 KiB of every chunk are resolved here), then markers -> bytes and CRCs in parallel
    std::vector<uint8_t> w = window;
    for (int j = 0; j < n; ++j) {
      ParChunk& c = chunks[(size_t)j];
      if (c.start == (size_t)-1) continue;
      c.window = w;
      // this chunk's last 32 KiB: its bytes, the marks among them resolved from the window before it
      const size_t m = c.cb.len, keep = std::min<size_t>(m, 32768);
      std::vector<uint8_t> nw(32768, 0);
      if (keep < 32768) memcpy(nw.data(), w.data() + keep, 32768 - keep);
      memcpy(nw.data() + 32768 - keep, c.cb.out.data() + c.cb.head + (m - keep), keep);
      c.cb.for_marks(m - keep, m, [&](size_t p, uint16_t wi) { nw[32768 - keep + (p - (m - keep))] = w[wi & 0x7fffu]; });
      w.swap(nw);
    }
    window = w;
    parallel(n, [&](int j) {
      ParChunk& c = chunks[(size_t)j];
      if (c.start == (size_t)-1) return;
      const size_t m = c.cb.len;
      char* o = c.cb.out.data() + head; const uint8_t* wv = c.window.data();            // (the buffer keeps its size: the chunk is cb.len bytes behind `head`)
      c.cb.for_marks(0, m, [&](size_t p, uint16_t wi) { o[p] = (char)wv[wi & 0x7fffu]; });
      { size_t wc = 0; uint16_t* wp = c.cb.release_widx(&wc); std::lock_guard<std::mutex> lk(wmu); if (wp && wpool.size() < 2 * (size_t)n) wpool.emplace_back(wp, wc); else free(wp); }
      c.cb.drop_marks();
      size_t at = 0;
      for (size_t k = 0; k <= c.ends.size(); ++k) {
        const size_t to = k < c.ends.size() ? c.ends[k].off : m;
        uint32_t cr = (uint32_t)crc32(0L, Z_NULL, 0);
        cr = c3crc::crc32_fast(cr, (const unsigned char*)o + at, to - at);
        c.seg_crc.push_back(cr); c.seg_len.push_back(to - at);
        at = to;
      }
    });
#ifdef C3_GZPAR_PROF
    const double T4 = now_();
#endif
    // ---- 5. every member that ended in this round: CRC-32 and length against its trailer
    for (int j = 0; j < n; ++j) {
      ParChunk& c = chunks[(size_t)j];
      if (c.start == (size_t)-1) continue;
      for (size_t k = 0; k <= c.ends.size(); ++k) {
        run_crc = (uint32_t)crc32_combine(run_crc, c.seg_crc[k], (z_off_t)c.seg_len[k]); run_len += c.seg_len[k];
        if (k < c.ends.size()) {
          if (run_crc != c.ends[k].crc || (uint32_t)run_len != c.ends[k].isize) { bad = true; return false; }
          run_crc = (uint32_t)crc32(0L, Z_NULL, 0); run_len = 0; ++members;
        }
      }
    }
#ifdef C3_GZPAR_PROF
    fprintf(stderr, "  find %.4f decode %.4f chain %.4f convert %.4f verify %.4f\n", T1 - T0, T2 - T1, T3 - T2, T4 - T3, now_() - T4);
#endif
    if (done && run_len != 0) { bad = true; return false; }      // the input ended inside a member
    return true;
  }
};

}  // namespace c3inf
