// k_polish_band.h -- the BANDED rows of k_window (included by k_polish.hip; round 3).
//
// A layer of ~500 bases against a window graph never leaves a narrow diagonal (measured on the synthetic configs: +-8 columns
// around the column predicted from the backbone position), but the unbanded rows fill all Q + 1 columns.  Here a row
// computes only BW = 64 * CB columns [lo(r), lo(r) + BW) around its expected column
//     c(r) = (bb(r) - begin + 1) * Q / (end - begin + 1),   bb(r) = highest backbone node among the rows <= r,
// lane owns band offsets CB*lane .. CB*lane + CB - 1 (column = lo(r) + offset), and the band slides right as the rows go down.
//
// EXACTNESS.  Cells outside the band count as -infinity, so the banded optimum S_b is the best score of the
// paths that stay inside.  While the rows run, a certificate bound is accumulated: an upper bound of the score of EVERY path
// with a cell outside the band.  Such a path leaves the band for the first time at a band-edge cell X whose banded value
// bounds its prefix, and the rest is bounded by the moves that are left (every remaining column scores at most `ups` = the
// best substitution score; a path that is behind the rows that remain must pay gaps for the columns it cannot match):
//   right exit (any row whose band ends before column Q; the path moves right out of the last cell):
//       H[r][hi] + ups * (Q - hi)
//   left exit (rows with a successor whose band starts further right; cells lo .. lo + leftspan - 1):
//       H[r][j] + ups * min(rem, nb) + gap * max(0, rem - nb),  rem = Q - j, nb = aligned blocks behind the row's block
//       (a path visits at most one node per aligned block, so at most nb more diagonal moves; the other columns are gaps)
//   rows hanging off the virtual start row whose band does not start at column 0: the same bound from cell (0, 0).
// If bound < S_b (strictly), the unbanded traceback path lies inside the band, the banded values along it equal the unbanded
// ones and every tie is broken the same way -- the result is bit-identical to the full matrix, which the oracle keeps
// filling.  A layer whose certificate fails is simply redone by the unbanded rows.
//
// Row loop.  hcur = row r-1 in ITS band coordinates.  The fast row (one predecessor = the row above, band shift 0 or 1,
// nothing to keep) is straight-line code, one variant per shift: neighbours from registers with one DPP move, substitution
// bytes of its columns prefetched from an LDS table during the row before, 2 bits of direction per cell appended to a
// per-lane accumulator that goes to memory once per RPW rows (ONE vector store per 4-8 rows: the row loop is bound by
// instruction issue, and a store costs the memory pipeline as much as sixteen ALU instructions cost the SIMD).  Every other
// predecessor comes from an LDS ring of the last WB_RING rows or, further back, from 16-bit rows in global memory.  The
// band-edge cells of a row (first 2*CB, last) are parked in LDS by three lanes and turned into the certificate bound
// once per 64 rows by the lane that holds the row's descriptor.
#pragma once

#define WB_PADL 4            /* shorts in front of cell 0 of a ring slot: [0..1] lo, [3] = -inf (offset -1) */
#define WB_PADR 28           /* -inf shorts behind the last cell: a predecessor is read at offsets up to BW - 1 + WB_MAXSHIFT + CB */
#define WB_MAXSHIFT 24       /* lo(row) - lo(predecessor) beyond this: the layer is not banded */
#ifndef WB_RING
#define WB_RING 4            /* rows kept in the LDS ring (power of two) */
#endif
#define WB_EROW 12           /* shorts per row of band-edge cells parked for the certificate: [0, 2*CB) first cells, [8, 8+CB) last lane */
#define WB_EROWS 32          /* rows parked before the certificate bound of those rows is evaluated */
__host__ __device__ __forceinline__ int wb_slot_shorts(int cb) { return WB_PADL + 64 * cb + WB_PADR; }
// LDS of the banded rows behind the row-type bitmasks, in dwords: substitution table, ring, edge cells of 64 rows
__host__ __device__ __forceinline__ int wb_lds_dwords(int Q, int cb) { return ((Q + 2) & ~1) + WB_RING * wb_slot_shorts(cb) / 2 + WB_EROWS * WB_EROW / 2; }
__device__ __forceinline__ int wb_left(int cb) { return cb == 2 ? 70 : cb == 3 ? 105 : 140; }     // columns left of the centre (the rest, 57 / 86 / 115, right)
__host__ __device__ __forceinline__ int wb_rpw(int cb) { return cb == 2 ? 8 : 4; }                 // rows per packed direction word (2*cb bits per row and lane)
__host__ __device__ __forceinline__ int wb_bit0(int cb) { return 32 - 2 * cb * wb_rpw(cb); }       // first used bit of a direction word (cb == 3: 8)

// Direction stream of a banded layer in the slot's D scratch, as dwords:
//   word g (rows g*RPW + 1 .. g*RPW + RPW) of lane l at  [g * 64 + l]:  row k of the word, cell cc of the lane -> 2 bits at
//   bit0 + 2*CB*k + 2*cc  (3 diagonal, 2 vertical, 1 horizontal)
//   predecessor index bytes of the rows with several predecessors (cell cc -> byte cc) at [(G + 1 + r) * 64 + l], G = words
__device__ __forceinline__ int wb_words(int R, int cb) { return (R + wb_rpw(cb) - 1) / wb_rpw(cb); }

// descriptors of the banded rows, built in ONE pass over the rows (64 per step; no second pass, no side arrays: every
// vector memory instruction of the graph phases costs the kernel about as much as twenty ALU instructions).
//   x = base*8 | np<<8 | ovf<<16 | far<<17 | isend<<18 | two<<19 | fast<<20 | shift<<22 (fast rows: 0/1) | wr<<23 | virt<<24
//   y, z = predecessor rows (as the unbanded descriptors)
//   w = lo | dist<<10 | bidx<<18
//   far:  some successor is more than WB_RING rows ahead -> the H row also goes to global memory
//   wr:   some successor will read the row from the LDS ring (i.e. is not the fast row right below)
//   virt: no masked predecessor (the row hangs off the virtual start row)
//   dist: rows to the farthest successor (0 = none).  lo is non-decreasing along the rows, so the band of that successor starts
//         furthest right: leftspan(r) = lo(r + dist) - lo(r) = sum of the shifts of rows r+1 .. r+dist, taken from the shift
//         bitmasks when the certificate is evaluated -- the row's first `leftspan` cells are where a path can leave the band
//   bidx: number of aligned blocks among rows 1..r
// d0 / d1: bit r-1 = bit 0 / 1 of lo(r) - lo(r-1).  Returns 0 when the layer cannot be banded.
#define WB_W_LO(w) ((int)((w) & 0x3ffu))
#define WB_W_DIST(w) ((int)(((w) >> 10) & 0xffu))
#define WB_W_BIDX(w) ((int)((w) >> 18))
__device__ int win_build_desc_band(WCtx& c, int R, int Q, int begin, int end, int blen, int CB, int lane,
                                   unsigned long long* m2, unsigned long long* ma, unsigned long long* d0, unsigned long long* d1, int* nblocks) {
  const int BW = 64 * CB, WLf = wb_left(CB), span = end - begin + 1, lomax = Q + 1 - BW;
  if (R >= 16000 || Q > 1000) return 0;
  int bbc = begin - 1, loc = 0, nbc = 0, gprev = -1, bad = 0;
#ifndef WB_DU
#define WB_DU 1
#endif
  constexpr int DU = WB_DU;                                               // 64-row chunks per iteration: the loads of one level go out together
  for (int rb = 1; rb <= R; rb += 64 * DU) {
    int v[DU], nin[DU], nout[DU], gr[DU], pe[DU][4], se[DU][4];
    unsigned vb[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) { const int r = rb + 64 * u + lane; v[u] = r <= R ? c.rows()[r] : -1; }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const bool live = v[u] >= 0;
      nin[u] = live ? c.n_in()[v[u]] : 0; nout[u] = live ? c.n_out()[v[u]] : 0;
      vb[u] = live ? (unsigned)c.base()[v[u]] & 3u : 0u; gr[u] = live ? c.grp()[v[u]] : -2;
    }
    int kmax = 0;
#pragma unroll
    for (int u = 0; u < DU; ++u) kmax = max(kmax, max(nin[u], nout[u]));
    kmax = min(wave_max(kmax), 4);
    // first four in- / out-edges of every row (slot-major adjacency: independent loads), then the rows of their nodes
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int u = 0; u < DU; ++u) { pe[u][k] = -1; se[u][k] = -1; }
      if (k < kmax) {
#pragma unroll
        for (int u = 0; u < DU; ++u) {
          if (k < nin[u]) pe[u][k] = c.in_from()[EI(v[u], k)];
          if (k < nout[u]) se[u][k] = c.out_to()[EI(v[u], k)];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < kmax) {
#pragma unroll
        for (int u = 0; u < DU; ++u) {
          if (pe[u][k] >= 0) pe[u][k] = c.rowof()[pe[u][k]];
          if (se[u][k] >= 0) se[u][k] = c.rowof()[se[u][k]];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int r0 = rb + 64 * u, r = r0 + lane;
      if (r0 > R) break;
      const bool live = v[u] >= 0;
      // backbone position reached so far -> band start (non-decreasing)
      const int bbs = max(wave_scan_max(live && v[u] < blen ? v[u] : -1), bbc);
      bbc = wave_bcast(bbs, 63);
      // (32-bit: 0 <= bbs - begin + 1 <= span <= Ncap and Q <= 1000, the product stays below 2^31 -- the 64-bit division was ~150 vector
      // instructions per 64 rows, as much as three fast rows)
      const int cen = (int)(((unsigned)(bbs - begin + 1) * (unsigned)Q + (unsigned)(span / 2)) / (unsigned)span);
      const int lo = min(max(cen - WLf, 0), lomax);
      const int lop = wave_shr1(lo, loc);                               // lo of the row above
      loc = wave_bcast(lo, 63);
      const int dl = live ? lo - lop : 0;
      if (dl > 3) bad = 1;
      // aligned blocks: runs of equal group ids in the row order
      const int grp_ = wave_shr1(gr[u], gprev);
      gprev = wave_bcast(gr[u], 63);
      const unsigned long long bs = __ballot(live && gr[u] != grp_);
      const int bidx = nbc + __popcll(bs & ((2ull << lane) - 1));
      nbc += __popcll(bs);
      bool two = false, adj = false;
      unsigned p[4] = {0, 0, 0, 0}, has = 0, far = 0, fast = 0, other = 0, next = 0, virt = 0, ovf = 0;
      int np = 0, dist = 0;
      if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (pe[u][k] >= 0) { p[np < 4 ? np : 3] = (unsigned)pe[u][k]; ++np; }
        for (int k = 4; k < nin[u]; ++k) { const int pr = c.rowof()[c.in_from()[EI(v[u], k)]]; if (pr >= 0) { if (np < 4) p[np] = (unsigned)pr; ++np; } }     // (masked-out predecessors in the first slots: a later edge can still be one of the first four ROWS)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (se[u][k] >= 0) { has = 1; dist = max(dist, se[u][k] - r); if (se[u][k] == r + 1) next = 1; else other = 1; }
        for (int k = 4; k < nout[u]; ++k) {
          const int sr = c.rowof()[c.out_to()[EI(v[u], k)]];
          if (sr >= 0) { has = 1; dist = max(dist, sr - r); if (sr == r + 1) next = 1; else other = 1; }
        }
        far = dist > WB_RING;
        if (dist > 64) bad = 1;                                          // (the certificate reads the shifts of at most 64 rows)
        ovf = np > 4; virt = np == 0;
        if (np == 0) np = 1;
        two = np == 1;
        adj = two && (int)p[0] == r - 1;
        fast = adj && !far && has && dl <= 1;
#ifdef C3_EXP_NOFAST
        fast = 0;
#endif
      }
      // wr: a successor other than the row below, or the row below is not a fast row (its kind sits one lane up; the last
      // lane of a step does not see it and writes the ring to be safe)
      const unsigned fastn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)fast, 0x130, 0xf, 0xf, false);       // wave_shl:1
      if (live) {
        const unsigned wr = other | (next & (fastn ^ 1u));
        uint4 d; d.x = vb[u] * 8u | ((unsigned)min(np, 255) << 8) | (ovf << 16) | (far << 17) | ((has ^ 1u) << 18) | ((unsigned)two << 19) | (fast << 20)
                       | ((unsigned)(fast ? dl : 0) << 22) | (wr << 23) | (virt << 24);
        d.y = p[0] | (p[1] << 16); d.z = p[2] | (p[3] << 16); d.w = (unsigned)lo | ((unsigned)min(dist, 255) << 10) | ((unsigned)bidx << 18);
        c.rdesc[r] = d;
      }
      const unsigned long long b2 = __ballot(two), ba = __ballot(adj), bd0 = __ballot(dl & 1), bd1 = __ballot((dl & 2) != 0);
      if (lane == 0) { m2[r0 >> 6] = b2; ma[r0 >> 6] = ba; d0[r0 >> 6] = bd0; d1[r0 >> 6] = bd1; }
    }
  }
  *nblocks = nbc;
  bad = __ballot(bad) != 0;
  WSYNC();
  return !bad;
}

// three lanes (0, 1, 63) park their cells of the row in LDS: EXEC is narrowed by hand (the compiler's version of the same
// `if` costs two compares and an and/or dance per row)
template <int CB>
__device__ __forceinline__ void wb_park_edge(unsigned eaddr, const int (&h)[CB]) {
  const unsigned long long emask = 0x8000000000000003ull;
  if (CB == 2)
    asm volatile("s_mov_b64 exec, %0\n\tds_write_b16 %1, %2\n\tds_write_b16 %1, %3 offset:2\n\ts_mov_b64 exec, -1"
                 :: "s"(emask), "v"(eaddr), "v"(h[0]), "v"(h[CB > 1 ? 1 : 0]) : "memory");
  else if (CB == 3)
    asm volatile("s_mov_b64 exec, %0\n\tds_write_b16 %1, %2\n\tds_write_b16 %1, %3 offset:2\n\tds_write_b16 %1, %4 offset:4\n\ts_mov_b64 exec, -1"
                 :: "s"(emask), "v"(eaddr), "v"(h[0]), "v"(h[CB > 1 ? 1 : 0]), "v"(h[CB > 2 ? 2 : 0]) : "memory");
  else
    asm volatile("s_mov_b64 exec, %0\n\tds_write_b16 %1, %2\n\tds_write_b16 %1, %3 offset:2\n\tds_write_b16 %1, %4 offset:4\n\tds_write_b16 %1, %5 offset:6\n\ts_mov_b64 exec, -1"
                 :: "s"(emask), "v"(eaddr), "v"(h[0]), "v"(h[CB > 1 ? 1 : 0]), "v"(h[CB > 2 ? 2 : 0]), "v"(h[CB > 3 ? 3 : 0]) : "memory");
}

// inclusive max-scan of signed 16-bit keys in the low halves (the high halves are don't-care on input, zero on output): v_max_i16 takes DPP
// like every VOP2, so the scan needs no sign extension first.  Hand-written because the hazard recogniser does not look inside inline
// assembly: a DPP instruction reading a register the previous vector instruction wrote needs two wait states (s_nop 1).
__device__ __forceinline__ int wb_scan_max16(int x) {
  asm volatile("s_nop 1\n\t"
               "v_max_i16_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i16_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i16_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i16_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i16_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
               "v_max_i16_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
               : "+v"(x));       // (the compiler adds the wait states in front of whatever reads x next: it knows the statement wrote it)
  return x;
}

// All banded rows of one layer.  Returns 0, 1 (a predecessor's band lies too far left: not banded) or -1 (scratch too small).
// Results in lob[0..2]: the certificate bound, the best end-row score H[r][Q] (INT32_MIN: none inside the band), its row.
template <int CB, bool SECOND>
__device__ __attribute__((noinline)) int win_rows_band(int* cI, int* cE, int32_t* cH, uint8_t* cD, uint4* crdesc, int cK, int cn, int cNcap, long long chcap,
                                                       int mt_, int mm_, int g_, const uint32_t* pk_, int qbeg_, int Q_, int R_, unsigned long long* dbg_, int lds_off_, int nblocks_) {
  WCtx c;
  c.I = uni_ptr(cI); c.E = uni_ptr(cE); c.B8 = nullptr; c.score = nullptr; c.H = uni_ptr(cH); c.D = uni_ptr(cD);
  c.rdesc = uni_ptr(crdesc); c.K = uni32(cK); c.n = uni32(cn); c.Ncap = uni32(cNcap);
  c.hcap = ((long long)uni32((int)(chcap >> 32)) << 32) | (unsigned)uni32((int)chcap);
  const uint32_t* pk = uni_ptr(pk_);
  unsigned long long* dbg = uni_ptr(dbg_);
  const int qbeg = uni32(qbeg_), Q = uni32(Q_), R = uni32(R_), NB = uni32(nblocks_);
  const int lane = wave_lane();
  extern __shared__ int lds_dyn[];
  constexpr int BW = 64 * CB, SLOT = WB_PADL + BW + WB_PADR /* shorts */, HS = BW + 8 /* shorts of a global H row: [0..1] lo, [4..] cells */;
  constexpr int RPW = CB == 2 ? 8 : 4, BPR = 2 * CB, BIT0 = 32 - BPR * RPW;
  unsigned* tbl = (unsigned*)lds_dyn + uni32(lds_off_);                      // [Q + 1] substitution bytes per column
  unsigned short* ring = (unsigned short*)(tbl + ((Q + 2) & ~1));
  unsigned short* ebuf = ring + WB_RING * SLOT;                              // [WB_EROWS][WB_EROW] band-edge cells of the current 32 rows
  const unsigned long long* d0bits = (const unsigned long long*)(lds_dyn + uni32(lds_off_) / 2);      // the four bitmask arrays fill [0, lds_off)
  const unsigned long long* d1bits = (const unsigned long long*)(lds_dyn + uni32(lds_off_) / 4 * 3);
  struct { int pol_match, pol_mismatch, pol_gap; } P = {uni32(mt_), uni32(mm_), uni32(g_)};
  const int K = c.K;
  const int G = (R + RPW - 1) / RPW;
  // direction words + predecessor-index rows in the D scratch (hcap bytes per slot), far H rows in the H scratch (4 * hcap bytes)
  if ((long long)(G + 2 + R) * 256 > c.hcap || (long long)(R + 1) * HS * 2 + 64 > c.hcap * 4 || R >= 65535) return -1;
  unsigned short* const H16 = (unsigned short*)c.H;
  unsigned* const DW = (unsigned*)c.D;
  unsigned* const PX = DW + (size_t)(G + 1) * 64;
  const int ups = max(P.pol_match, P.pol_mismatch), gap = P.pol_gap;
  const int mt3 = P.pol_match * 4 + 3, mm3 = P.pol_mismatch * 4 + 3, g4 = P.pol_gap * 4;
  const int mm3x4 = (mm3 & 255) * 0x01010101;
  for (int j = lane; j <= Q; j += 64) {
    const int qc = j >= 1 ? c3_code_at(pk, qbeg + j - 1) : 7;
    tbl[j] = qc < 4 ? (mm3x4 & ~(255 << (8 * qc))) | ((mt3 & 255) << (8 * qc)) : mm3x4;
  }
  if (lane == 0) tbl[Q + 1] = mm3x4;                                        // (prefetched past the last column, never used)
  for (int i = lane; i < WB_RING * (WB_PADR + 1); i += 64) {               // the -inf cells around every ring slot
    const int sl = i / (WB_PADR + 1), k = i % (WB_PADR + 1);
    ring[sl * SLOT + (k == 0 ? WB_PADL - 1 : WB_PADL + BW + k - 1)] = (unsigned short)W_NEG16;
  }
  int hcur[CB], g41[CB];
#pragma unroll
  for (int cc = 0; cc < CB; ++cc) {
    const int b = lane * CB + cc;
    hcur[cc] = b * g4;                                                      // virtual row 0, lo = 0
    g41[cc] = b * g4 + 1;                                                   // horizontal candidate = 4 * (best y + g * offset) + 1 (the band start cancels inside a row)
  }
  int cV = g4 + 2;
  VREG(cV);
  WSYNC();
  const unsigned* tp = tbl + CB * lane;
  int lo = 0;                                                               // band start of the row in hcur
  unsigned tn[CB + 1];                                                      // substitution bytes of columns lo + CB*lane + 0..CB (the next row needs CB of them)
#pragma unroll
  for (int k = 0; k <= CB; ++k) tn[k] = tp[k];
  unsigned dacc = 0;                                                        // direction bits of the rows of the current word
  unsigned dwoff = (unsigned)lane;                                          // dword index of this lane's next direction word
  const unsigned ebase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)ebuf + 2u * (lane < 2 ? (unsigned)(lane * CB) : 8u);
  int best = (BW - 1) * gap + ups * (Q - BW + 1);                           // certificate: right exit of the virtual row (its band ends at BW - 1 < Q)
  int ebs = INT32_MIN, ebr = INT32_MAX / 2, bandbad = 0;                    // best end row seen by this lane
#ifdef C3_PHASE_PROF
  unsigned long long pf_fast = 0, pf_d0 = 0, pf_d1 = 0, pf_c2 = 0, pf_c3 = 0, pf_c4 = 0;
#endif

  // one row ends: horizontal gap inside the band (in-lane prefix + one cross-lane max-scan over y = H - g * offset), new
  // hcur, 2 direction bits per cell on top of the accumulator.  PIDX: the predecessor that set the cell sits in bits 16..21
  // of key (rows with several predecessors) and goes to the row's index bytes.
  // Trimmed by hand in round 5 (-3 % of k_window at cfg2, -2 % at cfg3 / cfg4; the compiler's version of the fast row was ~56 vector
  // instructions, this is ~43): the first term starts the in-lane maximum (no -infinity to max against), the scan runs on the 16-bit
  // keys themselves (v_max_i16 takes DPP: no sign extension), the direction bits of the lane's cells are gathered with and / shift-or
  // and enter the accumulator through ONE v_alignbit (it was shift, and, shift, or per cell and a shift + or for the word), and the LDS
  // address of the parked edge cells is a register that counts up (it was a 64-bit multiply-add per row).  What is left of the gap to
  // a hand-written row are five to eight register copies where the three row kinds merge; giving the runs of fast rows a loop of
  // their own removes them there (36 instructions per row) and doubles them everywhere else: 28.5 -> 30.7 ms, not shipped.
#define WB_ROW_TAIL(PIDX)                                                                                        \
    {                                                                                                            \
      int yv[CB];                                                                                                \
      _Pragma("unroll") for (int cc = 0; cc < CB; ++cc) yv[cc] = key[cc] - g41[cc];                             \
      int run = yv[0];                                                                                           \
      _Pragma("unroll") for (int cc = 1; cc < CB; ++cc) run = max16(run, yv[cc]);                               \
      int ex = wave_shr1(wb_scan_max16(run), W_NEG16);                                                           \
      unsigned pidx = 0;                                                                                         \
      int k2[CB];                                                                                                \
      _Pragma("unroll") for (int cc = 0; cc < CB; ++cc) {                                                       \
        k2[cc] = max16(key[cc], (ex & ~3) + g41[cc]);                                                            \
        if (cc + 1 < CB) ex = max16(ex, yv[cc]);                                                                 \
        hcur[cc] = k2[cc] & ~3;                                                                                  \
        if (PIDX) pidx |= (((unsigned)key[cc] >> 16) & 63u) << (8 * cc);                                         \
      }                                                                                                          \
      unsigned tb = (unsigned)k2[CB - 1];                                                                        \
      _Pragma("unroll") for (int cc = CB - 2; cc >= 0; --cc) tb = (tb << 2) | ((unsigned)k2[cc] & 3u);          \
      dacc = __builtin_amdgcn_alignbit(tb, dacc, BPR);                                                           \
      if (PIDX) GP(unsigned, PX)[(unsigned)r * 64u + (unsigned)lane] = pidx;                                     \
      wb_park_edge<CB>(eaddr, hcur);                                                                             \
    }
#define WB_ROW_TAIL_FAST() WB_ROW_TAIL(false)
#define WB_RING_WRITE()                                                                                          \
    {                                                                                                            \
      unsigned short* sp_ = ring + (r & (WB_RING - 1)) * SLOT;                                                   \
      *(int*)sp_ = lo;                                                                                           \
      _Pragma("unroll") for (int cc = 0; cc < CB; ++cc) sp_[WB_PADL + CB * lane + cc] = (unsigned short)hcur[cc]; \
    }

  for (int rb = 1; rb <= R; rb += 64) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 dv = GP(const u32x4, c.rdesc)[min(rb + lane, R)];
    uint4 dblk = make_uint4(dv.x, dv.y, dv.z, dv.w);
    asm volatile("" : "+v"(dblk.x), "+v"(dblk.y), "+v"(dblk.z), "+v"(dblk.w));
    const int cnt = min(64, R - rb + 1);
    for (int half_ = 0; half_ < 64 && half_ < cnt; half_ += WB_EROWS) {
    const int hend_ = min(half_ + WB_EROWS, cnt);
    unsigned eaddr = ebase - 2u * WB_EROW;                                  // LDS address of this lane's parked edge cells of the row (slot li & 31): counts up
    for (int li = half_; li < hend_; ++li) {
      const int r = rb + li;
      const unsigned dx = (unsigned)__builtin_amdgcn_readlane((int)dblk.x, li);
      eaddr += 2u * WB_EROW;
      int key[CB];
      if (dx & (1u << 20)) {
        // FAST ROW: the row above, band shift 0 or 1
        const int vb8 = (int)(dx & 24u);
#ifdef C3_PHASE_PROF
        ++pf_fast;
#endif
        if (dx & (1u << 22)) {
          int tv[CB];
#pragma unroll
          for (int cc = 0; cc < CB; ++cc) tv[cc] = __builtin_amdgcn_sbfe((int)tn[cc + 1], vb8, 8);
          lo += 1;
#pragma unroll
          for (int k = 0; k <= CB; ++k) tn[k] = tp[lo + k];
          const int hnext = wave_shl1(hcur[0], W_NEG16);                    // offset CB*lane + CB of the row above
#pragma unroll
          for (int cc = 0; cc < CB; ++cc) {
            const int hv = cc + 1 < CB ? hcur[cc + 1 < CB ? cc + 1 : cc] : hnext;
            key[cc] = max16(hcur[cc] + tv[cc], hv + cV);
          }
          WB_ROW_TAIL_FAST()
        } else {
          int tv[CB];
#pragma unroll
          for (int cc = 0; cc < CB; ++cc) tv[cc] = __builtin_amdgcn_sbfe((int)tn[cc], vb8, 8);
          const int hleft = wave_shr1(hcur[CB - 1], W_NEG16);
#pragma unroll
          for (int cc = 0; cc < CB; ++cc) {
            const int hd = cc == 0 ? hleft : hcur[cc > 0 ? cc - 1 : 0];
            key[cc] = max16(hd + tv[cc], hcur[cc] + cV);
          }
          WB_ROW_TAIL_FAST()
        }
        if (dx & (1u << 23)) WB_RING_WRITE()
      } else {
#ifdef C3_PHASE_PROF
        const unsigned long long gen_t0 = __builtin_readcyclecounter();
#endif
        uint4 de;
        de.x = dx;
        de.y = __builtin_amdgcn_readlane(dblk.y, li);
        de.z = __builtin_amdgcn_readlane(dblk.z, li);
        de.w = __builtin_amdgcn_readlane(dblk.w, li);
        const int np = (de.x >> 8) & 0xff;
        const bool ovf = (de.x >> 16) & 1, two = (de.x >> 19) & 1;
        if (np > 64) return -1;
        lo = WB_W_LO(de.w);
        const int vb8 = (int)(dx & 24u);
        int tv[CB];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) tv[cc] = __builtin_amdgcn_sbfe((int)tp[lo + cc], vb8, 8);
#pragma unroll
        for (int k = 0; k <= CB; ++k) tn[k] = tp[lo + k];                   // for the row below
        int kedge = 0;
        for (int t = 0; t < np; ++t) {
          int prow;
          if ((de.x >> 24) & 1) prow = 0;
          else if (!ovf) prow = (t == 0) ? (de.y & 0xffff) : (t == 1) ? (de.y >> 16) : (t == 2) ? (de.z & 0xffff) : (de.z >> 16);
          else {
            const int v = GP(const int, c.rows().ptr())[r];
            prow = -1;
            while (kedge < GP(const int, c.n_in().ptr())[v]) { int pr = GP(const int, c.rowof().ptr())[GP(const int, c.in_from().ptr())[EI(v, kedge)]]; ++kedge; if (pr >= 0) { prow = pr; break; } }
            if (prow < 0) break;
          }
          int hpv[CB + 1];                                                   // H[prow][lo + CB*lane - 1 + k]
          if (prow == 0) {                                                   // the virtual start row: H[0][j] = j * gap
            const int b0 = (lo + CB * lane - 1) * g4;
#pragma unroll
            for (int k = 0; k <= CB; ++k) hpv[k] = b0 + k * g4;
            if (lo + CB * lane == 0) hpv[0] = W_NEG16;
          } else if (r - prow <= WB_RING) {
            const unsigned short* sp_ = ring + (prow & (WB_RING - 1)) * SLOT;
            const int sh = lo - *(const int*)sp_;
            if (sh > WB_MAXSHIFT) bandbad = 1;
            const unsigned short* cp = sp_ + (WB_PADL - 1) + CB * lane + min(sh, WB_MAXSHIFT);
#pragma unroll
            for (int k = 0; k <= CB; ++k) hpv[k] = (int)cp[k];
          } else {
            const auto* hp_ = GP(const unsigned short, H16) + (size_t)prow * HS;
            const int sh = lo - (int)((unsigned)hp_[0] | ((unsigned)hp_[1] << 16));
            if (sh > WB_MAXSHIFT) bandbad = 1;
#pragma unroll
            for (int k = 0; k <= CB; ++k) { const int ix = CB * lane + sh - 1 + k; hpv[k] = (ix >= 0 && ix < BW) ? (int)hp_[WB_PADL + ix] : W_NEG16; }
          }
          if (t == 0) {
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) key[cc] = max16(hpv[cc] + tv[cc], hpv[cc + 1] + cV);
          } else {
            const int tmark = t << 16;
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) {
              const int nk = max16(key[cc], max16(hpv[cc] + tv[cc], hpv[cc + 1] + cV));
              key[cc] = nk != (key[cc] & 0xffff) ? (nk | tmark) : key[cc];
            }
          }
        }
        if (two) WB_ROW_TAIL(false) else WB_ROW_TAIL(true)
        if (de.x & (1u << 23)) WB_RING_WRITE()
        if ((de.x >> 17) & 1) {
          auto* hrow = GP(unsigned short, H16) + (size_t)r * HS;
          hrow[0] = (unsigned short)lo; hrow[1] = (unsigned short)((unsigned)lo >> 16);
#pragma unroll
          for (int cc = 0; cc < CB; ++cc) hrow[WB_PADL + CB * lane + cc] = (unsigned short)hcur[cc];
        }
        if ((de.x >> 18) & 1) {                                             // an end row: H[r][Q] is a candidate end of the alignment (first maximum in row order)
#pragma unroll
          for (int cc = 0; cc < CB; ++cc) { const int sc = __builtin_amdgcn_sbfe(hcur[cc], 2, 14); if (lo + lane * CB + cc == Q && sc > ebs) { ebs = sc; ebr = r; } }
        }
#ifdef C3_PHASE_PROF
        { const bool pm1 = (de.y & 0xffff) == (unsigned)(r - 1); pf_d0 += (1ull << 32) + (two && pm1); pf_d1 += 1 + ((unsigned long long)!two << 32);
          const unsigned long long dt_ = __builtin_readcyclecounter() - gen_t0; if (!two) pf_c2 += dt_; else if (pm1) pf_c4 += dt_; else pf_c3 += dt_; }
#endif
      }
      if ((li & (RPW - 1)) == RPW - 1) { GP(unsigned, DW)[dwoff] = dacc; dwoff += 64; dacc = 0; }
    }
    if (hend_ == cnt && (cnt & (RPW - 1))) {                                 // the last, partial word of the layer
      const int miss = RPW - (cnt & (RPW - 1));
      GP(unsigned, DW)[dwoff] = dacc >> (BPR * miss); dwoff += 64; dacc = 0;
    }
    // certificate bound of these rows: lane li holds the descriptor of row rb + li, the edge cells sit in ebuf[li & 31]
    if (lane >= half_ && lane < hend_) {
      const unsigned short* e = ebuf + (lane & (WB_EROWS - 1)) * WB_EROW;
      const int r = rb + lane;
      const int lo_r = WB_W_LO(dblk.w), dist = WB_W_DIST(dblk.w), nb = NB - WB_W_BIDX(dblk.w);
      const int hi = lo_r + BW - 1;
      if (hi < Q) best = max(best, ((int)(short)e[8 + CB - 1] >> 2) + ups * (Q - hi));
      if (((dblk.x >> 24) & 1) && lo_r > 0) best = max(best, ups * min(Q, nb + 1) + gap * max(0, Q - nb - 1));      // entered from (0, j), j < lo
      if (dist > 0) {
        // leftspan = lo(r + dist) - lo(r): shift bits of rows r+1 .. r+dist = bit indices r .. r+dist-1
        const int w0 = r >> 6, off = r & 63;
        unsigned long long a0 = d0bits[w0] >> off, a1 = d1bits[w0] >> off;
        if (off) { a0 |= d0bits[w0 + 1] << (64 - off); a1 |= d1bits[w0 + 1] << (64 - off); }
        const unsigned long long dm = dist >= 64 ? ~0ull : ((1ull << dist) - 1ull);
        const int ls = __popcll(a0 & dm) + 2 * __popcll(a1 & dm);
        if (ls > 2 * CB) bandbad = 1;                                        // only the first 2*CB cells of a row are parked
        for (int b = 0; b < min(ls, 2 * CB); ++b) {
          const int rem = Q - lo_r - b;
          best = max(best, ((int)(short)e[b] >> 2) + ups * min(rem, nb) + gap * max(0, rem - nb));
        }
      }
    }
    }
  }
#undef WB_ROW_TAIL
#undef WB_ROW_TAIL_FAST
#undef WB_RING_WRITE
  best = wave_max(best);
  const int gbs = wave_max(ebs);
  const int gbr = wave_min(ebs == gbs ? ebr : INT32_MAX / 2);
  if (lane == 0) { auto* res = GP(int, c.lob().ptr()); res[0] = best; res[1] = gbs; res[2] = gbr; }
  bandbad = __ballot(bandbad) != 0;
#ifdef C3_PHASE_PROF
  dbg[0] += pf_d0; dbg[1] += pf_d1 + pf_fast; dbg[2] += pf_c2; dbg[3] += pf_c3; dbg[4] += pf_c4;
#endif
  WSYNC();
  return bandbad ? 1 : 0;
}

// Traceback through a banded layer, 64 rows at a time (same scheme as the unbanded one in k_window: lane k owns row rt - k,
// every step inside a block is an LDS read).  A block costs ONE memory round trip: the descriptors of its 64 rows, the
// direction words of its rows around the path (NW words x LW lanes) and -- for rows with several predecessors -- four
// dwords of predecessor-index bytes are all fetched at once; the band start of every row comes from the shift bitmasks
// in LDS, not from memory.  rq[q] = DP row aligned to query base q, 0 = insertion.
// (a real call, like the rows: inlined into k_window the eleven-column tables pushed the kernel's register allocation over
// the edge -- 130 more bytes of scratch per lane, every reload a vector load that waits for all older stores)
#ifndef WB_TB_SHIFT
#define WB_TB_SHIFT 3     /* columns the traceback windows of a block sit to the right of its diagonal */
#endif
template <bool SECOND>
__device__ __attribute__((noinline)) void win_traceback_band(int* cI, int* cE, uint8_t* cD, uint4* crdesc, int cK, int cn, int cNcap, int CB_, int R_, int Q_, int r_,
                                                             int mw_, int big_, unsigned long long* prof_) {
  WCtx c;
  c.I = uni_ptr(cI); c.E = uni_ptr(cE); c.B8 = nullptr; c.score = nullptr; c.H = nullptr; c.D = uni_ptr(cD);
  c.rdesc = uni_ptr(crdesc); c.K = uni32(cK); c.n = uni32(cn); c.Ncap = uni32(cNcap); c.hcap = 0;
  const int CB = uni32(CB_), R = uni32(R_), Q = uni32(Q_), MW = uni32(mw_);
  int r = uni32(r_);
  unsigned long long* prof = uni_ptr(prof_);
  const int lane = wave_lane();
  extern __shared__ int lds_dyn[];
  const unsigned long long* m2bits = (const unsigned long long*)lds_dyn; const unsigned long long* mabits = m2bits + MW;
  const unsigned long long* d0bits = mabits + MW; const unsigned long long* d1bits = d0bits + MW;
  unsigned* lds = (unsigned*)lds_dyn + 8 * MW;                              // traceback windows behind the four bitmask arrays
  const QArr rq = {(unsigned short*)(lds + W_TBW), c.opq(), uni32(big_) != 0};
  const int RPW = wb_rpw(CB), BPR = 2 * CB, BIT0 = wb_bit0(CB), BW = 64 * CB;
  const int NW = 64 / RPW + 1, LW = CB == 2 ? 14 : 7;                       // direction words of a block, lanes fetched per word
  const int cdiv = (65536 + CB - 1) / CB;                                    // o / CB == (o * cdiv) >> 16 for o < 2^13
  const int G = wb_words(R, CB);
  const unsigned* DW = (const unsigned*)c.D;
  const unsigned* PX = DW + (size_t)(G + 1) * 64;
  unsigned* WD = lds;                                                        // [NW * LW] direction words (<= 128)
  int* WL0 = (int*)(lds + 128);                                              // [NW] first lane of every word's window
  unsigned* WP = lds + 160;                                                  // [64][4] predecessor-index dwords
  int j = Q;
  int lo_t = r > 0 ? WB_W_LO(c.rdesc[r].w) : 0;                             // band start of the current row
  while (r > 0 || j > 0) {
    if (r == 0) { for (int q = lane; q < j; q += 64) rq.set(q, 0); break; }
    if (j == 0) break;                                   // only vertical moves remain
#ifdef C3_PHASE_PROF
    const unsigned long long tb_t0 = __builtin_readcyclecounter();
#endif
    const int rt = r, jt = j;
    // band shifts of the 64 rows above rt: bit 63 - x of sd0 / sd1 = shift bit of row rt - x
    unsigned long long sd0, sd1;
    {
      const int wh = (rt - 1) >> 6, sh = (rt - 1) & 63;
      sd0 = d0bits[wh] << (63 - sh); sd1 = d1bits[wh] << (63 - sh);
      if (sh != 63 && wh > 0) { sd0 |= d0bits[wh - 1] >> (sh + 1); sd1 |= d1bits[wh - 1] >> (sh + 1); }
    }
    const int rk = rt - lane;
    const bool rowv = rk >= 1;
    const int rbx = max(rk, 1) - 1;
    const bool two = rowv && ((m2bits[rbx >> 6] >> (rbx & 63)) & 1);
    const bool adj = two && ((mabits[rbx >> 6] >> (rbx & 63)) & 1);
    // lo(rt - k) = lo(rt) - sum of the shifts of rows rt - k + 1 .. rt
    const int lok = lane == 0 ? lo_t : lo_t - (__popcll(sd0 >> (64 - lane)) + 2 * __popcll(sd1 >> (64 - lane)));
    const uint4 de = c.rdesc[max(rk, 1)];
    // ---- direction words: slot s of the block = word (ghi - s / LW), lane window position s % LW
    const int ghi = (rt - 1) / RPW;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = lane * 2 + h;
      const int wi = s / LW, ln = s - wi * LW;
      const int g = ghi - wi;
      if (wi < NW && g >= 0) {
        // expected cell of the word's middle row on the diagonal through (rt, jt)
        const int rm = min(g * RPW + RPW / 2 + 1, rt);
        const int km = min(rt - rm, 63);
        const int lom = km == 0 ? lo_t : lo_t - (__popcll(sd0 >> (64 - km)) + 2 * __popcll(sd1 >> (64 - km)));
        const int om = min(max(jt - km - lom + WB_TB_SHIFT, 0), BW - 1);       // (+: a path falls BEHIND its block's diagonal -- a skipped row costs a lane and no column)
        const int l0 = min(max(((om * cdiv) >> 16) - LW / 2, 0), 64 - LW);
        WD[s] = DW[(size_t)g * 64 + l0 + ln];
        if (ln == 0) WL0[wi] = l0;
      }
    }
    // ---- predecessor-index bytes of this lane's row (rows with several predecessors only)
    const int oe = min(max(jt - lane - lok + WB_TB_SHIFT, 0), BW - 1);
    const int pl0 = min(max(((oe * cdiv) >> 16) - 1, 0), 60);
    if (rowv && !two) {
      const unsigned* src = PX + (size_t)rk * 64 + pl0;
      *(uint4*)(WP + lane * 4) = make_uint4(src[0], src[1], src[2], src[3]);
    }
    WSYNC();
#ifdef C3_PHASE_PROF
    const unsigned long long tb_t1 = __builtin_readcyclecounter();
    prof[0] += 1; prof[1] += tb_t1 - tb_t0;
#endif
    const int wik = ghi - (max(rk, 1) - 1) / RPW;                            // this lane's word in the block
    const int kin = (max(rk, 1) - 1) % RPW;                                  // ... and its row inside the word
    const int wl0 = (wik >= 0 && wik < NW) ? WL0[wik] : 0;
    // (Tried: per-lane tables of the row's cells at the 11 columns around the block's diagonal, built once per block, so that
    // a step is a shift + ballot.  The walk itself got a third cheaper in wave time, the kernel 4 % SLOWER: the table build is
    // ~330 vector instructions per block in a kernel whose throughput is set by vector issue, while the steps below mostly wait.)
    for (;;) {
#ifdef C3_PHASE_PROF
      prof[2] += 1;
#endif
      const int s = rt - r;                               // lane s holds the current row
      const int jk = j - (lane - s);
      const int ok_ = jk - lok;                           // band offset of the cell
      const bool val = rowv && lane >= s && jk >= 0 && ok_ >= 0 && ok_ < BW && wik < NW;
      const int oc = min(max(ok_, 0), BW - 1);
      const int lq = (oc * cdiv) >> 16, cw = oc - lq * CB;
      const int ix = lq - wl0;
      bool hit = val && ix >= 0 && ix < LW;
#ifdef C3_EXP_TBSLOW
      hit = false;
#endif
      const unsigned wv = WD[min(max(wik, 0), NW - 1) * LW + min(max(ix, 0), LW - 1)];
      const unsigned cell = (wv >> (BIT0 + BPR * kin + 2 * cw)) & 3u;
      int d = 63 + 64 * (int)cell, prow = -1;
      if (two) prow = adj ? rk - 1 : (int)(de.y & 0xffff);      // a single predecessor: this lane's descriptor names it (no reload at a break; 0 = the virtual start row)
      else {
        const int px = lq - pl0;
        if (px < 0 || px > 3) hit = false;
        const unsigned pv = WP[lane * 4 + min(max(px, 0), 3)];
        d -= (int)((pv >> (8 * cw)) & 63u);
        if (win_d_type(d) != 2) prow = ((de.x >> 16) & 1) ? -2 : win_pred_row(c, de, rk, win_d_pred(d));
      }
      const bool diag1 = hit && jk >= 1 && win_d_type(d) == 0 && prow == rk - 1;
      const unsigned long long bal = __ballot(diag1) >> s;
      const int m = (~bal) ? __builtin_ctzll(~bal) : 64;  // length of the diagonal run
      if (lane >= s && lane < s + m) rq.set(jk - 1, rk);
      r -= m; j -= m;
      if (s + m >= 64 || r <= 0 || j <= 0) break;
      // the breaking cell (r, j) sits in lane cl
      const int cl = rt - r;
      int db, pb;
      if (wave_bcast((int)hit, cl)) { db = wave_bcast(d, cl); pb = wave_bcast(prow, cl); }
      else {
        // outside the windows (the path drifted off this block's diagonal): direct loads of the one cell
        const int lor = wave_bcast(lok, cl);
        const int o0 = min(max(j - lor, 0), BW - 1), l0_ = o0 / CB, c0 = o0 % CB;
        const unsigned w0 = DW[(size_t)((r - 1) / RPW) * 64 + l0_];
        db = 63 + 64 * (int)((w0 >> (BIT0 + BPR * ((r - 1) % RPW) + 2 * c0)) & 3u);
        const bool two0 = (m2bits[(r - 1) >> 6] >> ((r - 1) & 63)) & 1;
        if (two0) pb = ((mabits[(r - 1) >> 6] >> ((r - 1) & 63)) & 1) ? r - 1 : -3;
        else { db -= (int)((PX[(size_t)r * 64 + l0_] >> (8 * c0)) & 63u); pb = -2; }
      }
      const int ty = win_d_type(db);
      if (ty == 2) { if (lane == 0) rq.set(j - 1, 0); --j; }
      else {
        if (pb <= -2) pb = win_pred_row(c, c.rdesc[r], r, win_d_pred(db));   // > 4 predecessors or a window miss
        if (ty == 0) { if (lane == 0) rq.set(j - 1, r); --j; }
        r = pb;
      }
      if (r <= 0 || j <= 0) break;
      const int drift = (jt - j) - (rt - r);
      if (rt - r >= 64 || drift > 5 - WB_TB_SHIFT || drift < -5 - WB_TB_SHIFT) break;
    }
#ifdef C3_PHASE_PROF
    prof[3] += __builtin_readcyclecounter() - tb_t1;
#endif
    // band start of the row the next block starts at
    if (r > 0) {
      const int back = rt - r;
      lo_t = back < 64 ? wave_bcast(lok, back) : WB_W_LO(c.rdesc[r].w);
    }
    WSYNC();
  }
}
