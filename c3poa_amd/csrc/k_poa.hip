// k_poa.hip -- K3: adaptive-band partial-order alignment (convex gap), graph fusion with an
// incrementally maintained topological order, heaviest-bundling consensus / 2-row MSA +
// quality-aware pairwise merge, and the subread -> draft coordinate map used by the polish.
//
// Replaces, per read (paths relative to /root/reference):
//   bin/determine_consensus.py:30-47  pyabpoa.msa_aligner(match=5).msa(...)  [abPOA 1.0.5, external]
//   bin/consensus.py:4-81             pairwise_consensus (2 subreads)
//   bin/determine_consensus.py:56-67  the kept-subread overlaps (derived from the POA paths)
// Spec: DESIGN.md 4.3/4.4; bit-exact with oracle/c3o_poa.c + c3o_pairwise.c.
//
// Mapping: ONE WAVE (= one 64-thread workgroup) PER READ.  DP rows are graph nodes in
// topological order, the 64 lanes are consecutive band columns; the horizontal-gap states are
// two DPP max-scans per 64-column chunk; row maxima (adaptive band) are DPP reductions.
#include "c3_dev.h"
#include "c3_args.h"
#include <type_traits>

#define WSYNC() __syncthreads()
// adjacency slot k of node v.  Slot-major (all first edges, then all second edges, ...): nearly every node has one or two
// edges, so the arrays a wave actually touches are contiguous runs of Ncap ints instead of one 64-byte line per node
#define EI(v, k) ((size_t)(k) * (size_t)c.Ncap + (size_t)(v))
#define C3_POA_NI 18       // int arrays of Ncap per slot in Ctx::I
#define SRC 0
#define SNK 1


// Per-slot scratch of k_poa.  Only a few base pointers are kept live; every array is base + constant multiple of Ncap
// (or Ncap*K, cells_cap), which keeps the uniform state in SGPRs instead of spilling it into VGPR lanes.
struct Ctx {
  int* I; int* E; char* C; uint8_t* B8; long long* score_; uint4* desc_; int* jump_; int* path_;
  int K, n, Ncap, cells_cap;
  int osel;                  // which of the two order buffers is current (g_reorder writes the other one and flips)
  const uint32_t* pk;        // packed read
#define CTX_I(name, k) __device__ __forceinline__ int* name() const { return I + (size_t)(k) * Ncap; }
  CTX_I(n_in, 0) CTX_I(n_out, 1) CTX_I(grp, 2) CTX_I(index, 5) CTX_I(gfirst, 6) CTX_I(glast, 7)
  CTX_I(rem, 8) CTX_I(mpl, 9) CTX_I(mpr, 10) CTX_I(rowm, 11) /* 3 ints per row: band begin, band end, cell offset (blocks 11..13) */ CTX_I(anchor, 14) CTX_I(col, 15)
  CTX_I(col2t, 16) CTX_I(nxt, 17)
#undef CTX_I
  __device__ __forceinline__ int* order() const { return I + (size_t)(3 + osel) * Ncap; }
  __device__ __forceinline__ int* order2() const { return I + (size_t)(4 - osel) * Ncap; }
  __device__ __forceinline__ int* in_from() const { return E; }
  __device__ __forceinline__ int* out_to() const { return E + (size_t)Ncap * K; }
  __device__ __forceinline__ int* out_w() const { return E + 2 * (size_t)Ncap * K; }
  __device__ __forceinline__ int32_t* H() const { return (int32_t*)C; }
  __device__ __forceinline__ int32_t* E1() const { return (int32_t*)(C + 4 * (size_t)cells_cap); }
  __device__ __forceinline__ int32_t* E2() const { return (int32_t*)(C + 8 * (size_t)cells_cap); }
  __device__ __forceinline__ uint32_t* D() const { return (uint32_t*)(C + 12 * (size_t)cells_cap); }
  __device__ __forceinline__ uint8_t* D8() const { return (uint8_t*)(C + 16 * (size_t)cells_cap); }     // direction bytes (rows of <= 4 predecessors)
  __device__ __forceinline__ uint8_t* P8() const { return (uint8_t*)(C + 17 * (size_t)cells_cap); }     // predecessor bytes (rows of 2..4 predecessors)
  __device__ __forceinline__ uint8_t* base() const { return B8; }
  __device__ __forceinline__ uint8_t* rows2() const { return B8 + (size_t)Ncap; }
  __device__ __forceinline__ long long* score() const { return score_; }
  __device__ __forceinline__ uint4* descA() const { return desc_; }
  __device__ __forceinline__ uint4* descB() const { return desc_ + (size_t)Ncap; }
  __device__ __forceinline__ int* jump() const { return jump_; }
  __device__ __forceinline__ int* path() const { return path_; }     // node of every base fused so far: Pcap = sum of the subread lengths
};

__device__ __forceinline__ void g_add_edge(Ctx& c, int u, int v, int w) {
  const int K = c.K;
  for (int k = 0; k < c.n_out()[u]; ++k)
    if (c.out_to()[EI(u, k)] == v) { c.out_w()[EI(u, k)] += w; return; }
  int no = c.n_out()[u], ni = c.n_in()[v];
  c.out_to()[EI(u, no)] = v; c.out_w()[EI(u, no)] = w; c.n_out()[u] = no + 1;
  c.in_from()[EI(v, ni)] = u; c.n_in()[v] = ni + 1;
}

// block extents from order/grp (parallel).  GU chunks of 64 positions per iteration: the two dependent levels (position -> node ->
// group) cost one memory latency per GU*64 nodes; no look-ahead loads (a block's last position is known when the next one starts).
#ifndef GU
#define GU 2
#endif
__device__ void g_blocks(Ctx& c, int lane) {
  // The members of an aligned block are CONTIGUOUS in the topological order (a new sibling is merged right behind its
  // block), so a block's extent is a run of equal group ids: where the id changes, the new block starts and the previous one
  // ends -- plain stores, no initialisation pass, no atomics.
  int gprev = -1;
  for (int i0 = 0; i0 < c.n; i0 += 64 * GU) {
    int v[GU], r[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) { const int i = i0 + 64 * u + lane; v[u] = i < c.n ? c.order()[i] : -1; }
#pragma unroll
    for (int u = 0; u < GU; ++u) r[u] = v[u] >= 0 ? c.grp()[v[u]] : -2;
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int i = i0 + 64 * u + lane;
      const int rp = wave_shr1(r[u], gprev);
      gprev = wave_bcast(r[u], 63);
      if (v[u] >= 0 && r[u] != rp) { c.gfirst()[r[u]] = i; if (i > 0) c.glast()[rp] = i - 1; }
      if (v[u] >= 0 && i + 1 == c.n) c.glast()[r[u]] = i;
    }
  }
  WSYNC();
}

// merge new nodes n_old..n-1 (creation order, anchor[k] = old order index they follow) into order
__device__ void g_reorder(Ctx& c, int n_old, int lane, int* lds, int lds_cap) {
  const int n_new = c.n - n_old;
  // old node at old index i moves to i + #(anchor < i); new node k goes to anchor[k] + 1 + k.  The (sorted) anchors of the new
  // nodes are staged in LDS first: the binary search per old node is then 6-8 LDS reads instead of 6-8 dependent global loads
  // per 64 nodes (the LDS scratch of the alignment is idle during the graph phases).  The new order goes to the OTHER order
  // buffer together with index[], and the two buffers swap roles: no copy-back pass.
  const bool inl = n_new <= lds_cap;
  if (inl) { for (int k = lane; k < n_new; k += 64) lds[k] = c.anchor()[k]; WSYNC(); }
  for (int i0 = 0; i0 < n_old; i0 += 64 * GU) {
    int v[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) { const int i = i0 + 64 * u + lane; v[u] = i < n_old ? c.order()[i] : -1; }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int i = i0 + 64 * u + lane;
      if (v[u] < 0) continue;
      int lo = 0, hi = n_new;                 // first k with anchor[k] >= i
      if (inl) { while (lo < hi) { int m = (lo + hi) >> 1; if (lds[m] < i) lo = m + 1; else hi = m; } }
      else { while (lo < hi) { int m = (lo + hi) >> 1; if (c.anchor()[m] < i) lo = m + 1; else hi = m; } }
      c.order2()[i + lo] = v[u]; c.index()[v[u]] = i + lo;
    }
  }
  for (int k = lane; k < n_new; k += 64) { const int pos = (inl ? lds[k] : c.anchor()[k]) + 1 + k; c.order2()[pos] = n_old + k; c.index()[n_old + k] = pos; }
  c.osel ^= 1;
  WSYNC();                                  // (the LDS words are free again)
  g_blocks(c, lane);
}

__device__ __forceinline__ int32_t rdcell(const Ctx& c, const int32_t* a, int pb, int pe, int po, int j) {
  return (j < pb || j > pe) ? C3_NEG : a[po + (j - pb)];
}

#ifdef C3_PHASE_PROF
#define PHA , unsigned long long& ph_t0_, unsigned long long (&ph_acc_)[16]
#define PHP , ph_t0_, ph_acc_
#else
#define PHA
#define PHP
#endif
// ---- DP of one alignment --------------------------------------------------------------------------
// LDS ring of the last PR rows (cells and band metadata): reading a predecessor row costs LDS
// latency only, and -- since gfx9 counts loads and stores in ONE in-order vmcnt -- the row loop
// carries no vector loads on its common path (descriptors arrive 64 rows at a time and are broadcast
// with v_readlane).  Rows with a successor more than PR-1 rows ahead, or wider than a ring slot,
// also go to the global arena; the direction words always do (4 B per cell).
#ifndef C3_NEAR
#define C3_NEAR 1
#endif
#define PW 128      // ring slot width (cells)
#define PR 4        // ring rows
#define PQW 112     // packed query words kept in LDS (1792 bases); longer subreads read the packed read
struct PoaLds { int H[PR][PW], E1[PR][PW], E2[PR][PW]; int4 meta[PR] /* band begin, end, leftmost / rightmost argmax */; unsigned qpk[PQW]; };
__shared__ PoaLds L;     // file scope: accesses stay in the LDS address space (ds_*, lgkmcnt only)

// scores are carried as score*512 (+ a 9-bit tag while candidates compete): one v_max per candidate
// implements "highest score, first candidate in order".  Unreachable cells use -(2^20) score units.
#define S9(x) ((x) * 512)
#define NEGS (-(1 << 29))
#define NEG2S (-(1 << 30))

// direction word (device-internal): mp[0..7] | e1code[8..16] | e2code[17..25] | hts[26..27] | hs[28..29]
// | f1x[30] | f2x[31];  e?code = 2*pred + ext;  hts: 0 M, 1 E1, 2 E2;  hs: 0 Ht, 1 F1, 2 F2

// Row descriptors of one alignment AND the "remaining length" of every node in ONE reverse sweep over the topological order,
// 64 positions at a time (lane = position).  rem[v] = hops from v to the sink along its heaviest out-edge (first maximum in
// out-list order) - 1; the chain of a node continues at a LATER position, so when the chunks are taken from the last to
// the first, a target outside the chunk is already final (one gather from hops[]), and the chains inside a chunk are
// resolved by pointer doubling between lanes (ds_bpermute, at most six rounds).  Replaces log2(n) rounds of pointer jumping
// over four node arrays in memory by four dependent memory levels per chunk.
// descriptor A: x = node, y = unused, z = base | nin<<8 | far<<16 | (nin>4)<<17 | fast-row candidate<<18 | sink<<19, w = qr = Q - rem;
// descriptor B: positions of the first four predecessors.  hops[] (by position) lives in col() (free until the MSA columns).
__device__ void poa_sweep_desc(Ctx& c, int lane, int Q, bool qlds) {
  const int n = c.n;
  int* hops = c.col();
  for (int c0 = ((n - 1) >> 6) << 6; c0 >= 0; c0 -= 64) {
    const int idx = c0 + lane;
    const bool live = idx < n;
    const int v = live ? c.order()[idx] : SNK;
    const int nin = live ? c.n_in()[v] : 0, nout = live ? c.n_out()[v] : 0;
    int bw = INT32_MIN, bt = SNK;
    unsigned far = 0;
    for (int k = 0; __builtin_amdgcn_ballot_w64(k < nout) != 0; ++k) {
      if (k < nout) {
        const int t = c.out_to()[EI(v, k)], ww = c.out_w()[EI(v, k)];
        if (ww > bw) { bw = ww; bt = t; }
        if (t == SNK || c.index()[t] - idx > PR - 1) far = 1;
      }
    }
    unsigned p[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < nin) p[k] = (unsigned)c.index()[c.in_from()[EI(v, k)]];
    // hops to the sink: d = hops to ptr, ptr = position still to follow (-1: d is final)
    int d = (v == SNK) ? 0 : 1, ptr = (v == SNK || !live) ? -1 : c.index()[bt];
    if (ptr >= c0 + 64) { d += hops[ptr]; ptr = -1; }                    // beyond this chunk: final already
    for (int r = 0; r < 6 && __builtin_amdgcn_ballot_w64(ptr >= 0) != 0; ++r) {
      const int src = (max(ptr, c0) - c0) << 2;
      const int dd = __builtin_amdgcn_ds_bpermute(src, d), pp = __builtin_amdgcn_ds_bpermute(src, ptr);
      if (ptr >= 0) { d += dd; ptr = pp; }
    }
    if (live) {
      hops[idx] = d;
      uint4 A; A.x = (unsigned)v; A.y = 0;
      A.z = (unsigned)c.base()[v] | ((unsigned)min(nin, 255) << 8) | (far << 16) | ((unsigned)(nin > 4) << 17);
      // bit 18: candidate for the fast row (one predecessor, the row above; query in LDS); bit 19: the sink (no DP row)
      A.z |= ((unsigned)(qlds && nin == 1 && (int)p[0] == idx - 1 && v != SRC && v != SNK) << 18) | ((unsigned)(v == SNK) << 19);
      A.w = (unsigned)(Q - (d - 1));                         // qr: the query column this node would sit on by distance to the sink
      uint4 B; B.x = p[0]; B.y = p[1]; B.z = p[2]; B.w = p[3];
      c.descA()[idx] = A; c.descB()[idx] = B;
    }
    WSYNC();                                                 // hops[] of this chunk is read by the next (earlier) chunk
  }
}

// banded global alignment of subread [qb, qb+Q) against the graph; ops are written BACKWARDS
// into opn/opq, returns their count (or <0 on failure)
__device__ int poa_align(Ctx& c, const C3Params& P, int qb, int Q, int lane, long long* cells PHA) {
  const int K = c.K, n = c.n;
  const int mt9 = S9(P.poa_match), mm9 = S9(-P.poa_mismatch);
  const int e1_9 = S9(P.e1), e2_9 = S9(P.e2), o1_9 = S9(P.o1), o2_9 = S9(P.o2), oe1_9 = S9(P.o1 + P.e1), oe2_9 = S9(P.o2 + P.e2);
  const int w = wave_first(P.band_b + (int)(P.band_f * (double)Q));
  const int le1 = e1_9 * lane, le2 = e2_9 * lane;                   // the F scans run in lane coordinates: e*(j-beg) = e*lane
  const int lo1 = le1 + o1_9, lo2 = le2 + o2_9;
  const bool qlds = Q <= PQW * 16;
  poa_sweep_desc(c, lane, Q, qlds);
  // the subread, 2-bit packed and re-aligned to its first base, goes to LDS: the row loop must not
  // touch global memory for it (a vector load would wait for every older row store)
  if (qlds) {
    for (int i = lane; i * 16 < Q; i += 64) {
      const long long b0 = (long long)qb + 16 * i;
      const unsigned w0 = c.pk[b0 >> 4], w1 = c.pk[(b0 >> 4) + 1];
      L.qpk[i] = __builtin_amdgcn_alignbit(w1, w0, (unsigned)(b0 & 15) * 2);
    }
  }
  WSYNC();
  PH_MARK(0)
  // Row loop.  The scalar ALU is ONE per CU (measured: 0.96 scalar instructions per cycle per CU against 1.5-1.7 vector
  // instructions; DPP forms at half the vector rate), so uniform per-row values -- the previous row's band and argmax span, the
  // cell counter -- live in VECTOR registers (every lane holds the same number) and the band arithmetic runs on the vector
  // ALU; the scalar unit only sees the loop, one descriptor test and two branches per row.
  int u_beg = 0, u_end = -1, u_left = 0, u_right = 0, u_ncell = 0;                    // previous row / cells used so far
  int pH = NEGS, pE1 = NEGS, pE2 = NEGS;                                              // its cells, lane = band column
  bool pv_ok = false;                                                                  // ... valid: the row above, <= 64 cells
  const int lane4 = lane * 4;
#define UNI(x) asm volatile("" : "+v"(x))       /* keep a uniform value in a vector register (no instruction) */
#ifdef C3_PHASE_PROF
  unsigned long long row_t0 = __builtin_readcyclecounter();
#endif
  for (int ib = 0; ib < n; ib += 64) {
  uint4 dA = c.descA()[min(ib + lane, n - 1)], dB = c.descB()[min(ib + lane, n - 1)];
  asm volatile("" : "+v"(dA.x), "+v"(dA.y), "+v"(dA.z), "+v"(dA.w), "+v"(dB.x), "+v"(dB.y), "+v"(dB.z), "+v"(dB.w));   // wait here, not in the row loop
  const int cnt = min(64, n - ib);
  for (int li = 0; li < cnt; ++li) {
    const int idx = ib + li;
    const int fl = __builtin_amdgcn_readlane(dA.z, li);
    if ((fl >> 19) & 1) { pv_ok = false; continue; }                                  // the sink has no row
    const int qr = __builtin_amdgcn_readlane(dA.w, li);
    const int vb = fl & 0xff;
    const bool far = (fl >> 16) & 1;
    // ---- FAST ROW: one predecessor = the previous row, whose H/E1/E2 are still in this wave's REGISTERS (lane = band
    // column).  The predecessor cells arrive by lane permutes (no LDS round trip through memory on the dependent chain),
    // the row maximum is taken from Ht in parallel with the two F scans (an F value is always strictly below some Ht to
    // its left, so max H == max Ht and both are attained in the same columns), and every tie order is a tag in the low
    // bits of the compared keys, so the direction cell is 1 byte of masked key bits
    // (DIRECTION BYTE, all rows): bit0 E1 opened (0 = extended), bit1 E2 opened, bits2-3 Ht source (2 M, 1 E1, 0 E2),
    // bits4-5 H source (2 Ht, 1 F1, 0 F2), bit6 F1 extended, bit7 F2 extended.
    if (((fl >> 18) & 1) && pv_ok) {
      UNI(u_beg); UNI(u_end); UNI(u_left); UNI(u_right); UNI(u_ncell);
      const bool nonempty = u_end >= u_beg;
      const int mplv = nonempty ? u_left + 1 : INT32_MAX / 2, mprv = nonempty ? u_right + 1 : 0;
      const int beg = max(max(0, min(mplv, qr) - w), u_beg);
      int end = min(min(Q, max(mprv, qr) + w), u_end + 1);
      end = max(end, beg - 1);
      const int wd = end - beg + 1, sh = beg - u_beg;
      // (64 cells of head room instead of wd: the stores below are not masked)
      if (__builtin_amdgcn_ballot_w64((unsigned)(wd - 1) < 64u && sh < 64 && u_ncell + 64 <= c.cells_cap) != 0) {
        const int slot = idx & (PR - 1);
        const int ro = u_ncell;
        const int j = beg + lane;
        const bool act = lane < wd;
        // query base of column j (LDS copy; issued first, consumed after the permutes)
        const int jq = max(j - 1, 0);
        const unsigned qw_ = L.qpk[min(jq >> 4, PQW - 1)];
        // previous row moved to this row's columns: hp = H[i-1][j] sits sh lanes to the right, hd = H[i-1][j-1] one less
        const int a_p = lane4 + sh * 4, a_d = a_p - 4;
        const int hd_ = __builtin_amdgcn_ds_bpermute(a_d, pH), hp_ = __builtin_amdgcn_ds_bpermute(a_p, pH);
        const int e1_ = __builtin_amdgcn_ds_bpermute(a_p, pE1), e2_ = __builtin_amdgcn_ds_bpermute(a_p, pE2);
        const bool vp = j <= u_end, vd = a_d >= 0;                              // (j - 1 <= u_end always: end <= u_end + 1)
        const int hd = vd ? hd_ : NEGS, hp = vp ? hp_ : NEGS, e1p = vp ? e1_ : NEGS, e2p = vp ? e2_ : NEGS;
        const int qc = (int)((qw_ >> ((jq & 15) * 2)) & 3);
        const int M9 = hd + ((vb == qc) ? mt9 + 8 : mm9 + 8);                   // tag 2 in bits 2-3
        const int E1t = max(hp - (oe1_9 - 1), e1p - e1_9);                     // bit 0 set: opened (open wins ties)
        const int E2t = max(hp - (oe2_9 - 2), e2p - e2_9);                     // bit 1 set: opened
        const int E1c = E1t & ~511, E2c = E2t & ~511;
        const int k2 = max(max(M9, E1c + 4), E2c);
        const int ht9 = k2 & ~511;
        const int htm = act ? ht9 : NEG2S;
        // F[j] = max_{k<j} Ht[k] - o - e*(j-k): scanned in lane coordinates (the e*beg term cancels)
        int s1 = htm + le1, s2 = htm + le2, s3 = htm;
        wave_scan_max3(s1, s2, s3);
        const int px1 = wave_shr1(s1, NEG2S), px2 = wave_shr1(s2, NEG2S);
        const int htl = wave_shr1(htm, NEGS);
        const int f1 = px1 - lo1, f2 = px2 - lo2;
        const int k3 = max(max(ht9 + 32, f1 + 16), f2);
        const int h9 = k3 & ~511;
        unsigned d = ((unsigned)E1t & 1u) | ((unsigned)E2t & 2u) | ((unsigned)k2 & 12u) | ((unsigned)k3 & 48u);
        d |= (((unsigned)(htl - oe1_9 - f1)) >> 25) & 64u;                      // f1 > its "open" candidate: extended
        d |= (((unsigned)(htl - oe2_9 - f2)) >> 24) & 128u;
        pH = act ? h9 : NEGS; pE1 = act ? E1c : NEGS; pE2 = act ? E2c : NEGS;
        // unmasked stores: lanes past the band write cells that the next rows overwrite / that no reader ever selects
        c.D8()[(unsigned)(ro + lane)] = (uint8_t)d;
        L.H[slot][lane] = h9; L.E1[slot][lane] = E1c; L.E2[slot][lane] = E2c;
        const int rb = __builtin_amdgcn_readlane(s3, 63);
        // first / last column holding the row maximum: columns are beg + lane, so one ballot replaces two reductions
        const unsigned long long mxm = __ballot(htm == rb);
        const int left = beg + __builtin_ctzll(mxm), right = beg + (63 - __builtin_clzll(mxm));
        if (lane == 0) {
          L.meta[slot] = make_int4(beg, end, left, right);
          int off = 3 * idx; UNI(off);
          int* rm = c.rowm() + off; rm[0] = beg; rm[1] = end; rm[2] = ro;                 // type 0: byte cells, one predecessor
        }
        if (far) {
          if (act) { c.H()[ro + lane] = h9; c.E1()[ro + lane] = E1c; c.E2()[ro + lane] = E2c; }
          if (lane == 0) { c.mpl()[idx] = left; c.mpr()[idx] = right; }
        }
        u_beg = beg; u_end = end; u_left = left; u_right = right; u_ncell = ro + wd;
#ifdef C3_PHASE_PROF
        { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_[8] += t_ - row_t0; row_t0 = t_; }
#endif
        continue;
      }
    }
    const int v = __builtin_amdgcn_readlane(dA.x, li);
    const int nin = (fl >> 8) & 0xff;
    const bool ovf = (fl >> 17) & 1;
    const int p0 = __builtin_amdgcn_readlane(dB.x, li), p1 = __builtin_amdgcn_readlane(dB.y, li);
    const int p2 = __builtin_amdgcn_readlane(dB.z, li), p3 = __builtin_amdgcn_readlane(dB.w, li);
#define PRED_IDX(k) ((k) == 0 ? p0 : (k) == 1 ? p1 : (k) == 2 ? p2 : (k) == 3 ? p3 : c.index()[c.in_from()[EI(v, (k))]])
    int ncell = wave_first(u_ncell);
    // ---- NEAR ROW: up to four predecessors, every one of them among the last PR-1 rows (so its cells and band record are in
    // the LDS ring) -- the usual member of an aligned block and the node after it -- and a band of at most 64 (one chunk) or
    // 128 columns (two chunks: the rows whose nominal column has drifted away from the predecessors' maxima).  Branch free: the
    // ring records and the predecessor cells are read unconditionally (clamped addresses), an absent predecessor is given an
    // empty band far to the right so that every mask derived from it is false, the band arithmetic runs on the vector ALU
    // (uniform values), and one ballot decides whether the row qualifies.  Tie order by tags exactly as in the general row
    // below; direction byte + predecessor byte.  One instance per (predecessor count, chunk count).
    if (C3_NEAR && !ovf && qlds && v != SRC && idx - p0 < PR && (nin < 2 || idx - p1 < PR) && (nin < 3 || idx - p2 < PR) && (nin < 4 || idx - p3 < PR)) {
      auto near_body = [&](auto NPc, auto NCHc) -> bool {
      constexpr int NP = decltype(NPc)::value, NCH = decltype(NCHc)::value;
      UNI(u_ncell);
      const int BIGB = 1 << 28;
      int4 m_[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) m_[k] = L.meta[(k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3) & (PR - 1)];
      int pb_[NP], pe_[NP];
      int mplv = INT32_MAX / 2, mprv = 0, minb = INT32_MAX, maxe = INT32_MIN, wmax = 0;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const bool has = k < nin;                                             // scalar condition, used as a select mask
        const int b = has ? m_[k].x : BIGB, e = has ? m_[k].y : -BIGB;
        pb_[k] = b; pe_[k] = e;
        minb = min(minb, b); maxe = max(maxe, e + 1); wmax = max(wmax, e - b);
        const bool ne = e >= b;
        mplv = ne ? min(mplv, m_[k].z + 1) : mplv; mprv = ne ? max(mprv, m_[k].w + 1) : mprv;
      }
      const int beg = max(max(0, min(mplv, qr) - w), minb);
      int end = min(min(Q, max(mprv, qr) + w), maxe);
      end = max(end, beg - 1);
      const int wd = end - beg + 1;
      if (__builtin_amdgcn_ballot_w64((unsigned)(wd - 1) < (unsigned)(64 * NCH) && wmax < PW && u_ncell + 64 * NCH <= c.cells_cap) == 0) return false;
      const int slot = idx & (PR - 1);
      const int ro = u_ncell;
      int carry1 = NEG2S, carry2 = NEG2S, prev_ht = NEGS;                       // scan carries from the first chunk
      int best = INT32_MIN, left = 0, right = 0;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = 64 * ch;
        const int j = beg + c0 + lane;
        const bool act = c0 + lane < wd;
        const int jq = max(j - 1, 0);
        const unsigned qw_ = L.qpk[min(jq >> 4, PQW - 1)];
        int hd_[NP], hp_[NP], e1_[NP], e2_[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int sl = (k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3) & (PR - 1);
          const int o = j - pb_[k];
          const int oc = min(max(o, 0), PW - 1), om = min(max(o - 1, 0), PW - 1);
          hd_[k] = L.H[sl][om]; hp_[k] = L.H[sl][oc]; e1_[k] = L.E1[sl][oc]; e2_[k] = L.E2[sl][oc];
        }
        int kM = INT32_MIN, kE1 = INT32_MIN, kE2 = INT32_MIN;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const bool vd = j - 1 >= pb_[k] && j - 1 <= pe_[k], vp = j >= pb_[k] && j <= pe_[k];       // (j - 1 >= b >= 0 implies j > 0)
          const int hd = vd ? hd_[k] : NEGS, hp = vp ? hp_[k] : NEGS, e1p = vp ? e1_[k] : NEGS, e2p = vp ? e2_[k] : NEGS;
          kM = max(kM, hd + (511 - k));
          kE1 = max(kE1, max(hp - oe1_9 + (511 - 2 * k), e1p - e1_9 + (510 - 2 * k)));
          kE2 = max(kE2, max(hp - oe2_9 + (511 - 2 * k), e2p - e2_9 + (510 - 2 * k)));
        }
        const int qc = (int)((qw_ >> ((jq & 15) * 2)) & 3);
        const int M9 = (j > 0) ? (kM & ~511) + ((vb == qc) ? mt9 : mm9) : NEGS;
        const int E1c = kE1 & ~511, E2c = kE2 & ~511;
        const int k2 = max(max(M9 + 2, E1c + 1), E2c);
        const int ht9 = k2 & ~511;
        const unsigned mp = 511u - ((unsigned)kM & 511u), c1 = 511u - ((unsigned)kE1 & 511u), c2 = 511u - ((unsigned)kE2 & 511u);
        unsigned d = ((~c1) & 1u) | (((~c2) & 1u) << 1) | (((unsigned)k2 & 3u) << 2);
        const unsigned pby = mp | ((c1 >> 1) << 2) | ((c2 >> 1) << 4);
        const int htm = act ? ht9 : NEG2S;
        const int cl1 = le1 + e1_9 * c0, cl2 = le2 + e2_9 * c0;               // e * (column - beg)
        int s1 = htm + cl1, s2 = htm + cl2, s3 = htm;
        wave_scan_max3(s1, s2, s3);
        const int px1 = max(wave_shr1(s1, NEG2S), carry1), px2 = max(wave_shr1(s2, NEG2S), carry2);
        const int htl = wave_shr1(htm, prev_ht);
        const int f1 = px1 - o1_9 - cl1, f2 = px2 - o2_9 - cl2;               // column beg: NEG2S - ... (never wins)
        if (NCH > 1) { carry1 = max(carry1, wave_bcast(s1, 63)); carry2 = max(carry2, wave_bcast(s2, 63)); prev_ht = wave_bcast(htm, 63); }
        const int k3 = max(max(ht9 + 2, f1 + 1), f2);
        const int h9 = k3 & ~511;
        d |= (((unsigned)k3 & 3u) << 4);
        d |= (((unsigned)(htl - oe1_9 - f1)) >> 25) & 64u;                      // f1 > its "open" candidate: extended
        d |= (((unsigned)(htl - oe2_9 - f2)) >> 24) & 128u;
        if (ch == 0) { pH = act ? h9 : NEGS; pE1 = act ? E1c : NEGS; pE2 = act ? E2c : NEGS; }
        // unmasked stores (see the fast row); every near row keeps a predecessor byte (type 1), also with one predecessor
        c.D8()[(unsigned)(ro + c0 + lane)] = (uint8_t)d; c.P8()[(unsigned)(ro + c0 + lane)] = (uint8_t)pby;
        L.H[slot][c0 + lane] = h9; L.E1[slot][c0 + lane] = E1c; L.E2[slot][c0 + lane] = E2c;
        if (far) { if (act) { c.H()[ro + c0 + lane] = h9; c.E1()[ro + c0 + lane] = E1c; c.E2()[ro + c0 + lane] = E2c; } }
        const int cmx = __builtin_amdgcn_readlane(s3, 63);                      // maximum of Ht over the chunk (== maximum of H)
        const unsigned long long mxm = __ballot(htm == cmx);
        if (NCH == 1 || cmx > best) { best = cmx; left = beg + c0 + __builtin_ctzll(mxm); right = beg + c0 + (63 - __builtin_clzll(mxm)); }
        else if (cmx == best && mxm) right = beg + c0 + (63 - __builtin_clzll(mxm));
      }
      pv_ok = NCH == 1;
      if (lane == 0) {
        L.meta[slot] = make_int4(beg, end, left, right);
        int off = 3 * idx; UNI(off);
        int* rm = c.rowm() + off; rm[0] = beg; rm[1] = end | (1 << 28); rm[2] = ro;
        if (far) { c.mpl()[idx] = left; c.mpr()[idx] = right; }
      }
      u_beg = beg; u_end = end; u_left = left; u_right = right; u_ncell = ro + wd;
#ifdef C3_PHASE_PROF
      { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_[10] += t_ - row_t0; row_t0 = t_; }
#endif
      return true;
      };
      typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 4> I4;
      bool handled = nin == 1 ? near_body(I1{}, I1{}) : nin == 2 ? near_body(I2{}, I1{}) : near_body(I4{}, I1{});
      if (!handled) handled = nin == 1 ? near_body(I1{}, I2{}) : nin == 2 ? near_body(I2{}, I2{}) : near_body(I4{}, I2{});
      if (handled) continue;
    }
    // ---- GENERAL ROW.  Adaptive band: gather the hints of the predecessors (abPOA scatters them to the successors).  The
    // ring metadata of the first four predecessors is fetched in ONE LDS round trip (one 16-byte read each, issued
    // together) and pinned to scalars; predecessors that left the ring (or a fifth, sixth ... one) take global loads.
    int beg, end;
    int pb_[4] = {0, 0, 0, 0}, pe_[4] = {-1, -1, -1, -1};          // band of predecessor k (k < 4)
    bool ring_[4] = {false, false, false, false};                    // ... and whether its cells are in the LDS ring
    if (v == SRC) { beg = 0; end = min(Q, max(qr, 0) + w); }
    else {
      int mplv = INT32_MAX / 2, mprv = 0, minb = INT32_MAX, maxe = INT32_MIN;
      int4 m_[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) m_[k] = L.meta[(k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3) & (PR - 1)];
      asm volatile("" : "+v"(m_[0].x), "+v"(m_[1].x), "+v"(m_[2].x), "+v"(m_[3].x));     // the LDS loads happen HERE, together
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < nin) {
          const int pi = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
          int b = m_[k].x, e = m_[k].y, l = m_[k].z, r = m_[k].w;
          // (written as an override, not as if/else: a select between an LDS and a global POINTER becomes a flat load)
          if (idx - pi >= PR) { b = c.rowm()[3 * pi]; e = c.rowm()[3 * pi + 1] & 0x0fffffff; l = c.mpl()[pi]; r = c.mpr()[pi]; }
          b = wave_first(b); e = wave_first(e); l = wave_first(l); r = wave_first(r);
          pb_[k] = b; pe_[k] = e; ring_[k] = idx - pi < PR && e - b + 1 <= PW;
          minb = min(minb, b); maxe = max(maxe, e + 1);
          if (e >= b) { mplv = min(mplv, l + 1); mprv = max(mprv, r + 1); }
        }
      }
      for (int k = 4; k < nin; ++k) {
        const int pi = PRED_IDX(k);
        const int sl = pi & (PR - 1);
        int4 m = L.meta[sl];
        asm volatile("" : "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));
        int b = m.x, e = m.y, l = m.z, r = m.w;
        if (idx - pi >= PR) { b = c.rowm()[3 * pi]; e = c.rowm()[3 * pi + 1] & 0x0fffffff; l = c.mpl()[pi]; r = c.mpr()[pi]; }
        minb = min(minb, b); maxe = max(maxe, e + 1);
        if (e >= b) { mplv = min(mplv, l + 1); mprv = max(mprv, r + 1); }
      }
      beg = max(0, min(mplv, qr) - w);
      end = min(Q, max(mprv, qr) + w);
      beg = max(beg, minb); end = min(end, maxe);
    }
    // the band comes out of LDS / global loads the compiler cannot see are uniform: pin it to scalars, or every
    // loop-carried row descriptor (and the fast row's band arithmetic and branches) turns into vector code
    beg = wave_first(beg); end = wave_first(end);
    if (end < beg) end = beg - 1;
    const int wd = end - beg + 1;
    if (ncell + wd > c.cells_cap) return -4;
    const int slot = idx & (PR - 1);
    const bool inl = wd <= PW, toglobal = far || !inl;             // wd is scalar (beg / end pinned above)
    const int ro = ncell;
    const int ty = ovf ? 2 : (nin >= 2 ? 1 : 0);                   // cell format: byte / byte + predecessor byte / 32-bit word
    ncell += wd;
    int best = INT32_MIN, bl = 0, br = 0;        // per-lane running row maximum
    int carry1 = NEG2S, carry2 = NEG2S;          // scan carries over previous chunks
    int prev_ht = NEGS;
    int gH = NEGS, gE1 = NEGS, gE2 = NEGS;       // the row's cells (first chunk) for a fast successor
    for (int c0 = 0; c0 < wd; c0 += 64) {
      const int j = beg + c0 + lane;
      const bool act = j <= end;
      int ht9, E1v, E2v; unsigned d = 0, pby = 0, dw = 0;
      if (v == SRC) { ht9 = (j == 0) ? 0 : NEGS; E1v = E2v = NEGS; d = 8; }
      else {
        int kM = INT32_MIN, kE1 = INT32_MIN, kE2 = INT32_MIN;
        // ring predecessors: unconditional reads on clamped addresses, all issued before the first use, masked afterwards
        int hd_[4], hp_[4], e1_[4], e2_[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int sl = (k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3) & (PR - 1);
          const int o = j - pb_[k];
          const int oc = min(max(o, 0), PW - 1), om = min(max(o - 1, 0), PW - 1);
          hd_[k] = L.H[sl][om]; hp_[k] = L.H[sl][oc]; e1_[k] = L.E1[sl][oc]; e2_[k] = L.E2[sl][oc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k < nin) {
            const int pi = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
            const int b = pb_[k], e = pe_[k];
            const bool vd = j > 0 && j - 1 >= b && j - 1 <= e, vp = j >= b && j <= e;
            int hd = NEGS, hp = NEGS, e1p = NEGS, e2p = NEGS;
            if (ring_[k]) {
              hd = vd ? hd_[k] : NEGS; hp = vp ? hp_[k] : NEGS; e1p = vp ? e1_[k] : NEGS; e2p = vp ? e2_[k] : NEGS;
            } else {
              const int po = c.rowm()[3 * pi + 2];
              if (vd) hd = c.H()[po + (j - 1 - b)];
              if (vp) { hp = c.H()[po + (j - b)]; e1p = c.E1()[po + (j - b)]; e2p = c.E2()[po + (j - b)]; }
            }
            kM = max(kM, hd + (511 - k));
            kE1 = max(kE1, max(hp - oe1_9 + (511 - 2 * k), e1p - e1_9 + (510 - 2 * k)));
            kE2 = max(kE2, max(hp - oe2_9 + (511 - 2 * k), e2p - e2_9 + (510 - 2 * k)));
          }
        }
        for (int k = 4; k < nin; ++k) {
          const int pi = PRED_IDX(k);
          int hd = NEGS, hp = NEGS, e1p = NEGS, e2p = NEGS;
          const int sl = pi & (PR - 1);
          int4 m = L.meta[sl];
          asm volatile("" : "+v"(m.x), "+v"(m.y));
          int b = m.x, e = m.y;
          if (idx - pi >= PR) { b = c.rowm()[3 * pi]; e = c.rowm()[3 * pi + 1] & 0x0fffffff; }
          if (idx - pi < PR && e - b + 1 <= PW) {
            if (j > 0 && j - 1 >= b && j - 1 <= e) hd = L.H[sl][j - 1 - b];
            if (j >= b && j <= e) { hp = L.H[sl][j - b]; e1p = L.E1[sl][j - b]; e2p = L.E2[sl][j - b]; }
          } else {
            const int po = c.rowm()[3 * pi + 2];
            if (j > 0 && j - 1 >= b && j - 1 <= e) hd = c.H()[po + (j - 1 - b)];
            if (j >= b && j <= e) { hp = c.H()[po + (j - b)]; e1p = c.E1()[po + (j - b)]; e2p = c.E2()[po + (j - b)]; }
          }
          kM = max(kM, hd + (511 - k));
          kE1 = max(kE1, max(hp - oe1_9 + (511 - 2 * k), e1p - e1_9 + (510 - 2 * k)));
          kE2 = max(kE2, max(hp - oe2_9 + (511 - 2 * k), e2p - e2_9 + (510 - 2 * k)));
        }
        int qc = 7;
        if (act && j > 0) qc = qlds ? (int)((L.qpk[(j - 1) >> 4] >> (((j - 1) & 15) * 2)) & 3) : c3_code_at(c.pk, qb + j - 1);
        const int M9 = (j > 0) ? (kM & ~511) + ((vb == qc) ? mt9 : mm9) : NEGS;
        E1v = kE1 & ~511; E2v = kE2 & ~511;
        const int k2 = max(max(M9 + 2, E1v + 1), E2v);
        ht9 = k2 & ~511;
        const unsigned mp = 511u - ((unsigned)kM & 511u), c1 = 511u - ((unsigned)kE1 & 511u), c2 = 511u - ((unsigned)kE2 & 511u);
        d = ((~c1) & 1u) | (((~c2) & 1u) << 1) | (((unsigned)k2 & 3u) << 2);
        pby = mp | ((c1 >> 1) << 2) | ((c2 >> 1) << 4);
        dw = mp | (c1 << 8) | (c2 << 17) | ((unsigned)(2 - (k2 & 3)) << 26);             // word format (> 4 predecessors)
      }
      // horizontal states: F[j] = max_{beg<=k<j} ht[k] - o - e*(j-k)
      const int htm = act ? ht9 : NEG2S;
      const int cl1 = le1 + e1_9 * c0, cl2 = le2 + e2_9 * c0;                           // e * (column - beg)
      int s1 = htm + cl1, s2 = htm + cl2, s3 = htm;
      wave_scan_max3(s1, s2, s3);
      const int px1 = max(wave_shr1(s1, NEG2S), carry1), px2 = max(wave_shr1(s2, NEG2S), carry2);
      const int htl = wave_shr1(htm, prev_ht);
      int f1, f2; unsigned f1x = 0, f2x = 0;
      if (j == beg) { f1 = f2 = NEG2S; }
      else {
        f1 = px1 - o1_9 - cl1; f2 = px2 - o2_9 - cl2;
        f1x = f1 != htl - oe1_9; f2x = f2 != htl - oe2_9;
      }
      carry1 = max(carry1, wave_bcast(s1, 63)); carry2 = max(carry2, wave_bcast(s2, 63));
      prev_ht = wave_bcast(htm, 63);
      const int k3 = max(max(ht9 + 2, f1 + 1), f2);
      const int h9 = k3 & ~511;
      d |= (((unsigned)k3 & 3u) << 4) | (f1x << 6) | (f2x << 7);
      dw |= ((unsigned)(2 - (k3 & 3)) << 28) | (f1x << 30) | (f2x << 31);
      if (c0 == 0) { gH = act ? h9 : NEGS; gE1 = act ? E1v : NEGS; gE2 = act ? E2v : NEGS; }
      if (act) {
        const int ci = c0 + lane;
        if (ty == 2) c.D()[ro + ci] = dw;
        else { c.D8()[(unsigned)(ro + ci)] = (uint8_t)d; if (ty == 1) c.P8()[(unsigned)(ro + ci)] = (uint8_t)pby; }
        if (inl) { L.H[slot][ci] = h9; L.E1[slot][ci] = E1v; L.E2[slot][ci] = E2v; }
        if (toglobal) { c.H()[ro + ci] = h9; c.E1()[ro + ci] = E1v; c.E2()[ro + ci] = E2v; }
      }
      // row maximum == maximum of Ht, attained in the same columns (see the fast row)
      const int cmx = wave_bcast(s3, 63);
      if (cmx > best) { best = cmx; const unsigned long long mm = __ballot(htm == cmx); bl = beg + c0 + __builtin_ctzll(mm); br = beg + c0 + 63 - __builtin_clzll(mm); }
      else if (cmx == best) { const unsigned long long mm = __ballot(htm == cmx); if (mm) br = beg + c0 + 63 - __builtin_clzll(mm); }
    }
    // leftmost / rightmost argmax -> band hints read by the successors
    const int left = wd > 0 ? bl : 0, right = wd > 0 ? br : 0;
    if (lane == 0) {
      L.meta[slot] = make_int4(beg, end, left, right);
      { int* rm = c.rowm() + 3 * idx; rm[0] = beg; rm[1] = end | (ty << 28); rm[2] = ro; }
      if (far) { c.mpl()[idx] = left; c.mpr()[idx] = right; }
    }
    u_beg = beg; u_end = end; u_left = left; u_right = right; u_ncell = ncell;
#ifdef C3_PHASE_PROF
    { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_[11] += t_ - row_t0; row_t0 = t_; }
#endif
    pH = gH; pE1 = gE1; pE2 = gE2; pv_ok = wd <= 64;
  }
  }
  WSYNC();
  *cells += wave_first(u_ncell);
  PH_MARK(1)
  // ---- end cell: best predecessor of the sink at column Q (first maximum in in-edge order)
  int bi = -1, bs = INT32_MIN;
  for (int k = 0; k < c.n_in()[SNK]; ++k) {
    const int pi = c.index()[c.in_from()[EI(SNK, k)]];
    const int pb = c.rowm()[3 * pi], pe = c.rowm()[3 * pi + 1] & 0x0fffffff;
    const int hh = (Q < pb || Q > pe) ? NEGS : c.H()[c.rowm()[3 * pi + 2] + (Q - pb)];
    if (hh > bs) { bs = hh; bi = pi; }
  }
  if (bi < 0 || bs <= NEGS / 2) return -1;
  // ---- traceback.  vq[q] = graph node aligned to base q (-1 = insertion); deletions leave no trace.
  // The walk needs, per row it crosses, the row record (band, cell offset, format), the node and its first four
  // predecessor rows, and the direction cell(s) around the column where the path crosses the row.  All of that is fetched
  // 64 ROWS AT A TIME: lane k owns row it-k, loads its records (coalesced) and a 32-byte window of its direction bytes
  // (and predecessor bytes) centred on the diagonal through the current cell, and parks the windows in LDS (the DP ring is
  // free by now).  Inside such a block every step is an LDS read: in state H the wave checks 64 cells down the diagonal at
  // once (lane k: "match move from the previous row"?), consumes the run, and resolves the cell that breaks it with the
  // scalar state machine from that lane's registers.  One memory round trip per 64 rows instead of one per break.
  int* vq = c.mpl();
  int rc = 0;
  {
    uint8_t* WD = (uint8_t*)&L.H[0][0];          // [64][32] direction-byte windows
    uint8_t* WP = (uint8_t*)&L.E1[0][0];         // [64][32] predecessor-byte windows
    int i = bi, j = Q, st = 0;   // st: 0 H, 1 Ht, 2 E1, 3 E2, 4 F1, 5 F2
    while (!(i == 0 && j == 0) && rc == 0) {
      const int it = i, jt = j;
      const int rk = it - lane;
      const int ic = max(rk, 0);
      const int b = c.rowm()[3 * ic], et = c.rowm()[3 * ic + 1], ro = c.rowm()[3 * ic + 2];
      const uint4 A = c.descA()[ic], B = c.descB()[ic];
      const int e = et & 0x0fffffff, ty = (int)((unsigned)et >> 28);
      const bool adj = rk >= 1 && ((A.z >> 8) & 0xff) == 1 && (int)B.x == rk - 1;       // one predecessor, the row above
      // window of 32 cells, 4-byte aligned in the arena, around column jt - lane (clamped into the slot's arena)
      int a0 = ro + (jt - lane - 12 - b);
      a0 = min(max(a0, 0), c.cells_cap - 32) & ~3;
      const int w0 = a0 - ro + b;                                                          // column of window byte 0
      {
        uint4 x0, x1;
        const uint32_t* src = (const uint32_t*)(c.D8() + a0);
        x0 = make_uint4(src[0], src[1], src[2], src[3]); x1 = make_uint4(src[4], src[5], src[6], src[7]);
        uint4 y0 = make_uint4(0, 0, 0, 0), y1 = y0;
        if (ty == 1) { const uint32_t* sp = (const uint32_t*)(c.P8() + a0); y0 = make_uint4(sp[0], sp[1], sp[2], sp[3]); y1 = make_uint4(sp[4], sp[5], sp[6], sp[7]); }
        uint4* wd_ = (uint4*)(WD + lane * 32); wd_[0] = x0; wd_[1] = x1;
        uint4* wp_ = (uint4*)(WP + lane * 32); wp_[0] = y0; wp_[1] = y1;
      }
      WSYNC();
      // ---- steps inside the block
      for (;;) {
        const int s = it - i;                                  // lane s holds the current row
        const int jk = j - (lane - s);                         // lane k >= s looks at cell (it-k, j-(k-s))
        const bool inb = rk >= 0 && lane >= s && jk >= b && jk <= e;
        const int wo = jk - w0;
        const bool hit = inb && wo >= 0 && wo < 32 && ty != 2;
        const int wa = lane * 32 + min(max(wo, 0), 31);
        unsigned d = WD[wa], pq = WP[wa];
        int m = 0;
        if (st == 0 && i > 0 && j > 0) {
          // H <- Ht <- M through the single, adjacent predecessor: bits 2-3 == 2 and bits 4-5 == 2
          const bool ok = hit && adj && jk >= 1 && ((d >> 2) & 15u) == 10u;
          const unsigned long long bal = __ballot(ok) >> s;
          m = (~bal) ? __builtin_ctzll(~bal) : 64;
          m = min(m, 64 - s);
          if (lane >= s && lane < s + m) vq[jk - 1] = (int)A.x;
          i -= m; j -= m;
          if (i == 0 && j == 0) break;
          if (s + m >= 64) break;                              // block used up: fetch the next 64 rows
        }
        // the current cell (i, j) sits in lane cl
        const int cl = it - i;
        if (!wave_bcast((int)inb, cl)) { rc = -2; break; }
        const int v = wave_bcast((int)A.x, cl);
        const int cty = wave_bcast(ty, cl);
        const int p0 = wave_bcast((int)B.x, cl), p1 = wave_bcast((int)B.y, cl), p2 = wave_bcast((int)B.z, cl), p3 = wave_bcast((int)B.w, cl);
        unsigned mp, c1, c2, hts, hs, f1x, f2x;
        if (cty == 2 || !wave_bcast((int)hit, cl)) {
          // outside the window (the path drifted off the diagonal of this block) or a row of 32-bit words: direct loads
          const int cb = wave_bcast(b, cl), cro = wave_bcast(ro, cl);
          if (cty == 2) {
            const unsigned wv = c.D()[cro + (j - cb)];
            mp = wv & 0xff; c1 = (wv >> 8) & 0x1ff; c2 = (wv >> 17) & 0x1ff; hts = (wv >> 26) & 3; hs = (wv >> 28) & 3; f1x = (wv >> 30) & 1; f2x = wv >> 31;
          } else {
            const unsigned db = c.D8()[(unsigned)(cro + (j - cb))];
            const unsigned pb = cty == 1 ? c.P8()[(unsigned)(cro + (j - cb))] : 0u;
            mp = pb & 3; c1 = (((pb >> 2) & 3) << 1) | ((~db) & 1u); c2 = (((pb >> 4) & 3) << 1) | (((~db) >> 1) & 1u);
            hts = 2u - ((db >> 2) & 3u); hs = 2u - ((db >> 4) & 3u); f1x = (db >> 6) & 1; f2x = db >> 7;
          }
        } else {
          const unsigned db = (unsigned)wave_bcast((int)d, cl), pb = (unsigned)wave_bcast((int)pq, cl);
          mp = pb & 3; c1 = (((pb >> 2) & 3) << 1) | ((~db) & 1u); c2 = (((pb >> 4) & 3) << 1) | (((~db) >> 1) & 1u);
          hts = 2u - ((db >> 2) & 3u); hs = 2u - ((db >> 4) & 3u); f1x = (db >> 6) & 1; f2x = db >> 7;
        }
#define TB_PRED(k) ((k) == 0 ? p0 : (k) == 1 ? p1 : (k) == 2 ? p2 : (k) == 3 ? p3 : c.index()[c.in_from()[EI(v, (k))]])
        for (bool same = true; same;) {
          if (st == 0) { st = hs == 0 ? 1 : (hs == 1 ? 4 : 5); }
          else if (st == 1) {
            if (hts == 0) { if (lane == 0) vq[j - 1] = v; const int k = (int)mp; i = TB_PRED(k); --j; st = 0; same = false; }
            else st = hts == 1 ? 2 : 3;
          }
          else if (st == 2) { const int k = (int)(c1 >> 1); i = TB_PRED(k); st = (c1 & 1) ? 2 : 0; same = false; }
          else if (st == 3) { const int k = (int)(c2 >> 1); i = TB_PRED(k); st = (c2 & 1) ? 3 : 0; same = false; }
          else if (st == 4) { if (lane == 0) vq[j - 1] = -1; st = f1x ? 4 : 1; --j; same = false; }
          else { if (lane == 0) vq[j - 1] = -1; st = f2x ? 5 : 1; --j; same = false; }
        }
#undef TB_PRED
        if (i == 0 && j == 0) break;
        if (i < 0 || j < 0) { rc = -2; break; }
        // leave the block when the row is no longer in it, or when the path has drifted too far from the block's diagonal
        const int drift = (jt - j) - (it - i);
        if (it - i >= 64 || drift > 8 || drift < -8) break;
      }
      WSYNC();
    }
  }
  WSYNC();
  PH_MARK(2)
  return rc;
}

// fuse the aligned subread into the graph, parallel over its bases: vq[q] = graph row node aligned to
// base q (-1 = insertion).  Every graph node is touched by at most one base, so targets, new-node
// ids (prefix sum), anchors (prefix max) and the Q+1 edges are all independent.
__device__ int poa_fuse(Ctx& c, bool first, int qb, int Q, int* path, int lane PHA) {
  const int n_old = c.n, K = c.K;
  int* vq = c.mpl();                                  // filled by poa_align's traceback
  if (first) { for (int q = lane; q < Q; q += 64) vq[q] = -1; WSYNC(); }
  int carry_anchor = 0 /* order index of SRC */, carry_new = 0;
  for (int q0 = 0; q0 < Q; q0 += 64) {
    const int q = q0 + lane;
    const bool act = q < Q;
    const int v = act ? vq[q] : -1;
    const int cb = act ? c3_code_at(c.pk, qb + q) : 0;
    int tgt = -1, gnew = -1, anc = -1;
    if (v >= 0) {
      const int rr = c.grp()[v];
      anc = c.glast()[rr];
      if (c.base()[v] == cb) tgt = v;
      else for (int i = c.gfirst()[rr]; i <= anc; ++i) { int x = c.order()[i]; if (c.base()[x] == cb) { tgt = x; break; } }
      if (tgt < 0) gnew = rr;
    }
    const int isnew = act && tgt < 0;
    const int as = max(wave_scan_max(anc), carry_anchor);
    carry_anchor = wave_bcast(as, 63);
    const int ps = wave_scan_add(isnew);
    const int k = carry_new + ps - isnew;
    carry_new += wave_bcast(ps, 63);
    if (isnew) {
      const int id = n_old + k;
      if (id < c.Ncap) { c.base()[id] = (uint8_t)cb; c.n_in()[id] = 0; c.n_out()[id] = 0; c.grp()[id] = gnew >= 0 ? gnew : id; c.anchor()[k] = as; }
      tgt = id;
    }
    if (act) path[q] = tgt;
  }
  const int nn = n_old + carry_new;
  if (nn > c.Ncap) return -1;
  WSYNC();
  for (int q = lane; q <= Q; q += 64) {
    const int u = q == 0 ? SRC : path[q - 1], v = q == Q ? SNK : path[q];
    const int no = c.n_out()[u];
    int hit = -1;
    for (int k = 0; k < no; ++k) if (c.out_to()[EI(u, k)] == v) { hit = k; break; }
    if (hit >= 0) c.out_w()[EI(u, hit)] += 1;
    else {
      const int ni = c.n_in()[v];
      c.out_to()[EI(u, no)] = v; c.out_w()[EI(u, no)] = 1; c.n_out()[u] = no + 1;
      c.in_from()[EI(v, ni)] = u; c.n_in()[v] = ni + 1;
    }
  }
  WSYNC();
  c.n = nn;
  PH_MARK(3)
  g_reorder(c, n_old, lane, &L.H[0][0], 3 * PR * PW);      // H, E1, E2 rings are contiguous and idle after the traceback
  PH_MARK(4)
  return 0;
}

// bin/consensus.py:50-74 on code rows (4 = gap); out has msa_len bytes
__device__ void normalize_len(const uint8_t* row, int msa_len, const uint8_t* qual, int qlen, uint8_t* out) {
  int si = 0, qi = 0, n = 0;
  while (qi < qlen) {
    if (row[si] != 4) { out[n++] = qual[qi]; ++qi; ++si; }
    else if (qi == 0) { out[n++] = qual[qi]; ++si; }
    else { out[n++] = (uint8_t)(((int)qual[qi - 1] + (int)qual[qi]) / 2); ++si; }
  }
  if (n != msa_len) { int gap = 0; while (gap < msa_len && row[msa_len - 1 - gap] == 4) { out[n] = out[n - 1]; ++n; ++gap; } }
}

// bin/consensus.py:4-48 (pairwise_consensus core) on code rows (4 = gap) with the normalised qualities qa/qb of
// normalize_len: equal columns are copied, mismatches take the base of higher quality, gap runs go to the row whose
// quality sum over the run is larger.  col2t (optional) receives the draft position of every emitted column.
__device__ int pairwise_merge(const uint8_t* rowA, const uint8_t* rowB, int ncol, const uint8_t* qa, const uint8_t* qb_,
                              uint8_t* draft, int* col2t) {
  int o = 0, i = 0;
  while (i != ncol) {
    const int A = rowA[i], B = rowB[i];
    if (A == B) { if (A != 4) { if (col2t) col2t[i] = o; draft[o++] = (uint8_t)A; } }
    if (A != B && A != 4 && B != 4) { if (col2t) col2t[i] = o; draft[o++] = (uint8_t)((qa[i] > qb_[i]) ? A : B); }
    if (A == 4 || B == 4) {
      int gl = 1; const uint8_t* gs = (A == 4) ? rowA : rowB;
      for (;;) { if (i + gl >= ncol) { gl = 1; break; } if (gs[i + gl] == 4) ++gl; else break; }
      long sa = 0, sb = 0;
      for (int k = i; k < i + gl && k < ncol; ++k) { sa += qa[k]; sb += qb_[k]; }
      const uint8_t* srcr = (sa > sb) ? rowA : rowB;
      for (int k = i; k < i + gl && k < ncol; ++k) if (srcr[k] != 4) { if (col2t) col2t[k] = o; draft[o++] = srcr[k]; }
      i += gl; continue;
    }
    ++i;
  }
  return o;
}

// stand-alone stage probe: pairwise_consensus(msa_rows, subreads, quals) (bin/consensus.py:76-81) for one pair.
// rows: 2 x ncol codes; scratch: 2 x ncol bytes; out: ncol bytes of codes, out_len[0] = their number.
__global__ void k_pairwise(const uint8_t* rows, int ncol, const uint8_t* qualA, int lenA, const uint8_t* qualB, int lenB,
                           uint8_t* scratch, uint8_t* out, int* out_len) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  normalize_len(rows, ncol, qualA, lenA, scratch);
  normalize_len(rows + ncol, ncol, qualB, lenB, scratch + ncol);
  out_len[0] = pairwise_merge(rows, rows + ncol, ncol, scratch, scratch + ncol, out, nullptr);
}
extern "C" void c3k_launch_pairwise(const uint8_t* rows, int ncol, const uint8_t* qa, int la, const uint8_t* qb, int lb,
                                    uint8_t* scratch, uint8_t* out, int* out_len, hipStream_t s) {
  hipLaunchKernelGGL(k_pairwise, dim3(1), dim3(64), 0, s, rows, ncol, qa, la, qb, lb, scratch, out, out_len);
}

// 6 waves/SIMD (80 VGPRs, 26 spilled outside the row loop) with a 4-row LDS ring (6.7 KB per wave, 24 waves per CU):
// 59.4 ms per 32768 cfg2 reads against 67.8 ms at 4 waves/SIMD with the 8-row ring -- the row loop is a dependent
// chain (scan -> next row), so resident waves are what hides its latency.
__global__ __launch_bounds__(64, 6) void k_poa(PoaArgs a) {
  const int lane = wave_lane();
  const int slot = blockIdx.x;
  Ctx c;
  const size_t N = (size_t)a.Ncap;
  c.I = a.ibase + (size_t)slot * C3_POA_NI * N; c.path_ = a.pbase + (size_t)slot * a.Pcap; c.E = a.ebase + (size_t)slot * 3 * N * a.K;
  c.C = a.cellsb + (size_t)slot * 18 * (size_t)a.cells_cap; c.B8 = a.bbase + (size_t)slot * 5 * N;
  c.score_ = a.score + (size_t)slot * N; c.desc_ = a.desc + (size_t)slot * 2 * N; c.jump_ = a.jump + (size_t)slot * C3_JUMP_LEVELS * N;
  c.K = a.K; c.Ncap = a.Ncap; c.cells_cap = a.cells_cap; c.osel = 0;
  PH_DECL

  for (;;) {
    int wi = 0;
    if (lane == 0) wi = atomicAdd(a.counter, 1);
    wi = wave_first(wi);
    if (wi >= a.n_work) break;
    const int rid = a.work[wi];
    C3Info* info = &a.info[rid];
    const int ns = info->n_sub;
    const int64_t off = a.b.off[rid];
    c.pk = a.b.pk + a.b.woff[rid];
    uint8_t* draft = a.draft + off;
    int32_t* tpos = a.tpos + off;
    const uint8_t* qual = a.b.qual + off;
    long long cells = 0;
    int C = 0, fail = 0;
    if (ns == 1) {
      const int qb = info->sub_beg[0]; C = info->sub_end[0] - qb;
      for (int k = lane; k < C; k += 64) { draft[k] = (uint8_t)c3_code_at(c.pk, qb + k); tpos[qb + k] = k; }
    } else {
      // ---- build the graph, one subread at a time
      c.osel = 0;
      if (lane == 0) {
        c.base()[SRC] = 0; c.base()[SNK] = 0; c.n_in()[SRC] = c.n_out()[SRC] = c.n_in()[SNK] = c.n_out()[SNK] = 0;
        c.grp()[SRC] = SRC; c.grp()[SNK] = SNK; c.order()[0] = SRC; c.order()[1] = SNK; c.index()[SRC] = 0; c.index()[SNK] = 1;
      }
      c.n = 2;
      WSYNC();
      g_blocks(c, lane);
      int poff = 0;
      for (int s = 0; s < ns && !fail; ++s) {
        const int qb = wave_first(info->sub_beg[s]), Q = wave_first(info->sub_end[s]) - qb;
        if (s > 0) { const int rc = poa_align(c, a.p, qb, Q, lane, &cells PHP); if (rc < 0) { fail = rc == -4 ? 2 : 1; break; } }
        if (poa_fuse(c, s == 0, qb, Q, c.path() + poff, lane PHP) < 0) { fail = 2; break; }                 // node capacity
        poff += Q;
      }
      PH_MARK(9)
      if (!fail) {
        // ---- MSA columns = aligned blocks in topological order
        int ncol_acc = 0;
        for (int i0 = 0; i0 < c.n; i0 += 64) {      // column = number of block starts up to here (prefix sum)
          const int i = i0 + lane;
          int v = -1, start = 0;
          if (i < c.n) {
            v = c.order()[i];
            if (v != SRC && v != SNK) start = (i == 0) || (c.grp()[v] != c.grp()[c.order()[i - 1]]) || c.order()[i - 1] == SRC;
          }
          const int ps = wave_scan_add(start);
          if (i < c.n) c.col()[v] = (v == SRC || v == SNK) ? -1 : ncol_acc + ps - 1;
          ncol_acc += wave_bcast(ps, 63);
        }
        WSYNC();
        PH_MARK(5)
        const int ncol = ncol_acc;
        for (int i = lane; i < ncol; i += 64) c.col2t()[i] = -1;
        if (a.msa_dbg) {                       // res.msa_seq rows (codes, 4 = gap), row-major
          uint8_t* dbg = a.msa_dbg + a.msa_off[rid];
          for (int i = lane; i < ns * ncol; i += 64) dbg[i] = 4;
          WSYNC();
          int po = 0;
          for (int s = 0; s < ns; ++s) {
            const int Q = info->sub_end[s] - info->sub_beg[s];
            for (int k = lane; k < Q; k += 64) { int v = c.path()[po + k]; dbg[(size_t)s * ncol + c.col()[v]] = c.base()[v]; }
            po += Q;
          }
          if (lane == 0) a.msa_len[rid] = ncol;
        }
        WSYNC();
        if (ns == 2) {
          // rows (codes, 4 = gap) -> bin/consensus.py pairwise_consensus
          uint8_t* rowA = c.rows2(); uint8_t* rowB = c.rows2() + ncol;
          uint8_t* qa = c.rows2() + 2 * (size_t)ncol; uint8_t* qb_ = c.rows2() + 3 * (size_t)ncol;
          for (int i = lane; i < 2 * ncol; i += 64) c.rows2()[i] = 4;
          WSYNC();
          const int b0 = info->sub_beg[0], l0 = info->sub_end[0] - b0, b1 = info->sub_beg[1], l1 = info->sub_end[1] - b1;
          for (int k = lane; k < l0; k += 64) { int v = c.path()[k]; rowA[c.col()[v]] = c.base()[v]; }
          for (int k = lane; k < l1; k += 64) { int v = c.path()[l0 + k]; rowB[c.col()[v]] = c.base()[v]; }
          WSYNC();
          if (lane == 0) {
            // seqDict collision: identical subreads share the later quality (consensus.py:77-79)
            bool same = (l0 == l1);
            for (int k = 0; same && k < l0; ++k) same = c3_code_at(c.pk, b0 + k) == c3_code_at(c.pk, b1 + k);
            normalize_len(rowA, ncol, same ? qual + b1 : qual + b0, l0, qa);
            normalize_len(rowB, ncol, qual + b1, l1, qb_);
            const int o = pairwise_merge(rowA, rowB, ncol, qa, qb_, draft, c.col2t());
            c.rem()[0] = o;
          }
          WSYNC();
          C = c.rem()[0];
        } else {
          // ---- abPOA heaviest bundling: nxt[v] = out-edge of maximum weight; among edges of equal (maximum) weight the one
          // whose target has the higher downstream score, the LATER edge on equality; score[v] = weight + score[nxt[v]].
          // ONE reverse sweep over the topological order, 64 positions at a time (lane = position): every successor sits at
          // a later position, so a target outside the chunk is final (one gather), chains inside the chunk are resolved by
          // pointer doubling between lanes (ds_bpermute; score and hop count ride along), and the few nodes with a tie for
          // the maximum -- the only ones that look at scores -- are resolved one by one from the highest lane down, each
          // after the lanes above it are final.  The consensus path is then marked by one FORWARD sweep (in-chunk doubling of
          // "is on the path"), and a node's draft position is C - hops(node): no pointer chase, no jump tables.
          {
            const int n = c.n;
            int* scp = c.jump(); int* hpp = c.jump() + (size_t)c.Ncap; int* nxp = c.jump() + 2 * (size_t)c.Ncap;     // by position: score, hops to the sink, position of nxt
            int* onf = c.jump() + 3 * (size_t)c.Ncap;                                                               // by position: on the consensus path
            for (int c0 = ((n - 1) >> 6) << 6; c0 >= 0; c0 -= 64) {
              const int idx = c0 + lane;
              const bool live = idx < n;
              const int v = live ? c.order()[idx] : SNK;
              const int no = (live && v != SNK) ? c.n_out()[v] : 0;
              int bw = INT32_MIN, bt = SNK, cm = 0;
              for (int k = 0; __builtin_amdgcn_ballot_w64(k < no) != 0; ++k) {
                if (k < no) {
                  const int ww = c.out_w()[EI(v, k)];
                  if (ww > bw) { bw = ww; bt = c.out_to()[EI(v, k)]; cm = 1; } else if (ww == bw) ++cm;
                }
              }
              bool tie = cm >= 2;
              // acc / hp: score and hops from this node to ptr; ptr = position still to follow, -1 = final
              int acc = no > 0 ? bw : (v == SNK ? 0 : INT32_MIN / 2), hp = no > 0 ? 1 : 0;
              int ptr = (no > 0 && !tie) ? c.index()[bt] : -1;
              int np = ptr;                                                     // position of nxt (tie lanes: set when resolved)
              bool done = no == 0;                                              // the sink (and padding lanes)
              if (ptr >= c0 + 64) { acc += scp[ptr]; hp += hpp[ptr]; ptr = -1; done = true; }
              for (;;) {
                // pointer doubling among the lanes that know their successor; a pointer to an unresolved tie lane waits
                for (int r = 0; r < 6 && __builtin_amdgcn_ballot_w64(!done && !tie) != 0; ++r) {
                  const int src = (max(ptr, c0) - c0) << 2;
                  const int a2 = __builtin_amdgcn_ds_bpermute(src, acc), h2 = __builtin_amdgcn_ds_bpermute(src, hp);
                  const int p2 = __builtin_amdgcn_ds_bpermute(src, ptr);
                  const int d2 = __builtin_amdgcn_ds_bpermute(src, (int)done), t2 = __builtin_amdgcn_ds_bpermute(src, (int)tie);
                  if (!done && !tie && ptr >= 0) {
                    if (d2) { acc += a2; hp += h2; ptr = -1; done = true; }
                    else if (!t2) { acc += a2; hp += h2; ptr = p2; }
                  }
                }
                const unsigned long long tm = __builtin_amdgcn_ballot_w64(tie && !done);
                if (!tm) break;
                const int T = 63 - __builtin_clzll(tm);                         // highest unresolved tie lane: everything above it is final
                const int vT = __builtin_amdgcn_readlane(v, T), noT = __builtin_amdgcn_readlane(no, T);
                int tbw = INT32_MIN, tbt = SNK, tsb = 0, thp = 0, tnp = -1;
                for (int k = 0; k < noT; ++k) {
                  const int wk = c.out_w()[EI(vT, k)], tk = c.out_to()[EI(vT, k)];
                  const int pk_ = c.index()[tk];
                  int sk, hk;
                  if (pk_ >= c0 + 64) { sk = scp[pk_]; hk = hpp[pk_]; }
                  else { sk = wave_bcast(acc, pk_ - c0); hk = wave_bcast(hp, pk_ - c0); }
                  if (wk > tbw) { tbw = wk; tbt = tk; tsb = sk; thp = hk; tnp = pk_; }
                  else if (wk == tbw && tsb <= sk) { tbt = tk; tsb = sk; thp = hk; tnp = pk_; }
                }
                if (lane == T) { acc = tbw + tsb; hp = 1 + thp; np = tnp; done = true; tie = false; ptr = -1; }
              }
              if (live) { scp[idx] = acc; hpp[idx] = hp; nxp[idx] = np; onf[idx] = 0; }
              WSYNC();
            }
            C = hpp[0] - 1;                                                     // SRC sits at position 0: hops to the sink - 1 = consensus length
            if (lane == 0) onf[0] = 1;
            WSYNC();
            for (int c0 = 0; c0 < n && C > 0; c0 += 64) {
              const int idx = c0 + lane;
              const bool live = idx < n;
              int on = live ? onf[idx] : 0;
              const int np = live ? nxp[idx] : -1;
              int J = (np >= c0 && np < c0 + 64) ? np - c0 : -1;                  // successor lane inside the chunk
              for (int r = 0; r < 6; ++r) {
                L.qpk[lane] = 0;
                WSYNC();
                if (on && J >= 0) L.qpk[J] = 1;
                WSYNC();
                on |= (int)L.qpk[lane];
                const int J2 = __builtin_amdgcn_ds_bpermute(max(J, 0) << 2, J);
                J = J >= 0 ? J2 : -1;
                WSYNC();
              }
              if (on && np >= c0 + 64) onf[np] = 1;
              if (on && live) {
                const int v = c.order()[idx];
                if (v != SRC && v != SNK) { const int pp = C - hpp[idx]; draft[pp] = c.base()[v]; c.col2t()[c.col()[v]] = pp; }
              }
              WSYNC();
            }
            if (C < 0) C = 0;
            WSYNC();
          }

        }
        PH_MARK(6)
        // ---- subread -> draft coordinates
        int poff2 = 0;
        for (int s = 0; s < ns; ++s) {
          const int qb = info->sub_beg[s], Q = info->sub_end[s] - qb;
          for (int k = lane; k < Q; k += 64) tpos[qb + k] = c.col2t()[c.col()[c.path()[poff2 + k]]];
          poff2 += Q;
        }
      }
    }
    WSYNC();
    PH_MARK(7)
    // fail == 2: the scratch of this slot was too small (cells / nodes).  The first pass runs with scratch sized for the
    // TYPICAL alignment (more resident waves); such reads are queued and redone by a second launch with worst-case scratch
    const bool redo = fail == 2 && a.overflow != nullptr;
    if (lane == 0) {
      if (redo) a.overflow[atomicAdd(a.counter + 4, 1)] = rid;
      else {
        info->draft_len = C;
        if (fail) { info->status = C3_ST_LIMIT; info->draft_len = 0; }
        else if (C == 0) info->status = C3_ST_NO_CONSENSUS;
        atomicAdd((unsigned long long*)(a.counter + 2), (unsigned long long)cells);
      }
    }
    WSYNC();
  }
  PH_FLUSH(a.phases)
}

extern "C" void c3k_launch_poa(const PoaArgs* a, int slots, hipStream_t stream) {
  hipLaunchKernelGGL(k_poa, dim3(slots), dim3(64), 0, stream, *a);
}
