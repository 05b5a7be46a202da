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

#ifdef C3_POA_MW
// k_poa_mw.hip compiles this file a second time for the LAST pass: a workgroup of C3_POA_MW waves per read.  Wave 0 runs every
// phase exactly as the single-wave kernel does (its WSYNC is a wave-local fence there, not a barrier); the wide general rows
// are computed by all waves together (mw_chunks), two phases between three real barriers.
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
// the barrier between the phases of a wide row: cells written by one wave are read by another -- all on one CU, through one
// vector L1: workgroup scope is enough (an agent-scope release writes the L2 back, microseconds per row)
#define MW_BARRIER() __syncthreads()
#else
#define WSYNC() __syncthreads()
#endif
// adjacency slot k of node v.  Slot-major (all first edges, then all second edges, ...): nearly every node has one or two
// edges, so the arrays a wave actually touches are contiguous runs of Ncap ints instead of one 64-byte line per node
#define EI(v, k) ((size_t)(k) * (size_t)c.Ncap + (size_t)(v))
#define C3_POA_NI 19       // int arrays of Ncap per slot in Ctx::I
#define SRC 0
#define SNK 1


// Per-slot scratch of k_poa.  Only a few base pointers are kept live; every array is base + constant multiple of Ncap
// (or Ncap*K, cells_cap), which keeps the uniform state in SGPRs instead of spilling it into VGPR lanes.
struct Ctx {
  int* I; int* E; char* C; uint8_t* B8; long long* score_; uint4* desc_; int* jump_; int* path_;
  int K, n, Ncap, cells_cap;
  int far_shift;             // far arena (32-bit H, E1, E2, direction words of the FEW rows that keep them): cells_cap >> far_shift cells, behind the byte cells
  int osel;                  // which of the two order buffers is current (g_reorder writes the other one and flips)
  int rb_span;               // RB_HI16 - RB_LO16 (smaller under the C3_DEBUG_POA_RBSPAN test hook: the base moves every few rows)
  const uint32_t* pk;        // packed read
#ifdef C3_DEBUG_PUNT
  unsigned long long* dbg;
#endif
#ifdef C3_MW_CHECK
  unsigned long long* chk;
#endif
#define CTX_I(name, k) __device__ __forceinline__ int* name() const { return I + (size_t)(k) * Ncap; }
  CTX_I(n_in, 0) CTX_I(n_out, 1) CTX_I(grp, 2) CTX_I(index, 5) CTX_I(gfirst, 6) CTX_I(glast, 7)
  CTX_I(rem, 8) CTX_I(mpl, 9) CTX_I(mpr, 10) CTX_I(rowm, 11) /* 3 ints per row: band begin, band end, cell offset (blocks 11..13) */ CTX_I(anchor, 14) CTX_I(col, 15)
  CTX_I(col2t, 16) CTX_I(nxt, 17) CTX_I(foff, 18) /* per row: offset of its cells in the far arena (far / wide / many-predecessor rows only) */
#undef CTX_I
  __device__ __forceinline__ int* order() const { return I + (size_t)(3 + osel) * Ncap; }
  __device__ __forceinline__ int* order2() const { return I + (size_t)(4 - osel) * Ncap; }
  __device__ __forceinline__ int* in_from() const { return E; }
  __device__ __forceinline__ int* out_to() const { return E + (size_t)Ncap * K; }
  __device__ __forceinline__ int* out_w() const { return E + 2 * (size_t)Ncap * K; }
  // 32-bit cells: only rows with a successor beyond the LDS ring (or the sink), rows wider than a ring slot and rows with more than
  // four predecessors keep them -- a few per cent of the cells; they have an arena and a cell counter of their own (round 4: the
  // arena used to reserve 16 bytes for EVERY cell, 42 of 44 MB per slot with 6 kb subreads, and the slots no longer fitted)
  __device__ __forceinline__ int far_cap() const { return cells_cap >> far_shift; }
  __device__ __forceinline__ int32_t* H() const { return (int32_t*)(C + 2 * (size_t)cells_cap); }
  __device__ __forceinline__ int32_t* E1() const { return (int32_t*)(C + 2 * (size_t)cells_cap + 4 * (size_t)far_cap()); }
  __device__ __forceinline__ int32_t* E2() const { return (int32_t*)(C + 2 * (size_t)cells_cap + 8 * (size_t)far_cap()); }
  __device__ __forceinline__ uint32_t* D() const { return (uint32_t*)(C + 2 * (size_t)cells_cap + 12 * (size_t)far_cap()); }
  __device__ __forceinline__ uint8_t* D8() const { return (uint8_t*)C; }                                 // direction bytes (rows of <= 4 predecessors)
  __device__ __forceinline__ uint8_t* P8() const { return (uint8_t*)(C + (size_t)cells_cap); }           // predecessor bytes (rows of 2..4 predecessors)
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
// also go to the global arena (32-bit); the direction bytes always do.
//
// Round 4: the ring and the fast / near rows work on 16-BIT CELLS.  A cell is kept relative to a per-row base (the row's
// own maximum; the absolute base of every ring row sits beside its band record), as a BIASED UNSIGNED key score*8 + 3 tag
// bits: every value is a non-negative number below 2^16, so the cheap 16-bit VOP2 max (v_max_u16: 2.3 cycles against 4.2 for
// the 32-bit max) and plain 32-bit adds / DPP max-scans can be mixed freely (a zero-extended unsigned key IS its 32-bit
// value).  Measured with the oracle on the config shapes (tools/poa_row_stats.py): no reachable cell lies more than 391
// below its row's maximum and consecutive row maxima differ by at most 38, against +-4095 of range.  Unreachable cells are a
// low constant (NEG16); a ring row is padded with it on both sides, so a predecessor cell outside the predecessor's band
// reads as unreachable WITHOUT any validity compare (clamped offset -> pad).  Exactness is guarded, not assumed: every stored
// H must be either plainly reachable (>= GLO16) or plainly unreachable (<= ZHI16); a read with a value in between (a
// reachable score that fell out of range, or a chain of unreachable cells that drifted up) is handed to the second pass,
// whose kernel instance (W32) computes every row with the 32-bit general row and no ring at all.
#ifndef C3_NEAR
#define C3_NEAR 1
#endif
#ifndef C3_STEADY
#define C3_STEADY 1            /* steady rows (round 6); 0 = the round-5 row loop, instruction for instruction */
#endif
#ifndef C3_NEAR1_TYPE0
#define C3_NEAR1_TYPE0 1       /* near rows with ONE predecessor keep no predecessor byte (it is always 0): row type 0, as the fast rows (round 6) */
#endif
#ifndef C3_STEADY_RING
#define C3_STEADY_RING 0       /* 1: a steady run may start at a row whose one predecessor sits 2 .. PR-1 rows up (cells from the LDS ring).  Built, parity-green,
                                  SLOWER: cfg2 +6.4 %, cfg3 +3.9 %, cfg4 +2.5 % on distinct reads -- every row kind got slower with it, steady rows 1 977 -> 2 319 cycles, fast rows
                                  2 934 -> 3 112, although their listings are unchanged: profiles/r06_ab_poa_steady_rows.txt */
#endif
#ifndef C3_STEADY_FORK
#define C3_STEADY_FORK 0       /* steady rows also for rows that a row further down reads (they write the ring at once) */
#endif
#ifndef C3_EXP_CALL_NARROW
#define C3_EXP_CALL_NARROW 0   /* experiment: poa_align as a real call in the NARROW instances too */
#endif
// Ring geometry, two instances of the kernel (same LDS footprint): NARROW = 6 rows of 128 cells (two chunks of 64; a predecessor
// up to 5 rows back is read from LDS: 99.7 % of them at cfg4) for subreads up to 1792 bases, whose bands stay below 128 columns,
// with the whole packed subread in LDS; WIDE = 3 rows of 256 cells (up to four chunks) with a SLIDING query window of 960
// bases for longer subreads (w = 10 + Q/100: a 6 kb insert has bands of 145-250 columns; its graphs hold 3-5 subreads, so a
// predecessor is rarely more than two rows back).  The LDS footprint decides the kernel's speed beyond what the wave count
// explains (measured: 5.6 KB per wave 37.2 ms, 6.6 KB 39.5 ms per 32 768 cfg2 reads, 24 waves per CU fitting both ways), so
// the WIDE geometry is cut to what the NARROW one needs anyway.
#define PADL 3      // unreachable cells on both sides of a ring row
#define PADR 3
#ifndef C3_NARROW_PR
#define C3_NARROW_PR 6
#endif
#define RING_CELLS_N (C3_NARROW_PR * (PADL + 128 + PADR))      // 804
#ifndef C3_WIDE_PW
#define C3_WIDE_PW 256
#endif
#define RING_CELLS_W (3 * (PADL + C3_WIDE_PW + PADR))      // 786
#define PQW_N 112   // NARROW: packed query words in LDS (1792 bases = the whole subread); they sit behind its (smaller) rings
#define PQW_W 60    // WIDE: a window of 960 bases that follows the band (60 words: the struct is 6656 bytes = 13 LDS granules of 512, 24 waves per CU)
struct PoaLds { unsigned short ring[3 * (RING_CELLS_W > RING_CELLS_N + PQW_N * 2 / 3 + 2 ? RING_CELLS_W : RING_CELLS_N + PQW_N * 2 / 3 + 2)]; int4 meta[6] /* band begin, end, leftmost / rightmost argmax */; int base8[8] /* absolute base * 8 of the row */; unsigned qpk[PQW_W]; };
__shared__ PoaLds L;     // file scope: accesses stay in the LDS address space (ds_*, lgkmcnt only)
static_assert(sizeof(PoaLds) <= 6656, "24 waves per CU need <= 13 LDS granules of 512 bytes per wave");
static_assert(3 * RING_CELLS_N * 2 + PQW_N * 4 <= sizeof(L.ring), "the NARROW query copy must fit behind the NARROW rings");
#define POA_LDS_INTS ((int)(3 * RING_CELLS_N * sizeof(unsigned short) / sizeof(int)))      // ring bytes used as scratch by the graph phases (the smaller geometry's: 4.7 KB)

// general rows (and the global arena): scores are carried as score*512 (+ a 9-bit tag while candidates compete): one v_max per
// candidate implements "highest score, first candidate in order".  Unreachable cells use -(2^20) score units.
#define S9(x) ((x) * 512)
#define NEGS (-(1 << 29))
#define NEG2S (-(1 << 30))
#ifdef C3_POA_MW
// ---- the wide general rows of the last pass, by all waves of the workgroup.  A read whose band has blown up to the width of the
// subread (a missed or an extra peak: one subread twice as long as the others) has tens of thousands of rows of up to Q columns;
// one wave takes 300-600 ms for it.  The row's 64-column chunks are dealt round-robin to the waves.  Phase A: the vertical /
// diagonal states of the chunk (Ht, E1, E2, their tags) from the predecessor rows -- no dependence between chunks -- and the
// chunk's scan aggregates into LDS; phase B: the horizontal states from the aggregates of the chunks to the left (what the
// single-wave loop carries from chunk to chunk), the final H and direction cells.  Bit for bit the arithmetic of the loop below.
#define MW_CH 512          /* chunks of a row (rows of more columns take the single-wave loop) */
#define MW_MINCH 3         /* ... and rows of fewer chunks than this as well: three barriers cost more than two chunks */
#define MW_NP 16           /* predecessors of a row handed to the other waves */
struct MwTask {
  int cmd, beg, end, nin, vb, fo, ro, ty, qb, pad_;
  unsigned long long pk;
  int pb[MW_NP], pe[MW_NP], po[MW_NP];
  int agg1[MW_CH], agg2[MW_CH], lastht[MW_CH], cmx[MW_CH], fpos[MW_CH], lpos[MW_CH];
};
__shared__ MwTask MWT;
struct MwConst { int mt9_, mm9_, e1_9_, e2_9_, o1_9_, o2_9_; };
template <int PHASE>
__device__ void mw_chunks(const Ctx& c, const MwConst& kc, int wave, int lane) {
  const int beg = MWT.beg, end = MWT.end, nin = MWT.nin, vb = MWT.vb, fo = MWT.fo, ro = MWT.ro, ty = MWT.ty, qb = MWT.qb;
  const uint32_t* pk = (const uint32_t*)MWT.pk;
  const int wd = end - beg + 1, nch = (wd + 63) >> 6;
  const int oe1 = kc.o1_9_ + kc.e1_9_, oe2 = kc.o2_9_ + kc.e2_9_;
  for (int ci = wave; ci < nch; ci += C3_POA_MW) {
    const int c0 = 64 * ci;
    const int j = beg + c0 + lane;
    const bool act = j <= end;
    const int cix = c0 + lane;
    const int cl1 = kc.e1_9_ * (lane + c0), cl2 = kc.e2_9_ * (lane + c0);             // e * (column - beg)
    if (PHASE == 0) {
      int kM = INT32_MIN, kE1 = INT32_MIN, kE2 = INT32_MIN;
      for (int k = 0; k < nin; ++k) {
        const int b = MWT.pb[k], e = MWT.pe[k], po = MWT.po[k];
        const bool vd = j > 0 && j - 1 >= b && j - 1 <= e, vp = j >= b && j <= e;
        int hd = NEGS, hp = NEGS, e1p = NEGS, e2p = NEGS;
        if (vd) hd = c.H()[po + (j - 1 - b)];
        if (vp) { hp = c.H()[po + (j - b)]; e1p = c.E1()[po + (j - b)]; e2p = c.E2()[po + (j - b)]; }
        kM = max(kM, hd + (511 - k));
        kE1 = max(kE1, max(hp - oe1 + (511 - 2 * k), e1p - kc.e1_9_ + (510 - 2 * k)));
        kE2 = max(kE2, max(hp - oe2 + (511 - 2 * k), e2p - kc.e2_9_ + (510 - 2 * k)));
      }
      int qc = 7;
      if (act && j > 0) qc = c3_code_at(pk, qb + j - 1);
      const int M9 = (j > 0) ? (kM & ~511) + ((vb == qc) ? kc.mt9_ : kc.mm9_) : NEGS;
      const int E1v = kE1 & ~511, E2v = kE2 & ~511;
      const int k2 = max(max(M9 + 2, E1v + 1), E2v);
      const int ht9 = k2 & ~511;
      const unsigned mp = 511u - ((unsigned)kM & 511u), c1 = 511u - ((unsigned)kE1 & 511u), c2 = 511u - ((unsigned)kE2 & 511u);
      const unsigned d = ((~c1) & 1u) | (((~c2) & 1u) << 1) | (((unsigned)k2 & 3u) << 2);
      const unsigned pby = mp | ((c1 >> 1) << 2) | ((c2 >> 1) << 4);
      const unsigned dw = mp | (c1 << 8) | (c2 << 17) | ((unsigned)(2 - (k2 & 3)) << 26);
      const int htm = act ? ht9 : NEG2S;
      int s1 = htm + cl1, s2 = htm + cl2, s3 = htm;
      wave_scan_max3(s1, s2, s3);
      // (read out of the scans HERE, by every lane: taken inside the one-lane branch below, the scans' DPP steps were sunk into it
      // with it and ran with one lane enabled)
      const int cm = wave_bcast(s3, 63), g1 = wave_bcast(s1, 63), g2 = wave_bcast(s2, 63), lh = wave_bcast(htm, 63);
      const unsigned long long mm = __ballot(htm == cm);
      if (lane == 0) {
        MWT.agg1[ci] = g1; MWT.agg2[ci] = g2; MWT.lastht[ci] = lh; MWT.cmx[ci] = cm;
        MWT.fpos[ci] = beg + c0 + __builtin_ctzll(mm); MWT.lpos[ci] = beg + c0 + 63 - __builtin_clzll(mm);
      }
      if (act) {
        if (ty == 2) c.D()[fo + cix] = dw;
        else { c.D8()[(unsigned)(ro + cix)] = (uint8_t)d; if (ty == 1) c.P8()[(unsigned)(ro + cix)] = (uint8_t)pby; }
        c.H()[fo + cix] = ht9; c.E1()[fo + cix] = E1v; c.E2()[fo + cix] = E2v;          // (H: Ht until phase B)

      }
    } else {
      // what the chunks to the left hand over: the maxima of their two scans, the last Ht
      int a1 = NEG2S, a2 = NEG2S;
      for (int x = lane; x < ci; x += 64) { a1 = max(a1, MWT.agg1[x]); a2 = max(a2, MWT.agg2[x]); }
      const int carry1 = wave_max(a1), carry2 = wave_max(a2);
      const int prev_ht = ci > 0 ? MWT.lastht[ci - 1] : NEGS;
      const int ht9 = act ? c.H()[fo + cix] : 0;
      const int htm = act ? ht9 : NEG2S;
      int s1 = htm + cl1, s2 = htm + cl2, s3 = htm;
      wave_scan_max3(s1, s2, s3);
      const int px1 = max(wave_shr1(s1, NEG2S), carry1), px2 = max(wave_shr1(s2, NEG2S), carry2);
      const int htl = wave_shr1(htm, prev_ht);
      int f1, f2; unsigned f1x = 0, f2x = 0;
      if (j == beg) { f1 = f2 = NEG2S; }
      else {
        f1 = px1 - kc.o1_9_ - cl1; f2 = px2 - kc.o2_9_ - cl2;
        f1x = f1 != htl - oe1; f2x = f2 != htl - oe2;
      }
      const int k3 = max(max(ht9 + 2, f1 + 1), f2);
      const int h9 = k3 & ~511;
      if (act) {
        if (ty == 2) c.D()[fo + cix] |= ((unsigned)(2 - (k3 & 3)) << 28) | (f1x << 30) | (f2x << 31);
        else c.D8()[(unsigned)(ro + cix)] |= (uint8_t)((((unsigned)k3 & 3u) << 4) | (f1x << 6) | (f2x << 7));
        c.H()[fo + cix] = h9;
      }
    }
  }
}
#endif
// fast / near rows and the ring: biased unsigned 16-bit keys, (score - row base) * 8 + BIAS16 (+ 3 tag bits while candidates compete)
#define S3(x) ((x) * 8)
#define BIAS16 32768
#define NEG16 6000          // unreachable
#define NEG2_16 2000        // "no left neighbour" of the horizontal states / idle lanes of the scans
#define FLOOR16 5000        // stored H never sinks below this (a chain of mismatching unreachable cells loses 4 a row)
#define ZHI16 12000         // a stored H up to here is unreachable ...
#define GLO16 14400         // ... from here on reachable; in between: not representable
#define ZLO16 13200         // conversion threshold between the two (E1 / E2 sit up to 25 units below an H)
#define RB_LO16 (BIAS16 - 400 * 8)   // a row whose maximum leaves [RB_LO16, RB_HI16] moves its base to the maximum (about every 600 rows: the
#define RB_HI16 (BIAS16 + 3000 * 8)  // score grows by at most 5 a row); all other rows keep the base of the row before them.  A reachable
                                     // cell may sit (RB_LO16 - GLO16) / 8 = 1896 units below its row maximum (measured: 391 at cfg2-cfg4, 760 with 6 kb inserts)
// (volatile: hipcc sinks a plain asm that feeds one arm of a select into an EXEC-masked branch -- two scalar branches per row)
#ifdef C3_EXP_ASMNV
#define C3_ASMV
#else
#define C3_ASMV volatile
#endif
__device__ __forceinline__ int maxu16(int a, int b) { int d; asm C3_ASMV("v_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ int minu16(int a, int b) { int d; asm C3_ASMV("v_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ int cv_16to9(int v, int b8) { return v >= ZLO16 ? (int)((unsigned)(v - BIAS16 + b8) << 6) : NEGS; }
__device__ __forceinline__ int cv_9to16(int x9, int b8, int& bad) {
  if (x9 <= NEGS / 2) return NEG16;
  const int r = (x9 >> 6) - b8 + BIAS16;
  if (r < GLO16 || r > 62000) bad = 1;
  return r;
}

// direction word (device-internal): mp[0..7] | e1code[8..16] | e2code[17..25] | hts[26..27] | hs[28..29]
// | f1x[30] | f2x[31];  e?code = 2*pred + ext;  hts: 0 M, 1 E1, 2 E2;  hs: 0 Ht, 1 F1, 2 F2

// Row descriptors of one alignment AND the "remaining length" of every node in ONE reverse sweep over the topological order,
// 64 positions at a time (lane = position).  rem[v] = hops from v to the sink along its heaviest out-edge (first maximum in
// out-list order) - 1; the chain of a node continues at a LATER position, so when the chunks are taken from the last to
// the first, a target outside the chunk is already final (one gather from hops[]), and the chains inside a chunk are
// resolved by pointer doubling between lanes (ds_bpermute, at most six rounds).  Replaces log2(n) rounds of pointer jumping
// over four node arrays in memory by four dependent memory levels per chunk.
// descriptor A: x = node, y = ring slots (position mod PR) of the first four predecessors, 4 bits each, z = base | nin<<8 | far<<16 | (nin>4)<<17 | fast-row candidate<<18 | sink<<19 | fork<<20 | one predecessor in the ring<<21 | its slot<<22, w = qr = Q - rem;
// descriptor B: positions of the first four predecessors.  hops[] (by position) lives in col() (free until the MSA columns).
__device__ void poa_sweep_desc(Ctx& c, int lane, int Q, bool qlds, const int PR) {
  const int n = c.n;
  int* hops = c.col();
  for (int c0 = ((n - 1) >> 6) << 6; c0 >= 0; c0 -= 64) {
    const int idx = c0 + lane;
    const bool live = idx < n;
    const int v = live ? c.order()[idx] : SNK;
    const int nin = live ? c.n_in()[v] : 0, nout = live ? c.n_out()[v] : 0;
    int bw = INT32_MIN, bt = SNK;
    unsigned far = 0, fork = 0;
    for (int k = 0; __builtin_amdgcn_ballot_w64(k < nout) != 0; ++k) {
      if (k < nout) {
        const int t = c.out_to()[EI(v, k)], ww = c.out_w()[EI(v, k)];
        if (ww > bw) { bw = ww; bt = t; }
        const int ti = t == SNK ? INT32_MAX : c.index()[t];
        if (ti - idx > PR - 1) far = 1;
        if (ti != idx + 1) fork = 1;                            // a reader of this row other than the row below (or the sink)
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
      uint4 A; A.x = (unsigned)v;
      A.y = (p[0] % PR) | ((p[1] % PR) << 4) | ((p[2] % PR) << 8) | ((p[3] % PR) << 12);
      A.z = (unsigned)c.base()[v] | ((unsigned)min(nin, 255) << 8) | (far << 16) | ((unsigned)(nin > 4) << 17);
      // bit 18: candidate for the fast row (one predecessor, the row above; query in LDS); bit 19: the sink (no DP row)
      // bit 20: some row other than the next one reads this row's cells (the steady row keeps them out of the LDS ring otherwise)
      A.z |= ((unsigned)(qlds && nin == 1 && (int)p[0] == idx - 1 && v != SRC && v != SNK) << 18) | ((unsigned)(v == SNK) << 19) | (fork << 20);
      // bit 21: one predecessor, 2 .. PR-1 rows up (its cells are in the LDS ring: a steady row can start from there); bits 22-24: its ring slot
      A.z |= ((unsigned)(qlds && nin == 1 && idx - (int)p[0] >= 2 && idx - (int)p[0] < PR && v != SRC && v != SNK) << 21) | ((p[0] % PR) << 22);
      A.w = (unsigned)(Q - (d - 1));                         // qr: the query column this node would sit on by distance to the sink
      uint4 B; B.x = p[0]; B.y = p[1]; B.z = p[2]; B.w = p[3];
      c.descA()[idx] = A; c.descB()[idx] = B;
    }
    WSYNC();                                                 // hops[] of this chunk is read by the next (earlier) chunk
  }
}

// banded global alignment of subread [qb, qb+Q) against the graph; ops are written BACKWARDS
// into opn/opq, returns their count (or <0 on failure: -4 scratch too small, -5 a value that 16-bit cells cannot hold)
// W32 (second pass): every row is a general row on 32-bit cells from the global arena, the ring is not used
// DEF: the scoring is abPOA's default with match 5 (what the reference runs: bin/determine_consensus.py:30) -- every score is an
// immediate operand then; otherwise they are read from the parameters (SGPRs, most of them spilled into VGPR lanes: one
// v_readlane per use in the row loop)
template <bool W32, bool DEF, bool WIDE>
__device__ __forceinline__ int poa_align(Ctx& c, const C3Params& P, int qb, int Q, int lane, long long* cells PHA) {
  constexpr int PR = WIDE ? 3 : C3_NARROW_PR, PW = WIDE ? C3_WIDE_PW : 128, PWT = PADL + PW + PADR, NCHMAX = PW / 64;
  constexpr int RING_CELLS = WIDE ? RING_CELLS_W : RING_CELLS_N, PQW = WIDE ? PQW_W : PQW_N;
  static_assert(PR * PWT == RING_CELLS, "ring geometry");
  unsigned* const Lqpk = WIDE ? &L.qpk[0] : (unsigned*)&L.ring[3 * RING_CELLS_N];
  const int K = c.K, n = c.n;
  const int mt8 = DEF ? S3(5) : S3(P.poa_match), mm8 = DEF ? S3(-4) : S3(-P.poa_mismatch);
  const int e1_8 = DEF ? S3(2) : S3(P.e1), e2_8 = DEF ? S3(1) : S3(P.e2), o1_8 = DEF ? S3(4) : S3(P.o1), o2_8 = DEF ? S3(24) : S3(P.o2);
  const int oe1_8 = o1_8 + e1_8, oe2_8 = o2_8 + e2_8;
  // (the general row's score*512 constants: derived where that row needs them)
#define mt9 (mt8 << 6)
#define mm9 (mm8 * 64)
#define e1_9 (e1_8 << 6)
#define e2_9 (e2_8 << 6)
#define o1_9 (o1_8 << 6)
#define o2_9 (o2_8 << 6)
#define oe1_9 (oe1_8 << 6)
#define oe2_9 (oe2_8 << 6)
#ifdef C3_POA_MW
  const MwConst mwk = {mt9, mm9, e1_9, e2_9, o1_9, o2_9};
#endif
  const int w = wave_first(P.band_b + (int)(P.band_f * (double)Q));
  const int le1_8 = e1_8 * lane, le2_8 = e2_8 * lane;               // the F scans run in lane coordinates: e*(j-beg) = e*lane
#define le1 (le1_8 << 6)
#define le2 (le2_8 << 6)
  const int lo1_8 = le1_8 + o1_8, lo2_8 = le2_8 + o2_8;
  const bool qlds = WIDE || Q <= PQW_N * 16;
  poa_sweep_desc(c, lane, Q, qlds, PR);
  // the subread, 2-bit packed and re-aligned to its first base, goes to LDS: the row loop must not
  // touch global memory for it (a vector load would wait for every older row store).  WIDE: a window of PQW * 16 bases
  // from base qwb on; it follows the band (the row loop moves it, rarely: once per ~1400 rows)
  int qwb = 0;
  auto stage_query = [&](int from) {
    for (int i = lane; i < PQW && from + 16 * i < Q; i += 64) {
      const long long b0 = (long long)qb + from + 16 * i;
      const unsigned w0 = c.pk[b0 >> 4], w1 = c.pk[(b0 >> 4) + 1];
      Lqpk[i] = __builtin_amdgcn_alignbit(w1, w0, (unsigned)(b0 & 15) * 2);
    }
  };
  if (qlds) stage_query(0);
  // every ring cell starts unreachable (the pads stay so: rows only write their PW cells)
  if (!W32) { unsigned* r32 = (unsigned*)&L.ring[0]; for (int i = lane; i < 3 * RING_CELLS / 2; i += 64) r32[i] = NEG16 | (NEG16 << 16); }
  WSYNC();
  PH_MARK(0)
  // Row loop.  The scalar ALU is ONE per CU (measured: 0.96 scalar instructions per cycle per CU against 1.5-1.7 vector
  // instructions; DPP forms at half the vector rate), so uniform per-row values -- the previous row's band and argmax span, the
  // cell counter -- live in VECTOR registers (every lane holds the same number) and the band arithmetic runs on the vector
  // ALU; the scalar unit only sees the loop, one descriptor test and two branches per row.
  int u_beg = 0, u_end = -1, u_left = 0, u_right = 0, u_ncell = 0;                    // previous row / cells used so far
  int u_b8 = 0;                                                                        // ... and its base (absolute score * 8)
  int u_nfar = 0;                                                                      // cells used in the far arena
  int pH = NEG16, pE1 = NEG16, pE2 = NEG16;                                           // its 16-bit cells, lane = band column
  bool pv_ok = false;                                                                  // ... valid: the row above, <= 64 cells
  int gacc = 0xffff;                                                                   // guard: lowest (stored H - ZHI16 - 1) mod 2^16 per lane
  int punt = 0;                                                                        // a value the 16-bit cells cannot hold was seen
  const int lane4 = lane * 4;
  unsigned short* const LH = &L.ring[0]; unsigned short* const LE1 = &L.ring[RING_CELLS]; unsigned short* const LE2 = &L.ring[2 * RING_CELLS];
#ifdef C3_EXP_SBAND
#define UNI(x)                                   /* experiment: band arithmetic of the fast row on the scalar unit */
#else
#define UNI(x) asm volatile("" : "+v"(x))       /* keep a uniform value in a vector register (no instruction) */
#endif
#ifdef C3_PHASE_PROF
  unsigned long long row_t0 = __builtin_readcyclecounter();
#endif
  int slot = PR - 1;                                                                   // ring slot of the row: position mod PR
  // steady rows (round 6, below): the previous row's band in SCALAR registers, the query window of the lanes, what the ring still lacks
  bool st_on = false, ring_stale = false;
  int s_beg = 0, s_end = 0, s_left = 0, s_right = 0, s_ro = 0, s_idx0 = 0;
  unsigned qw = 0;
  // band records (begin, end, cell offset: what the traceback reads) of a run of steady rows, rows s_idx0 .. iend - 1, written when the run
  // ends -- one lane per row: within a run the band moves one column per row and the width stays, so the last row's record gives them all
  // (a lane-0 store per row cost three scalar-to-vector moves, the address arithmetic and an EXEC switch in every steady row)
  // the previous row was a steady row whose cells are still in registers only: into the ring, exactly as the fast row's tail writes them
  auto flush_ring = [&](int cur_slot) {
    const int ps = cur_slot == 0 ? PR - 1 : cur_slot - 1;
    const int cb = ps * PWT + PADL + lane;
    LH[cb] = (unsigned short)pH; LE1[cb] = (unsigned short)pE1; LE2[cb] = (unsigned short)pE2;
#pragma unroll
    for (int f = 1; f < NCHMAX; ++f) { LH[cb + 64 * f] = NEG16; LE1[cb + 64 * f] = NEG16; LE2[cb + 64 * f] = NEG16; }
    if (lane == 0) { L.meta[ps] = make_int4(u_beg, u_end, u_left, u_right); L.base8[ps] = u_b8; }
  };
  auto steady_rowm = [&](int iend) {
    const int R = iend - s_idx0, swd = s_end - s_beg + 1;
    for (int t0 = 0; t0 < R; t0 += 64) {
      const int t = t0 + lane;
      if (t < R) { int* rm = c.rowm() + 3 * (iend - 1 - t); rm[0] = s_beg - t; rm[1] = s_end - t; rm[2] = s_ro - swd * (t + 1); }
    }
  };
  for (int ib = 0; ib < n; ib += 64) {
  uint4 dA = c.descA()[min(ib + lane, n - 1)], dB = c.descB()[min(ib + lane, n - 1)];
  asm volatile("" : "+v"(dA.x), "+v"(dA.y), "+v"(dA.z), "+v"(dA.w), "+v"(dB.x), "+v"(dB.y), "+v"(dB.z), "+v"(dB.w));   // wait here, not in the row loop
  const int cnt = min(64, n - ib);
  for (int li = 0; li < cnt; ++li) {
    const int idx = ib + li;
    slot = slot == PR - 1 ? 0 : slot + 1;
    const int fl = __builtin_amdgcn_readlane(dA.z, li);
    if ((fl >> 19) & 1) { pv_ok = false; continue; }                                  // the sink has no row
    const int qr = __builtin_amdgcn_readlane(dA.w, li);
    const int vb = fl & 0xff;
    const bool far = W32 || ((fl >> 16) & 1);
    if (WIDE && qr + w + 72 > qwb + PQW * 16 && qwb + PQW * 16 < Q) {                // the band nears the end of the query window: move it
      qwb = max(0, qr - w - 320) & ~15;
      WSYNC(); stage_query(qwb); WSYNC();
    }
    // ---- FAST ROW: one predecessor = the previous row, whose H/E1/E2 are still in this wave's REGISTERS (lane = band
    // column).  The predecessor cells arrive by lane permutes (no LDS round trip through memory on the dependent chain),
    // the row maximum is taken from Ht in parallel with the two F scans (an F value is always strictly below some Ht to
    // its left, so max H == max Ht and both are attained in the same columns), and every tie order is a tag in the low
    // bits of the compared keys.  No validity compares: idle lanes of the previous row hold NEG16, and the row only qualifies
    // when no ACTIVE lane's permute wraps around the wave (wd + shift <= 64; lane 0's diagonal source, lane 63 of the previous
    // row, must be idle when the band did not move)
    // (DIRECTION BYTE, all rows): bit0 E1 opened (0 = extended), bit1 E2 opened, bits2-3 Ht source (2 M, 1 E1, 0 E2),
    // bits4-5 H source (2 Ht, 1 F1, 0 F2), bit6 F1 extended, bit7 F2 extended.
#if C3_STEADY
    // ---- STEADY ROW (round 6): a fast row whose band is the previous row's moved ONE column to the right at both ends -- 83 % of the
    // fast rows at cfg2 (66 % of all rows), 68 % at cfg4 (29 %): tools/poa_run_lengths.py.  What the fast row spends on the general case is
    // then known in advance and the scalar unit only CHECKS it (a dozen scalar instructions instead of 27 vector ones): beg = s_beg + 1,
    // end = s_end + 1, the same width, shift 1.  With shift 1 the cell above sits one lane to the right (a DPP move instead of ds_bpermute
    // and its addresses) and the diagonal in the lane itself; the query bases of the lane's next 16 columns ride in one register (a shift
    // per row, refilled every 16 rows, instead of an LDS lookup per row); the band records of a run are written when it ends
    // (steady_rowm); and a row whose cells nobody but the next row reads (descriptor bit 20 clear) writes nothing to the LDS ring -- when
    // the NEXT row turns out not to be a fast / steady row, the cells still in registers go there then (ring_stale, below).  Rows with a
    // successor beyond the ring (far) keep the fast row.  The arithmetic of the cells is the fast row's, instruction for instruction.
    const unsigned stm = (unsigned)fl & ((1u << 16) | (1u << 18) | (1u << 19) | (1u << 21) | (C3_STEADY_FORK ? 0u : 1u << 20));
    // (cand2, C3_STEADY_RING: the one predecessor sits 2 .. PR-1 rows up -- the second sibling of a bubble, 25-32 % of cfg4's near rows: a run can
    // START there, its predecessor's cells and band record come out of the LDS ring instead of the registers)
    const bool cand2 = C3_STEADY_RING && stm == (1u << 21);
    if (!W32 && !WIDE && ((stm == (1u << 18) && pv_ok) || cand2)) {
      int rslot = 0, rb8 = 0;
      if (cand2) {
        if (st_on) { steady_rowm(idx); u_beg = s_beg; u_end = s_end; u_left = s_left; u_right = s_right; u_ncell = s_ro; st_on = false; }      // a run ends here
        if (ring_stale) { flush_ring(slot); ring_stale = false; }        // (cannot be: the row before would have no reader)
        rslot = (fl >> 22) & 7;
        const int4 m = L.meta[rslot]; rb8 = wave_first(L.base8[rslot]);
        s_beg = wave_first(m.x); s_end = wave_first(m.y); s_left = wave_first(m.z); s_right = wave_first(m.w); s_ro = wave_first(u_ncell); s_idx0 = idx;
      } else if (!st_on) { s_beg = wave_first(u_beg); s_end = wave_first(u_end); s_left = wave_first(u_left); s_right = wave_first(u_right); s_ro = wave_first(u_ncell); s_idx0 = idx; }
      const int qr1 = qr - 1, wd = s_end - s_beg + 1;
      if (min(s_left, qr1) == s_beg + w && s_end < Q && max(s_right, qr1) >= s_end - w && (unsigned)(wd - 1) < 63u && s_ro + 64 <= c.cells_cap) {
        const int beg = s_beg + 1, end = s_end + 1, ro = s_ro;
        // the lane's next 16 query bases (columns beg + lane - 1 ...): fetched on entering a run and wherever the band start crosses a
        // multiple of 16 -- never more than 16 rows apart, no counter
        if (!st_on || (s_beg & 15) == 0) {
          const int cq = s_beg + lane;
          const unsigned w0 = Lqpk[min(cq >> 4, PQW - 1)], w1 = Lqpk[min((cq >> 4) + 1, PQW - 1)];
          qw = __builtin_amdgcn_alignbit(w1, w0, (unsigned)(cq & 15) * 2);
        }
        st_on = true;
        if (cand2) {                                                       // the predecessor's cells: lane = its band column, as a row above in registers would sit
          const int cb = rslot * PWT + PADL + lane;
          pH = LH[cb]; pE1 = LE1[cb]; pE2 = LE2[cb]; u_b8 = rb8; pv_ok = true;
        }
        const unsigned long long am = __ballot(lane < wd);                 // active lanes: ONE compare, every select below names this mask
#define SEL(a, b) ({ int d_; asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d_) : "v"(b), "v"(a), "s"(am)); d_; })      /* act ? a : b */
        const int qc = (int)(qw & 3u);
        qw >>= 2;
        // the cells above sit one lane to the right; the vertical candidates are formed in the lane that holds them and THEN moved: two DPP
        // moves instead of three (lane 63 reads 0: it is never active -- wd <= 63 -- and every result of an inactive lane is replaced below)
        const int E1t = wave_shl1z(maxu16(pH - (oe1_8 - 1), pE1 - e1_8));
        const int E2t = wave_shl1z(maxu16(pH - (oe2_8 - 2), pE2 - e2_8));
        const int M = pH + ((vb == qc) ? mt8 + 2 : mm8 + 2);
        const int E1c = E1t & ~7, E2c = E2t & ~7;
        const int k2 = maxu16(maxu16(M, E1c + 1), E2c);
        const int ht = k2 & ~7;
        const int c_neg2 = NEG2_16, c_neg = NEG16;
        const int htm = SEL(ht, c_neg2);
        int s1 = htm + le1_8, s2 = htm + le2_8, s3 = htm;
        wave_scan_max3(s1, s2, s3);
        const int px1 = wave_shr1(s1, NEG2_16), px2 = wave_shr1(s2, NEG2_16);
        const int htl = wave_shr1(htm, NEG16);
        const int f1 = px1 - lo1_8, f2 = px2 - lo2_8;
        const int k3 = maxu16(maxu16(ht + 2, f1 + 1), f2);
        const int h = k3 & ~7;
        unsigned d = ((unsigned)E1t & 1u) | ((unsigned)E2t & 2u) | (((unsigned)k2 & 3u) << 2) | (((unsigned)k3 & 3u) << 4);
        d |= (((unsigned)(htl - oe1_8 - f1)) >> 25) & 64u;
        d |= (((unsigned)(htl - oe2_8 - f2)) >> 24) & 128u;
        const int rb = __builtin_amdgcn_readlane(s3, 63);
        const unsigned long long mxm = __ballot(htm == rb);
        const int left = beg + __builtin_ctzll(mxm), right = beg + (63 - __builtin_clzll(mxm));
        pH = maxu16(SEL(h, c_neg), FLOOR16); pE1 = SEL(E1c, c_neg); pE2 = SEL(E2c, c_neg);
        if ((unsigned)(rb - RB_LO16) > (unsigned)c.rb_span) {         // rare: the base follows the row maximum (see the fast row)
          const int m1 = rb - BIAS16, fl1 = FLOOR16 + max(m1, 0);
          u_b8 += m1; punt |= (int)(rb < GLO16);
          const int a_ = maxu16(h, fl1) - m1, b_ = maxu16(E1c, fl1) - m1, c_ = maxu16(E2c, fl1) - m1;
          pH = SEL(a_, c_neg); pE1 = SEL(b_, c_neg); pE2 = SEL(c_, c_neg);
        }
#undef SEL
        gacc = minu16(gacc, pH - (ZHI16 + 1));
        c.D8()[(unsigned)(ro + lane)] = (uint8_t)d;
        // (the row's band record: written for the whole run when it ends -- steady_rowm)
        if ((fl >> 20) & 1) {                                             // a row further down reads this one: into the ring at once, as the fast row does
          const int cb = slot * PWT + PADL + lane;
          LH[cb] = (unsigned short)pH; LE1[cb] = (unsigned short)pE1; LE2[cb] = (unsigned short)pE2;
#pragma unroll
          for (int f = 1; f < NCHMAX; ++f) { LH[cb + 64 * f] = NEG16; LE1[cb + 64 * f] = NEG16; LE2[cb + 64 * f] = NEG16; }
          if (lane == 0) { L.meta[slot] = make_int4(beg, end, left, right); L.base8[slot] = u_b8; }
          ring_stale = false;
        } else ring_stale = true;
        s_beg = beg; s_end = end; s_left = left; s_right = right; s_ro = ro + wd;
#ifdef C3_PHASE_PROF
        { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_[12] += t_ - row_t0; row_t0 = t_; ph_acc_[13] += 1; }
#endif
        continue;
      }
    }
    if (st_on) { steady_rowm(idx); u_beg = s_beg; u_end = s_end; u_left = s_left; u_right = s_right; u_ncell = s_ro; st_on = false; }      // leaving the steady rows
#endif
    if (!W32 && ((fl >> 18) & 1) && pv_ok) {
      UNI(u_beg); UNI(u_end); UNI(u_left); UNI(u_right); UNI(u_ncell);
      const bool nonempty = u_end >= u_beg;
      const int mplv = nonempty ? u_left + 1 : INT32_MAX / 2, mprv = nonempty ? u_right + 1 : 0;
      const int beg = max(max(0, min(mplv, qr) - w), u_beg);
      int end = min(min(Q, max(mprv, qr) + w), u_end + 1);
      end = max(end, beg - 1);
      const int wd = end - beg + 1, sh = beg - u_beg;
      // (64 cells of head room instead of wd: the stores below are not masked)
      if (__builtin_amdgcn_ballot_w64(((int)((unsigned)(wd - 1) < 64u) & (int)(wd + sh <= 64) & ((int)(sh >= 1) | (int)(u_end - u_beg < 63)) & (int)(u_ncell + 64 <= c.cells_cap) & (int)(!far || u_nfar + 64 <= c.far_cap()) & (int)(!WIDE || (max(beg - 1, 0) >= qwb && end <= qwb + PQW * 16))) != 0) != 0) {
        const int ro = u_ncell;
        const int j = beg + lane;
        const bool act = lane < wd;
        // query base of column j (LDS copy; issued first, consumed after the permutes)
        const int jq = max(j - 1, 0) - (WIDE ? qwb : 0);
        const unsigned qw_ = Lqpk[min(jq >> 4, PQW - 1)];
        // previous row moved to this row's columns: hp = H[i-1][j] sits sh lanes to the right, hd = H[i-1][j-1] one less
        const int a_p = lane4 + sh * 4, a_d = a_p - 4;
        const int hd = __builtin_amdgcn_ds_bpermute(a_d, pH), hp = __builtin_amdgcn_ds_bpermute(a_p, pH);
        const int e1p = __builtin_amdgcn_ds_bpermute(a_p, pE1), e2p = __builtin_amdgcn_ds_bpermute(a_p, pE2);
        const int qc = (int)((qw_ >> ((jq & 15) * 2)) & 3);
        const int M = hd + ((vb == qc) ? mt8 + 2 : mm8 + 2);                   // Ht source in bits 0-1: M 2, E1 1, E2 0
        const int E1t = maxu16(hp - (oe1_8 - 1), e1p - e1_8);                  // bit 0 set: opened (open wins ties)
        const int E2t = maxu16(hp - (oe2_8 - 2), e2p - e2_8);                  // bit 1 set: opened
        const int E1c = E1t & ~7, E2c = E2t & ~7;
        const int k2 = maxu16(maxu16(M, E1c + 1), E2c);
        const int ht = k2 & ~7;
        const int htm = act ? ht : NEG2_16;
        // F[j] = max_{k<j} Ht[k] - o - e*(j-k): scanned in lane coordinates (the e*beg term cancels)
        int s1 = htm + le1_8, s2 = htm + le2_8, s3 = htm;
        wave_scan_max3(s1, s2, s3);
        const int px1 = wave_shr1(s1, NEG2_16), px2 = wave_shr1(s2, NEG2_16);
        const int htl = wave_shr1(htm, NEG16);
        const int f1 = px1 - lo1_8, f2 = px2 - lo2_8;
        const int k3 = maxu16(maxu16(ht + 2, f1 + 1), f2);                     // H source in bits 0-1: Ht 2, F1 1, F2 0
        const int h = k3 & ~7;
        unsigned d = ((unsigned)E1t & 1u) | ((unsigned)E2t & 2u) | (((unsigned)k2 & 3u) << 2) | (((unsigned)k3 & 3u) << 4);
        d |= (((unsigned)(htl - oe1_8 - f1)) >> 25) & 64u;                      // f1 > its "open" candidate: extended
        d |= (((unsigned)(htl - oe2_8 - f2)) >> 24) & 128u;
        const int rb = __builtin_amdgcn_readlane(s3, 63);                       // row maximum (of Ht == of H)
        // first / last column holding the row maximum: columns are beg + lane, so one ballot replaces two reductions
        const unsigned long long mxm = __ballot(htm == rb);
        const int left = beg + __builtin_ctzll(mxm), right = beg + (63 - __builtin_clzll(mxm));
        // the row leaves relative to its own maximum; idle lanes unreachable
        pH = maxu16(act ? h : NEG16, FLOOR16); pE1 = act ? E1c : NEG16; pE2 = act ? E2c : NEG16;      // (select first: NEG16 > FLOOR16)
        int nb8 = u_b8;
        if ((unsigned)(rb - RB_LO16) > (unsigned)c.rb_span) {         // rare: the base follows the row maximum
          // (saturating: an unreachable cell sits far below any base shift and must stay at the floor, not wrap around)
          const int m1 = rb - BIAS16, fl1 = FLOOR16 + max(m1, 0);
          nb8 += m1; punt |= (int)(rb < GLO16);
          pH = act ? maxu16(h, fl1) - m1 : NEG16; pE1 = act ? maxu16(E1c, fl1) - m1 : NEG16; pE2 = act ? maxu16(E2c, fl1) - m1 : NEG16;
        }
        gacc = minu16(gacc, pH - (ZHI16 + 1));
#ifdef C3_DEBUG_PUNT
        if (pH > ZHI16 && pH < GLO16) { atomicAdd(c.dbg + 8, 1ull); atomicMax(c.dbg + 9, ((unsigned long long)(65535 - pH) << 40) | ((unsigned long long)idx << 16) | ((unsigned long long)lane << 8) | (unsigned long long)(wd & 255)); atomicMax(c.dbg + 10, ((unsigned long long)(65535 - pH) << 40) | ((unsigned long long)(unsigned short)rb << 16) | (unsigned)(sh & 255) << 8 | (unsigned)(u_end - u_beg + 1)); }
#endif
        // unmasked stores: lanes past the band write cells that the next rows overwrite / that no reader ever selects
        c.D8()[(unsigned)(ro + lane)] = (uint8_t)d;
        { const int cb = slot * PWT + PADL + lane;
          LH[cb] = (unsigned short)pH; LE1[cb] = (unsigned short)pE1; LE2[cb] = (unsigned short)pE2;
#pragma unroll
          for (int f = 1; f < NCHMAX; ++f) { LH[cb + 64 * f] = NEG16; LE1[cb + 64 * f] = NEG16; LE2[cb + 64 * f] = NEG16; }      // (a one-chunk row: its other chunks read as unreachable)
        }
        if (lane == 0) {
          L.meta[slot] = make_int4(beg, end, left, right); L.base8[slot] = nb8;
          int off = 3 * idx; UNI(off);
          int* rm = c.rowm() + off; rm[0] = beg; rm[1] = end; rm[2] = ro;                 // type 0: byte cells, one predecessor
        }
        if (far) {
          const int fo = wave_first(u_nfar);
          if (act) { c.H()[fo + lane] = cv_16to9(pH, nb8); c.E1()[fo + lane] = cv_16to9(pE1, nb8); c.E2()[fo + lane] = cv_16to9(pE2, nb8); }
          if (lane == 0) { c.mpl()[idx] = left; c.mpr()[idx] = right; c.foff()[idx] = fo; }
          u_nfar = fo + wave_first(wd);
        }
        u_beg = beg; u_end = end; u_left = left; u_right = right; u_ncell = ro + wd; u_b8 = nb8;
#if C3_STEADY
        ring_stale = false;                          // (this row is in the ring; the one before it has no other reader)
#endif
#ifdef C3_PHASE_PROF
        { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_[8] += t_ - row_t0; row_t0 = t_; ph_acc_[14] += 1; }
#endif
        continue;
      }
    }
#if C3_STEADY
    if (ring_stale) {                                                       // the previous row was a steady row and this one reads the ring
      flush_ring(slot);
      ring_stale = false;
#ifdef C3_PHASE_PROF
      ph_acc_[15] += 1;
#endif
    }
#endif
    const int v = __builtin_amdgcn_readlane(dA.x, li);
    const int nin = (fl >> 8) & 0xff;
    const bool ovf = (fl >> 17) & 1;
    const int p0 = __builtin_amdgcn_readlane(dB.x, li), p1 = __builtin_amdgcn_readlane(dB.y, li);
    const int p2 = __builtin_amdgcn_readlane(dB.z, li), p3 = __builtin_amdgcn_readlane(dB.w, li);
    const int psl = __builtin_amdgcn_readlane(dA.y, li);
    const int ps0 = psl & 15, ps1 = (psl >> 4) & 15, ps2 = (psl >> 8) & 15, ps3 = (psl >> 12) & 15;      // their ring slots
#define PRED_IDX(k) ((k) == 0 ? p0 : (k) == 1 ? p1 : (k) == 2 ? p2 : (k) == 3 ? p3 : c.index()[c.in_from()[EI(v, (k))]])
#define PRED_SLOT(k) ((k) == 0 ? ps0 : (k) == 1 ? ps1 : (k) == 2 ? ps2 : ps3)
    int ncell = wave_first(u_ncell);
    // ---- NEAR ROW: up to four predecessors, every one of them among the last PR-1 rows (so its cells and band record are in
    // the LDS ring) -- the usual member of an aligned block and the node after it -- and a band of at most 64 (one chunk) or
    // 128 columns (two chunks: the rows whose nominal column has drifted away from the predecessors' maxima).  Branch free and
    // free of validity compares: a predecessor cell is read at a clamped offset, outside the predecessor's band that is a pad
    // or an idle cell and holds NEG16; an absent predecessor is given a band far to the right, so all of its reads land in the
    // left pad.  The bases of the predecessor rows differ by a few score units: the differences ride in the per-predecessor
    // constants that carry the tie-order tags anyway.  Tie order by tags exactly as in the general row below; direction byte +
    // predecessor byte.  One instance per (predecessor count, chunk count).
    if (!W32 && C3_NEAR && !ovf && qlds && v != SRC && idx - p0 < PR && (nin < 2 || idx - p1 < PR) && (nin < 3 || idx - p2 < PR) && (nin < 4 || idx - p3 < PR)) {
      auto near_body = [&](auto NPc, auto NCHc) -> bool {
      constexpr int NP = decltype(NPc)::value, NCH = decltype(NCHc)::value;
      UNI(u_ncell);
      const int BIGB = 1 << 28;
      int4 m_[NP]; int b8_[NP], sl_[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) { sl_[k] = k < nin ? PRED_SLOT(k) : 0; m_[k] = L.meta[sl_[k]]; b8_[k] = L.base8[sl_[k]]; }
      int pb_[NP], dk_[NP];
      int mplv = INT32_MAX / 2, mprv = 0, minb = INT32_MAX, maxe = INT32_MIN, wmax = 0, dabs = 0;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const bool has = k < nin;                                             // scalar condition, used as a select mask
        const int b = has ? m_[k].x : BIGB, e = has ? m_[k].y : -BIGB;
        pb_[k] = b;
        dk_[k] = has ? b8_[k] - b8_[0] : 0;                                   // this predecessor's cells against the first one's base
        dabs = max(dabs, max(dk_[k], -dk_[k]));
        minb = min(minb, b); maxe = max(maxe, e + 1); wmax = max(wmax, e - b);
        const bool ne = e >= b;
        mplv = ne ? min(mplv, m_[k].z + 1) : mplv; mprv = ne ? max(mprv, m_[k].w + 1) : mprv;
      }
      const int beg = max(max(0, min(mplv, qr) - w), minb);
      int end = min(min(Q, max(mprv, qr) + w), maxe);
      end = max(end, beg - 1);
      const int wd = end - beg + 1;
      if (__builtin_amdgcn_ballot_w64(((int)((unsigned)(wd - 1) < (unsigned)(64 * NCH)) & (int)(wmax < PW) & (int)(dabs < 2048) & (int)(u_ncell + 64 * NCH <= c.cells_cap) & (int)(!far || u_nfar + 64 * NCH <= c.far_cap()) & (int)(!WIDE || (max(beg - 1, 0) >= qwb && end <= qwb + PQW * 16))) != 0) == 0) return false;
      const int ro = u_ncell;
      const int W8 = wave_first(b8_[0]);
      // per predecessor: base difference + tie-order tag of each candidate (first predecessor wins: highest tag; open before extend)
      int cM[NP], c1o[NP], c1x[NP], c2o[NP], c2x[NP], sb_[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        cM[k] = dk_[k] + (3 - k);
        c1o[k] = dk_[k] - oe1_8 + (7 - 2 * k); c1x[k] = dk_[k] - e1_8 + (6 - 2 * k);
        c2o[k] = dk_[k] - oe2_8 + (7 - 2 * k); c2x[k] = dk_[k] - e2_8 + (6 - 2 * k);
        sb_[k] = sl_[k] * PWT + PADL - pb_[k];                                // ring index of column 0 of this predecessor
      }
      int carry1 = 0, carry2 = 0, prev_ht = NEG16;                              // scan carries from the first chunk
      int best = 0, left = 0, right = 0;
      int cH[NCH], cE1[NCH], cE2[NCH];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = 64 * ch;
        const int j = beg + c0 + lane;
        const bool act = c0 + lane < wd;
        const int jq = max(j - 1, 0) - (WIDE ? qwb : 0);
        const unsigned qw_ = Lqpk[min(jq >> 4, PQW - 1)];
        int hd_[NP], hp_[NP], e1_[NP], e2_[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          // offset in the predecessor's row, clamped into [-2, PW+1]: everything outside the row's cells is a pad
          const int lo_ = sl_[k] * PWT + PADL - 2, hi_ = sl_[k] * PWT + PADL + PW + 1;
          const int ix = min(max(j + sb_[k], lo_), hi_);
          hd_[k] = LH[ix - 1]; hp_[k] = LH[ix]; e1_[k] = LE1[ix]; e2_[k] = LE2[ix];
        }
        int kM = 0, kE1 = 0, kE2 = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          kM = maxu16(kM, hd_[k] + cM[k]);
          kE1 = maxu16(kE1, maxu16(hp_[k] + c1o[k], e1_[k] + c1x[k]));
          kE2 = maxu16(kE2, maxu16(hp_[k] + c2o[k], e2_[k] + c2x[k]));
        }
        const int qc = (int)((qw_ >> ((jq & 15) * 2)) & 3);
        const int M = (kM & ~7) + ((vb == qc) ? mt8 + 2 : mm8 + 2);            // (column 0 has no diagonal: its source is the left pad)
        const int E1c = kE1 & ~7, E2c = kE2 & ~7;
        const int k2 = maxu16(maxu16(M, E1c + 1), E2c);
        const int ht = k2 & ~7;
        const unsigned mp = 3u - ((unsigned)kM & 3u), c1 = 7u - ((unsigned)kE1 & 7u), c2 = 7u - ((unsigned)kE2 & 7u);
        unsigned d = ((~c1) & 1u) | (((~c2) & 1u) << 1) | (((unsigned)k2 & 3u) << 2);
        const unsigned pby = mp | ((c1 >> 1) << 2) | ((c2 >> 1) << 4);                 // (one predecessor: always 0 -- such a row keeps no predecessor byte, C3_NEAR1_TYPE0)
        const int htm = act ? ht : NEG2_16;
        const int cl1 = le1_8 + e1_8 * c0, cl2 = le2_8 + e2_8 * c0;           // e * (column - beg)
        int s1 = htm + cl1, s2 = htm + cl2, s3 = htm;
        wave_scan_max3(s1, s2, s3);
        const int px1 = max(wave_shr1(s1, NEG2_16), carry1), px2 = max(wave_shr1(s2, NEG2_16), carry2);
        const int htl = wave_shr1(htm, prev_ht);
        const int f1 = px1 - o1_8 - cl1, f2 = px2 - o2_8 - cl2;               // column beg: NEG2_16 - ... (never wins)
        if (NCH > 1) { carry1 = max(carry1, wave_bcast(s1, 63)); carry2 = max(carry2, wave_bcast(s2, 63)); prev_ht = wave_bcast(htm, 63); }
        const int k3 = maxu16(maxu16(ht + 2, f1 + 1), f2);
        const int h = k3 & ~7;
        d |= (((unsigned)k3 & 3u) << 4);
        d |= (((unsigned)(htl - oe1_8 - f1)) >> 25) & 64u;                      // f1 > its "open" candidate: extended
        d |= (((unsigned)(htl - oe2_8 - f2)) >> 24) & 128u;
        // unmasked stores (see the fast row); a near row with several predecessors keeps a predecessor byte per cell (type 1)
        c.D8()[(unsigned)(ro + c0 + lane)] = (uint8_t)d;
        if (NP > 1 || !C3_NEAR1_TYPE0) c.P8()[(unsigned)(ro + c0 + lane)] = (uint8_t)pby;
        cH[ch] = h; cE1[ch] = E1c; cE2[ch] = E2c;
        const int cmx = __builtin_amdgcn_readlane(s3, 63);                      // maximum of Ht over the chunk (== maximum of H)
        const unsigned long long mxm = __ballot(htm == cmx);
        if (NCH == 1 || cmx > best) { best = cmx; left = beg + c0 + __builtin_ctzll(mxm); right = beg + c0 + (63 - __builtin_clzll(mxm)); }
        else if (cmx == best && mxm) right = beg + c0 + (63 - __builtin_clzll(mxm));
      }
      // the row leaves relative to its own maximum
      int nb8 = W8;
      if ((unsigned)(best - RB_LO16) > (unsigned)c.rb_span) {       // rare (see the fast row): the base follows the row maximum, saturating
        const int m1 = best - BIAS16, fl1 = FLOOR16 + max(m1, 0);
        nb8 += m1; punt |= (int)(best < GLO16);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) { cH[ch] = maxu16(cH[ch], fl1) - m1; cE1[ch] = maxu16(cE1[ch], fl1) - m1; cE2[ch] = maxu16(cE2[ch], fl1) - m1; }
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const bool act = 64 * ch + lane < wd;
        const int vH = maxu16(act ? cH[ch] : NEG16, FLOOR16), vE1 = act ? cE1[ch] : NEG16, vE2 = act ? cE2[ch] : NEG16;
        gacc = minu16(gacc, vH - (ZHI16 + 1));
#ifdef C3_DEBUG_PUNT
        if (vH > ZHI16 && vH < GLO16) { atomicAdd(c.dbg + 11, 1ull); atomicMax(c.dbg + 12, ((unsigned long long)(65535 - vH) << 40) | ((unsigned long long)idx << 16) | ((unsigned long long)lane << 8) | (unsigned long long)(wd & 255)); }
#endif
        const int cb = slot * PWT + PADL + 64 * ch + lane;
        LH[cb] = (unsigned short)vH; LE1[cb] = (unsigned short)vE1; LE2[cb] = (unsigned short)vE2;
        if (ch == NCH - 1) {
#pragma unroll
          for (int f = 1; f <= NCHMAX - NCH; ++f) { LH[cb + 64 * f] = NEG16; LE1[cb + 64 * f] = NEG16; LE2[cb + 64 * f] = NEG16; }      // the chunks this row does not have
        }
        if (ch == 0) { pH = vH; pE1 = vE1; pE2 = vE2; }
        if (far) { if (act) { c.H()[u_nfar + 64 * ch + lane] = cv_16to9(vH, nb8); c.E1()[u_nfar + 64 * ch + lane] = cv_16to9(vE1, nb8); c.E2()[u_nfar + 64 * ch + lane] = cv_16to9(vE2, nb8); } }
      }
      pv_ok = NCH == 1;                          // (WIDE: never -- its instances start at two chunks)
      if (lane == 0) {
        L.meta[slot] = make_int4(beg, end, left, right); L.base8[slot] = nb8;
        int off = 3 * idx; UNI(off);
        int* rm = c.rowm() + off; rm[0] = beg; rm[1] = end | ((NP > 1 || !C3_NEAR1_TYPE0) ? (1 << 28) : 0); rm[2] = ro;      // (type 0 = direction bytes only: the traceback takes predecessor 0)
        if (far) { c.mpl()[idx] = left; c.mpr()[idx] = right; c.foff()[idx] = u_nfar; }
      }
      if (far) u_nfar = wave_first(u_nfar) + wave_first(wd);
#ifdef C3_EXP_SBAND
      u_beg = wave_first(beg); u_end = wave_first(end); u_left = wave_first(left); u_right = wave_first(right); u_ncell = wave_first(ro + wd); u_b8 = nb8;
#else
      u_beg = beg; u_end = end; u_left = left; u_right = right; u_ncell = ro + wd; u_b8 = nb8;
#endif
#ifdef C3_PHASE_PROF
      { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_[10] += t_ - row_t0; row_t0 = t_; }
#endif
      return true;
      };
      typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3; typedef std::integral_constant<int, 4> I4;
      // chunk counts tried: 1, 2 (NARROW) / 2, 3, 4 (WIDE: long subreads have no bands below 64 columns to speak of)
      typedef std::integral_constant<int, WIDE ? 2 : 1> CA; typedef std::integral_constant<int, WIDE ? 3 : 2> CB; typedef std::integral_constant<int, 4> CC;
      bool handled = nin == 1 ? near_body(I1{}, CA{}) : nin == 2 ? near_body(I2{}, CA{}) : nin == 3 ? near_body(I3{}, CA{}) : near_body(I4{}, CA{});
      if (!handled) handled = nin == 1 ? near_body(I1{}, CB{}) : nin == 2 ? near_body(I2{}, CB{}) : nin == 3 ? near_body(I3{}, CB{}) : near_body(I4{}, CB{});
      if (WIDE && C3_WIDE_PW >= 256) { if (!handled && nin <= 2) handled = nin == 1 ? near_body(I1{}, CC{}) : near_body(I2{}, CC{}); }      // (four chunks: one or two predecessors only)
      if (handled) continue;
    }
    // ---- GENERAL ROW (32-bit cells).  Adaptive band: gather the hints of the predecessors (abPOA scatters them to the
    // successors).  The ring metadata of the first four predecessors is fetched in ONE LDS round trip (one 16-byte read each,
    // issued together) and pinned to scalars; predecessors that left the ring (or a fifth, sixth ... one) take global loads.
    // Ring cells are converted on the way in (16-bit relative -> score*512) and on the way out.
    int beg, end;
    int pb_[4] = {0, 0, 0, 0}, pe_[4] = {-1, -1, -1, -1};          // band of predecessor k (k < 4)
    int rb8_[4] = {0, 0, 0, 0};                                      // ... the base of its ring row
    bool ring_[4] = {false, false, false, false};                    // ... and whether its cells are in the LDS ring
    if (v == SRC) { beg = 0; end = min(Q, max(qr, 0) + w); }
    else {
      int mplv = INT32_MAX / 2, mprv = 0, minb = INT32_MAX, maxe = INT32_MIN;
      int4 m_[4]; int b8m_[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { m_[k] = L.meta[PRED_SLOT(k)]; b8m_[k] = L.base8[PRED_SLOT(k)]; }
      asm volatile("" : "+v"(m_[0].x), "+v"(m_[1].x), "+v"(m_[2].x), "+v"(m_[3].x));     // the LDS loads happen HERE, together
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < nin) {
          const int pi = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
          int b = m_[k].x, e = m_[k].y, l = m_[k].z, r = m_[k].w;
          // (written as an override, not as if/else: a select between an LDS and a global POINTER becomes a flat load)
          if (W32 || idx - pi >= PR) { b = c.rowm()[3 * pi]; e = c.rowm()[3 * pi + 1] & 0x0fffffff; l = c.mpl()[pi]; r = c.mpr()[pi]; }
          b = wave_first(b); e = wave_first(e); l = wave_first(l); r = wave_first(r);
          pb_[k] = b; pe_[k] = e; ring_[k] = !W32 && idx - pi < PR && e - b + 1 <= PW; rb8_[k] = wave_first(b8m_[k]);
          minb = min(minb, b); maxe = max(maxe, e + 1);
          if (e >= b) { mplv = min(mplv, l + 1); mprv = max(mprv, r + 1); }
        }
      }
      for (int k = 4; k < nin; ++k) {
        const int pi = PRED_IDX(k);
        const int sl = pi % PR;
        int4 m = L.meta[sl];
        asm volatile("" : "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));
        int b = m.x, e = m.y, l = m.z, r = m.w;
        if (W32 || idx - pi >= PR) { b = c.rowm()[3 * pi]; e = c.rowm()[3 * pi + 1] & 0x0fffffff; l = c.mpl()[pi]; r = c.mpr()[pi]; }
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
#ifdef C3_DEBUG_PUNT
    if (ncell + wd > c.cells_cap && lane == 0) atomicAdd(c.dbg + 12, 1ull);
#endif
    if (ncell + wd > c.cells_cap) return -4;                       // (cannot wrap: the host keeps every capacity max_q + 512 below INT_MAX -- run_poa; rewriting the compare cost the fast rows 3 % through the register allocation)
    const bool inl = !W32 && wd <= PW, toglobal = far || !inl;     // wd is scalar (beg / end pinned above)
    const bool qrow = qlds && (!WIDE || (max(beg - 1, 0) >= qwb && end <= qwb + PQW * 16));      // the row's query bases are in the LDS window
    const int fo = wave_first(u_nfar);                               // its cells in the far arena (when it keeps 32-bit cells / direction words)
#ifdef C3_DEBUG_PUNT
    if ((toglobal || ovf) && fo + wd > c.far_cap() && lane == 0) { atomicAdd(c.dbg + 13, 1ull); atomicMax(c.dbg + 11, ((unsigned long long)(unsigned)fo << 32) | (unsigned)c.far_cap()); }
#endif
    if (toglobal || ovf) { if (fo + wd > c.far_cap()) return (!W32 && WIDE) ? -6 : -4; u_nfar = fo + wd; }      // (-6: the far arena of a 16-bit pass over LONG subreads -- a band that blew up to thousands of columns; such a read goes straight to the last pass, whose wide rows eight waves share)
    const int ro = ncell;
    const int ty = ovf ? 2 : (nin >= 2 ? 1 : 0);                   // cell format: byte / byte + predecessor byte / 32-bit word
    ncell += wd;
    int best = INT32_MIN, bl = 0, br = 0;        // per-lane running row maximum
    int carry1 = NEG2S, carry2 = NEG2S;          // scan carries over previous chunks
    int prev_ht = NEGS;
    int gH = NEG16, gE1 = NEG16, gE2 = NEG16;    // the row's 16-bit cells (first chunk) for a fast successor
    int Wg = 0;                                  // the base its ring cells are kept against: the maximum of its first chunk
    int bad = 0;
#ifdef C3_POA_MW
    const int nch_ = (wd + 63) >> 6;
    const bool mw_row = W32 && v != SRC && nch_ >= MW_MINCH && nch_ <= MW_CH && nin <= MW_NP;      // (narrower rows: wave 0 alone, no barrier)
    if (mw_row) {
      if (lane == 0) {
        MWT.cmd = 1; MWT.beg = beg; MWT.end = end; MWT.nin = nin; MWT.vb = vb; MWT.fo = fo; MWT.ro = ro; MWT.ty = ty; MWT.qb = qb;
        MWT.pk = (unsigned long long)c.pk;
      }
      if (lane < nin) {                           // lane k fetches predecessor k: one round trip for all of them
        const int pi = lane == 0 ? p0 : lane == 1 ? p1 : lane == 2 ? p2 : lane == 3 ? p3 : c.index()[c.in_from()[EI(v, lane)]];
        MWT.pb[lane] = c.rowm()[3 * pi]; MWT.pe[lane] = c.rowm()[3 * pi + 1] & 0x0fffffff; MWT.po[lane] = c.foff()[pi];
      }
      MW_BARRIER();
      mw_chunks<0>(c, mwk, 0, lane);
      MW_BARRIER();
      mw_chunks<1>(c, mwk, 0, lane);
      MW_BARRIER();
      // the row's maximum and its leftmost / rightmost column: first chunk that reaches it, last chunk that equals it
      int bx = INT32_MIN;
      for (int x = lane; x < nch_; x += 64) bx = max(bx, MWT.cmx[x]);
      bx = wave_max(bx);
      int fx = INT32_MAX, lx = -1;
      for (int x = lane; x < nch_; x += 64) if (MWT.cmx[x] == bx) { fx = min(fx, x); lx = max(lx, x); }
      fx = wave_min(fx); lx = wave_max(lx);
      best = bx; bl = MWT.fpos[fx]; br = MWT.lpos[lx];
      Wg = u_b8;
    }
#ifdef C3_MW_CHECK
    const int mw_best = best, mw_bl = bl, mw_br = br;
    if (mw_row) { best = INT32_MIN; bl = br = 0; }
#else
    else
#endif
#endif
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
          const int sl = PRED_SLOT(k);
          const int o = j - pb_[k];
          const int oc = sl * PWT + PADL + min(max(o, 0), PW - 1), om = sl * PWT + PADL + min(max(o - 1, 0), PW - 1);
          hd_[k] = LH[om]; hp_[k] = LH[oc]; e1_[k] = LE1[oc]; e2_[k] = LE2[oc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k < nin) {
            const int pi = k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
            const int b = pb_[k], e = pe_[k];
            const bool vd = j > 0 && j - 1 >= b && j - 1 <= e, vp = j >= b && j <= e;
            int hd = NEGS, hp = NEGS, e1p = NEGS, e2p = NEGS;
            if (ring_[k]) {
              hd = vd ? cv_16to9(hd_[k], rb8_[k]) : NEGS; hp = vp ? cv_16to9(hp_[k], rb8_[k]) : NEGS;
              e1p = vp ? cv_16to9(e1_[k], rb8_[k]) : NEGS; e2p = vp ? cv_16to9(e2_[k], rb8_[k]) : NEGS;
            } else {
              const int po = c.foff()[pi];
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
          const int sl = pi % PR;
          int4 m = L.meta[sl]; int pb8 = L.base8[sl];
          asm volatile("" : "+v"(m.x), "+v"(m.y), "+v"(pb8));
          int b = m.x, e = m.y;
          if (W32 || idx - pi >= PR) { b = c.rowm()[3 * pi]; e = c.rowm()[3 * pi + 1] & 0x0fffffff; }
          if (!W32 && idx - pi < PR && e - b + 1 <= PW) {
            if (j > 0 && j - 1 >= b && j - 1 <= e) hd = cv_16to9(LH[sl * PWT + PADL + j - 1 - b], pb8);
            if (j >= b && j <= e) { hp = cv_16to9(LH[sl * PWT + PADL + j - b], pb8); e1p = cv_16to9(LE1[sl * PWT + PADL + j - b], pb8); e2p = cv_16to9(LE2[sl * PWT + PADL + j - b], pb8); }
          } else {
            const int po = c.foff()[pi];
            if (j > 0 && j - 1 >= b && j - 1 <= e) hd = c.H()[po + (j - 1 - b)];
            if (j >= b && j <= e) { hp = c.H()[po + (j - b)]; e1p = c.E1()[po + (j - b)]; e2p = c.E2()[po + (j - b)]; }
          }
          kM = max(kM, hd + (511 - k));
          kE1 = max(kE1, max(hp - oe1_9 + (511 - 2 * k), e1p - e1_9 + (510 - 2 * k)));
          kE2 = max(kE2, max(hp - oe2_9 + (511 - 2 * k), e2p - e2_9 + (510 - 2 * k)));
        }
        int qc = 7;
        if (act && j > 0) qc = qrow ? (int)((Lqpk[(j - 1 - (WIDE ? qwb : 0)) >> 4] >> (((j - 1) & 15) * 2)) & 3) : c3_code_at(c.pk, qb + j - 1);
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
#if defined(C3_POA_MW) && defined(C3_MW_CHECK)
      if (mw_row && lane == 0) {              // the chunk aggregates of phase A against this loop's
        if (wave_bcast(s1, 63) != MWT.agg1[c0 >> 6]) atomicAdd(c.chk + 8, 1ull);
        if (wave_bcast(s2, 63) != MWT.agg2[c0 >> 6]) atomicAdd(c.chk + 9, 1ull);
        if (wave_bcast(htm, 63) != MWT.lastht[c0 >> 6]) atomicAdd(c.chk + 10, 1ull);
      }
#endif
      carry1 = max(carry1, wave_bcast(s1, 63)); carry2 = max(carry2, wave_bcast(s2, 63));
      prev_ht = wave_bcast(htm, 63);
      const int k3 = max(max(ht9 + 2, f1 + 1), f2);
      const int h9 = k3 & ~511;
      d |= (((unsigned)k3 & 3u) << 4) | (f1x << 6) | (f2x << 7);
      dw |= ((unsigned)(2 - (k3 & 3)) << 28) | (f1x << 30) | (f2x << 31);
      // row maximum == maximum of Ht, attained in the same columns (see the fast row)
      const int cmx = wave_bcast(s3, 63);
      // base of the row's ring cells: the running base (that of the row before it in the order) -- a general row between rows
      // that share a base must not start a new one, or every row next to it sees predecessors on different bases and turns
      // general too; only when its own maximum does not fit that base does it take the maximum (as the other rows do)
      if (c0 == 0) { Wg = u_b8; const int r_ = (cmx >> 6) - Wg + BIAS16; if (cmx > NEGS / 2 && (unsigned)(r_ - RB_LO16) > (unsigned)c.rb_span) Wg = cmx >> 6; }
#if defined(C3_POA_MW) && defined(C3_MW_CHECK)
      if (act && mw_row) {
        const int ci = c0 + lane;
#define MWCHK(mem, val, kind) if ((mem) != (val)) { atomicAdd(c.chk + (kind), 1ull); }
        if (ty == 2) { MWCHK(c.D()[fo + ci], dw, 0) } else { MWCHK(c.D8()[(unsigned)(ro + ci)], (uint8_t)d, 1) if (ty == 1) { MWCHK(c.P8()[(unsigned)(ro + ci)], (uint8_t)pby, 2) } }
        MWCHK(c.H()[fo + ci], h9, 3) MWCHK(c.E1()[fo + ci], E1v, 4) MWCHK(c.E2()[fo + ci], E2v, 5)
      }
#endif
      if (act) {
        const int ci = c0 + lane;
        if (ty == 2) c.D()[fo + ci] = dw;
        else { c.D8()[(unsigned)(ro + ci)] = (uint8_t)d; if (ty == 1) c.P8()[(unsigned)(ro + ci)] = (uint8_t)pby; }
        if (toglobal) { c.H()[fo + ci] = h9; c.E1()[fo + ci] = E1v; c.E2()[fo + ci] = E2v; }
      }
      if (inl) {
        // ring cells: 16 bits against the base Wg; idle lanes unreachable (every lane stores)
        const int vH = act ? cv_9to16(h9, Wg, bad) : NEG16, vE1 = act ? cv_9to16(E1v, Wg, bad) : NEG16, vE2 = act ? cv_9to16(E2v, Wg, bad) : NEG16;
        const int cb = slot * PWT + PADL + c0 + lane;
        LH[cb] = (unsigned short)vH; LE1[cb] = (unsigned short)vE1; LE2[cb] = (unsigned short)vE2;
        if (c0 + 64 >= wd) { for (int f = 64; c0 + f < PW; f += 64) { LH[cb + f] = NEG16; LE1[cb + f] = NEG16; LE2[cb + f] = NEG16; } }      // the chunks this row does not have
        if (c0 == 0) { gH = vH; gE1 = vE1; gE2 = vE2; }
      }
      if (cmx > best) { best = cmx; const unsigned long long mm = __ballot(htm == cmx); bl = beg + c0 + __builtin_ctzll(mm); br = beg + c0 + 63 - __builtin_clzll(mm); }
      else if (cmx == best) { const unsigned long long mm = __ballot(htm == cmx); if (mm) br = beg + c0 + 63 - __builtin_clzll(mm); }
    }
#if defined(C3_POA_MW) && defined(C3_MW_CHECK)
    if (mw_row && lane == 0) { if (mw_best != best) atomicAdd(c.chk + 6, 1ull); if (mw_bl != bl || mw_br != br) { atomicAdd(c.chk + 7, 1ull); } }
#endif
    if (!W32 && __builtin_amdgcn_ballot_w64(bad != 0) != 0) punt |= 2;
    // leftmost / rightmost argmax -> band hints read by the successors
    const int left = wd > 0 ? bl : 0, right = wd > 0 ? br : 0;
    if (lane == 0) {
      L.meta[slot] = make_int4(beg, end, left, right); L.base8[slot] = Wg;
      { int* rm = c.rowm() + 3 * idx; rm[0] = beg; rm[1] = end | (ty << 28); rm[2] = ro; }
      if (far) { c.mpl()[idx] = left; c.mpr()[idx] = right; }
      if (toglobal || ovf) c.foff()[idx] = fo;
    }
    u_beg = beg; u_end = end; u_left = left; u_right = right; u_ncell = ncell; u_b8 = Wg;
#ifdef C3_PHASE_PROF
    { unsigned long long t_ = __builtin_readcyclecounter(); ph_acc_[11] += t_ - row_t0; row_t0 = t_; }
#endif
    pH = gH; pE1 = gE1; pE2 = gE2; pv_ok = inl && wd <= 64;
  }
  }
#if C3_STEADY
  if (st_on) { steady_rowm(n); u_ncell = s_ro; st_on = false; }              // (cannot happen: the rows before the sink have an edge to it and are not steady)
#endif
  WSYNC();
  *cells += wave_first(u_ncell);
  PH_MARK(1)
  // a read with a value outside what the 16-bit cells represent exactly goes to the 32-bit pass (never seen on the config shapes)
#ifdef C3_DEBUG_PUNT
  if (!W32 && lane == 0) { if (punt) atomicAdd(c.dbg + 14, 1ull); if (__builtin_amdgcn_ballot_w64((unsigned)(gacc & 0xffff) < (unsigned)(GLO16 - ZHI16 - 1)) != 0) atomicAdd(c.dbg + 15, 1ull); }
#endif
  if (!W32 && (punt != 0 || __builtin_amdgcn_ballot_w64((unsigned)(gacc & 0xffff) < (unsigned)(GLO16 - ZHI16 - 1)) != 0)) return -5;
  // ---- end cell: best predecessor of the sink at column Q (first maximum in in-edge order)
  int bi = -1, bs = INT32_MIN;
  for (int k = 0; k < c.n_in()[SNK]; ++k) {
    const int pi = c.index()[c.in_from()[EI(SNK, k)]];
    const int pb = c.rowm()[3 * pi], pe = c.rowm()[3 * pi + 1] & 0x0fffffff;
    const int hh = (Q < pb || Q > pe) ? NEGS : c.H()[c.foff()[pi] + (Q - pb)];
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
    uint8_t* WD = (uint8_t*)&L.ring[0];          // [64][32] direction-byte windows
    uint8_t* WP = WD + 64 * 32;                  // [64][32] predecessor-byte windows (the three rings are contiguous: 4.7 KB)
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
      // (the path can only fall BEHIND the diagonal of the block -- a row it skips, a sibling or a vertical gap, costs a lane and no
      // column -- except for inserted bases; so the window starts 8 columns before the diagonal and reaches 23 beyond it)
      int a0 = ro + (jt - lane - 8 - b);
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
        // ---- an M move through a row that broke the run (its predecessor is not the row above, or it has several): H <- Ht <- M
        // again, only the predecessor differs.  Taken here with a handful of lane reads instead of the state machine below (at cfg4
        // half of the path's rows are of this kind)
        if (st == 0 && j > 0 && wave_bcast((int)hit, cl)) {
          const unsigned db = (unsigned)wave_bcast((int)d, cl);
          if (((db >> 2) & 15u) == 10u) {
            const unsigned mp1 = wave_bcast(ty, cl) == 1 ? ((unsigned)wave_bcast((int)pq, cl) & 3u) : 0u;
            const int pr = wave_bcast(mp1 == 0 ? (int)B.x : mp1 == 1 ? (int)B.y : mp1 == 2 ? (int)B.z : (int)B.w, cl);
            const int v1 = wave_bcast((int)A.x, cl);
            if (lane == 0) vq[j - 1] = v1;
            i = pr; --j;
            if (i == 0 && j == 0) break;
            if (i < 0) { rc = -2; break; }
            const int drift1 = (jt - j) - (it - i);
            if (it - i >= 64 || drift1 > 4 || drift1 < -20) break;
            continue;
          }
        }
        const int v = wave_bcast((int)A.x, cl);
        const int cty = wave_bcast(ty, cl);
        const int p0 = wave_bcast((int)B.x, cl), p1 = wave_bcast((int)B.y, cl), p2 = wave_bcast((int)B.z, cl), p3 = wave_bcast((int)B.w, cl);
        unsigned mp, c1, c2, hts, hs, f1x, f2x;
        if (cty == 2 || !wave_bcast((int)hit, cl)) {
          // outside the window (the path drifted off the diagonal of this block) or a row of 32-bit words: direct loads
          const int cb = wave_bcast(b, cl), cro = wave_bcast(ro, cl);
          if (cty == 2) {
            const unsigned wv = c.D()[c.foff()[i] + (j - cb)];
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
        if (it - i >= 64 || drift > 4 || drift < -20) break;
      }
      WSYNC();
    }
  }
  WSYNC();
  PH_MARK(2)
  return rc;
}
#undef mt9
#undef mm9
#undef e1_9
#undef e2_9
#undef o1_9
#undef o2_9
#undef oe1_9
#undef oe2_9
#undef le1
#undef le2

// fuse the aligned subread into the graph, parallel over its bases: vq[q] = graph row node aligned to
// base q (-1 = insertion).  Every graph node is touched by at most one base, so targets, new-node
// ids (prefix sum), anchors (prefix max) and the Q+1 edges are all independent.
// poa_align as a REAL CALL, for the WIDE instance (subreads beyond 1 792 bases: cfg4, cfgL).  Inlined into the kernel that instance spilled
// 176-220 bytes per lane around its row loops; as a function of its own -- the slot pointers and sizes copied into scalars at entry, so
// that nothing in the row loop goes through the Ctx reference -- it runs cfgL's alignments 11 % faster and cfg4's 2.7 %
// (profiles/r05_ab_poa_align_call.txt).  The NARROW instance is the other way round (+10 % as a call) and stays inlined.
__device__ __forceinline__ void* uni_ptr64(const void* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (void*)(((unsigned long long)hi << 32) | lo);
}
template <bool W32, bool DEF, bool WIDE>
__device__ __attribute__((noinline)) int poa_align_call(Ctx c, const C3Params& P_in, int qb_in, int Q_in, int lane, long long* cells PHA) {
  const C3Params P = P_in;            // (Ctx by VALUE: a reference would put the caller's copy in memory for the rest of the kernel)
  const int qb = __builtin_amdgcn_readfirstlane(qb_in), Q = __builtin_amdgcn_readfirstlane(Q_in);
  c.I = (int*)uni_ptr64(c.I); c.E = (int*)uni_ptr64(c.E); c.C = (char*)uni_ptr64(c.C); c.B8 = (uint8_t*)uni_ptr64(c.B8);
  c.score_ = (long long*)uni_ptr64(c.score_); c.desc_ = (uint4*)uni_ptr64(c.desc_); c.jump_ = (int*)uni_ptr64(c.jump_); c.path_ = (int*)uni_ptr64(c.path_);
  c.pk = (const uint32_t*)uni_ptr64(c.pk);
  c.K = __builtin_amdgcn_readfirstlane(c.K); c.n = __builtin_amdgcn_readfirstlane(c.n); c.Ncap = __builtin_amdgcn_readfirstlane(c.Ncap);
  c.cells_cap = __builtin_amdgcn_readfirstlane(c.cells_cap); c.far_shift = __builtin_amdgcn_readfirstlane(c.far_shift);
  c.osel = __builtin_amdgcn_readfirstlane(c.osel); c.rb_span = __builtin_amdgcn_readfirstlane(c.rb_span);
  return poa_align<W32, DEF, WIDE>(c, P, qb, Q, lane, cells PHP);
}

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
  g_reorder(c, n_old, lane, (int*)&L.ring[0], POA_LDS_INTS);      // H, E1, E2 rings are contiguous and idle after the traceback
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
#ifndef C3_POA_MW
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
#endif

// 6 waves/SIMD (80 VGPRs, 26 spilled outside the row loop) with a 4-row LDS ring (6.7 KB per wave, 24 waves per CU):
// 59.4 ms per 32768 cfg2 reads against 67.8 ms at 4 waves/SIMD with the 8-row ring -- the row loop is a dependent
// chain (scan -> next row), so resident waves are what hides its latency.
// W32 = false: the first pass (16-bit ring, fast / near rows); W32 = true: the second pass for reads whose scratch overflowed the
// typical size or whose scores left the 16-bit range -- 32-bit general rows only
#ifndef C3_POA_WAVES
#define C3_POA_WAVES 6
#endif
template <bool W32, bool DEF, bool WIDE>
#ifdef C3_POA_MW
__global__ __launch_bounds__(64 * C3_POA_MW) void k_poa(PoaArgs a) {
#else
__global__ __launch_bounds__(64, C3_POA_WAVES) void k_poa(PoaArgs a) {
#endif
  const int lane = wave_lane();
  const int slot = blockIdx.x;
  Ctx c;
  const size_t N = (size_t)a.Ncap;
  c.I = a.ibase + (size_t)slot * C3_POA_NI * N; c.path_ = a.pbase + (size_t)slot * a.Pcap; c.E = a.ebase + (size_t)slot * 3 * N * a.K;
  c.far_shift = W32 ? 0 : 2;
  c.C = a.cellsb + (size_t)slot * (2 * (size_t)a.cells_cap + 16 * (size_t)(a.cells_cap >> c.far_shift)); c.B8 = a.bbase + (size_t)slot * 5 * N;
  c.score_ = a.score + (size_t)slot * N; c.desc_ = a.desc + (size_t)slot * 2 * N; c.jump_ = a.jump + (size_t)slot * C3_JUMP_LEVELS * N;
  c.K = a.K; c.Ncap = a.Ncap; c.cells_cap = a.cells_cap; c.osel = 0; c.rb_span = a.rb_span > 0 ? a.rb_span : RB_HI16 - RB_LO16;
#ifdef C3_DEBUG_PUNT
  c.dbg = a.phases;
#endif
#ifdef C3_MW_CHECK
  c.chk = a.phases;
#endif
#ifdef C3_POA_MW
  // (this pass may share its CUs with the persistent k_prep / k_window waves of the rest of the batch: its few waves are the long pole)
  __builtin_amdgcn_s_setprio(3);
  if (threadIdx.x >= 64) {                      // the other waves: wide rows on request, nothing else
    const int wave = (int)(threadIdx.x >> 6);
    const int m8 = S3(a.p.poa_match), x8 = S3(-a.p.poa_mismatch);
    const MwConst kc = {m8 << 6, x8 * 64, S3(a.p.e1) << 6, S3(a.p.e2) << 6, S3(a.p.o1) << 6, S3(a.p.o2) << 6};
    for (;;) {
      MW_BARRIER();
      if (MWT.cmd == 2) return;
      mw_chunks<0>(c, kc, wave, lane);
      MW_BARRIER();
      mw_chunks<1>(c, kc, wave, lane);
      MW_BARRIER();
    }
  }
#endif
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
    int C = 0, fail = 0; bool punted = false;
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
        if (s > 0) { int rc; if constexpr ((WIDE || C3_EXP_CALL_NARROW) && !W32) rc = poa_align_call<W32, DEF, WIDE>(c, a.p, qb, Q, lane, &cells PHP); else rc = poa_align<W32, DEF, WIDE>(c, a.p, qb, Q, lane, &cells PHP); if (rc < 0) { fail = (rc == -4 || rc == -5 || rc == -6) ? 2 : 1; punted = rc == -5 || rc == -6; break; } }
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
            unsigned* const Lf = (unsigned*)&L.ring[0];                          // 64 flag words (the rings are idle here)
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
                Lf[lane] = 0;
                WSYNC();
                if (on && J >= 0) Lf[J] = 1;
                WSYNC();
                on |= (int)Lf[lane];
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
    // a read the slot's scratch was too small for is redone by the same kernel with worst-case scratch (list `overflow`); one whose
    // scores left the 16-bit cells by the 32-bit instance (list `overflow16`); a pass without the list a read needs ends it
    // (a read that still does not fit in pass 2 -- its far arena is a quarter of the cells -- goes on to the 32-bit pass, whose arenas are complete)
    const bool to16 = punted || a.overflow == nullptr;
    int* const olist = to16 ? a.overflow16 : a.overflow;
    const bool redo = fail == 2 && olist != nullptr;
    if (lane == 0) {
      if (redo) olist[atomicAdd(a.counter + (to16 ? 5 : 4), 1)] = rid;
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
#ifdef C3_POA_MW
  if (lane == 0) MWT.cmd = 2;
  MW_BARRIER();
#endif
}

#ifdef C3_POA_MW
extern "C" void c3k_launch_poa_mw(const PoaArgs* a, int slots, hipStream_t stream) {
  hipLaunchKernelGGL((k_poa<true, false, false>), dim3(slots), dim3(64 * C3_POA_MW), 0, stream, *a);
}
#else
extern "C" void c3k_launch_poa(const PoaArgs* a, int slots, int wide32, int wide_ring, hipStream_t stream) {
  const C3Params& p = a->p;
  const bool def = p.poa_match == 5 && p.poa_mismatch == 4 && p.o1 == 4 && p.e1 == 2 && p.o2 == 24 && p.e2 == 1;
#define C3_POA_LAUNCH(W, D, R) hipLaunchKernelGGL((k_poa<W, D, R>), dim3(slots), dim3(64), 0, stream, *a)
  if (wide32) { if (def) C3_POA_LAUNCH(true, true, false); else C3_POA_LAUNCH(true, false, false); }
  else if (wide_ring) { if (def) C3_POA_LAUNCH(false, true, true); else C3_POA_LAUNCH(false, false, true); }
  else { if (def) C3_POA_LAUNCH(false, true, false); else C3_POA_LAUNCH(false, false, false); }
#undef C3_POA_LAUNCH
}
#endif
