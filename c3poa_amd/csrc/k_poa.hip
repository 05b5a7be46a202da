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

#define WSYNC() __syncthreads()
#define SRC 0
#define SNK 1


struct Ctx {
  uint8_t* base; int *n_in, *n_out, *in_from, *out_to, *out_w, *grp, *order, *order2, *index;
  int *gfirst, *glast, *rem, *mpl, *mpr, *rbeg, *rend, *roff, *opn, *opq, *anchor, *path, *col, *col2t, *nxt;
  long long* score;
  int32_t *H, *E1, *E2; uint32_t* D; uint8_t* rows2;
  int K, n, Ncap, cells_cap;
  const uint32_t* pk;        // packed read
};

__device__ __forceinline__ void g_add_edge(Ctx& c, int u, int v, int w) {
  const int K = c.K;
  for (int k = 0; k < c.n_out[u]; ++k)
    if (c.out_to[u * K + k] == v) { c.out_w[u * K + k] += w; return; }
  int no = c.n_out[u], ni = c.n_in[v];
  c.out_to[u * K + no] = v; c.out_w[u * K + no] = w; c.n_out[u] = no + 1;
  c.in_from[v * K + ni] = u; c.n_in[v] = ni + 1;
}

// block extents from order/grp (parallel)
__device__ void g_blocks(Ctx& c, int lane) {
  for (int i = lane; i < c.n; i += 64) { c.gfirst[i] = 1 << 30; c.glast[i] = -1; }
  WSYNC();
  for (int i = lane; i < c.n; i += 64) {
    int r = c.grp[c.order[i]];
    atomicMin(&c.gfirst[r], i); atomicMax(&c.glast[r], i);
  }
  WSYNC();
}

// merge new nodes n_old..n-1 (creation order, anchor[k] = old order index they follow) into order
__device__ void g_reorder(Ctx& c, int n_old, int lane) {
  const int n_new = c.n - n_old;
  // old node at old index i moves to i + #(anchor < i); new node k goes to anchor[k] + 1 + k
  for (int i = lane; i < n_old; i += 64) {
    int lo = 0, hi = n_new;                 // first k with anchor[k] >= i
    while (lo < hi) { int m = (lo + hi) >> 1; if (c.anchor[m] < i) lo = m + 1; else hi = m; }
    c.order2[i + lo] = c.order[i];
  }
  for (int k = lane; k < n_new; k += 64) c.order2[c.anchor[k] + 1 + k] = n_old + k;
  WSYNC();
  for (int i = lane; i < c.n; i += 64) { int v = c.order2[i]; c.order[i] = v; c.index[v] = i; }
  WSYNC();
  g_blocks(c, lane);
}

__device__ __forceinline__ int32_t rdcell(const Ctx& c, const int32_t* a, int pb, int pe, int po, int j) {
  return (j < pb || j > pe) ? C3_NEG : a[po + (j - pb)];
}

// banded global alignment of subread [qb, qb+Q) against the graph; ops are written BACKWARDS
// into opn/opq, returns their count (or <0 on failure)
#ifdef C3_PHASE_PROF
#define PHA , unsigned long long& ph_t0_, unsigned long long (&ph_acc_)[12]
#define PHP , ph_t0_, ph_acc_
#else
#define PHA
#define PHP
#endif
__device__ int poa_align(Ctx& c, const C3Params& P, int qb, int Q, int lane, long long* cells PHA) {
  const int K = c.K, n = c.n;
  const int mt = P.poa_match, mm = -P.poa_mismatch;
  const int e1 = P.e1, e2 = P.e2, oe1 = P.o1 + P.e1, oe2 = P.o2 + P.e2;
  const int w = P.band_b + (int)(P.band_f * (double)Q);
  // remaining length along the heaviest out-edge (reverse sweep; lane 0)
  if (lane == 0) {
    for (int i = n - 1; i >= 0; --i) {
      int v = c.order[i];
      if (v == SNK) { c.rem[v] = -1; continue; }
      int bw = INT32_MIN, bt = SNK;
      for (int k = 0; k < c.n_out[v]; ++k) { int ww = c.out_w[v * K + k]; if (ww > bw) { bw = ww; bt = c.out_to[v * K + k]; } }
      c.rem[v] = c.rem[bt] + 1;
    }
  }
  for (int v = lane; v < n; v += 64) { c.mpl[v] = INT32_MAX / 2; c.mpr[v] = 0; }
  WSYNC();
  PH_MARK(0)
  int ncell = 0;
  for (int idx = 0; idx < n; ++idx) {
    const int v = c.order[idx];
    if (v == SNK) { if (lane == 0) { c.rbeg[idx] = 0; c.rend[idx] = -1; c.roff[idx] = ncell; } continue; }
    int beg, end;
    const int qr = Q - c.rem[v];
    const int nin = c.n_in[v];
    if (v == SRC) { beg = 0; end = min(Q, max(qr, 0) + w); }
    else {
      beg = max(0, min(c.mpl[v], qr) - w);
      end = min(Q, max(c.mpr[v], qr) + w);
      int minb = INT32_MAX, maxe = INT32_MIN;
      for (int k = 0; k < nin; ++k) {
        int pi = c.index[c.in_from[v * K + k]];
        minb = min(minb, c.rbeg[pi]); maxe = max(maxe, c.rend[pi] + 1);
      }
      beg = max(beg, minb); end = min(end, maxe);
    }
    if (end < beg) end = beg - 1;
    const int wd = end - beg + 1;
    if (ncell + wd > c.cells_cap) return -4;
    if (lane == 0) { c.rbeg[idx] = beg; c.rend[idx] = end; c.roff[idx] = ncell; }
    const int ro = ncell;
    ncell += wd;
    const int vb = c.base[v];
    int best = INT32_MIN, bl = 0, br = 0;      // per-lane running row maximum
    int carry1 = C3_NEG2, carry2 = C3_NEG2;     // scan carries: max of ht[k] + e*k over previous chunks
    int prev_ht = C3_NEG;                        // ht of the column left of the chunk
    for (int c0 = 0; c0 < wd; c0 += 64) {
      const int j = beg + c0 + lane;
      const bool act = j <= end;
      int ht, E1v, E2v; uint32_t d = 0;
      if (v == SRC) { ht = (j == 0) ? 0 : C3_NEG; E1v = E2v = C3_NEG; }
      else {
        int M = INT32_MIN, e1b = INT32_MIN, e2b = INT32_MIN, mp = 0, e1p = 0, e2p = 0, e1x = 0, e2x = 0;
        for (int k = 0; k < nin; ++k) {
          const int pi = c.index[c.in_from[v * K + k]];
          const int pb = c.rbeg[pi], pe = c.rend[pi], po = c.roff[pi];
          int hd = (j > 0) ? rdcell(c, c.H, pb, pe, po, j - 1) : C3_NEG;
          if (hd > M) { M = hd; mp = k; }
          int hp = rdcell(c, c.H, pb, pe, po, j);
          int a = hp - oe1, bb = rdcell(c, c.E1, pb, pe, po, j) - e1;
          int cnd = a >= bb ? a : bb, x = bb > a;
          if (cnd > e1b) { e1b = cnd; e1p = k; e1x = x; }
          a = hp - oe2; bb = rdcell(c, c.E2, pb, pe, po, j) - e2;
          cnd = a >= bb ? a : bb; x = bb > a;
          if (cnd > e2b) { e2b = cnd; e2p = k; e2x = x; }
        }
        if (j > 0) { int qc = act ? c3_code_at(c.pk, qb + j - 1) : 0; M += (vb == qc) ? mt : mm; } else M = C3_NEG;
        int hts;
        if (M >= e1b && M >= e2b) { hts = 0; ht = M; }
        else if (e1b >= e2b) { hts = 1; ht = e1b; }
        else { hts = 2; ht = e2b; }
        E1v = e1b; E2v = e2b;
        d = (uint32_t)mp | ((uint32_t)e1p << 8) | ((uint32_t)e2p << 16) | ((uint32_t)hts << 24)
            | ((uint32_t)e1x << 28) | ((uint32_t)e2x << 29);
      }
      // horizontal states: F[j] = max_{beg<=k<j} ht[k] - o - e*(j-k)
      const int htm = act ? ht : C3_NEG2;                    // inactive lanes must not win the scans
      const int x1 = htm + e1 * j, x2 = htm + e2 * j;
      const int s1 = wave_scan_max(x1), s2 = wave_scan_max(x2);
      const int px1 = max(wave_shr1(s1, C3_NEG2), carry1);   // exclusive prefix incl. carry
      const int px2 = max(wave_shr1(s2, C3_NEG2), carry2);
      const int htl = wave_shr1(htm, prev_ht);                 // ht[j-1]
      int f1, f2, f1x = 0, f2x = 0;
      if (j == beg) { f1 = f2 = C3_NEG2; }
      else {
        f1 = px1 - P.o1 - e1 * j; f2 = px2 - P.o2 - e2 * j;
        f1x = f1 != htl - oe1; f2x = f2 != htl - oe2;
      }
      carry1 = max(carry1, wave_bcast(s1, 63)); carry2 = max(carry2, wave_bcast(s2, 63));
      prev_ht = wave_bcast(htm, 63);
      int hs, hh;
      if (ht >= f1 && ht >= f2) { hs = 0; hh = ht; }
      else if (f1 >= f2) { hs = 1; hh = f1; }
      else { hs = 2; hh = f2; }
      d |= ((uint32_t)hs << 26) | ((uint32_t)f1x << 30) | ((uint32_t)f2x << 31);
      if (act) {
        const int ci = ro + c0 + lane;
        c.H[ci] = hh; c.E1[ci] = E1v; c.E2[ci] = E2v; c.D[ci] = d;
        if (hh > best) { best = hh; bl = br = j; } else if (hh == best) br = j;
      }
    }
    // row maximum: leftmost / rightmost argmax -> adaptive band hints of the successors
    const int rb = wave_max(best);
    const int left = wave_min(best == rb ? bl : INT32_MAX / 2);
    const int right = wave_max(best == rb ? br : -1);
    if (wd > 0 && lane == 0) {
      for (int k = 0; k < c.n_out[v]; ++k) {
        int t = c.out_to[v * K + k];
        if (right + 1 > c.mpr[t]) c.mpr[t] = right + 1;
        if (left + 1 < c.mpl[t]) c.mpl[t] = left + 1;
      }
    }
    WSYNC();
  }
  *cells += ncell;
  PH_MARK(1)
  // end cell + traceback (lane 0), ops stored backwards
  int nops = 0;
  if (lane == 0) {
    int bi = -1, bs = INT32_MIN;
    for (int k = 0; k < c.n_in[SNK]; ++k) {
      int pi = c.index[c.in_from[SNK * K + k]];
      int hh = rdcell(c, c.H, c.rbeg[pi], c.rend[pi], c.roff[pi], Q);
      if (hh > bs) { bs = hh; bi = pi; }
    }
    if (bi < 0 || bs <= C3_NEG / 2) nops = -1;
    else {
      int i = bi, j = Q, st = 0;
      while (!(i == 0 && j == 0)) {
        if (j < c.rbeg[i] || j > c.rend[i]) { nops = -2; break; }
        const uint32_t d = c.D[c.roff[i] + (j - c.rbeg[i])];
        const int v = c.order[i];
        if (st == 0) { int hs = (d >> 26) & 3; st = hs == 0 ? 1 : (hs == 1 ? 4 : 5); }
        else if (st == 1) {
          int hts = (d >> 24) & 3;
          if (hts == 0) { c.opn[nops] = v; c.opq[nops] = j - 1; ++nops; i = c.index[c.in_from[v * K + (d & 0xff)]]; --j; st = 0; }
          else st = hts == 1 ? 2 : 3;
        }
        else if (st == 2) { c.opn[nops] = v; c.opq[nops] = -1; ++nops; i = c.index[c.in_from[v * K + ((d >> 8) & 0xff)]]; st = ((d >> 28) & 1) ? 2 : 0; }
        else if (st == 3) { c.opn[nops] = v; c.opq[nops] = -1; ++nops; i = c.index[c.in_from[v * K + ((d >> 16) & 0xff)]]; st = ((d >> 29) & 1) ? 3 : 0; }
        else if (st == 4) { c.opn[nops] = -1; c.opq[nops] = j - 1; ++nops; st = ((d >> 30) & 1) ? 4 : 1; --j; }
        else { c.opn[nops] = -1; c.opq[nops] = j - 1; ++nops; st = ((d >> 31) & 1) ? 5 : 1; --j; }
      }
    }
  }
  nops = wave_first(nops);
  WSYNC();
  PH_MARK(2)
  return nops;
}

// fuse the aligned subread (ops backwards in opn/opq; nops<0 means "first sequence": all inserts)
__device__ void poa_fuse(Ctx& c, int nops, int qb, int Q, int* path, int lane PHA) {
  const int n_old = c.n;
  if (lane == 0) {
    int n_new = 0, prev = SRC, cur_anchor = 0, nn = c.n;
    const int total = nops < 0 ? Q : nops;
    for (int t = 0; t < total; ++t) {
      int v, qp;
      if (nops < 0) { v = -1; qp = t; } else { v = c.opn[nops - 1 - t]; qp = c.opq[nops - 1 - t]; }
      if (qp < 0) continue;
      const int cb = c3_code_at(c.pk, qb + qp);
      int tnode;
      if (v >= 0) {
        const int r = c.grp[v];
        cur_anchor = c.glast[r];
        tnode = -1;
        if (c.base[v] == cb) tnode = v;
        else
          for (int i = c.gfirst[r]; i <= c.glast[r]; ++i) { int x = c.order[i]; if (c.base[x] == cb) { tnode = x; break; } }
        if (tnode < 0) { tnode = nn++; c.base[tnode] = (uint8_t)cb; c.n_in[tnode] = 0; c.n_out[tnode] = 0; c.grp[tnode] = r; c.anchor[n_new++] = cur_anchor; }
      } else {
        tnode = nn++; c.base[tnode] = (uint8_t)cb; c.n_in[tnode] = 0; c.n_out[tnode] = 0; c.grp[tnode] = tnode; c.anchor[n_new++] = cur_anchor;
      }
      g_add_edge(c, prev, tnode, 1);
      path[qp] = tnode;
      prev = tnode;
    }
    g_add_edge(c, prev, SNK, 1);
    c.rem[0] = nn;    // hand the new node count to the other lanes through scratch
  }
  WSYNC();
  c.n = c.rem[0];
  WSYNC();
  PH_MARK(3)
  g_reorder(c, n_old, lane);
  PH_MARK(4)
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

__global__ __launch_bounds__(64) void k_poa(PoaArgs a) {
  const int lane = wave_lane();
  const int slot = blockIdx.x;
  Ctx c;
  const size_t N = (size_t)a.Ncap, NK = (size_t)a.Ncap * a.K;
  c.base = a.base + slot * N; c.n_in = a.n_in + slot * N; c.n_out = a.n_out + slot * N;
  c.in_from = a.in_from + slot * NK; c.out_to = a.out_to + slot * NK; c.out_w = a.out_w + slot * NK;
  c.grp = a.grp + slot * N; c.order = a.order + slot * N; c.order2 = a.order2 + slot * N; c.index = a.index + slot * N;
  c.gfirst = a.gfirst + slot * N; c.glast = a.glast + slot * N; c.rem = a.rem + slot * N;
  c.mpl = a.mpl + slot * N; c.mpr = a.mpr + slot * N; c.rbeg = a.rbeg + slot * N; c.rend = a.rend + slot * N; c.roff = a.roff + slot * N;
  c.opn = a.opn + slot * 2 * N; c.opq = a.opq + slot * 2 * N; c.anchor = a.anchor + slot * N;
  c.path = a.path + (size_t)slot * a.Pcap; c.col = a.col + slot * N; c.col2t = a.col2t + slot * N; c.nxt = a.nxt + slot * N;
  c.score = a.score + slot * N;
  c.H = a.H + (size_t)slot * a.cells_cap; c.E1 = a.E1 + (size_t)slot * a.cells_cap; c.E2 = a.E2 + (size_t)slot * a.cells_cap;
  c.D = a.D + (size_t)slot * a.cells_cap; c.rows2 = a.rows2 + slot * 4 * N;
  c.K = a.K; c.Ncap = a.Ncap; c.cells_cap = a.cells_cap;
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
      if (lane == 0) {
        c.base[SRC] = 0; c.base[SNK] = 0; c.n_in[SRC] = c.n_out[SRC] = c.n_in[SNK] = c.n_out[SNK] = 0;
        c.grp[SRC] = SRC; c.grp[SNK] = SNK; c.order[0] = SRC; c.order[1] = SNK; c.index[SRC] = 0; c.index[SNK] = 1;
      }
      c.n = 2;
      WSYNC();
      g_blocks(c, lane);
      int poff = 0;
      for (int s = 0; s < ns && !fail; ++s) {
        const int qb = info->sub_beg[s], Q = info->sub_end[s] - qb;
        int nops = -1;
        if (s > 0) { nops = poa_align(c, a.p, qb, Q, lane, &cells PHP); if (nops < 0) { fail = 1; break; } }
        poa_fuse(c, nops, qb, Q, c.path + poff, lane PHP);
        poff += Q;
      }
      PH_MARK(9)
      if (!fail) {
        // ---- MSA columns = aligned blocks in topological order
        if (lane == 0) {
          int nc = 0, prev_rep = -1;
          for (int i = 0; i < c.n; ++i) {
            int v = c.order[i];
            if (v == SRC || v == SNK) { c.col[v] = -1; continue; }
            if (c.grp[v] != prev_rep) { ++nc; prev_rep = c.grp[v]; }
            c.col[v] = nc - 1;
          }
          c.rem[0] = nc;
        }
        WSYNC();
        PH_MARK(5)
        const int ncol = c.rem[0];
        for (int i = lane; i < ncol; i += 64) c.col2t[i] = -1;
        if (a.msa_dbg) {                       // res.msa_seq rows (codes, 4 = gap), row-major
          uint8_t* dbg = a.msa_dbg + a.msa_off[rid];
          for (int i = lane; i < ns * ncol; i += 64) dbg[i] = 4;
          WSYNC();
          int po = 0;
          for (int s = 0; s < ns; ++s) {
            const int Q = info->sub_end[s] - info->sub_beg[s];
            for (int k = lane; k < Q; k += 64) { int v = c.path[po + k]; dbg[(size_t)s * ncol + c.col[v]] = c.base[v]; }
            po += Q;
          }
          if (lane == 0) a.msa_len[rid] = ncol;
        }
        WSYNC();
        if (ns == 2) {
          // rows (codes, 4 = gap) -> bin/consensus.py pairwise_consensus
          uint8_t* rowA = c.rows2; uint8_t* rowB = c.rows2 + ncol;
          uint8_t* qa = c.rows2 + 2 * (size_t)ncol; uint8_t* qb_ = c.rows2 + 3 * (size_t)ncol;
          for (int i = lane; i < 2 * ncol; i += 64) c.rows2[i] = 4;
          WSYNC();
          const int b0 = info->sub_beg[0], l0 = info->sub_end[0] - b0, b1 = info->sub_beg[1], l1 = info->sub_end[1] - b1;
          for (int k = lane; k < l0; k += 64) { int v = c.path[k]; rowA[c.col[v]] = c.base[v]; }
          for (int k = lane; k < l1; k += 64) { int v = c.path[l0 + k]; rowB[c.col[v]] = c.base[v]; }
          WSYNC();
          if (lane == 0) {
            // seqDict collision: identical subreads share the later quality (consensus.py:77-79)
            bool same = (l0 == l1);
            for (int k = 0; same && k < l0; ++k) same = c3_code_at(c.pk, b0 + k) == c3_code_at(c.pk, b1 + k);
            normalize_len(rowA, ncol, same ? qual + b1 : qual + b0, l0, qa);
            normalize_len(rowB, ncol, qual + b1, l1, qb_);
            int o = 0, i = 0;
            while (i != ncol) {
              const int A = rowA[i], B = rowB[i];
              if (A == B) { if (A != 4) { c.col2t[i] = o; draft[o++] = (uint8_t)A; } }
              if (A != B && A != 4 && B != 4) { c.col2t[i] = o; draft[o++] = (uint8_t)((qa[i] > qb_[i]) ? A : B); }
              if (A == 4 || B == 4) {
                int gl = 1; const uint8_t* gs = (A == 4) ? rowA : rowB;
                for (;;) { if (i + gl >= ncol) { gl = 1; break; } if (gs[i + gl] == 4) ++gl; else break; }
                long sa = 0, sb = 0;
                for (int k = i; k < i + gl && k < ncol; ++k) { sa += qa[k]; sb += qb_[k]; }
                const uint8_t* srcr = (sa > sb) ? rowA : rowB;
                for (int k = i; k < i + gl && k < ncol; ++k) if (srcr[k] != 4) { c.col2t[k] = o; draft[o++] = srcr[k]; }
                i += gl; continue;
              }
              ++i;
            }
            c.rem[0] = o;
          }
          WSYNC();
          C = c.rem[0];
        } else {
          // ---- heaviest bundling (reverse sweep, lane 0)
          if (lane == 0) {
            const int K = c.K;
            for (int i = c.n - 1; i >= 0; --i) {
              int v = c.order[i];
              if (v == SNK) { c.score[v] = 0; c.nxt[v] = -1; continue; }
              int bw = INT32_MIN, bt = -1;
              for (int k = 0; k < c.n_out[v]; ++k) {
                int t = c.out_to[v * K + k], ww = c.out_w[v * K + k];
                if (ww > bw) { bw = ww; bt = t; }
                else if (ww == bw && c.score[bt] <= c.score[t]) bt = t;
              }
              c.nxt[v] = bt; c.score[v] = (long long)bw + c.score[bt];
            }
            int o = 0;
            for (int v = c.nxt[SRC]; v != SNK && v >= 0; v = c.nxt[v]) { draft[o] = c.base[v]; c.col2t[c.col[v]] = o; ++o; }
            c.rem[0] = o;
          }
          WSYNC();
          C = c.rem[0];
        }
        PH_MARK(6)
        // ---- subread -> draft coordinates
        int poff2 = 0;
        for (int s = 0; s < ns; ++s) {
          const int qb = info->sub_beg[s], Q = info->sub_end[s] - qb;
          for (int k = lane; k < Q; k += 64) tpos[qb + k] = c.col2t[c.col[c.path[poff2 + k]]];
          poff2 += Q;
        }
      }
    }
    WSYNC();
    PH_MARK(7)
    if (lane == 0) {
      info->draft_len = C;
      if (fail) { info->status = C3_ST_LIMIT; info->draft_len = 0; }
      else if (C == 0) info->status = C3_ST_NO_CONSENSUS;
      atomicAdd((unsigned long long*)(a.counter + 2), (unsigned long long)cells);
    }
    WSYNC();
  }
  PH_FLUSH(a.phases)
}

extern "C" void c3k_launch_poa(const PoaArgs* a, int slots, hipStream_t stream) {
  hipLaunchKernelGGL(k_poa, dim3(slots), dim3(64), 0, stream, *a);
}
