// k_polish.hip -- K4: racon-equivalent windowed, quality-weighted POA polish of the draft.
//
// Replaces, per read (paths relative to /root/reference):
//   bin/determine_consensus.py:56-82  mappy overlaps of kept + dangling subreads vs the draft
//   bin/determine_consensus.py:87-99  racon <subreads.fastq> <overlaps.paf> <draft.fasta> -q 5 -t 1
// (mappy and racon are external, un-vendored; spec frozen in DESIGN.md 4.5/4.6 and restated by
// oracle/c3o_polish.c -- bit-exact with it.)
//
//   k_prep    one wave per read: anchored banded extension of the dangling pieces against the
//             draft, racon's layer filters, layers cut at 500-nt window boundaries.
//   k_window  one wave per window: spoa-style global linear-gap POA of every layer (sub-graph
//             when a layer does not span the window), heaviest-bundle consensus, coverage trim.
//             DP rows are register-blocked: each lane owns CPL consecutive columns, the row of a
//             predecessor is fetched once with CPL coalesced loads, the horizontal gap is an
//             in-lane prefix plus ONE DPP max-scan per row.
//   k_stitch  one wave per read: window consensi concatenated into the final sequence.
#include "c3_dev.h"
#include "c3_args.h"

#define WSYNC() __syncthreads()



// anchored banded extension alignment (oracle/c3o_polish.c: extend_align).  piece base k is
// read position pb + dir_*k, draft base t is draft position db + dir_*t (dir_ = -1 for the front
// piece: both sequences reversed).  Writes tpos for the aligned piece bases.
__device__ long long extend_align(const PrepArgs& a, const uint32_t* pk, const uint8_t* draft, int C,
                                  int pb, int n, int dir_, int32_t* tpos, int32_t* H, uint8_t* D, int lane) {
  const int W = a.p.dang_band, bw = 2 * W + 1;
  const int mt = a.p.pol_match, mm = a.p.pol_mismatch, g = a.p.pol_gap;
  const int db = dir_ > 0 ? 0 : C - 1;
  for (int k = lane; k < n; k += 64) tpos[pb + dir_ * k] = -1;
  if ((long long)(n + 1) * bw > a.ecap) return -1;
  int best = 0, bi = 0, bj = 0;
  long long cells = 0;
  for (int i = 0; i <= n; ++i) {
    const int jlo = max(0, i - W), jhi = min(C, i + W);
    const int pc = i > 0 ? c3_code_at(pk, pb + dir_ * (i - 1)) : 0;
    int carry = C3_NEG2;          // max over previous chunks of (Hv - g*b)
    for (int c0 = 0; c0 < bw; c0 += 64) {
      const int bb = c0 + lane, j = i - W + bb;
      const bool act = bb < bw && j >= jlo && j <= jhi;
      int hv = C3_NEG2, dirv = 0;
      if (act) {
        if (i == 0) { hv = C3_NEG2; }
        else {
          // diag (i-1, j-1): same band offset; up (i-1, j): offset bb+1
          const bool dok = j > 0 && (j - 1) >= max(0, i - 1 - W) && (j - 1) <= min(C, i - 1 + W);
          const bool uok = j >= max(0, i - 1 - W) && j <= min(C, i - 1 + W);
          int b2 = INT32_MIN;
          if (dok) { b2 = H[(size_t)(i - 1) * bw + bb] + ((pc == (int)draft[db + dir_ * (j - 1)]) ? mt : mm); dirv = 0; }
          if (uok) { int u = H[(size_t)(i - 1) * bw + bb + 1] + g; if (u > b2) { b2 = u; dirv = 1; } }
          hv = b2 == INT32_MIN ? C3_NEG2 : b2;
        }
      }
      int hh;
      if (i == 0) { hh = j * g; dirv = 2; }
      else {
        // left neighbour through a max-scan: H[b] = max(hv[b], max_{b'<b, same row} hv[b'] + g*(b-b'))
        const int x = act ? hv - g * bb : C3_NEG2;
        const int s = wave_scan_max(x);
        const int ex = max(wave_shr1(s, C3_NEG2), carry);
        carry = max(carry, wave_bcast(s, 63));
        const int lf = (j > jlo) ? ex + g * bb : INT32_MIN;
        hh = hv;
        if (lf > hv) { hh = lf; dirv = 2; }
      }
      if (act) {
        H[(size_t)i * bw + bb] = hh; D[(size_t)i * bw + bb] = (uint8_t)dirv;
        ++cells;
        if (i > 0 && j > 0 && hh > best) { best = hh; bi = i; bj = j; }
      }
    }
    WSYNC();
  }
  // first maximum in row-major order
  const int gb = wave_max(best);
  const int gi = wave_min(best == gb ? bi : INT32_MAX / 2);
  const int gj = wave_min((best == gb && bi == gi) ? bj : INT32_MAX / 2);
  if (gb > 0 && lane == 0) {
    int i = gi, j = gj;
    while (i > 0 || j > 0) {
      const int d = D[(size_t)i * bw + (j - i + W)];
      if (d == 0) { tpos[pb + dir_ * (i - 1)] = db + dir_ * (j - 1); --i; --j; }
      else if (d == 1) --i;
      else --j;
    }
  }
  WSYNC();
  long long tot = cells;
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  return tot;
}

__global__ __launch_bounds__(64) void k_prep(PrepArgs a) {
  const int lane = wave_lane();
  const int slot = blockIdx.x;
  int32_t* eH = a.eH + (size_t)slot * a.ecap;
  uint8_t* eD = a.eD + (size_t)slot * a.ecap;
  int* lwf = a.lw_first + (size_t)slot * a.NLcap * a.NWcap;
  int* lwl = a.lw_last + (size_t)slot * a.NLcap * a.NWcap;
  const int WL = a.p.pol_window;
  for (;;) {
    int wi = 0;
    if (lane == 0) wi = atomicAdd(a.counter, 1);
    wi = wave_first(wi);
    if (wi >= a.n_work) break;
    const int rid = a.work[wi];
    C3Info* info = &a.info[rid];
    if (info->status != C3_ST_OK || info->draft_len <= 0) { if (lane == 0) { info->n_win = 0; a.win_base[rid] = 0; } continue; }
    const int64_t off = a.b.off[rid];
    const int L = (int)(a.b.off[rid + 1] - off);
    const uint32_t* pk = a.b.pk + a.b.woff[rid];
    const uint8_t* qual = a.b.qual + off;
    const uint8_t* draft = a.draft + off;
    int32_t* tpos = a.tpos + off;
    const int C = info->draft_len, ns = info->n_sub;
    const int hf = info->has_front, ht = info->has_tail;
    long long cells = 0;
    // ---- dangling pieces (tail: anchored at the draft start; front: at the draft end)
    if (ht) { long long r = extend_align(a, pk, draft, C, info->tail_beg, L - info->tail_beg, +1, tpos, eH, eD, lane); if (r > 0) cells += r; }
    if (hf) { long long r = extend_align(a, pk, draft, C, info->front_end - 1, info->front_end, -1, tpos, eH, eD, lane); if (r > 0) cells += r; }
    // ---- layers: kept subreads, front, tail
    const int nl = ns + hf + ht;
    const int nwin = (C + WL - 1) / WL;
    if (nwin > a.NWcap || nl > a.NLcap) { if (lane == 0) { info->status = C3_ST_LIMIT; info->n_win = 0; a.win_base[rid] = 0; } continue; }
    long tl = 0;
    for (int i = 0; i < nl; ++i) {
      int lb, le;
      if (i < ns) { lb = info->sub_beg[i]; le = info->sub_end[i]; }
      else if (i == ns && hf) { lb = 0; le = info->front_end; }
      else { lb = info->tail_beg; le = L; }
      tl += le - lb;
    }
    const int tgs = tl > 1000L * nl;
    int wbase = 0;
    if (lane == 0) wbase = atomicAdd(a.n_windows, nwin);
    wbase = wave_first(wbase);
    if (wbase + nwin > a.wcap) { if (lane == 0) { info->status = C3_ST_LIMIT; info->n_win = 0; a.win_base[rid] = 0; } continue; }
    for (int i = lane; i < nl * nwin; i += 64) { lwf[i] = INT32_MAX; lwl[i] = -1; }
    for (int w = lane; w < nwin; w += 64) {
      WinRec r; r.rid = rid; r.w = w; r.n_layers = 0; r.blen = (w * WL + WL <= C) ? WL : C - w * WL; r.tgs = tgs; r.out_len = 0; r.polished = 0; r.pad_ = 0;
      a.wrec[wbase + w] = r;
    }
    WSYNC();
    for (int i = 0; i < nl; ++i) {
      int lb, le;
      if (i < ns) { lb = info->sub_beg[i]; le = info->sub_end[i]; }
      else if (i == ns && hf) { lb = 0; le = info->front_end; }
      else { lb = info->tail_beg; le = L; }
      // first / last aligned base of the layer, and per window
      int qf = INT32_MAX, ql = -1;
      for (int k = lb + lane; k < le; k += 64) {
        int t = tpos[k];
        if (t >= 0) { qf = min(qf, k); ql = max(ql, k); int w = t / WL; atomicMin(&lwf[i * nwin + w], k); atomicMax(&lwl[i * nwin + w], k); }
      }
      qf = wave_min(qf); ql = wave_max(ql);
      WSYNC();
      if (ql < 0) continue;
      const int qs = ql + 1 - qf, ts = tpos[ql] + 1 - tpos[qf];
      const double err = 1.0 - (double)min(qs, ts) / (double)max(qs, ts);
      if (err > 0.3) continue;                 // racon error threshold
      for (int w = 0; w < nwin; ++w) {
        const int fq = lwf[i * nwin + w], lq = lwl[i * nwin + w];
        if (lq < 0) continue;
        const int seglen = lq + 1 - fq;
        if ((double)seglen < 0.02 * WL) continue;
        long qsum = 0;
        for (int x = fq + lane; x <= lq; x += 64) qsum += (int)qual[x] - 33;
        for (int o = 32; o > 0; o >>= 1) qsum += __shfl_xor(qsum, o);
        if (qsum < (long)a.p.pol_q * seglen) continue;
        if (lane == 0) {
          WinRec* r = &a.wrec[wbase + w];
          WLayer* dst = &a.wlay[(size_t)(wbase + w) * a.NLcap + r->n_layers];
          dst->qbeg = fq; dst->len = seglen; dst->begin = tpos[fq] - w * WL; dst->end = tpos[lq] - w * WL;
          r->n_layers++;
        }
      }
      WSYNC();
    }
    if (lane == 0) {
      info->n_win = nwin; a.win_base[rid] = wbase;
      atomicAdd((unsigned long long*)(a.counter + 2), (unsigned long long)cells);
    }
    WSYNC();
  }
}

// ------------------------------------------------------------------------------------------

struct WCtx {
  uint8_t* base; int *n_in, *n_out, *in_from, *in_w, *out_to, *out_w, *grp, *order, *order2, *index;
  int *gfirst, *glast, *ncov, *rowof, *rows, *anchor, *opn, *opq, *pred; uint8_t* mask; long long* score;
  int32_t* H; uint16_t* D; uint4* rdesc;
  int K, n, Ncap; long long hcap;
};

__device__ void w_blocks(WCtx& c, int lane) {
  for (int i = lane; i < c.n; i += 64) { c.gfirst[i] = 1 << 30; c.glast[i] = -1; }
  WSYNC();
  for (int i = lane; i < c.n; i += 64) { int r = c.grp[c.order[i]]; atomicMin(&c.gfirst[r], i); atomicMax(&c.glast[r], i); }
  WSYNC();
}
__device__ void w_reorder(WCtx& c, int n_old, int lane) {
  const int n_new = c.n - n_old;
  for (int i = lane; i < n_old; i += 64) {
    int lo = 0, hi = n_new;
    while (lo < hi) { int m = (lo + hi) >> 1; if (c.anchor[m] < i) lo = m + 1; else hi = m; }
    c.order2[i + lo] = c.order[i];
  }
  for (int k = lane; k < n_new; k += 64) c.order2[c.anchor[k] + 1 + k] = n_old + k;
  WSYNC();
  for (int i = lane; i < c.n; i += 64) { int v = c.order2[i]; c.order[i] = v; c.index[v] = i; }
  WSYNC();
  w_blocks(c, lane);
}

// Row descriptors of one alignment, built in parallel before the DP so that the row loop has no
// dependent graph loads: x = base | np<<8 | overflow<<16, y = p0 | p1<<16, z = p2 | p3<<16
// (p = DP row of a masked predecessor in in-edge order; row 0 = the virtual start row).
__device__ void win_build_desc(WCtx& c, int R, int lane) {
  const int K = c.K;
  for (int r = 1 + lane; r <= R; r += 64) {
    const int v = c.rows[r];
    const int nin = c.n_in[v];
    unsigned p[4] = {0, 0, 0, 0};
    int np = 0;
    for (int k = 0; k < nin; ++k) {
      const int pr = c.rowof[c.in_from[v * K + k]];
      if (pr < 0) continue;
      if (np < 4) p[np] = (unsigned)pr;
      ++np;
    }
    unsigned ovf = np > 4;
    if (np == 0) np = 1;                      // no masked predecessor: virtual row 0
    uint4 d; d.x = (unsigned)c.base[v] | ((unsigned)min(np, 255) << 8) | (ovf << 16);
    d.y = p[0] | (p[1] << 16); d.z = p[2] | (p[3] << 16); d.w = 0;
    c.rdesc[r] = d;
  }
  WSYNC();
}

// All DP rows of one layer.  Lane owns columns lane*CPL .. lane*CPL+CPL-1 (element (lane,cc) of a
// row lives at cc*64+lane), so every lane only ever re-reads cells it stored itself: no barrier in
// the row loop.  The row just computed stays in registers and feeds the next row directly (the
// common predecessor); older predecessor rows are fetched with CPL coalesced loads.
// D cell = type (0 diag, 1 vertical, 2 horizontal) | (row - predecessor row) << 2.
template <int CPL>
__device__ int win_rows(WCtx& c, const C3Params& P, const uint32_t* pk, int qbeg, int Q, int R, int lane) {
  const int RS = 64 * CPL, K = c.K;
  const int mt = P.pol_match, mm = P.pol_mismatch, g = P.pol_gap;
  if ((long long)(R + 1) * RS > c.hcap || R >= 16383) return -1;
  int qc[CPL], hcur[CPL];
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    const int j = lane * CPL + cc;
    qc[cc] = (j >= 1 && j <= Q) ? c3_code_at(pk, qbeg + j - 1) : 7;
    hcur[cc] = j * g;                                                       // virtual row 0
    if (j <= Q) { c.H[cc * 64 + lane] = j * g; c.D[cc * 64 + lane] = 2; }
  }
  uint4 dn = c.rdesc[1 <= R ? 1 : 0];
  for (int r = 1; r <= R; ++r) {
    const uint4 de = dn;
    if (r < R) dn = c.rdesc[r + 1];                                         // prefetch the next descriptor
    const int vb = de.x & 0xff, np = (de.x >> 8) & 0xff;
    const bool ovf = (de.x >> 16) & 1;
    int bd[CPL], dd[CPL], bv[CPL], dv[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) { bd[cc] = INT32_MIN; bv[cc] = INT32_MIN; dd[cc] = 0; dv[cc] = 0; }
    int seen = 0, kedge = 0;
    for (int t = 0; t < np; ++t) {
      int prow;
      if (!ovf) prow = (t == 0) ? (de.y & 0xffff) : (t == 1) ? (de.y >> 16) : (t == 2) ? (de.z & 0xffff) : (de.z >> 16);
      else {                                                                 // >4 predecessors: walk the in-edges
        const int v = c.rows[r];
        prow = -1;
        while (kedge < c.n_in[v]) { int pr = c.rowof[c.in_from[v * K + kedge]]; ++kedge; if (pr >= 0) { prow = pr; break; } }
        if (prow < 0) break;
      }
      ++seen;
      const int dl = (r - prow) << 2;
      int hp[CPL];
      if (prow == r - 1) {
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) hp[cc] = hcur[cc];
      } else {
        const int32_t* hp_ = c.H + (size_t)prow * RS;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) hp[cc] = hp_[cc * 64 + lane];
      }
      const int hleft = wave_shr1(hp[CPL - 1], INT32_MIN);                   // column lane*CPL-1 of the predecessor
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int hd = cc == 0 ? hleft : hp[cc - 1];
        if (hd != INT32_MIN) { int cnd = hd + ((vb == qc[cc]) ? mt : mm); if (cnd > bd[cc]) { bd[cc] = cnd; dd[cc] = 0 | dl; } }
        int cv = hp[cc] + g;
        if (cv > bv[cc]) { bv[cc] = cv; dv[cc] = 1 | dl; }
      }
    }
    // vertical beats diagonal only when strictly greater (in place: bd/dd become the pre-gap H / dir)
    int run = C3_NEG2;
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
      const int j = lane * CPL + cc;
      if (bv[cc] > bd[cc]) { bd[cc] = bv[cc]; dd[cc] = dv[cc]; }
      const int y = (j <= Q) ? bd[cc] - g * j : C3_NEG2;
      run = max(run, y);
    }
    // horizontal gap: in-lane prefix + one cross-lane max-scan
    const int s = wave_scan_max(run);
    int ex = wave_shr1(s, C3_NEG2);                                          // max over all previous lanes
    int32_t* hrow = c.H + (size_t)r * RS; uint16_t* drow = c.D + (size_t)r * RS;
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
      const int j = lane * CPL + cc;
      int hh = bd[cc], d = dd[cc];
      const int y = (j <= Q) ? hh - g * j : C3_NEG2;
      const int lf = ex + g * j;
      if (j > 0 && lf > hh) { hh = lf; d = 2; }
      ex = max(ex, y);
      hcur[cc] = hh;
      if (j <= Q) { hrow[cc * 64 + lane] = hh; drow[cc * 64 + lane] = (uint16_t)d; }
    }
  }
  WSYNC();
  return 0;
}

// generic fallback for very long segments: linear layout (element j at j), 64 columns per step
__device__ int win_rows_lin(WCtx& c, const C3Params& P, const uint32_t* pk, int qbeg, int Q, int R, int lane) {
  const int RS = ((Q + 1 + 63) / 64) * 64, K = c.K;
  const int mt = P.pol_match, mm = P.pol_mismatch, g = P.pol_gap;
  if ((long long)(R + 1) * RS > c.hcap || R >= 16383) return -1;
  for (int j = lane; j <= Q; j += 64) { c.H[j] = j * g; c.D[j] = 2; }
  WSYNC();
  for (int r = 1; r <= R; ++r) {
    const int v = c.rows[r];
    const int vb = c.base[v];
    const int nin = c.n_in[v];
    int32_t* hrow = c.H + (size_t)r * RS; uint16_t* drow = c.D + (size_t)r * RS;
    int carry = C3_NEG2;
    for (int c0 = 0; c0 <= Q; c0 += 64) {
      const int j = c0 + lane;
      const bool act = j <= Q;
      const int qc = (act && j >= 1) ? c3_code_at(pk, qbeg + j - 1) : 7;
      int bd = INT32_MIN, dd = 0, bv = INT32_MIN, dv = 0, np = 0;
      for (int k = 0; k <= nin; ++k) {
        int prow;
        if (k < nin) { prow = c.rowof[c.in_from[v * K + k]]; if (prow < 0) continue; ++np; }
        else { if (np > 0) break; prow = 0; }
        const int dl = (r - prow) << 2;
        const int32_t* hp_ = c.H + (size_t)prow * RS;
        if (act) {
          if (j > 0) { int cnd = hp_[j - 1] + ((vb == qc) ? mt : mm); if (cnd > bd) { bd = cnd; dd = 0 | dl; } }
          int cv = hp_[j] + g;
          if (cv > bv) { bv = cv; dv = 1 | dl; }
        }
      }
      int hv = bd, dir = dd;
      if (bv > hv) { hv = bv; dir = dv; }
      const int y = act ? hv - g * j : C3_NEG2;
      const int s = wave_scan_max(y);
      const int ex = max(wave_shr1(s, C3_NEG2), carry);
      carry = max(carry, wave_bcast(s, 63));
      const int lf = ex + g * j;
      int hh = hv;
      if (j > 0 && lf > hh) { hh = lf; dir = 2; }
      if (act) { hrow[j] = hh; drow[j] = (uint16_t)dir; }
    }
    WSYNC();
  }
  return 0;
}

// element index of column j inside a row for the layout chosen by win_rows_dispatch
__device__ __forceinline__ int win_idx(int j, int cpl) { return cpl ? (j % cpl) * 64 + j / cpl : j; }

__device__ int win_rows_dispatch(WCtx& c, const C3Params& P, const uint32_t* pk, int qbeg, int Q, int R, int lane, int* cpl_out, int* rs_out) {
  const int need = (Q + 1 + 63) / 64;
  int cpl;
  if (need <= 2) cpl = 2; else if (need <= 4) cpl = 4; else if (need <= 6) cpl = 6; else if (need <= 8) cpl = 8;
  else if (need <= 10) cpl = 10; else cpl = 0;
  *cpl_out = cpl; *rs_out = cpl ? 64 * cpl : need * 64;
  if (cpl) win_build_desc(c, R, lane);
  switch (cpl) {
    case 2: return win_rows<2>(c, P, pk, qbeg, Q, R, lane);
    case 4: return win_rows<4>(c, P, pk, qbeg, Q, R, lane);
    case 6: return win_rows<6>(c, P, pk, qbeg, Q, R, lane);
    case 8: return win_rows<8>(c, P, pk, qbeg, Q, R, lane);
    case 10: return win_rows<10>(c, P, pk, qbeg, Q, R, lane);
    default: return win_rows_lin(c, P, pk, qbeg, Q, R, lane);
  }
}

__global__ __launch_bounds__(64) void k_window(WinArgs a) {
  const int lane = wave_lane();
  const int slot = blockIdx.x;
  WCtx c;
  const size_t N = (size_t)a.Ncap, NK = (size_t)a.Ncap * a.K;
  c.base = a.base + slot * N; c.n_in = a.n_in + slot * N; c.n_out = a.n_out + slot * N;
  c.in_from = a.in_from + slot * NK; c.in_w = a.in_w + slot * NK; c.out_to = a.out_to + slot * NK; c.out_w = a.out_w + slot * NK;
  c.grp = a.grp + slot * N; c.order = a.order + slot * N; c.order2 = a.order2 + slot * N; c.index = a.index + slot * N;
  c.gfirst = a.gfirst + slot * N; c.glast = a.glast + slot * N; c.ncov = a.ncov + slot * N;
  c.rowof = a.rowof + slot * N; c.rows = a.rows + slot * (N + 1); c.anchor = a.anchor + slot * N;
  c.opn = a.opn + slot * 2 * N; c.opq = a.opq + slot * 2 * N; c.pred = a.pred + slot * N;
  c.mask = a.mask + slot * N; c.score = a.score + slot * N;
  c.H = a.H + (size_t)slot * a.hcap; c.D = a.D + (size_t)slot * a.hcap; c.rdesc = a.rdesc + slot * (N + 1);
  c.K = a.K; c.Ncap = a.Ncap; c.hcap = a.hcap;
  const C3Params& P = a.p;
  PH_DECL

  for (;;) {
    int wi = 0;
    if (lane == 0) wi = atomicAdd(a.counter, 1);
    wi = wave_first(wi);
    if (wi >= a.n_win) break;
    PH_MARK(9)
    const WinRec rec = a.wrec_in[wi];
    const int rid = rec.rid, blen = rec.blen, nl = rec.n_layers;
    const int64_t off = a.b.off[rid];
    const uint32_t* pk = a.b.pk + a.b.woff[rid];
    const uint8_t* qual = a.b.qual + off;
    const uint8_t* bb = a.draft + off + (size_t)rec.w * P.pol_window;
    uint8_t* out = a.wout + (size_t)wi * a.wout_cap;
    const WLayer* lay = a.wlay + (size_t)wi * a.NLcap;
    long long cells = 0;
    int olen = 0, polished = 0, fail = 0;
    if (nl + 1 < 3) {
      if (blen > a.wout_cap) fail = 1;
      else { for (int i = lane; i < blen; i += 64) out[i] = bb[i]; olen = blen; }
    } else {
      // ---- backbone chain (weight-0 edges, coverage 1)
      for (int i = lane; i < blen; i += 64) {
        c.base[i] = bb[i]; c.grp[i] = i; c.order[i] = i; c.index[i] = i; c.ncov[i] = 1;
        c.n_in[i] = i > 0; c.n_out[i] = i + 1 < blen;
        if (i > 0) { c.in_from[i * c.K] = i - 1; c.in_w[i * c.K] = 0; }
        if (i + 1 < blen) { c.out_to[i * c.K] = i + 1; c.out_w[i * c.K] = 0; }
      }
      c.n = blen;
      WSYNC();
      w_blocks(c, lane);
      PH_MARK(0)
      // ---- stable order of the layers by begin position (tiny; every lane computes it)
      const int offset = (int)(0.01 * (double)blen);
      for (int t = 0; t < nl && !fail; ++t) {
        // rank t = the layer with t layers before it in (begin, index) order
        int li = -1;
        for (int x = 0; x < nl; ++x) {
          int before = 0;
          for (int y = 0; y < nl; ++y) before += (lay[y].begin < lay[x].begin) || (lay[y].begin == lay[x].begin && y < x);
          if (before == t) { li = x; break; }
        }
        const WLayer l = lay[li];
        const int Q = l.len;
        const bool full = l.begin < offset && l.end > blen - offset;
        // ---- rows of this alignment (masked sub-graph or everything)
        if (!full) {
          for (int i = lane; i < c.n; i += 64) c.mask[i] = 0;
          WSYNC();
          if (lane == 0) {                  // spoa Graph::subgraph: reverse block sweep
            const int K = c.K;
            int i = c.glast[c.grp[l.end]];
            while (i >= 0) {
              const int r = c.grp[c.order[i]], f = c.gfirst[r], la = c.glast[r];
              int seed = 0;
              for (int t2 = f; t2 <= la && !seed; ++t2) {
                int x = c.order[t2];
                if (x < l.begin) continue;
                if (x == l.end) { seed = 1; break; }
                for (int k = 0; k < c.n_out[x]; ++k) if (c.mask[c.out_to[x * K + k]]) { seed = 1; break; }
              }
              if (seed) for (int t2 = f; t2 <= la; ++t2) { int x = c.order[t2]; if (x >= l.begin) c.mask[x] = 1; }
              i = f - 1;
            }
          }
          WSYNC();
        }
        PH_MARK(1)
        int R = 0;
        for (int i0 = 0; i0 < c.n; i0 += 64) {        // order-preserving compaction
          const int i = i0 + lane;
          const int v = i < c.n ? c.order[i] : 0;
          const bool in = i < c.n && (full || c.mask[v]);
          const unsigned long long bal = __ballot(in);
          if (i < c.n) {
            if (in) { int r = R + 1 + __popcll(bal & ((1ull << lane) - 1)); c.rows[r] = v; c.rowof[v] = r; }
            else c.rowof[v] = -1;
          }
          R += __popcll(bal);
        }
        WSYNC();
        PH_MARK(2)
        int cpl = 0, RS = 0;
        if (win_rows_dispatch(c, P, pk, l.qbeg, Q, R, lane, &cpl, &RS) < 0) { fail = 1; break; }
        PH_MARK(3)
        cells += (long long)(R + 1) * (Q + 1);
        // ---- end row: masked nodes without masked successors; first maximum in order
        int bs = INT32_MIN, br = INT32_MAX / 2;
        for (int r = 1 + lane; r <= R; r += 64) {
          const int v = c.rows[r];
          int has = 0;
          for (int k = 0; k < c.n_out[v]; ++k) if (c.rowof[c.out_to[v * c.K + k]] >= 0) { has = 1; break; }
          if (has) continue;
          const int sc = c.H[(size_t)r * RS + win_idx(Q, cpl)];
          if (sc > bs) { bs = sc; br = r; }
        }
        const int gbs = wave_max(bs);
        const int gbr = wave_min(bs == gbs ? br : INT32_MAX / 2);
        PH_MARK(4)
        // ---- traceback (lane 0; one dependent load per step): rq[q] = DP row aligned to query base q, 0 = insertion
        int* rq = c.opq; int* tq = c.opn;
        if (lane == 0) {
          int r = (gbs == INT32_MIN) ? 0 : gbr, j = Q;
          while (r > 0 || j > 0) {
            const int d = c.D[(size_t)r * RS + win_idx(j, cpl)], ty = d & 3;
            if (ty == 2) { rq[j - 1] = 0; --j; continue; }
            if (ty == 0) { rq[j - 1] = r; --j; }
            r -= d >> 2;
          }
        }
        WSYNC();
        PH_MARK(5)
        // ---- fusion, parallel over the query bases (every graph node is touched by at most one base)
        const int n_old = c.n;
        int carry_anchor = -1, carry_new = 0;
        for (int q0 = 0; q0 < Q; q0 += 64) {
          const int q = q0 + lane;
          const bool act = q < Q;
          const int r = act ? rq[q] : 0;
          const int v = r > 0 ? c.rows[r] : -1;
          const int cb = act ? c3_code_at(pk, l.qbeg + q) : 0;
          int tgt = -1, gnew = -1, anc = -1;
          if (v >= 0) {
            const int rr = c.grp[v];
            anc = c.glast[rr];
            if (c.base[v] == cb) tgt = v;
            else for (int i = c.gfirst[rr]; i <= anc; ++i) { int x = c.order[i]; if (c.base[x] == cb) { tgt = x; break; } }
            if (tgt < 0) gnew = rr;
          }
          const int isnew = act && tgt < 0;
          const int as = max(wave_scan_max(anc), carry_anchor);       // anchors are non-decreasing along the path
          carry_anchor = wave_bcast(as, 63);
          const int ps = wave_scan_add(isnew);
          const int k = carry_new + ps - isnew;
          carry_new += wave_bcast(ps, 63);
          if (isnew) {
            const int id = n_old + k;
            if (id < c.Ncap) {
              c.base[id] = (uint8_t)cb; c.n_in[id] = 0; c.n_out[id] = 0; c.grp[id] = gnew >= 0 ? gnew : id; c.ncov[id] = 0;
              c.anchor[k] = as;
            }
            tgt = id;
          }
          if (act) tq[q] = tgt;
        }
        const int nn = n_old + carry_new;
        if (nn > c.Ncap) { fail = 1; break; }
        WSYNC();
        const int K = c.K;
        for (int q = lane; q < Q; q += 64) {
          const int v = tq[q];
          c.ncov[v] += 1;
          if (q == 0) continue;
          const int u = tq[q - 1];
          const int w = ((int)qual[l.qbeg + q - 1] - 33) + ((int)qual[l.qbeg + q] - 33);
          const int no = c.n_out[u];
          int hit = -1;
          for (int k = 0; k < no; ++k) if (c.out_to[u * K + k] == v) { hit = k; break; }
          if (hit >= 0) {
            c.out_w[u * K + hit] += w;
            for (int t2 = 0; t2 < c.n_in[v]; ++t2) if (c.in_from[v * K + t2] == u) { c.in_w[v * K + t2] += w; break; }
          } else {
            const int ni = c.n_in[v];
            c.out_to[u * K + no] = v; c.out_w[u * K + no] = w; c.n_out[u] = no + 1;
            c.in_from[v * K + ni] = u; c.in_w[v * K + ni] = w; c.n_in[v] = ni + 1;
          }
        }
        WSYNC();
        c.n = nn;
        PH_MARK(6)
        w_reorder(c, n_old, lane);
        PH_MARK(7)
      }
      if (!fail) {
        // ---- spoa heaviest bundle + branch completion (lane 0), coverage trim
        if (lane == 0) {
          const int K = c.K, n = c.n;
          for (int v = 0; v < n; ++v) { c.score[v] = -1; c.pred[v] = -1; }
          int max_id = 0;
          for (int i = 0; i < n; ++i) {
            const int v = c.order[i];
            for (int k = 0; k < c.n_in[v]; ++k) {
              const int u = c.in_from[v * K + k]; const long long w = c.in_w[v * K + k];
              if (c.score[v] < w || (c.score[v] == w && c.score[c.pred[v]] <= c.score[u])) { c.score[v] = w; c.pred[v] = u; }
            }
            if (c.pred[v] != -1) c.score[v] += c.score[c.pred[v]];
            if (c.score[max_id] < c.score[v]) max_id = v;
          }
          while (c.n_out[max_id] > 0) {
            const int v = max_id;
            for (int k = 0; k < c.n_out[v]; ++k) {
              const int t2 = c.out_to[v * K + k];
              for (int e = 0; e < c.n_in[t2]; ++e) { int u = c.in_from[t2 * K + e]; if (u != v) c.score[u] = -1; }
            }
            long long ms = 0; int mid = -1;
            for (int i = c.index[v] + 1; i < n; ++i) {
              const int x = c.order[i];
              c.score[x] = -1; c.pred[x] = -1;
              for (int k = 0; k < c.n_in[x]; ++k) {
                const int u = c.in_from[x * K + k]; const long long w = c.in_w[x * K + k];
                if (c.score[u] == -1) continue;
                if (c.score[x] < w || (c.score[x] == w && c.score[c.pred[x]] <= c.score[u])) { c.score[x] = w; c.pred[x] = u; }
              }
              if (c.pred[x] != -1) c.score[x] += c.score[c.pred[x]];
              if (ms < c.score[x]) { ms = c.score[x]; mid = x; }
            }
            if (mid < 0) break;
            max_id = mid;
          }
          // consensus path backwards into opn, then trim + emit
          int nc = 0;
          for (int v = max_id; v != -1; v = c.pred[v]) c.opn[nc++] = v;
          int b = 0, e = nc - 1;      // positions in forward order: forward[i] = opn[nc-1-i]
          if (rec.tgs) {
            const int avg = nl / 2;
            for (; b < nc; ++b) if (c.ncov[c.opn[nc - 1 - b]] >= avg) break;
            for (; e >= 0; --e) if (c.ncov[c.opn[nc - 1 - e]] >= avg) break;
            if (b >= e) { b = 0; e = nc - 1; }
          }
          int o = 0;
          if (e - b + 1 > a.wout_cap) o = -1;
          else for (int i = b; i <= e; ++i) out[o++] = c.base[c.opn[nc - 1 - i]];
          c.pred[0] = o;
        }
        WSYNC();
        olen = c.pred[0];
        WSYNC();
        PH_MARK(8)
        if (olen < 0) { fail = 1; olen = 0; } else polished = 1;
      }
    }
    if (lane == 0) {
      WinRec* r = &a.wrec[wi];
      r->out_len = fail ? -1 : olen; r->polished = polished;
      atomicAdd((unsigned long long*)(a.counter + 2), (unsigned long long)cells);
    }
    WSYNC();
  }
  PH_FLUSH(a.phases)
}

// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_stitch(StitchArgs a) {
  const int lane = wave_lane();
  for (int wi = blockIdx.x; wi < a.n_work; wi += gridDim.x) {
    const int rid = a.work[wi];
    C3Info* info = &a.info[rid];
    if (info->status != C3_ST_OK) continue;
    const int64_t off = a.b.off[rid];
    const int L = (int)(a.b.off[rid + 1] - off);
    const int nwin = info->n_win, wb = a.win_base[rid];
    char* cons = a.cons + off;
    int olen = 0, any = 0, bad = 0;
    for (int w = 0; w < nwin; ++w) {
      const WinRec r = a.wrec[wb + w];
      if (r.out_len < 0) { bad = 1; break; }
      if (olen + r.out_len > L) { bad = 1; break; }
      const uint8_t* src = a.wout + (size_t)(wb + w) * a.wout_cap;
      for (int i = lane; i < r.out_len; i += 64) cons[olen + i] = "ACGT"[src[i] & 3];
      olen += r.out_len; any |= r.polished;
    }
    if (lane == 0) {
      if (bad) { info->status = C3_ST_LIMIT; info->cons_len = 0; }
      else if (!any || olen == 0) { info->status = C3_ST_NO_CONSENSUS; info->cons_len = 0; }   // racon drops unpolished targets
      else info->cons_len = olen;
    }
  }
}

extern "C" void c3k_launch_prep(const PrepArgs* a, int slots, hipStream_t s) { hipLaunchKernelGGL(k_prep, dim3(slots), dim3(64), 0, s, *a); }
extern "C" void c3k_launch_window(const WinArgs* a, int slots, hipStream_t s) { hipLaunchKernelGGL(k_window, dim3(slots), dim3(64), 0, s, *a); }
extern "C" void c3k_launch_stitch(const StitchArgs* a, int grid, hipStream_t s) { hipLaunchKernelGGL(k_stitch, dim3(grid), dim3(64), 0, s, *a); }
