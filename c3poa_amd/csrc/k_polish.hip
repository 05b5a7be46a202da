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
#include <algorithm>

#define WSYNC() __syncthreads()
// adjacency slot k of node v.  Slot-major (all first edges, then all second edges, ...): nearly every node has one or two
// edges, so the arrays a wave actually touches are contiguous runs of Ncap ints instead of one 64-byte line per node
#define EI(v, k) ((size_t)(k) * (size_t)c.Ncap + (size_t)(v))
#ifndef C3_WIN_RING
#define C3_WIN_RING 1          /* LDS ring of the last four kept H rows in k_window's row loop */
#endif



// anchored banded extension alignment (oracle/c3o_polish.c: extend_align).  piece base k is read
// position pb + dir_*k, draft base t is draft position db + dir_*t (dir_ = -1 for the front piece:
// both sequences reversed).  Writes tpos for the aligned piece bases.
//
// Lane owns EC consecutive band offsets (bb = lane*EC + cc; cell (i, j = i - W + bb)); the previous
// row lives in registers (diagonal = same offset, up = offset+1), the left gap is an in-lane prefix
// plus one DPP max-scan, cells are KEY = score*4 + tag (3 diag, 2 up, 1 left) so one v_max per
// candidate keeps the oracle's tie order.  Only the directions go to memory, as a 2-BIT STREAM: a lane's five cells of a row are
// 10 bits, three rows share one dword (word (i-1)/3 of the lane, bits 10*((i-1)%3) + 2*cc; tag 3 diag, 2 up, 1 left, 0 none) that
// is stored once per three rows (round 4; it was 8 bytes per lane and row: 512 bytes per row against 85).
// The draft sits in LDS and the piece bases arrive 64 rows at a time, so the row loop has no loads.
#ifndef C3_PREP_WAVES
#define C3_PREP_WAVES 5
#endif
#define EC 5
#define EXT_DCAP 4096        // draft bases kept in LDS; longer drafts read the global copy
// DL: the draft fits the LDS copy.  A template parameter, not a run-time flag: `dl ? ldraft[i] : draft[i]` makes the
// compiler select between an LDS and a global pointer and emit flat loads, whose vmcnt wait drains every outstanding
// direction-byte store -- once per row.
template <bool DL>
__device__ long long extend_align(const PrepArgs& a, const uint32_t* pk, const uint8_t* draft, const uint8_t* ldraft, int C,
                                  int pb, int n, int dir_, int32_t* tpos, uint8_t* D, int lane) {
  const int W = a.p.dang_band, bw = 2 * W + 1;
  const int mt4 = a.p.pol_match * 4, mm4 = a.p.pol_mismatch * 4, g4 = a.p.pol_gap * 4;
  const int db = dir_ > 0 ? 0 : C - 1;
  const int NEGK = -(1 << 28);
  for (int k = lane; k < n; k += 64) tpos[pb + dir_ * k] = -1;
  if (bw > 64 * EC || ((long long)(n + 2) / 3 + 1) * 256 > a.ecap) return -1;
  unsigned* const DW = (unsigned*)D;
  unsigned dacc = 0;
  int hprev[EC];
#pragma unroll
  for (int cc = 0; cc < EC; ++cc) {           // row 0: H[0][j] = j*g for 0 <= j <= min(C, W)
    const int bb = lane * EC + cc, j = bb - W;
    hprev[cc] = (bb < bw && j >= 0 && j <= C) ? j * g4 : NEGK;
  }
  int best = 0, bi = 0, bj = 0;
  long long cells = lane == 0 ? min(C, W) + 1 : 0;          // row 0
#ifndef C3_EXT_OLD
  if (DL) {
    // BRANCH-FREE ROWS (the draft is in LDS).  An invalid cell -- left of column 0, right of column C, beyond the band -- always
    // holds NEGK, so every candidate that comes from one loses by itself: no masks on the diagonal / up / left candidates, ONE
    // select per cell (valid ? key : NEGK) before the key is split into score and tag.  The draft codes of the lane's five
    // columns travel in one register (a nibble each, shifted by one column per row; the new nibble is the row's only LDS read),
    // the first maximum is tracked per band offset (compare, max, select), the cell count is closed-form per row on the scalar
    // unit.  (The masked version compiled into two EXEC-mask branches per cell: ~450 instructions per row; this is ~140.)
    const int bb0 = lane * EC;
    int g4bb1[EC], bestc[EC], bic[EC];
#pragma unroll
    for (int cc = 0; cc < EC; ++cc) { g4bb1[cc] = g4 * (bb0 + cc) + 1; bestc[cc] = 0; bic[cc] = 0; }
    auto code_at = [&](int q) { return (unsigned)ldraft[db + dir_ * min(max(q, 0), C - 1)]; };
    unsigned dcp = 0;                                                      // nibble cc = draft code of column j - 1 = i - W + bb - 1
#pragma unroll
    for (int cc = 0; cc < EC; ++cc) dcp |= code_at(1 - W + bb0 + cc - 1) << (4 * cc);
    long long cells_s = 0;
    int ph3 = 0;                                                           // (i - 1) % 3
    for (int ib = 1; ib <= n; ib += 64) {
      int pcs = 0;
      if (ib + lane <= n) pcs = c3_code_at(pk, pb + dir_ * (ib + lane - 1));
      asm volatile("" : "+v"(pcs));
      const int cnt = min(64, n - ib + 1);
      for (int li = 0; li < cnt; ++li) {
        const int i = ib + li;
        const int pc = __builtin_amdgcn_readlane(pcs, li);
        const unsigned nxt_code = code_at(i - W + bb0 + EC - 1);          // nibble EC-1 of row i + 1 (consumed at the end of the row)
        const int vlo = W - i;                                             // valid offsets: max(0, vlo) .. vhi
        const int vhi = min(bw - 1, C - i + W);
        const unsigned span = (unsigned)(vhi - vlo);                      // (vhi >= vlo whenever a valid cell exists; else every compare below fails)
        cells_s += max(0, vhi - max(0, vlo) + 1);
        const int nxt0 = __builtin_amdgcn_update_dpp(NEGK, hprev[0], 0x130, 0xf, 0xf, false);   // wave_shl:1
        int key[EC], y[EC];
        bool valm[EC];
        int run = NEGK;
#pragma unroll
        for (int cc = 0; cc < EC; ++cc) {
          const int dcode = (int)((dcp >> (4 * cc)) & 15u);
          const int kd = hprev[cc] + ((pc == dcode) ? mt4 + 3 : mm4 + 3);
          const int up = (cc + 1 < EC) ? hprev[cc + 1 < EC ? cc + 1 : cc] : nxt0;
          const int k = max(kd, up + (g4 + 2));
          const bool val = (unsigned)(bb0 + cc - vlo) <= span && vhi >= vlo && bb0 + cc >= 0;
          valm[cc] = val;
          key[cc] = val ? k : NEGK;
          y[cc] = (key[cc] & ~3) - (g4bb1[cc] - 1);
          run = max(run, y[cc]);
        }
        const int s = wave_scan_max(run);
        int ex = wave_shr1(s, NEGK);
        unsigned d0 = 0;
#pragma unroll
        for (int cc = 0; cc < EC; ++cc) {
          // (no test for the row's first column: everything to its left is NEGK, so the left candidate loses by itself;
          // an invalid cell is re-masked because ex can be a real score)
          int k2 = max(key[cc], ex + g4bb1[cc]);
          ex = max(ex, y[cc]);
          k2 = valm[cc] ? k2 : NEGK;                                        // (the first loop's mask again: comparing the key with NEGK cost a v_cmp per cell, 6 % of the kernel)
          const int hh = k2 & ~3;
          hprev[cc] = hh;
          const unsigned tag = (unsigned)k2 & 3u;                           // 3 diag, 2 up, 1 left, 0 invalid
          d0 |= tag << (2 * cc);
          const bool up_ = hh > bestc[cc];                                 // (column 0 and invalid cells are <= 0: never)
          bestc[cc] = max(bestc[cc], hh);
          bic[cc] = up_ ? i : bic[cc];
        }
        dacc = (dacc >> 10) | (d0 << 20);                                  // three rows per dword: the oldest row ends up in bits 0-9
        if (ph3 == 2 || i == n) {                                          // (uniform: a scalar branch)
          DW[(unsigned)(i - 1) / 3u * 64u + lane] = dacc >> (10 * (2 - ph3));
          dacc = 0;
        }
        ph3 = ph3 == 2 ? 0 : ph3 + 1;
        dcp = (dcp >> 4) | (nxt_code << (4 * (EC - 1)));
      }
    }
    // the lane's first maximum in row-major order: smallest row, then smallest offset
#pragma unroll
    for (int cc = 0; cc < EC; ++cc) {
      const bool better = bestc[cc] > best || (bestc[cc] == best && bestc[cc] > 0 && bic[cc] < bi);
      if (better) { best = bestc[cc]; bi = bic[cc]; bj = bic[cc] - W + bb0 + cc; }
    }
    cells += lane == 0 ? cells_s : 0;
  } else
#endif
  for (int ib = 1; ib <= n; ib += 64) {
    int pcs = 0;                              // piece base of row ib+lane
    if (ib + lane <= n) pcs = c3_code_at(pk, pb + dir_ * (ib + lane - 1));
    asm volatile("" : "+v"(pcs));
    const int cnt = min(64, n - ib + 1);
    for (int li = 0; li < cnt; ++li) {
      const int i = ib + li;
      const int pc = __builtin_amdgcn_readlane(pcs, li);
      const int jlo = max(0, i - W);
      // up neighbour of the last owned offset = first offset of the next lane
      const int nxt0 = __builtin_amdgcn_update_dpp(NEGK, hprev[0], 0x130, 0xf, 0xf, false);   // wave_shl:1
      int key[EC], y[EC];
      int run = NEGK;
#pragma unroll
      for (int cc = 0; cc < EC; ++cc) {
        const int bb = lane * EC + cc, j = i - W + bb;
        const bool val = bb < bw && j >= 0 && j <= C;
        int k = INT32_MIN;
        if (val) {
          if (j > 0) {
            const int dcode = DL ? (int)ldraft[db + dir_ * (j - 1)] : (int)draft[db + dir_ * (j - 1)];
            k = hprev[cc] + ((pc == dcode) ? mt4 : mm4) + 3;
          }
          const int up = (cc + 1 < EC) ? hprev[cc + 1 < EC ? cc + 1 : cc] : nxt0;
          if (bb + 1 < bw) k = max(k, up + g4 + 2);
        }
        key[cc] = val ? k : NEGK;
        y[cc] = val ? (k & ~3) - g4 * bb : NEGK;
        run = max(run, y[cc]);
      }
      const int s = wave_scan_max(run);
      int ex = wave_shr1(s, NEGK);
      unsigned d0 = 0;
#pragma unroll
      for (int cc = 0; cc < EC; ++cc) {
        const int bb = lane * EC + cc, j = i - W + bb;
        const bool val = bb < bw && j >= 0 && j <= C;
        int k2 = key[cc];
        if (val && j > jlo) k2 = max(k2, ex + g4 * bb + 1);
        ex = max(ex, y[cc]);
        const int hh = k2 & ~3;
        hprev[cc] = val ? hh : NEGK;
        const unsigned tag = val ? (unsigned)(k2 & 3) : 0u;            // 3 diag, 2 up, 1 left, 0 none
        d0 |= tag << (2 * cc);
        if (val) { ++cells; if (j > 0 && hh > best) { best = hh; bi = i; bj = j; } }
      }
      const int ph = (i - 1) % 3;
      dacc = (dacc >> 10) | (d0 << 20);
      if (ph == 2 || i == n) { DW[(unsigned)(i - 1) / 3u * 64u + lane] = dacc >> (10 * (2 - ph)); dacc = 0; }
    }
  }
  WSYNC();
  // first maximum in row-major order
  const int gb = wave_max(best);
  const int gi = wave_min(best == gb ? bi : INT32_MAX / 2);
  const int gj = wave_min((best == gb && bi == gi) ? bj : INT32_MAX / 2);
  if (gb > 0) {
    // traceback; diagonal runs keep the band offset, so 64 steps are checked per round
    int i = gi, j = gj;
    while (i > 0 || j > 0) {
      if (i == 0) break;                                   // row 0: only left moves, nothing to record
      const int bb = j - i + W;
      const int ik = i - lane;
      int d = 3;                                                   // 0 diag, 1 up, 2 left, 3 none
      if (ik >= 1) d = 3 - (int)((DW[(unsigned)(ik - 1) / 3u * 64u + bb / EC] >> (10 * ((ik - 1) % 3) + 2 * (bb % EC))) & 3u);
      const unsigned long long bal = __ballot(ik >= 1 && d == 0 && j - lane >= 1);
      const int m = (~bal) ? __builtin_ctzll(~bal) : 64;
      if (lane < m) tpos[pb + dir_ * (ik - 1)] = db + dir_ * (j - lane - 1);
      i -= m; j -= m;
      if (m < 64 && i > 0) {
        const int dbk = wave_bcast(d, m);
        if (dbk == 1) --i; else if (dbk == 2) --j; else break;       // 0 here would mean j == 0: cannot happen
      }
    }
  }
  WSYNC();
  long long tot = cells;
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  return tot;
}

// one read of k_prep.  Kept as a function with early returns: a `continue` straight after the work-queue
// pop made hipcc re-enter the persistent loop with a partial EXEC mask (lane 0 missing), so the
// readfirstlane of the queue index never advanced -- the kernel hung on the first skipped read.
__device__ __forceinline__ void prep_one(const PrepArgs& a, int wi, int lane, uint8_t* ldraft, uint8_t* eD, int* lwf, int* lwl, int WL) {
    const int rid = a.work[wi];
    C3Info* info = &a.info[rid];
    // skip paths store from every lane (same value): no divergent branch right before the early return
    if (info->status != C3_ST_OK || info->draft_len <= 0) { info->n_win = 0; a.win_base[rid] = 0; return; }
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
    if ((ht || hf) && C <= EXT_DCAP) { for (int t = lane; t < C; t += 64) ldraft[t] = draft[t]; WSYNC(); }
#ifdef C3_EXP_X2_EXT
    if (C <= EXT_DCAP) {
      if (ht) { long long r = extend_align<true>(a, pk, draft, ldraft, C, info->tail_beg, L - info->tail_beg, +1, tpos, eD, lane); if (r > 1LL << 60) cells += r; }
      if (hf) { long long r = extend_align<true>(a, pk, draft, ldraft, C, info->front_end - 1, info->front_end, -1, tpos, eD, lane); if (r > 1LL << 60) cells += r; }
    }
#endif
    if (C <= EXT_DCAP) {
      if (ht) { long long r = extend_align<true>(a, pk, draft, ldraft, C, info->tail_beg, L - info->tail_beg, +1, tpos, eD, lane); if (r > 0) cells += r; }
      if (hf) { long long r = extend_align<true>(a, pk, draft, ldraft, C, info->front_end - 1, info->front_end, -1, tpos, eD, lane); if (r > 0) cells += r; }
    } else {
      if (ht) { long long r = extend_align<false>(a, pk, draft, ldraft, C, info->tail_beg, L - info->tail_beg, +1, tpos, eD, lane); if (r > 0) cells += r; }
      if (hf) { long long r = extend_align<false>(a, pk, draft, ldraft, C, info->front_end - 1, info->front_end, -1, tpos, eD, lane); if (r > 0) cells += r; }
    }
    // ---- layers: kept subreads, front, tail
    const int nl = ns + hf + ht;
    const int nwin = (C + WL - 1) / WL;
    if (nwin > a.NWcap || nl > a.NLcap) { info->status = C3_ST_LIMIT; info->n_win = 0; a.win_base[rid] = 0; return; }
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
    if (wbase + nwin > a.wcap) {
      // the reservation cannot be undone: the slots below wcap become empty windows (no layers, no backbone), so k_window
      // and k_stitch never see a record nobody wrote
      for (int w = lane; w < nwin && wbase + w < a.wcap; w += 64) {
        WinRec r; r.rid = rid; r.w = w; r.n_layers = 0; r.blen = 0; r.tgs = 0; r.out_len = 0; r.polished = 0; r.pad_ = 0;
        a.wrec[wbase + w] = r;
      }
      info->status = C3_ST_LIMIT; info->n_win = 0; a.win_base[rid] = 0; return;
    }
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
      // Only the first / last base of every run of equal window numbers inside a 64-base chunk touches memory: 64
      // lanes hammering one address with atomicMin/atomicMax (two L2 atomics per base) was the bulk of this kernel.
      int qf = INT32_MAX, ql = -1;
      for (int k0 = lb; k0 < le; k0 += 64) {
        const int k = k0 + lane;
        const int t = k < le ? tpos[k] : -1;
        const bool valid = t >= 0;
        const int w = valid ? t / WL : -1;
        const unsigned long long vm = __ballot(valid);
        const unsigned long long below = vm & ((1ull << lane) - 1ull);
        const unsigned long long above = lane == 63 ? 0ull : (vm & ~((2ull << lane) - 1ull));
        const int pl = below ? 63 - __builtin_clzll(below) : lane, nx = above ? __builtin_ctzll(above) : lane;
        const int wp = __shfl(w, pl), wn = __shfl(w, nx);
        if (valid) {
          qf = min(qf, k); ql = max(ql, k);
          if (!below || wp != w) atomicMin(&lwf[i * nwin + w], k);
          if (!above || wn != w) atomicMax(&lwl[i * nwin + w], k);
        }
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
    // what k_window's full-size launch must be able to hold: the longest layer of any window (DP columns) and the most nodes a
    // window graph can reach (every base of every layer a new node -- the bound the reference's containers grow to by themselves)
    int mq = 0, mn = 0;
    for (int w = lane; w < nwin; w += 64) {
      const WinRec* r = &a.wrec[wbase + w];
      int sum = r->blen;
      for (int x = 0; x < r->n_layers; ++x) { const int len = a.wlay[(size_t)(wbase + w) * a.NLcap + x].len; sum += len; mq = max(mq, len); }
      mn = max(mn, sum);
    }
    mq = wave_max(mq); mn = wave_max(mn);
    if (lane == 0) {
      info->n_win = nwin; a.win_base[rid] = wbase;
      atomicAdd((unsigned long long*)(a.counter + 2), (unsigned long long)cells);
      atomicMax(a.counter + 10, mn); atomicMax(a.counter + 11, mq);
    }
    WSYNC();
}

__global__ __launch_bounds__(64, C3_PREP_WAVES) void k_prep(PrepArgs a) {
  const int lane = wave_lane();
  const int slot = blockIdx.x;
  __shared__ uint8_t ldraft[EXT_DCAP];
  uint8_t* eD = a.eD + (size_t)slot * a.ecap;
  int* lwf = a.lw_first + (size_t)slot * a.NLcap * a.NWcap;
  int* lwl = a.lw_last + (size_t)slot * a.NLcap * a.NWcap;
  const int WL = a.p.pol_window;
  for (;;) {
    int wi = 0;
    if (lane == 0) wi = atomicAdd(a.counter, 1);
    wi = wave_first(wi);
    if (wi >= a.n_work) break;
    prep_one(a, wi, lane, ldraft, eD, lwf, lwl, WL);
  }
}

// ------------------------------------------------------------------------------------------

// Per-slot scratch of k_window.  Only the base pointers are kept live: an array accessor returns (base, element offset) and
// its operator[] forms the address as base + 32-bit byte offset, i.e. a scalar base with a vector offset (`global_load v, v,
// s[..]`).  Handing out base + k*Ncap as POINTERS made the compiler keep some twenty-five derived 64-bit pointers alive:
// it spilled ~190 SGPRs into vector lanes and re-read them with v_readlane all over the graph phases.
template <class T> struct WArr {
  T* base; unsigned off;
  __device__ __forceinline__ T& at(unsigned i) const { return *(T*)((char*)base + ((i + off) * (unsigned)sizeof(T))); }
  __device__ __forceinline__ T& operator[](int i) const { return at((unsigned)i); }
  __device__ __forceinline__ T& operator[](unsigned i) const { return at(i); }
  __device__ __forceinline__ T& operator[](long long i) const { return at((unsigned)i); }
  __device__ __forceinline__ T& operator[](size_t i) const { return at((unsigned)i); }
  __device__ __forceinline__ T* ptr() const { return base + off; }
};
// rq / tq of one layer (DP row and node of every query base): in LDS when the layer has at most W_QCAP bases (every layer of a
// 500-base window in practice), in the slot's global arrays otherwise.  The traceback's stores and the fusion's loads of these
// two small arrays were a sixth of the graph phases' vector memory instructions.
#define W_QCAP 704
#define W_TBW 448            /* dwords of traceback windows in front of rq / tq */
struct QArr {
  unsigned short* l; WArr<int> g; bool big;
  __device__ __forceinline__ int get(int i) const { return big ? g[i] : (int)l[i]; }
  __device__ __forceinline__ void set(int i, int v) const { if (big) g[i] = v; else l[i] = (unsigned short)v; }
};
struct WCtx {
  int* I; int* E; uint8_t* B8; long long* score; int32_t* H; uint8_t* D; uint4* rdesc;
  int K, n, Ncap; long long hcap;
  int osel = 0;                  // which of the two order buffers is current (w_reorder writes the other one and flips)
#define W_I(name, k) __device__ __forceinline__ WArr<int> name() const { return {I, (unsigned)(k) * (unsigned)Ncap}; }
  W_I(n_in, 0) W_I(n_out, 1) W_I(grp, 2) W_I(index, 5) W_I(gfirst, 6) W_I(glast, 7) W_I(ncov, 8)
  W_I(rowof, 9) W_I(anchor, 10) W_I(pred, 11) W_I(hend, 11) /* shares pred (disjoint lifetimes) */ W_I(opn, 12) /* 2N */ W_I(opq, 14) /* 2N */
  W_I(rows, 16) /* N+1 */
  __device__ __forceinline__ WArr<int> order() const { return {I, (unsigned)(3 + osel) * (unsigned)Ncap}; }
  __device__ __forceinline__ WArr<int> order2() const { return {I, (unsigned)(4 - osel) * (unsigned)Ncap}; }
  __device__ __forceinline__ WArr<int> lob() const { return {I, 17u * (unsigned)Ncap + 8u}; }   /* N+1: band start | block index << 16 of every DP row */
#undef W_I
  __device__ __forceinline__ WArr<int> in_from() const { return {E, 0u}; }
  __device__ __forceinline__ WArr<int> in_w() const { return {E, (unsigned)Ncap * (unsigned)K}; }
  __device__ __forceinline__ WArr<int> out_to() const { return {E, 2u * (unsigned)Ncap * (unsigned)K}; }
  __device__ __forceinline__ WArr<int> out_w() const { return {E, 3u * (unsigned)Ncap * (unsigned)K}; }
  __device__ __forceinline__ WArr<uint8_t> base() const { return {B8, 0u}; }
  __device__ __forceinline__ WArr<uint8_t> mask() const { return {B8, (unsigned)Ncap}; }
};
#define W_INTS 19   // ints of Ncap per slot in WCtx::I (18*Ncap + 9 used)

// The graph phases are chains of dependent memory round trips (position -> node -> its group / edges ...), and a wave that
// waits for one has nothing else to do: every loop over the nodes therefore takes WU chunks of 64 per iteration, all loads of
// one level issued back to back, so a level costs ONE memory latency per WU*64 nodes instead of one per 64.
#ifndef WU
#define WU 4
#endif
__device__ void w_blocks(WCtx& c, int lane) {
  // The members of an aligned block are CONTIGUOUS in the topological order (a new sibling is merged right behind its
  // block), so a block's extent is a run of equal group ids: where the id changes, the new block starts and the previous one
  // ends -- plain stores, no initialisation pass, no atomics, no look-ahead.
  int gprev = -1;                                                       // group of position i0 - 1
  for (int i0 = 0; i0 < c.n; i0 += 64 * WU) {
    int v[WU], r[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) { const int i = i0 + 64 * u + lane; v[u] = i < c.n ? c.order()[i] : -1; }
#pragma unroll
    for (int u = 0; u < WU; ++u) r[u] = v[u] >= 0 ? c.grp()[v[u]] : -2;
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int i = i0 + 64 * u + lane;
      const int rp = wave_shr1(r[u], gprev);
      gprev = wave_bcast(r[u], 63);
      if (v[u] >= 0 && r[u] != rp) { c.gfirst()[r[u]] = i; if (i > 0) c.glast()[rp] = i - 1; }
      if (v[u] >= 0 && i + 1 == c.n) c.glast()[r[u]] = i;
    }
  }
  WSYNC();
}
__device__ void w_reorder(WCtx& c, int n_old, int lane, int* lds, int lds_cap) {
  const int n_new = c.n - n_old;
  // old node at old index i moves to i + #(anchor < i); new node k goes to anchor[k] + 1 + k.  The (sorted) anchors of the new
  // nodes are staged in LDS first: the binary search per old node is then 6-8 LDS reads instead of 6-8 dependent global loads
  // per 64 nodes (the LDS scratch of the alignment is idle during the graph phases).  The new order is written to the OTHER
  // order buffer together with index[], and the buffers swap roles: no copy-back pass.
  const bool inl = n_new <= lds_cap;
  if (inl) { for (int k = lane; k < n_new; k += 64) lds[k] = c.anchor()[k]; WSYNC(); }
  for (int i0 = 0; i0 < n_old; i0 += 64 * WU) {
    int v[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) { const int i = i0 + 64 * u + lane; v[u] = i < n_old ? c.order()[i] : -1; }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
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
  w_blocks(c, lane);
}

// Row descriptors of one alignment, built in parallel before the DP so that the row loop has no
// dependent graph loads:
//   x = base | np<<8 | overflow<<16 | needH<<17 | isend<<18 | twobit<<19,  y = p0 | p1<<16,  z = p2 | p3<<16
// p = DP row of a masked predecessor in in-edge order (row 0 = the virtual start row);
// needH: some masked successor is neither of the next two rows, so the H row must be kept in memory;
// isend: no masked successor, the row is a candidate end of the global alignment.
// m2[i] bit b: DP row 1+64i+b has exactly one masked predecessor (its direction cells are stored with 2 bits);
// ma[i] bit b: ... and that predecessor is the previous row.  Both live in LDS so the traceback can classify the 64 rows
// of a round without a dependent global load (allow2 = false: every row uses byte cells, e.g. the linear fallback).
__device__ void win_build_desc(WCtx& c, int R, int lane, unsigned long long* m2, unsigned long long* ma, bool allow2) {
  const int K = c.K;
  const WArr<int> kept = c.opn();               // kept[r] = number of kept (needH) rows before row r (free between two tracebacks)
  int nkept = 0;
  for (int r0 = 1; r0 <= R; r0 += 64) {
    const int r = r0 + lane;
    bool two = false, adj = false;
    unsigned needh = 0;
    if (r <= R) {
      const int v = c.rows()[r];
      const int nin = c.n_in()[v];
      unsigned p[4] = {0, 0, 0, 0};
      int np = 0;
      for (int k = 0; k < nin; ++k) {
        const int pr = c.rowof()[c.in_from()[EI(v, k)]];
        if (pr < 0) continue;
        if (np < 4) p[np] = (unsigned)pr;
        ++np;
      }
      unsigned has = 0;
      for (int k = 0; k < c.n_out()[v]; ++k) {
        const int sr = c.rowof()[c.out_to()[EI(v, k)]];
        if (sr >= 0) { has = 1; if (sr != r + 1 && sr != r + 2) needh = 1; }     // rows r-1 and r-2 stay in registers
      }
      unsigned ovf = np > 4;
      if (np == 0) np = 1;                      // no masked predecessor: virtual row 0
      two = allow2 && np == 1;
      adj = two && (int)p[0] == r - 1;
      // bit 20: the row loop's fast row -- one predecessor, the row right above, nothing to keep for later (no far successor, not an end row)
      const unsigned fast = adj && !needh && has;
      uint4 d; d.x = (unsigned)c.base()[v] | ((unsigned)min(np, 255) << 8) | (ovf << 16) | (needh << 17) | ((has ^ 1u) << 18) | ((unsigned)two << 19) | (fast << 20);
      d.y = p[0] | (p[1] << 16); d.z = p[2] | (p[3] << 16); d.w = 0;
      c.rdesc[r] = d;
      c.hend()[r] = INT32_MIN;
    }
    const unsigned long long b2 = __ballot(two), ba = __ballot(adj), bk = __ballot(needh != 0);
    if (r <= R) kept[r] = nkept + __popcll(bk & ((1ull << lane) - 1));
    nkept += __popcll(bk);
    if (lane == 0) { m2[r0 >> 6] = b2; ma[r0 >> 6] = ba; }
  }
  WSYNC();
  // bit 21 of a kept row: some far successor will NOT find it in the LDS ring any more, so its H row must also go to global
  // memory.  The ring holds the last four kept rows in the order they were written: row p is still there at row s iff at most
  // three kept rows lie between them.  Most kept rows are consumed within a few rows and never leave the CU.
  if (C3_WIN_RING) {
    for (int r = 1 + lane; r <= R; r += 64) {
      const unsigned dx = ((const unsigned*)(c.rdesc + r))[0];
      if (!((dx >> 17) & 1)) continue;
      const int v = c.rows()[r], kr = kept[r];
      unsigned miss = 0;
      for (int k = 0; k < c.n_out()[v]; ++k) {
        const int sr = c.rowof()[c.out_to()[EI(v, k)]];
        if (sr >= 0 && sr != r + 1 && sr != r + 2 && kept[sr] - kr - 1 > 3) miss = 1;
      }
      if (miss) ((unsigned*)(c.rdesc + r))[0] = dx | (1u << 21);
    }
    WSYNC();
  }
}

// predecessor row number t of DP row r (descriptor order); rows with more than 4 masked
// predecessors walk the in-edge list
__device__ int win_pred_row(const WCtx& c, const uint4& de, int r, int t) {
  // (branch-free: as a chain of ?: the compiler built four nested EXEC-mask branches into every traceback step)
  if (!((de.x >> 16) & 1)) { const unsigned w = (t & 2) ? de.z : de.y; return (int)((w >> ((t & 1) << 4)) & 0xffffu); }
  const int v = c.rows()[r];
  int seen = 0;
  for (int k = 0; k < c.n_in()[v]; ++k) { int pr = c.rowof()[c.in_from()[EI(v, k)]]; if (pr >= 0) { if (seen == t) return pr; ++seen; } }
  return 0;
}

// All DP rows of one layer.  Lane owns columns lane*CPL .. lane*CPL+CPL-1 (element (lane,cc) of an
// H row lives at cc*64+lane), so every lane only ever re-reads cells it stored itself: no barrier in
// the row loop.  The row just computed stays in registers and feeds the next row directly (the
// common predecessor); H rows go to memory only when a non-adjacent successor will need them.
//
// Cells are 16-bit KEYS = score*4 + type (3 diagonal, 2 vertical, 1 horizontal) in the low half of a register; the high
// half is "don't care".  On gfx950 the 16-bit VOP2 v_max_i16 and 32-bit add / sub / and issue at twice the rate of
// v_max_i32, v_max3, compares, selects, DPP and every VALU instruction with an SGPR operand (tools/ubench/valu_cost.hip),
// so the row is written in exactly those: one v_max_i16 per candidate implements "highest score, diagonal before vertical
// before horizontal"; among several predecessors the FIRST one that reaches the maximum keeps the cell (strict >), which
// only rows with more than one predecessor have to track.  The D byte of a cell is tag = 255 - p, p = 0..63 diagonal from
// predecessor p, 64..127 vertical from predecessor p-64, 128..191 horizontal; single-predecessor rows store the 2-bit type.
// (Columns past Q compute garbage that only ever flows to the right: never read.)
#define W_NEG16 (-32000)
#define W_TAG_H 127           /* byte rows of the linear fallback: 255 - 128 */
#define VREG(x) asm volatile("" : "+v"(x))       /* keep a uniform value in a vector register (no instruction) */
__device__ __forceinline__ int win_d_type(int tag) { return (255 - tag) >> 6; }       // 0 diag 1 vert 2 horiz
__device__ __forceinline__ int win_d_pred(int tag) { return (255 - tag) & 63; }

// Arguments of a real (not inlined) call arrive in vector registers: make the uniform ones scalar again.  Pointers arrive
// generic: GP() names the global address space again (a flat load could be private memory, so the compiler must treat its
// result as divergent and the access as FLAT).
#define GP(T, p) ((__attribute__((address_space(1))) T*)(p))
__device__ __forceinline__ int uni32(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <class T> __device__ __forceinline__ T* uni_ptr(T* p) {
  const unsigned long long v = (unsigned long long)p;
  return (T*)(((unsigned long long)(unsigned)uni32((int)(v >> 32)) << 32) | (unsigned)uni32((int)v));
}

// NOT inlined on purpose: k_window carries some thirty vector registers of per-window state across the layer loop, and
// inlined here the register allocator spilled the row loop's own arrays (eight scratch reloads per row, each waiting for
// every older store).  As a function the rows get a register file of their own; the call costs a few dozen instructions
// per LAYER.
template <int CPL, bool SECOND>
// (SECOND: the two launches of k_window have different register budgets -- 80 VGPRs for the first, 96 for the full-size second -- and
// a function shared by both would be compiled for the smaller occupancy: every launch gets row functions of its own)
// (the context travels as scalars: a struct passed by value gets an 80-byte stack slot PER CALL SITE, and the kernel's scratch
// segment -- sized for every wave slot of the device -- decides how long the process's first launch waits for the runtime)
__device__ __attribute__((noinline)) int win_rows(int* cI, int* cE, int32_t* cH, uint8_t* cD, uint4* crdesc, int cK, int cn, int cNcap, long long chcap,
                                                  int mt_, int mm_, int g_, const uint32_t* pk_, int qbeg_, int Q_, int R_, unsigned long long* dbg_, int ring_off_) {
  WCtx c;
  c.I = uni_ptr(cI); c.E = uni_ptr(cE); c.B8 = nullptr; c.score = nullptr; c.H = uni_ptr(cH); c.D = uni_ptr(cD);
  c.rdesc = uni_ptr(crdesc); c.K = uni32(cK); c.n = uni32(cn); c.Ncap = uni32(cNcap);
  c.hcap = ((long long)uni32((int)(chcap >> 32)) << 32) | (unsigned)uni32((int)chcap);
  const uint32_t* pk = uni_ptr(pk_);
  unsigned long long* dbg = uni_ptr(dbg_);
  const int qbeg = uni32(qbeg_), Q = uni32(Q_), R = uni32(R_);
  const int lane = wave_lane();
  extern __shared__ int lds_dyn[];
  unsigned* ring = (unsigned*)lds_dyn + uni32(ring_off_);
  struct { int pol_match, pol_mismatch, pol_gap; } P = {uni32(mt_), uni32(mm_), uni32(g_)};
  const int RS = 64 * CPL, K = c.K;
  constexpr int DS = (CPL + 3) & ~3, RSD = 64 * DS;                        // D row: natural column order, DS bytes per lane
  if ((long long)(R + 1) * RSD > c.hcap || R >= 65535) return -1;
  const int mt3 = P.pol_match * 4 + 3, mm3 = P.pol_mismatch * 4 + 3, g4 = P.pol_gap * 4;
  const int mm3x4 = (mm3 & 255) * 0x01010101;
  int tbl[CPL], hcur[CPL], hp2[CPL], g41[CPL];                              // hcur = row r-1, hp2 = row r-2 (score * 4)
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc) {
    const int j = lane * CPL + cc;
    const int qc = (j >= 1 && j <= Q) ? c3_code_at(pk, qbeg + j - 1) : 7;
    // byte b of tbl: 4 * substitution score + 3 (the diagonal type) of this column against graph base b
    tbl[cc] = qc < 4 ? (mm3x4 & ~(255 << (8 * qc))) | ((mt3 & 255) << (8 * qc)) : mm3x4;
    hp2[cc] = 0;
    hcur[cc] = j * g4;                                                      // virtual row 0 (never stored: a row that needs it again takes g41 - 1)
    g41[cc] = j * g4 + 1;                                                   // horizontal candidate = 4 * (best y + g * j) + 1
  }
  int cV = g4 + 2;                                                          // vertical candidate = H4[pred][j] + 4 * g + 2
  VREG(cV);
  // gfx9 has ONE in-order vmcnt for loads and stores: consuming any load waits for every older
  // store.  So the row loop carries no vector loads on its common path -- descriptors come 64 rows at
  // a time (one per lane) and are broadcast with v_readlane; only rows with a non-adjacent
  // predecessor touch memory.
  // LDS ring of the last four rows that a far successor will need (16-bit scores, two columns per dword): a far
  // predecessor is usually a handful of rows back, and an LDS hit spares the row loop a global load -- which on gfx9
  // waits for every older direction / H store (one in-order vmcnt)
  int rt0 = -1, rt1 = -1, rt2 = -1, rt3 = -1, rnext = 0;
  constexpr int RW = (CPL + 1) / 2 * 64;                                    // dwords per ring slot
  unsigned doff = (CPL <= 8 ? 2 : 4) * lane;                                 // byte offset of this lane's 2-bit word in D row r (kept in a vector register: + RSD per row)
  for (int rb = 1; rb <= R; rb += 64) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 dv = GP(const u32x4, c.rdesc)[min(rb + lane, R)];
  uint4 dblk = make_uint4(dv.x, dv.y, dv.z, dv.w);
  // pin the wait for this load HERE (the asm "uses" the registers), not inside the row loop
  asm volatile("" : "+v"(dblk.x), "+v"(dblk.y), "+v"(dblk.z));
  const int cnt = min(64, R - rb + 1);
  for (int li = 0; li < cnt; ++li) {
    const int r = rb + li;
    uint4 de;
    de.x = __builtin_amdgcn_readlane(dblk.x, li);
    doff += RSD;
    if (de.x & (1u << 20)) {
      // FAST ROW (most rows): one predecessor = the row above, held in hcur; nothing stored but the 2-bit directions.  One
      // scalar test, no further descriptor words, no scalar address arithmetic: the scalar unit is shared by the CU's
      // four SIMDs, and a row of the general path below spends some seventy scalar instructions and a dozen branches
      int vb8 = (de.x & 3) * 8;
      VREG(vb8);
#ifdef C3_PHASE_PROF
      dbg[1]++;
#endif
      int key[CPL];
      const int hleft = wave_shr1(hcur[CPL - 1], W_NEG16);
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int hd = cc == 0 ? hleft : hcur[cc - 1];
        key[cc] = max16(hd + __builtin_amdgcn_sbfe(tbl[cc], vb8, 8), hcur[cc] + cV);
      }
      int run = W_NEG16;
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) run = max16(run, key[cc] - g41[cc]);
      int ex = wave_shr1(wave_scan_max(__builtin_amdgcn_sbfe(run, 0, 16)), W_NEG16);
      unsigned w2 = 0;
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int k2 = max16(key[cc], (ex & ~3) + g41[cc]);
        ex = max16(ex, key[cc] - g41[cc]);
        hp2[cc] = hcur[cc];
        hcur[cc] = k2 & ~3;
        w2 |= ((unsigned)k2 & 3u) << (2 * cc);
      }
      if (CPL <= 8) *GP(unsigned short, c.D + doff) = (unsigned short)w2;
      else *GP(unsigned, c.D + doff) = w2;
      continue;
    }
#ifdef C3_PHASE_PROF
    const unsigned long long gen_t0 = __builtin_readcyclecounter();
#endif
    de.y = __builtin_amdgcn_readlane(dblk.y, li);
    de.z = __builtin_amdgcn_readlane(dblk.z, li); de.w = 0;
    const int vb = de.x & 0xff, np = (de.x >> 8) & 0xff;
    const bool ovf = (de.x >> 16) & 1, needh = (de.x >> 17) & 1, isend = (de.x >> 18) & 1, two = (de.x >> 19) & 1;
    if (np > 64) return -1;
    int key[CPL];                                                            // bits 16..21: the predecessor that set the cell (rows with several)
    int vb8 = (vb & 3) * 8;                                                  // (graph bases are 2-bit codes)
    VREG(vb8);
    int kedge = 0;
    for (int t = 0; t < np; ++t) {
      int prow;
      if (!ovf) prow = (t == 0) ? (de.y & 0xffff) : (t == 1) ? (de.y >> 16) : (t == 2) ? (de.z & 0xffff) : (de.z >> 16);
      else {                                                                 // >4 predecessors: walk the in-edges
        const int v = GP(const int, c.rows().ptr())[r];
        prow = -1;
        while (kedge < GP(const int, c.n_in().ptr())[v]) { int pr = GP(const int, c.rowof().ptr())[GP(const int, c.in_from().ptr())[EI(v, kedge)]]; ++kedge; if (pr >= 0) { prow = pr; break; } }
        if (prow < 0) break;
      }
#ifdef C3_PHASE_PROF
      // row census: [0] low = general rows with the single predecessor r-1 (kept for later), high = all general rows;
      // [1] low = all rows, high = rows with several predecessors
      if (t == 0) { dbg[0] += (1ull << 32) + (two && prow == r - 1); dbg[1] += 1 + ((unsigned long long)!two << 32); }
#endif
      int hp[CPL];
      if (prow == r - 1) {
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) hp[cc] = hcur[cc];
      } else if (prow == r - 2) {                                            // the usual "skip one sibling" edge
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) hp[cc] = hp2[cc];
      } else if (prow == 0) {                                                // the virtual start row: H[0][j] = j * gap
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) hp[cc] = g41[cc] - 1;
      } else {
        const int sl = !C3_WIN_RING ? -1 : prow == rt0 ? 0 : prow == rt1 ? 1 : prow == rt2 ? 2 : prow == rt3 ? 3 : -1;
        if (sl >= 0) {
          const unsigned* rp = ring + sl * RW;
#pragma unroll
          for (int m = 0; m < (CPL + 1) / 2; ++m) {
            const unsigned x = rp[m * 64 + lane];
            hp[2 * m] = (int)x;                                              // (the high half is never looked at)
            if (2 * m + 1 < CPL) hp[2 * m + 1] = (int)(x >> 16);
          }
        } else {
          const auto* hp_ = GP(const unsigned short, c.H) + (size_t)prow * RS;
#pragma unroll
          for (int cc = 0; cc < CPL; ++cc) hp[cc] = (int)hp_[cc * 64 + lane];
        }
      }
      const int hleft = wave_shr1(hp[CPL - 1], W_NEG16);                     // column lane*CPL-1 of the predecessor
      if (t == 0) {
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
          const int hd = cc == 0 ? hleft : hp[cc - 1];
          key[cc] = max16(hd + __builtin_amdgcn_sbfe(tbl[cc], vb8, 8), hp[cc] + cV);
        }
      } else {
        const int tmark = t << 16;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
          const int hd = cc == 0 ? hleft : hp[cc - 1];
          const int nk = max16(key[cc], max16(hd + __builtin_amdgcn_sbfe(tbl[cc], vb8, 8), hp[cc] + cV));
          key[cc] = nk != (key[cc] & 0xffff) ? (nk | tmark) : key[cc];      // strictly better only: the first predecessor keeps a tie
        }
      }
    }
    // horizontal gap: in-lane prefix + one cross-lane max-scan over y = H - g*j (carried as key - g41: the low bits stay
    // below 4, so the order of the y is the order of the keys)
    int run = W_NEG16;
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) run = max16(run, key[cc] - g41[cc]);
    const int s = wave_scan_max(__builtin_amdgcn_sbfe(run, 0, 16));
    int ex = wave_shr1(s, W_NEG16);                                          // max over all previous lanes
    if (two) {
      // single-predecessor row: 2 bits per cell (3 diag, 2 vert, 1 horiz), one word per lane, 128 or 256 contiguous bytes
      // per row instead of 64*DS
      unsigned w2 = 0;
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int k2 = max16(key[cc], (ex & ~3) + g41[cc]);                  // for j == 0 ex is W_NEG16: never wins
        ex = max16(ex, key[cc] - g41[cc]);
        hp2[cc] = hcur[cc];
        hcur[cc] = k2 & ~3;
        w2 |= ((unsigned)k2 & 3u) << (2 * cc);
      }
      if (CPL <= 8) GP(unsigned short, c.D + (size_t)r * RSD)[lane] = (unsigned short)w2;         // 16 bits hold 8 cells: 128 bytes per row
      else GP(unsigned, c.D + (size_t)r * RSD)[lane] = w2;
    } else {
      unsigned dpk[DS / 4];
#pragma unroll
      for (int w = 0; w < DS / 4; ++w) dpk[w] = 0;
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) {
        const int k2 = max16(key[cc], (ex & ~3) + g41[cc]);
        ex = max16(ex, key[cc] - g41[cc]);
        hp2[cc] = hcur[cc];
        hcur[cc] = k2 & ~3;
        const unsigned tag = (((unsigned)k2 & 3u) << 6) + 63u - ((unsigned)key[cc] >> 16);      // horizontal: 127 - p, still type 2
        dpk[cc / 4] |= tag << (8 * (cc & 3));
      }
      auto* drow = GP(unsigned, c.D + (size_t)r * RSD) + lane * (DS / 4);
#pragma unroll
      for (int w = 0; w < DS / 4; ++w) drow[w] = dpk[w];
    }
    if (needh) {
      if (!C3_WIN_RING || ((de.x >> 21) & 1)) {                            // only rows that outlive their stay in the ring
        auto* hrow = GP(short, c.H) + (size_t)r * RS;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) hrow[cc * 64 + lane] = (short)hcur[cc];           // (columns past Q: inside the row, never read)
      }
      if (C3_WIN_RING) {
      const int sl = rnext;
      rnext = (rnext + 1) & 3;
      rt0 = sl == 0 ? r : rt0; rt1 = sl == 1 ? r : rt1; rt2 = sl == 2 ? r : rt2; rt3 = sl == 3 ? r : rt3;
      unsigned* wp = ring + sl * RW;
#pragma unroll
      for (int m = 0; m < (CPL + 1) / 2; ++m)
        wp[m * 64 + lane] = __builtin_amdgcn_perm((unsigned)(2 * m + 1 < CPL ? hcur[2 * m + 1] : 0), (unsigned)hcur[2 * m], 0x05040100u);
      }
    }
    if (isend) {
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) if (lane * CPL + cc == Q) GP(int, c.hend().ptr())[r] = __builtin_amdgcn_sbfe(hcur[cc], 2, 14);
    }
#ifdef C3_PHASE_PROF
    // cycles of the general rows by kind: [2] several predecessors, [3] one predecessor that is not r-1, [4] r-1 but kept; [5] all rows
    dbg[!two ? 2 : ((de.y & 0xffff) == (unsigned)(r - 1) ? 4 : 3)] += __builtin_readcyclecounter() - gen_t0;
#endif
  }
  }
  WSYNC();
  return 0;
}

// generic fallback for very long segments: linear layout (element j at j), 64 columns per step
__device__ int win_rows_lin(WCtx& c, const C3Params& P, const uint32_t* pk, int qbeg, int Q, int R, int lane) {
  const int RS = ((Q + 1 + 63) / 64) * 64, K = c.K;
  const int mt = P.pol_match, mm = P.pol_mismatch, g = P.pol_gap;
  if ((long long)(R + 1) * RS > c.hcap || R >= 65535) return -1;
  for (int j = lane; j <= Q; j += 64) { c.H[j] = j * g; c.D[j] = 2; }
  WSYNC();
  for (int r = 1; r <= R; ++r) {
    const int v = c.rows()[r];
    const int vb = c.base()[v];
    const int nin = c.n_in()[v];
    int32_t* hrow = c.H + (size_t)r * RS; uint8_t* drow = c.D + (size_t)r * RS;
    const bool isend = (c.rdesc[r].x >> 18) & 1;
    int carry = C3_NEG2;
    for (int c0 = 0; c0 <= Q; c0 += 64) {
      const int j = c0 + lane;
      const bool act = j <= Q;
      const int qc = (act && j >= 1) ? c3_code_at(pk, qbeg + j - 1) : 7;
      int bd = INT32_MIN, dd = 0, bv = INT32_MIN, dv = 0, np = 0;
      for (int k = 0; k <= nin; ++k) {
        int prow;
        if (k < nin) { prow = c.rowof()[c.in_from()[EI(v, k)]]; if (prow < 0) continue; ++np; }
        else { if (np > 0) break; prow = 0; }
        const int tt = min(np > 0 ? np - 1 : 0, 63);
        const int32_t* hp_ = c.H + (size_t)prow * RS;
        if (act) {
          if (j > 0) { int cnd = hp_[j - 1] + ((vb == qc) ? mt : mm); if (cnd > bd) { bd = cnd; dd = 255 - tt; } }
          int cv = hp_[j] + g;
          if (cv > bv) { bv = cv; dv = 191 - tt; }
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
      if (j > 0 && lf > hh) { hh = lf; dir = W_TAG_H; }
      if (act) { hrow[j] = hh; drow[j] = (uint8_t)dir; }
      if (isend && j == Q) c.hend()[r] = hh;
    }
    WSYNC();
  }
  return 0;
}

// dynamic LDS of k_window: the consensus sweep arrays (Lcap scores + 16-bit predecessors) share it with four row bitmasks
// (row kinds, band shifts) followed by either the H ring of the unbanded rows (the traceback windows reuse it) or the banded
// rows' substitution table (up to 640 columns), ring and edge cells (sized for the widest band)
#include "k_polish_band.h"
__host__ __device__ __forceinline__ int win_mask_words(int Ncap) { return ((Ncap + 64) >> 6) + 1; }
// `first`: the FIRST launch (k_window<false>).  The LDS allocator of gfx950 hands out granules of 1 280 bytes (160 KB / 128; measured
// in round 5: at 6 656 bytes = six granules only 21 waves fit a CU, at 6 400 = five granules 24 do, and k_window runs 8 % faster: profiles/
// r05_ab_window_6_waves_lds_granule.txt).  The first launch has no wide unbanded rows (their ring would need 5 KB), its consensus arrays are
// capped by the host (Lcap) and the banded rows' LDS by C3_WIN_LDS_FIRST: a layer whose substitution table + ring do not fit runs unbanded
// (checked per layer in win_rows_dispatch).  Graphs with many layers (large row bitmasks) simply take more LDS and fewer waves.
#ifndef C3_WIN_LDS_FIRST
#define C3_WIN_LDS_FIRST 6400
#endif
__host__ __device__ __forceinline__ size_t win_lds_bytes(int Lcap, int Ncap, bool first) {
  const size_t masks = (size_t)32 * win_mask_words(Ncap);
  const size_t a = (size_t)Lcap * 6 + 16, b = masks + 4 * (first ? 2 : 5) * 64 * 4 + 64, d = masks + 4 * (size_t)(W_TBW + W_QCAP);
  size_t c = masks + 4 * (size_t)wb_lds_dwords(639, 4);
  size_t m = a > b ? a : b; m = m > d ? m : d;
  if (first) { const size_t cap = m > (size_t)C3_WIN_LDS_FIRST ? m : (size_t)C3_WIN_LDS_FIRST; if (c > cap) c = cap; }
  m = m > c ? m : c;
  return m;
}

// byte index of column j inside a D row for the layout chosen by win_rows_dispatch
__device__ __forceinline__ int win_idx(int j, int cpl) { return cpl ? (j / cpl) * ((cpl + 3) & ~3) + j % cpl : j; }

// cb_io: in = 0: not banded, k >= 1: banded with at least k cells per lane (a retry after a failed certificate asks for a wider
// band); out = cells per lane of the band that ran (0 = the unbanded rows ran)
template <bool SECOND>
__device__ int win_rows_dispatch(WCtx& c, const C3Params& P, const uint32_t* pk, int qbeg, int Q, int R, int lane, int* cpl_out, int* rs_out, unsigned long long* dbg,
                                 unsigned long long* m2, unsigned long long* ma, unsigned long long* d0, unsigned long long* d1, int ring_off, int lds_ints, int begin, int end, int blen, int* cb_io, int* nblocks) {
  const int need = (Q + 1 + 63) / 64;
  int cpl;
  // 16-bit keys (score * 4 + type) of the register-blocked rows: |score| <= pm * max(R, Q), |score - g * j| <= (match + |g|) * Q,
  // and the substitution scores live in signed bytes
  const int pm = max(max(abs(P.pol_match), abs(P.pol_mismatch)), abs(P.pol_gap));
  const bool ok16 = !(4 * pm * (max(R, Q) + 4) >= 31000 || 4 * (abs(P.pol_match) + abs(P.pol_gap)) * (Q + 4) >= 31000 || pm > 30 || P.pol_gap >= 0);
  if (!ok16) cpl = 0;
  else if (need <= 2) cpl = 2; else if (need <= 4) cpl = 4; else if (need <= 6) cpl = 6; else if (need <= 8) cpl = 8;
  else if (need <= 10) cpl = 10; else cpl = 0;
  // band width by the inflation of the graph (rows per backbone position of the layer): the more alternative nodes, the
  // weaker the certificate's bounds and the wider the band it needs (tools/band_model.py)
  const int span = end - begin + 1;
  int cb = R * 4 < span * 5 ? 2 : R * 2 < span * 3 ? 3 : 4;             // (starting narrower -- 1.3 / 1.7 -- and relying on the retry was measured 1 % slower at cfg4)
  cb = max(cb, *cb_io);
  if (cb > 4) cb = 0;
  if (!*cb_io || !cb || !ok16 || need > 10 || need <= cb || span < 1 || max(P.pol_match, P.pol_mismatch) <= 0 || ring_off + wb_lds_dwords(Q, cb) > lds_ints) cb = 0;
#ifdef C3_BAND_OFF
  cb = 0;
#endif
#ifdef C3_PHASE_PROF
  const unsigned long long bd_t0 = __builtin_readcyclecounter();
#endif
  if (cb && !win_build_desc_band(c, R, Q, begin, end, blen, cb, lane, m2, ma, d0, d1, nblocks)) cb = 0;
#ifdef C3_EXP_X2_DESC
  if (cb && !win_build_desc_band(c, R, Q, begin, end, blen, cb, lane, m2, ma, d0, d1, nblocks)) cb = 0;
#endif
#ifdef C3_PHASE_PROF
  dbg[5] += __builtin_readcyclecounter() - bd_t0;
#endif
  *cb_io = cb;
  if (cb) {
    *cpl_out = cb; *rs_out = 256;
    int rc = 0;
#ifdef C3_EXP_X2_ROWS
    for (int rep_ = 0; rep_ < 2; ++rep_)
#endif
    switch (cb) {
      case 2: rc = win_rows_band<2, SECOND>(c.I, c.E, c.H, c.D, c.rdesc, c.K, c.n, c.Ncap, c.hcap, P.pol_match, P.pol_mismatch, P.pol_gap, pk, qbeg, Q, R, dbg, ring_off, *nblocks); break;
      case 3: rc = win_rows_band<3, SECOND>(c.I, c.E, c.H, c.D, c.rdesc, c.K, c.n, c.Ncap, c.hcap, P.pol_match, P.pol_mismatch, P.pol_gap, pk, qbeg, Q, R, dbg, ring_off, *nblocks); break;
      default: rc = win_rows_band<4, SECOND>(c.I, c.E, c.H, c.D, c.rdesc, c.K, c.n, c.Ncap, c.hcap, P.pol_match, P.pol_mismatch, P.pol_gap, pk, qbeg, Q, R, dbg, ring_off, *nblocks); break;
    }
    if (rc < 0) return rc;
    if (rc > 0) c.lob()[0] = INT32_MAX;        // a band that cannot hold a predecessor: the certificate fails by definition
    return 0;
  }
  *cpl_out = cpl; *rs_out = cpl ? 64 * ((cpl + 3) & ~3) : need * 64;    // D row stride in bytes
#ifdef C3_PHASE_PROF
  const unsigned long long bd_t1 = __builtin_readcyclecounter();
#endif
  win_build_desc(c, R, lane, m2, ma, cpl != 0);
#ifdef C3_PHASE_PROF
  dbg[5] += __builtin_readcyclecounter() - bd_t1;
#endif
  // the FIRST launch carries no unbanded rows wider than 256 columns and no linear fallback (its 80 registers and five LDS granules are
  // sized for the banded rows): such a layer -- a band that failed three certificates, a layer the band geometry refuses, scores beyond
  // 16-bit keys -- sends its window to the full-size launch, exactly as a layer beyond the first launch's DP scratch does
  if (!SECOND && cpl != 2 && cpl != 4) return -1;
  switch (cpl) {
    case 2: return win_rows<2, SECOND>(c.I, c.E, c.H, c.D, c.rdesc, c.K, c.n, c.Ncap, c.hcap, P.pol_match, P.pol_mismatch, P.pol_gap, pk, qbeg, Q, R, dbg, ring_off);
    case 4: return win_rows<4, SECOND>(c.I, c.E, c.H, c.D, c.rdesc, c.K, c.n, c.Ncap, c.hcap, P.pol_match, P.pol_mismatch, P.pol_gap, pk, qbeg, Q, R, dbg, ring_off);
    case 6: if (SECOND) return win_rows<6, true>(c.I, c.E, c.H, c.D, c.rdesc, c.K, c.n, c.Ncap, c.hcap, P.pol_match, P.pol_mismatch, P.pol_gap, pk, qbeg, Q, R, dbg, ring_off); return -1;
    case 8: if (SECOND) return win_rows<8, true>(c.I, c.E, c.H, c.D, c.rdesc, c.K, c.n, c.Ncap, c.hcap, P.pol_match, P.pol_mismatch, P.pol_gap, pk, qbeg, Q, R, dbg, ring_off); return -1;
    case 10: if (SECOND) return win_rows<10, true>(c.I, c.E, c.H, c.D, c.rdesc, c.K, c.n, c.Ncap, c.hcap, P.pol_match, P.pol_mismatch, P.pol_gap, pk, qbeg, Q, R, dbg, ring_off); return -1;
    default: if (SECOND) return win_rows_lin(c, P, pk, qbeg, Q, R, lane); return -1;
  }
}

// spoa heaviest bundle of one window graph -> consensus bases in out[]; returns their number (-1: does not fit).
// Scores / predecessors live in s_score / s_pred: the LDS arrays when the graph fits them (the usual case), a slice of
// the slot's DP scratch in global memory otherwise -- the function is inlined at both call sites so each copy keeps its
// own address space.  The forward sweep takes the nodes 64 at a time: their in-edges are fetched in parallel, then
// consumed in order with lane broadcasts (no dependent global loads on the serial path).
__device__ __forceinline__ int win_consensus(WCtx& c, int* s_score, unsigned short* s_pred, int tgs, int nl, uint8_t* out, int wout_cap, int lane) {
  int olen = 0;
  const int K = c.K, n = c.n;
  // spoa's forward sweep: pred[v] = in-edge of maximum weight, among equal weights the predecessor with the higher score, the
  // LATER edge on equality; score[v] = weight + score[pred[v]] (a node without in-edges scores -1); the consensus ends at the
  // first node of maximal score.  Done 64 positions of the topological order at a time (lane = position): a predecessor
  // outside the chunk is final (one lookup), chains inside the chunk are resolved by pointer doubling between lanes, and
  // the few nodes with a tie for the maximum weight -- the only ones that compare scores -- one by one from the lowest
  // lane up, each after the lanes below it are final.  (The serial sweep spent ~5 dependent LDS round trips per node.)
  int max_id = 0, max_sc = -1;
  bool have_max = false;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int idx = c0 + lane;
    const bool live = idx < n;
    const int v = live ? c.order()[idx] : 0;
    const int nin = live ? c.n_in()[v] : 0;
    int bw = -1, bu = -1, cm = 0;
    {
      // the first two in-edges (nearly every node has no more) are fetched together, weight and source, needed or not: one round
      // trip where the loop made up to four
      const int w0 = c.in_w()[EI(v, 0)], w1 = c.in_w()[EI(v, 1)], u0 = c.in_from()[EI(v, 0)], u1 = c.in_from()[EI(v, 1)];
      if (nin > 0) { bw = w0; bu = u0; cm = 1; }
      if (nin > 1) { if (w1 > bw) { bw = w1; bu = u1; cm = 1; } else if (w1 == bw) ++cm; }
    }
    for (int k = 2; __builtin_amdgcn_ballot_w64(k < nin) != 0; ++k) {
      if (k < nin) {
        const int w = c.in_w()[EI(v, k)];
        if (w > bw) { bw = w; bu = c.in_from()[EI(v, k)]; cm = 1; } else if (w == bw) ++cm;
      }
    }
    bool tie = cm >= 2;
    int acc = nin > 0 ? bw : -1, pr = (nin > 0 && !tie) ? bu : -1;
    int ptr = (nin > 0 && !tie) ? c.index()[bu] : -1;              // position of the predecessor still to follow, -1 = final
    bool done = nin == 0;
    if (!done && !tie && ptr < c0) { acc += s_score[bu]; ptr = -1; done = true; }
    for (;;) {
      for (int r = 0; r < 6 && __builtin_amdgcn_ballot_w64(!done && !tie) != 0; ++r) {
        const int src = (max(ptr, c0) - c0) << 2;
        const int a2 = __builtin_amdgcn_ds_bpermute(src, acc), p2 = __builtin_amdgcn_ds_bpermute(src, ptr);
        const int d2 = __builtin_amdgcn_ds_bpermute(src, (int)done), t2 = __builtin_amdgcn_ds_bpermute(src, (int)tie);
        if (!done && !tie && ptr >= 0) {
          if (d2) { acc += a2; ptr = -1; done = true; }
          else if (!t2) { acc += a2; ptr = p2; }
        }
      }
      const unsigned long long tm = __builtin_amdgcn_ballot_w64(tie && !done);
      if (!tm) break;
      const int T = __builtin_ctzll(tm);                              // lowest unresolved tie lane: everything below it is final
      const int vT = __builtin_amdgcn_readlane(v, T), nT = __builtin_amdgcn_readlane(nin, T);
      int sc = -1, tp = -1, tps = 0;                                   // running best weight, its predecessor and that one's score
      for (int k = 0; k < nT; ++k) {
        const int u = c.in_from()[EI(vT, k)], w = c.in_w()[EI(vT, k)];
        const int pu = c.index()[u];
        const int su = pu < c0 ? s_score[u] : wave_bcast(acc, pu - c0);
        if (sc < w || (sc == w && tps <= su)) { sc = w; tp = u; tps = su; }
      }
      if (lane == T) { acc = sc + tps; pr = tp; done = true; tie = false; ptr = -1; }
    }
    if (live) { s_score[v] = acc; s_pred[v] = pr < 0 ? (unsigned short)0xffff : (unsigned short)pr; }
    // first node of maximal score (strictly greater than everything before it)
    const int cm_ = wave_max(live ? acc : INT32_MIN);
    if (cm_ > max_sc || !have_max) {
      if (cm_ > max_sc) {
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(live && acc == cm_);
        max_id = __builtin_amdgcn_readlane(v, __builtin_ctzll(mm)); max_sc = cm_;
      }
      have_max = true;
    }
    WSYNC();
  }
  if (max_sc < 0) max_id = 0;
  WSYNC();
  if (c.n_out()[max_id] > 0) {
    // branch completion (rare): spill to the global arrays and run spoa's re-scoring there
    for (int v = lane; v < n; v += 64) { c.score[v] = s_score[v]; c.pred()[v] = s_pred[v] == 0xffff ? -1 : (int)s_pred[v]; }
    WSYNC();
    if (lane == 0) {
      while (c.n_out()[max_id] > 0) {
        const int v = max_id;
        for (int k = 0; k < c.n_out()[v]; ++k) {
          const int t2 = c.out_to()[EI(v, k)];
          for (int e = 0; e < c.n_in()[t2]; ++e) { int u = c.in_from()[EI(t2, e)]; if (u != v) c.score[u] = -1; }
        }
        long long ms = 0; int mid = -1;
        for (int i = c.index()[v] + 1; i < n; ++i) {
          const int x = c.order()[i];
          c.score[x] = -1; c.pred()[x] = -1;
          for (int k = 0; k < c.n_in()[x]; ++k) {
            const int u = c.in_from()[EI(x, k)]; const long long w = c.in_w()[EI(x, k)];
            if (c.score[u] == -1) continue;
            if (c.score[x] < w || (c.score[x] == w && c.score[c.pred()[x]] <= c.score[u])) { c.score[x] = w; c.pred()[x] = u; }
          }
          if (c.pred()[x] != -1) c.score[x] += c.score[c.pred()[x]];
          if (ms < c.score[x]) { ms = c.score[x]; mid = x; }
        }
        if (mid < 0) break;
        max_id = mid;
      }
      c.anchor()[0] = max_id;
    }
    WSYNC();
    max_id = c.anchor()[0];
    for (int v = lane; v < n; v += 64) s_pred[v] = c.pred()[v] < 0 ? 0xffff : (unsigned short)c.pred()[v];
    WSYNC();
  }
  // consensus path (LDS pointer chase), coverage trim, emit
  int nc = 0;
  for (int v = max_id; v != 0xffff; v = s_pred[v]) { if (lane == 0) c.opn()[nc] = v; ++nc; }
  WSYNC();
  int b = 0, e = nc - 1;      // positions in forward order: forward[i] = opn[nc-1-i]
  if (tgs) {
    const int avg = nl / 2;
    for (; b < nc; ++b) if (c.ncov()[c.opn()[nc - 1 - b]] >= avg) break;
    for (; e >= 0; --e) if (c.ncov()[c.opn()[nc - 1 - e]] >= avg) break;
    if (b >= e) { b = 0; e = nc - 1; }
  }
  if (e - b + 1 > wout_cap) olen = -1;
  else { for (int i = b + lane; i <= e; i += 64) out[i - b] = c.base()[c.opn()[nc - 1 - i]]; olen = e - b + 1; }
  return olen;
}

// Occupancy is what hides the latency of the dependent row chain: 3 waves/SIMD (168 VGPRs) 98.3 ms, 4 waves (128 VGPRs,
// 38 spilled outside the row loops) 86.9 ms per 32768 cfg2 reads; with the LDS sweep sized for 2*WL+30*NL nodes (6.9 KB
// per wave, larger graphs fall back to global scratch) 5 waves/SIMD run 65.5 ms against 72.2 ms; 6 waves spill into the
// row loops (76 ms).
// Round 5: the first launch at SIX waves per SIMD (80 VGPRs; five LDS granules of 1 280 bytes): 30.9 -> 28.4 ms (cfg2), 59.1 -> 54.4 (cfg4),
// 33.9 -> 31.1 (cfg3).  Round 4 had measured "6 waves: nothing" -- with 6 656 bytes of LDS, i.e. six granules and 21 resident waves.
#ifndef C3_BAND_THIN
#define C3_BAND_THIN 48        /* certificate margin (score units) below which the next layer of the window starts one band wider */
#endif
#ifndef C3_WIN_WAVES
#define C3_WIN_WAVES 6
#endif
#ifndef C3_WIN_WAVES2
#define C3_WIN_WAVES2 5
#endif
template <bool SECOND>
__global__ __launch_bounds__(64, SECOND ? C3_WIN_WAVES2 : C3_WIN_WAVES) void k_window(WinArgs a) {
  const int lane = wave_lane();
  const int slot = blockIdx.x;
  WCtx c;
  const size_t N = (size_t)a.Ncap;
  c.I = a.ibase + (size_t)slot * W_INTS * N; c.E = a.ebase + (size_t)slot * 4 * N * a.K;
  c.B8 = a.base + (size_t)slot * 2 * N; c.score = a.score + slot * N;
  c.H = a.H + (size_t)slot * a.hcap; c.D = (uint8_t*)a.D + (size_t)slot * a.hcap; c.rdesc = a.rdesc + slot * (N + 1);
  c.K = a.K; c.Ncap = a.Ncap; c.hcap = a.hcap;
  const C3Params& P = a.p;
  extern __shared__ int lds_dyn[];                      // [Ncap] scores + [Ncap] u16 predecessors
  int* s_score = lds_dyn; unsigned short* s_pred = (unsigned short*)(lds_dyn + a.Lcap);
  // the same LDS holds the row-type bitmasks of the layer being aligned (DP rows + traceback; the consensus sweep comes later)
  const int MW = win_mask_words(a.Ncap);
  unsigned long long* m2bits = (unsigned long long*)lds_dyn; unsigned long long* mabits = m2bits + MW;
  unsigned long long* d0bits = mabits + MW; unsigned long long* d1bits = d0bits + MW;         // band shift bits of every DP row (banded layers)
  const int lds_ints = (int)(win_lds_bytes(a.Lcap, a.Ncap, !SECOND) / 4);    // = the launch's dynamic LDS
  PH_DECL
  if (!SECOND && lane == 0) __hip_atomic_store(a.counter + W_CNT_START, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (for the consumer launch beside this one: see below)

  for (;;) {
    int wi = 0, qi = 0;
    if (lane == 0) {
      if (SECOND && a.done_flag) {
        // consumer beside the first launch (round 6): a window that overflows there used to wait for that launch to drain and then ran ALONE on
        // an idle device -- 1.5-8 ms of single-wave latency behind every batch (2 ms of cfg2's 81, 16 of cfg4's 591, more on noisy reads).  Now
        // it is picked up while the first launch still runs: an entry is claimed (CAS on the queue head) only when the list holds one, so a
        // consumer that gives up -- the flag "no further entry will come" is set by the host behind the first launch; the wait is bounded all
        // the same: a host that never sets it costs seconds, not a hung device -- leaves nothing half taken for the launch that mops up
        wi = -1;
        // ... and first of all it must see the first launch RUNNING (every wave of it raises counter[W_CNT_START]): when the two kernels do not
        // overlap -- the first k_window launch of a process waits for the queue's scratch to be (re)allocated, which waits for this kernel;
        // counter passes of a profiler serialise dispatches -- half a millisecond is all this launch may cost
        bool started = false;
        for (int it = 0; it < 256 && !(started = __hip_atomic_load(a.counter + W_CNT_START, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0); ++it) __builtin_amdgcn_s_sleep(16);
        for (int it = 0; started && it < (1 << 22); ++it) {
          const int done = __hip_atomic_load(a.done_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);          // (read BEFORE the count: with the flag set the count is final)
          const int nl_ = __hip_atomic_load(a.n_win_dev, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT), q = __hip_atomic_load(a.counter + W_CNT_Q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (q < nl_) {
            if (atomicCAS(a.counter + W_CNT_Q2, q, q + 1) != q) continue;
            qi = q;
            // The producer takes its index first and writes the entry right after: a moment, but not none -- and on this device a moment can be
            // long: a plain store stays in the producing XCD's L2 until that kernel ends, invisible to a consumer on another XCD (the first
            // version bounded this wait and, when the first launch ran for seconds, gave up on entries it had CLAIMED: six windows of a 250-read
            // fuzz batch lost, profiles/r06_fuzz_parity.txt).  The entry is stored and loaded with agent scope, and a claimed entry is waited
            // for without a bound: its producer is a running wave that waits for nothing
            while ((wi = __hip_atomic_load(a.wlist + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < 0) __builtin_amdgcn_s_sleep(8);
            break;
          }
          if (done) break;
          __builtin_amdgcn_s_sleep(64);
        }
      } else if (SECOND) {                          // the windows the first launch could not hold: list and count in device memory
        qi = wi = atomicAdd(a.counter + W_CNT_Q2, 1);
        wi = wi < *(const volatile int*)a.n_win_dev ? a.wlist[wi] : -1;
      } else {
        wi = atomicAdd(a.counter, 1);
        if (wi >= a.n_win) wi = -1;
      }
    }
    wi = wave_first(wi); qi = wave_first(qi);
    if (wi < 0) break;
    PH_MARK(9)
    const WinRec rec = a.wrec_in[wi];
    const int rid = rec.rid, blen = rec.blen, nl = rec.n_layers;
    const int64_t off = a.b.off[rid];
    const uint32_t* pk = a.b.pk + a.b.woff[rid];
    const uint8_t* qual = a.b.qual + off;
    const uint8_t* bb = a.draft + off + (size_t)rec.w * P.pol_window;
    // (second launch: its own output slots, as long as a graph can have nodes)
    const int ocap = SECOND ? (qi < a.wout2_n ? a.wout2_cap : 0) : a.wout_cap;
    uint8_t* out = SECOND ? a.wout2 + (size_t)min(qi, max(a.wout2_n - 1, 0)) * a.wout2_cap : a.wout + (size_t)wi * a.wout_cap;
    const WLayer* lay = a.wlay + (size_t)wi * a.NLcap;
    long long cells = 0, cells_done = 0;          // cells of the full matrices (what the oracle counts) / cells actually computed
    int olen = 0, polished = 0, fail = 0, n_band = 0, n_fallback = 0;
    // band width the NEXT layer of this window starts with (cells per lane; the inflation rule of win_rows_dispatch is a floor under it).
    // The layers of a window come from one read and share its error rate, and that -- not the graph -- is what the certificate's margin
    // depends on: an optimal path scores ~2.25 per column at 10 % errors and ~1.9 at 15 %, the bound of a path that leaves the band assumes 3
    // for everything still to come, so noisier reads need the wider band's deeper edge cells.  At 15 % errors 27 % of the layers failed the
    // 128-column band and ALL of them passed at 192 (tools/experiments/band_stats.py): a layer that fails, or passes with a thin margin, sends
    // the layers after it straight to the width it needed.  A hint only: every accepted band reproduces the full matrix.
    int cb_hint = 1;
    if (nl + 1 < 3) {
      if (blen > ocap) { fail = SECOND ? 1 : 2; if (SECOND && lane == 0) atomicAdd(a.counter + W_CNT_WHY, 1); }
      else { for (int i = lane; i < blen; i += 64) out[i] = bb[i]; olen = blen; }
    } else {
      // ---- backbone chain (weight-0 edges, coverage 1)
      c.osel = 0;
      for (int i = lane; i < blen; i += 64) {
        c.base()[i] = bb[i]; c.grp()[i] = i; c.order()[i] = i; c.index()[i] = i; c.ncov()[i] = 1;
        c.n_in()[i] = i > 0; c.n_out()[i] = i + 1 < blen;
        if (i > 0) { c.in_from()[EI(i, 0)] = i - 1; c.in_w()[EI(i, 0)] = 0; }
        if (i + 1 < blen) { c.out_to()[EI(i, 0)] = i + 1; c.out_w()[EI(i, 0)] = 0; }
      }
      c.n = blen;
      WSYNC();
      w_blocks(c, lane);
      PH_MARK(0)
      // ---- stable order of the layers by begin position (tiny; every lane computes it)
      const int offset = (int)(0.01 * (double)blen);
      // (lane x holds the begin of layer x and counts the layers before it once; the t loop then finds the layer of rank t with
      // one ballot -- the plain triple loop re-read the begins from memory nl^3 times: 2 744 loads per window at cfg4's 14 layers)
      const int lbeg = lane < nl ? lay[min(lane, nl - 1)].begin : INT32_MAX;
      int lrank = 0;
      if (nl <= 64) for (int y = 0; y < nl; ++y) { const int by = __builtin_amdgcn_readlane(lbeg, y); lrank += (by < lbeg) || (by == lbeg && y < lane); }
      for (int t = 0; t < nl && !fail; ++t) {
        // rank t = the layer with t layers before it in (begin, index) order
        int li = -1;
        if (nl <= 64) li = __builtin_ctzll(__ballot(lane < nl && lrank == t) | (1ull << 63));
        else for (int x = 0; x < nl; ++x) {
          int before = 0;
          for (int y = 0; y < nl; ++y) before += (lay[y].begin < lay[x].begin) || (lay[y].begin == lay[x].begin && y < x);
          if (before == t) { li = x; break; }
        }
        const WLayer l = lay[li];
        const int Q = l.len;
        // (layers beyond W_QCAP bases keep rq / tq in the slot's opq / opn arrays, 2 * Ncap ints each: a longer layer -- a 1 200-base
        // insertion against a 100-base window -- does not fit the first launch's arrays; the full-size launch has Ncap >= every layer)
        if (Q > 2 * c.Ncap) { fail = SECOND ? 1 : 2; if (SECOND && lane == 0) atomicAdd(a.counter + W_CNT_WHY + 1, 1); break; }
        const bool full = l.begin < offset && l.end > blen - offset;
        // ---- rows of this alignment (masked sub-graph or everything)
        if (!full) {
          // spoa Graph::subgraph(begin, end): nodes with id >= begin that reach backbone node `end`
          // through edges / aligned-block links.  Computed as a shrinking fixpoint over whole aligned
          // blocks (the block graph is a DAG, so greatest == least fixpoint); a few parallel sweeps.
          const int K = c.K;
          const int eidx = c.glast()[c.grp()[l.end]];
          const WArr<int> gseed = c.rowof();
          for (int i = lane; i < c.n; i += 64) { const int v = c.order()[i]; c.mask()[v] = (v >= l.begin && i <= eidx) ? 1 : 0; }
          WSYNC();
          // (every sweep takes WU chunks of 64 nodes per turn and issues the loads of one level for all of them together: the
          // sweeps are chains of dependent gathers -- node -> out-edges -> mask of the targets -- and were all waiting)
          for (;;) {
            for (int v0 = 0; v0 < c.n; v0 += 64 * WU) {
              int m_[WU], g_[WU];
#pragma unroll
              for (int u = 0; u < WU; ++u) { const int v = min(v0 + 64 * u + lane, c.n - 1); m_[u] = c.mask()[v]; g_[u] = c.grp()[v]; }
#pragma unroll
              for (int u = 0; u < WU; ++u) if (v0 + 64 * u + lane < c.n && m_[u]) gseed[g_[u]] = 0;
            }
            WSYNC();
            for (int v0 = 0; v0 < c.n; v0 += 64 * WU) {
              int m_[WU], g_[WU], no_[WU], t0_[WU], t1_[WU], s0_[WU], s1_[WU];
#pragma unroll
              for (int u = 0; u < WU; ++u) {
                const int v = min(v0 + 64 * u + lane, c.n - 1);
                m_[u] = c.mask()[v]; g_[u] = c.grp()[v]; no_[u] = c.n_out()[v]; t0_[u] = c.out_to()[EI(v, 0)]; t1_[u] = c.out_to()[EI(v, 1)];
              }
#pragma unroll
              for (int u = 0; u < WU; ++u) {        // (slots beyond the out-degree hold stale node ids: clamped, read, ignored)
                s0_[u] = c.mask()[min(max(t0_[u], 0), c.Ncap - 1)]; s1_[u] = c.mask()[min(max(t1_[u], 0), c.Ncap - 1)];
              }
#pragma unroll
              for (int u = 0; u < WU; ++u) {
                const int v = v0 + 64 * u + lane;
                if (v >= c.n || !m_[u]) continue;
                int seed = v == l.end || (no_[u] > 0 && s0_[u]) || (no_[u] > 1 && s1_[u]);
                for (int k = 2; k < no_[u] && !seed; ++k) seed = c.mask()[c.out_to()[EI(v, k)]];
                if (seed) gseed[g_[u]] = 1;
              }
            }
            WSYNC();
            int changed = 0;
            for (int v0 = 0; v0 < c.n; v0 += 64 * WU) {
              int m_[WU], g_[WU], gs_[WU];
#pragma unroll
              for (int u = 0; u < WU; ++u) { const int v = min(v0 + 64 * u + lane, c.n - 1); m_[u] = c.mask()[v]; g_[u] = c.grp()[v]; }
#pragma unroll
              for (int u = 0; u < WU; ++u) gs_[u] = gseed[g_[u]];
#pragma unroll
              for (int u = 0; u < WU; ++u) { const int v = v0 + 64 * u + lane; if (v < c.n && m_[u] && !gs_[u]) { c.mask()[v] = 0; changed = 1; } }
            }
            changed = __ballot(changed) != 0;
            WSYNC();
            if (!changed) break;
          }
        }
        PH_MARK(1)
        int R = 0;
#ifdef C3_EXP_X2_COMP
        for (int rep_ = 0; rep_ < 2; ++rep_) { R = 0;
#endif
        for (int i0 = 0; i0 < c.n; i0 += 64 * WU) {   // order-preserving compaction, WU chunks of 64 positions per iteration
          int v[WU]; bool in[WU];
#pragma unroll
          for (int u = 0; u < WU; ++u) { const int i = i0 + 64 * u + lane; v[u] = i < c.n ? c.order()[i] : -1; }
#pragma unroll
          for (int u = 0; u < WU; ++u) in[u] = v[u] >= 0 && (full || c.mask()[v[u]]);
#pragma unroll
          for (int u = 0; u < WU; ++u) {
            const unsigned long long bal = __ballot(in[u]);
            if (v[u] >= 0) {
              if (in[u]) { int r = R + 1 + __popcll(bal & ((1ull << lane) - 1)); c.rows()[r] = v[u]; c.rowof()[v[u]] = r; }
              else c.rowof()[v[u]] = -1;
            }
            R += __popcll(bal);
          }
        }
#ifdef C3_EXP_X2_COMP
        WSYNC(); }
#endif
        WSYNC();
        PH_MARK(2)
        const int ring_off = 8 * MW;                                          // LDS behind the four row bitmasks, in dwords
        // (LDS behind the traceback windows: W_TBW dwords in, 2 x W_QCAP shorts)
        const QArr rq = {(unsigned short*)((unsigned*)lds_dyn + ring_off + W_TBW), c.opq(), Q > W_QCAP};
        const QArr tq = {(unsigned short*)((unsigned*)lds_dyn + ring_off + W_TBW) + W_QCAP, c.opn(), Q > W_QCAP};
        // (band_mode 3, test hook: every layer the certificate accepted is aligned a second time with the full matrix and the two
        // tracebacks are compared base by base -- the certificate's claim, checked on the device)
        unsigned long long tbp_[4] = {0, 0, 0, 0};
        bool verify = false;
        for (int vpass = 0; vpass < 2 && !fail; ++vpass) {
        int cpl = 0, RS = 0, cb = (a.band_mode != 1 && vpass == 0) ? (a.band_mode == 0 ? cb_hint : 1) : 0, gbs = INT32_MIN, gbr = 0;
        for (int attempt = 0; attempt < 4; ++attempt) {
          unsigned long long dbg_[6] = {0, 0, 0, 0, 0, 0};
          int nblocks = 0;
          if (win_rows_dispatch<SECOND>(c, P, pk, l.qbeg, Q, R, lane, &cpl, &RS, dbg_, m2bits, mabits, d0bits, d1bits, ring_off, lds_ints, l.begin, l.end, blen, &cb, &nblocks) < 0) { fail = SECOND ? 1 : 2; if (SECOND && lane == 0) atomicAdd(a.counter + W_CNT_WHY + 1, 1); break; }      // (2: the layer needs more DP scratch than this launch has -- the window goes to the full-size launch)
#ifdef C3_PHASE_PROF
          ph_acc_[10] += dbg_[0]; ph_acc_[11] += dbg_[1];
#endif
          PH_MARK(3)
          cells_done += (long long)(R + 1) * (cb ? 64 * cb : Q + 1);
          // ---- end row: candidate rows (no masked successor) left H[r][Q] in hend; first maximum in order
          if (cb) {
            // banded rows: the row loop kept the best end row itself and left the certificate bound of every path outside the
            // band; the band result is accepted only if that bound stays strictly below the banded optimum
            WSYNC();
            const int bound = c.lob()[0];
            gbs = c.lob()[1]; gbr = c.lob()[2];
#ifdef C3_EXP_NOCERT
            if (cb) { ++n_band; break; }
#endif
            if (gbs != INT32_MIN && bound < gbs && a.band_mode != 2) {
              ++n_band;
              if (a.band_mode == 0 && C3_BAND_THIN >= 0) cb_hint = (gbs - bound < C3_BAND_THIN && cb < 4) ? cb + 1 : cb;      // accepted by a thin margin: the next layer one wider
              break;
            }
          } else {
            int bs = INT32_MIN, br = INT32_MAX / 2;
            for (int r = 1 + lane; r <= R; r += 64) { const int sc = c.hend()[r]; if (sc > bs) { bs = sc; br = r; } }
            gbs = wave_max(bs);
            gbr = wave_min(bs == gbs ? br : INT32_MAX / 2);
            break;
          }
          // a failed certificate: one more try with the next wider band (half again / twice the margin for the bounds, still well
          // below the full matrix), then the full matrix
          ++n_fallback; cb = (cb < 4 && a.band_mode == 0) ? cb + 1 : 0;
          if (a.band_mode == 0 && C3_BAND_THIN >= 0) cb_hint = cb ? cb : 4;
        }
        if (fail) break;
        cells += (long long)(R + 1) * (Q + 1);
        PH_MARK(4)
        // ---- traceback.  rq[q] = DP row aligned to query base q, 0 = insertion.
        // 64 ROWS AT A TIME: lane k owns row rt-k, loads its descriptor and a 16-byte window of its direction cells around
        // the column where the diagonal through the current cell crosses that row (32-64 cells of a 2-bit row, 12-16 of a
        // byte row) and parks the window in LDS (the consensus arrays are not live yet).  Inside the block every step is an
        // LDS read: the wave checks 64 cells down the diagonal at once ("diagonal move from the row above"?), consumes the
        // run, and resolves the cell that breaks it.  One memory round trip per 64 rows instead of one per break -- and
        // the traceback no longer fetches about as many bytes as the fill wrote.
#ifdef C3_PHASE_PROF
        unsigned long long tbc_[3] = {0, 0, 0};
#endif
#ifdef C3_EXP_X2_TB
        for (int tbrep = 0; tbrep < 2; ++tbrep)
#endif
        if (cb) win_traceback_band<SECOND>(c.I, c.E, c.D, c.rdesc, c.K, c.n, c.Ncap, cb, R, Q, (gbs == INT32_MIN) ? 0 : gbr, MW, Q > W_QCAP, tbp_);
        else
        {
          unsigned* WD = (unsigned*)lds_dyn + ring_off;       // [64][4] dwords, behind the row-type bitmasks
          const int cdiv = cpl ? (65536 + cpl - 1) / cpl : 0;                   // j / cpl == (j * cdiv) >> 16 for j < 2^13
          const int tb = cpl <= 8 ? 2 : 4;                                       // bytes per lane of a 2-bit row
          int r = (gbs == INT32_MIN) ? 0 : gbr, j = Q;
          while (r > 0 || j > 0) {
            if (r == 0) { for (int q = lane; q < j; q += 64) rq.set(q, 0); break; }
            if (j == 0) break;                                   // only vertical moves remain
            const int rt = r, jt = j;
            const int rk = rt - lane;
            const bool rowv = rk >= 1;
            const int rb = max(rk, 1) - 1;
            const bool two = rowv && ((m2bits[rb >> 6] >> (rb & 63)) & 1);
            const bool adj = two && ((mabits[rb >> 6] >> (rb & 63)) & 1);
            const uint4 de = c.rdesc[max(rk, 1)];
            // byte offset of cell `col` inside a D row: two-bit rows keep one dword per lane (lane = col / cpl), byte rows
            // ds_ bytes per lane; the window starts 4-byte aligned a little left of the expected column
            const int ce = max(jt - lane, 0);
            const int le = cpl ? (ce * cdiv) >> 16 : 0;
            int wb;                                                             // window start (byte offset in the row)
            if (two) wb = (max(le - 2, 0) * tb) & ~3;
            else wb = max((cpl ? le * (((cpl + 3) & ~3)) + (ce - le * cpl) : ce) - 6, 0) & ~3;
            wb = min(wb, max(RS - 16, 0));
            {
              const unsigned* src = (const unsigned*)(c.D + (size_t)max(rk, 1) * RS + wb);
              uint4 x = make_uint4(0, 0, 0, 0);
              if (rowv) x = make_uint4(src[0], src[1], src[2], src[3]);
              *(uint4*)(WD + lane * 4) = x;
            }
            WSYNC();
            for (;;) {
              const int s = rt - r;                               // lane s holds the current row
              const int jk = j - (lane - s);
              const bool val = rowv && lane >= s && jk >= 0;
              const int jc = max(jk, 0);
              const int lq = cpl ? (jc * cdiv) >> 16 : 0, cw = jc - lq * cpl;
              const int bo = (two ? lq * tb : (cpl ? lq * ((cpl + 3) & ~3) + cw : jc)) - wb;      // byte offset in the window
              const bool hit = val && bo >= 0 && bo + (two ? tb : 1) <= 16;
              const unsigned wv = WD[lane * 4 + (min(max(bo, 0), 15) >> 2)];
              int d, prow = -1;
              if (two) { const unsigned cellw = tb == 2 ? (wv >> (8 * (bo & 2))) & 0xffffu : wv; d = 63 + 64 * (int)((cellw >> (2 * cw)) & 3u); prow = adj ? rk - 1 : (int)(de.y & 0xffff); }     /* a single predecessor: the descriptor this lane holds names it (no reload at a break) */
              else {
                d = (int)((wv >> (8 * (bo & 3))) & 0xffu);
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
                // outside the window (the path drifted off this block's diagonal): direct loads of the one cell
                const bool two0 = (m2bits[(r - 1) >> 6] >> ((r - 1) & 63)) & 1;
                if (two0) {
                  const unsigned w0 = tb == 2 ? (unsigned)((const unsigned short*)(c.D + (size_t)r * RS))[j / cpl] : ((const unsigned*)(c.D + (size_t)r * RS))[j / cpl];
                  db = 63 + 64 * (int)((w0 >> (2 * (j % cpl))) & 3u);
                  pb = ((mabits[(r - 1) >> 6] >> ((r - 1) & 63)) & 1) ? r - 1 : -3;
                } else {
                  db = c.D[(size_t)r * RS + win_idx(j, cpl)];
                  pb = -2;
                }
              }
              const int ty = win_d_type(db);
              if (ty == 2) { if (lane == 0) rq.set(j - 1, 0); --j; }
              else {
                if (pb <= -2) pb = win_pred_row(c, c.rdesc[r], r, win_d_pred(db));   // > 4 predecessors, a 2-bit row whose predecessor is not r-1, or a window miss
                if (ty == 0) { if (lane == 0) rq.set(j - 1, r); --j; }
                r = pb;
              }
              if (r <= 0 || j <= 0) break;
              const int drift = (jt - j) - (rt - r);
              if (rt - r >= 64 || drift > 5 || drift < -5) break;
            }
            WSYNC();
          }
        }
        if (!(a.band_mode == 3 && vpass == 0 && cb)) break;
        verify = true;
        WSYNC();
        for (int q = lane; q < Q; q += 64) c.opq()[q] = rq.get(q);          // (global: the unbanded rows reuse the LDS, their descriptor build uses opn)
        WSYNC();
        }
        if (verify && !fail) {
          int diff = 0;
          for (int q = lane; q < Q; q += 64) diff |= c.opq()[q] != rq.get(q);
          const unsigned long long dm = __ballot(diff);
          if (dm != 0) {
            // the highest differing query base (the walks start at the end): enough to find the row where they part
            int qd = -1;
            for (int q = lane; q < Q; q += 64) if (c.opq()[q] != rq.get(q)) qd = q;
            qd = wave_max(qd);
            if (lane == 0) { atomicAdd(a.counter + 8, 1); a.counter[9] = wi; a.counter[10] = t; a.counter[11] = R; a.counter[12] = qd; a.counter[13] = c.opq()[qd]; a.counter[14] = rq.get(qd);
                             a.counter[15] = qd + 1 < Q ? rq.get(qd + 1) : -1; }
          }
        }
        WSYNC();
        if (fail) break;                            // (no traceback happened: rq holds nothing -- the fusion below must not run)
        PH_MARK(5)
#ifdef C3_PHASE_PROF
        ph_acc_[12] += tbp_[0]; ph_acc_[13] += tbp_[1]; ph_acc_[14] += tbp_[2]; ph_acc_[15] += tbp_[3];       // traceback census: blocks, steps, window misses (overrides the row-kind cycles)
#endif
        // ---- fusion, parallel over the query bases (every graph node is touched by at most one base)
        const int n_old = c.n;
        int carry_anchor = -1, carry_new = 0;
        // (two 64-base chunks per turn, the loads of one level for both chunks issued together -- rows -> group + base -> block
        // extent -- then the scans and the stores chunk by chunk, in order: the carries run along the path)
        for (int q0 = 0; q0 < Q; q0 += 128) {
          int v_[2], cb_[2], rr_[2], bs_[2], anc_[2], gf_[2], tgt_[2], gnew_[2];
          bool act_[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int q = q0 + 64 * h2 + lane;
            act_[h2] = q < Q;
            const int r = act_[h2] ? rq.get(q) : 0;
            v_[h2] = r > 0 ? c.rows()[r] : -1;
            cb_[h2] = act_[h2] ? c3_code_at(pk, l.qbeg + q) : 0;
          }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) { const int vv = max(v_[h2], 0); rr_[h2] = c.grp()[vv]; bs_[h2] = c.base()[vv]; }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) { anc_[h2] = c.glast()[rr_[h2]]; gf_[h2] = c.gfirst()[rr_[h2]]; }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            int tgt = -1, gnew = -1;
            if (v_[h2] >= 0) {
              if (bs_[h2] == cb_[h2]) tgt = v_[h2];
              else for (int i = gf_[h2]; i <= anc_[h2]; ++i) { int x = c.order()[i]; if (c.base()[x] == cb_[h2]) { tgt = x; break; } }
              if (tgt < 0) gnew = rr_[h2];
            } else anc_[h2] = -1;
            tgt_[h2] = tgt; gnew_[h2] = gnew;
          }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int q = q0 + 64 * h2 + lane;
            if (q0 + 64 * h2 >= Q) break;
            const bool act = act_[h2];
            const int cb = cb_[h2], gnew = gnew_[h2], anc = anc_[h2];
            int tgt = tgt_[h2];
            const int isnew = act && tgt < 0;
            const int as = max(wave_scan_max(anc), carry_anchor);       // anchors are non-decreasing along the path
            carry_anchor = wave_bcast(as, 63);
            const int ps = wave_scan_add(isnew);
            const int k = carry_new + ps - isnew;
            carry_new += wave_bcast(ps, 63);
            if (isnew) {
              const int id = n_old + k;
              if (id < c.Ncap) {
                c.base()[id] = (uint8_t)cb; c.n_in()[id] = 0; c.n_out()[id] = 0; c.grp()[id] = gnew >= 0 ? gnew : id; c.ncov()[id] = 0;
                c.anchor()[k] = as;
              }
              tgt = id;
            }
            if (act) tq.set(q, tgt);
          }
        }
        const int nn = n_old + carry_new;
        if (nn > c.Ncap) { fail = SECOND ? 1 : 2; if (SECOND && lane == 0) atomicAdd(a.counter + W_CNT_WHY + 2, 1); break; }      // (2: more nodes than the first launch's graph arrays hold -- the full-size launch has the worst case)
        WSYNC();
        const int K = c.K;
        // edges of the path, two 64-base chunks per turn.  Every base owns the out-list of its left neighbour's node and the in-list
        // of its own node (a node is touched by at most one base), so the chunks are independent; the loads of one level -- counts
        // and the first two slots of both lists, speculatively -- are issued together for both chunks: three dependent round trips
        // per turn where the plain loop made six per chunk (this phase was 15 % of the wave time, all of it waiting)
        for (int q0 = 0; q0 < Q; q0 += 128) {
          int v_[2], u_[2], w_[2], no_[2], ni_[2], nc_[2], o0_[2], o1_[2], i0_[2], i1_[2];
          bool ed_[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int q = q0 + 64 * h2 + lane;
            const bool act = q < Q;
            ed_[h2] = act && q > 0;
            v_[h2] = act ? tq.get(q) : 0;
            u_[h2] = ed_[h2] ? tq.get(q - 1) : 0;
            const int qq = l.qbeg + (ed_[h2] ? q : 1);
            w_[h2] = ((int)qual[qq - 1] - 33) + ((int)qual[qq] - 33);
          }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            nc_[h2] = c.ncov()[v_[h2]]; no_[h2] = c.n_out()[u_[h2]]; ni_[h2] = c.n_in()[v_[h2]];
            o0_[h2] = c.out_to()[EI(u_[h2], 0)]; o1_[h2] = c.out_to()[EI(u_[h2], 1)];
            i0_[h2] = c.in_from()[EI(v_[h2], 0)]; i1_[h2] = c.in_from()[EI(v_[h2], 1)];
          }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int q = q0 + 64 * h2 + lane;
            if (q < Q) c.ncov()[v_[h2]] = nc_[h2] + 1;
            if (!ed_[h2]) continue;
            const int u = u_[h2], v = v_[h2], w = w_[h2], no = no_[h2], ni = ni_[h2];
            int hit = (no > 0 && o0_[h2] == v) ? 0 : (no > 1 && o1_[h2] == v) ? 1 : -1;
            if (hit < 0) for (int k = 2; k < no; ++k) if (c.out_to()[EI(u, k)] == v) { hit = k; break; }
            if (hit >= 0) {
              int hin = (ni > 0 && i0_[h2] == u) ? 0 : (ni > 1 && i1_[h2] == u) ? 1 : -1;
              if (hin < 0) for (int t2 = 2; t2 < ni; ++t2) if (c.in_from()[EI(v, t2)] == u) { hin = t2; break; }
              c.out_w()[EI(u, hit)] += w;
              if (hin >= 0) c.in_w()[EI(v, hin)] += w;
            } else {
              c.out_to()[EI(u, no)] = v; c.out_w()[EI(u, no)] = w; c.n_out()[u] = no + 1;
              c.in_from()[EI(v, ni)] = u; c.in_w()[EI(v, ni)] = w; c.n_in()[v] = ni + 1;
            }
          }
        }
        WSYNC();
        c.n = nn;
        PH_MARK(6)
        w_reorder(c, n_old, lane, lds_dyn, lds_ints);     // the whole dynamic LDS is idle between traceback and consensus
        PH_MARK(7)
      }
      if (!fail) {
        // ---- consensus: LDS-resident sweep; graphs larger than the LDS arrays use the slot's DP scratch instead
        if (c.n <= a.Lcap) olen = win_consensus(c, s_score, s_pred, rec.tgs, nl, out, ocap, lane);
        else olen = win_consensus(c, (int*)c.H, (unsigned short*)(c.H + c.Ncap), rec.tgs, nl, out, ocap, lane);
        PH_MARK(8)
        if (olen < 0) { fail = SECOND ? 1 : 2; olen = 0; if (SECOND && lane == 0) atomicAdd(a.counter + W_CNT_WHY + 3, 1); } else polished = 1;      // (2: a consensus longer than the first launch's output slot)
      }
    }
    if (fail == 2 && a.ovf_list) {
      // (nothing of this window has been published; the entry with agent scope: the consumer beside this launch may sit on another XCD)
      if (lane == 0) {
        const int k_ = atomicAdd(a.counter + W_CNT_OVF, 1);
        if (a.dbg_ovf_delay) for (long long t_ = __builtin_readcyclecounter() + (long long)a.dbg_ovf_delay * 100000; (long long)__builtin_readcyclecounter() < t_;) __builtin_amdgcn_s_sleep(64);      // (test hook; s_memtime counts at 100 MHz)
        __hip_atomic_store(a.ovf_list + k_, wi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (lane == 0) {
      WinRec* r = &a.wrec[wi];
      r->out_len = fail ? -1 : olen; r->polished = polished; r->pad_ = SECOND ? qi + 1 : 0;        // (fail == 2 with no second launch to take it: the window is over the limits)
      atomicAdd((unsigned long long*)(a.counter + 2), (unsigned long long)cells);
      atomicAdd((unsigned long long*)(a.counter + 4), (unsigned long long)cells_done);
      if (n_band) atomicAdd(a.counter + 6, n_band);
      if (n_fallback) atomicAdd(a.counter + 7, n_fallback);
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
    if (a.zflag && a.zflag[rid]) continue;             // zero-repeat rescue: already final, never polished
    const int64_t off = a.b.off[rid];
    const int L = (int)(a.b.off[rid + 1] - off);
    const int nwin = info->n_win, wb = a.win_base[rid];
    char* cons = a.cons + off;
    int olen = 0, any = 0, bad = 0;
    for (int w = 0; w < nwin; ++w) {
      const WinRec r = a.wrec[wb + w];
      if (r.out_len < 0) { bad = 1; break; }
      if (olen + r.out_len > L) { bad = 2; break; }          // longer than the read: the caller's buffer ends there (the oracle: no consensus)
      const uint8_t* src = r.pad_ ? a.wout2 + (size_t)(r.pad_ - 1) * a.wout2_cap : a.wout + (size_t)(wb + w) * a.wout_cap;
      for (int i = lane; i < r.out_len; i += 64) cons[olen + i] = "ACGT"[src[i] & 3];
      olen += r.out_len; any |= r.polished;
    }
    if (lane == 0) {
      if (bad == 1) { info->status = C3_ST_LIMIT; info->cons_len = 0; }
      else if (bad == 2) { info->status = C3_ST_NO_CONSENSUS; info->cons_len = 0; }
      else if (!any || olen == 0) { info->status = C3_ST_NO_CONSENSUS; info->cons_len = 0; }   // racon drops unpolished targets
      else info->cons_len = olen;
    }
  }
}

extern "C" void c3k_launch_prep(const PrepArgs* a, int slots, hipStream_t s) { hipLaunchKernelGGL(k_prep, dim3(slots), dim3(64), 0, s, *a); }
extern "C" void c3k_launch_window(const WinArgs* a, int slots, hipStream_t s) {
  const size_t lds = win_lds_bytes(a->Lcap, a->Ncap, a->wlist == nullptr);
  if (a->wlist) hipLaunchKernelGGL(k_window<true>, dim3(slots), dim3(64), lds, s, *a);
  else hipLaunchKernelGGL(k_window<false>, dim3(slots), dim3(64), lds, s, *a);
}
extern "C" void c3k_launch_stitch(const StitchArgs* a, int grid, hipStream_t s) { hipLaunchKernelGGL(k_stitch, dim3(grid), dim3(64), 0, s, *a); }
