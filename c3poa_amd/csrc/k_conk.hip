// k_conk.hip -- K1: splint x read local-alignment score track, summed per diagonal.
//
// Replaces conk.conk(splint, seq, 20) at /root/reference/C3POa.py:123 (conk is an un-vendored
// Cython dependency; spec frozen in DESIGN.md 4.1 and restated by oracle/c3o_signal.c:c3o_conk).
//
// Mapping (MI355X-first, no atomics; LDS only as a per-lane score table): ONE WAVE PER READ.  Lane l owns R consecutive
// splint rows (64*R >= S; padding rows sit on top and can never score), and at step t works on
// read column j = t - l, so the 64 lanes form a systolic anti-diagonal.  The only cross-lane
// traffic is two DPP wave_shr:1 moves per step: the bottom-row H of the lane above, and the
// running diagonal sum.  A diagonal's partial sum rides down the lanes as a token (one hop
// every R+1 steps) and leaves lane 63 complete, where it is stored exactly once: the track is
// written with plain stores, never read-modified (16 diagonals per block of 16 steps, under one predicate).  Read bases
// arrive 2-bit packed, 16 per dword, re-aligned once per 16 steps with v_alignbit.
#include "c3_dev.h"
#include "c3_args.h"


// a - b over unsigned 16-bit values, saturating at zero (result zero-extended)
__device__ __forceinline__ int subsat_u16(int a, int b) { int d; asm("v_sub_u16_e64 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }

// One anti-diagonal step of the R cells a lane owns.  The kernel is bound by vector issue and nothing else (102 % of the issue
// cycles, profiles/r05_sq_counters_*), so whatever can leave the vector ALU does: the substitution score of a cell is a signed
// BYTE READ FROM LDS -- the lane's table row k holds the four scores of splint row k as one dword (byte r = score against read
// base r), the address is the lane's table + the read base of this step, the row is the instruction's immediate offset -- issued
// on the LDS port beside the other waves' vector instructions (it was a v_bfe_i32 per cell: 4.2 issue cycles of 15.8).
// trc = this lane's table + read base.
template <int R, bool CHECK>
__device__ __forceinline__ void conk_step(const signed char* trc, bool kill, int (&hprev)[R], int (&P)[R], int& W,
                                          int& up_prev, int penalty) {
  int sc[R];
#pragma unroll
  for (int k = 0; k < R; ++k) sc[k] = trc[k * 256];
  int up = wave_shr1z(hprev[R - 1]);
  int recv = wave_shr1z(W);
  int u = up, d = up_prev;
  up_prev = up;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    // H <= match * splint length fits 16 bits: v_max_i16 issues at twice the rate of v_max_i32 / v_max3 (tools/ubench/valu_cost.hip);
    // its result is zero-extended, so the 32-bit diagonal sums below add clean values.
    // The gap candidate leaves its subtraction saturated at zero (v_sub_u16 with the clamp bit, same issue rate as the plain one), so
    // the cell's own "max with 0" is gone: max(d + s, m, 0) == max(d + s, max(m, 0)).  Five vector instructions per cell (seven in round 4).
    int m = subsat_u16(max16(u, hprev[k]), penalty);
    int hh = max16(d + sc[k], m);
    if (CHECK) hh = kill ? 0 : hh;
    d = hprev[k];
    hprev[k] = hh;
    u = hh;
  }
  int wn = P[R - 1];
#pragma unroll
  for (int k = R - 1; k >= 1; --k) P[k] = P[k - 1] + hprev[k];
  P[0] = recv + hprev[0];
  W = wn;
}

// SCAN = false: the track of every read against ITS splint/strand is written (the hot path).
// SCAN = true : work item = (read, splint, strand) for every splint on both strands; only max, argmax and
//               sum of the track are kept -- the splint/strand finder that replaces blat (preprocess.py).
template <int R, bool SCAN>
__global__ __launch_bounds__(256) void k_conk(ConkArgs a) {
  const int lane = wave_lane();
  const int n_items = SCAN ? a.b.n * a.n_spl * 2 : a.b.n;
  // score tables of the four waves of the block: [wave][row k][lane] dwords.  A lane only ever reads the dwords it wrote itself, and a
  // wave's LDS operations complete in order: no barrier anywhere (lanes l and l + 32 share a bank and never the same half-wave pass)
  __shared__ unsigned score_tab[4 * R * 64];
  unsigned* const T = score_tab + (threadIdx.x >> 6) * (R * 64);
  const signed char* const tbase = (const signed char*)(T + lane);
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(a.counter, 1);
    item = wave_first(item);
    if (item >= n_items) break;
    const int rid = SCAN ? item / (a.n_spl * 2) : item;
    const int64_t off = a.b.off[rid];
    const int L = (int)(a.b.off[rid + 1] - off);
    const int st = SCAN ? ((item & 1) ? '-' : '+') : a.b.strand[rid];
    // (stored by every lane: a divergent `if (lane == 0)` right before `continue` can livelock the
    // persistent loop -- see prep_one in k_polish.hip)
    if (!SCAN && st != '+' && st != '-') { a.info[rid].status = C3_ST_NOT_ASSIGNED; continue; }
    if (!SCAN) a.info[rid].status = C3_ST_OK;          // a re-run after c3_batch_assign must not see a stale NOT_ASSIGNED
    const int sid = SCAN ? (item >> 1) % a.n_spl : a.b.splint_id[rid];
    const int S = a.sp_len[sid];
    const uint8_t* sp = a.sp_codes + ((size_t)sid * 2 + (st == '-')) * C3_SPLINT_MAX;
    const uint32_t* pk = a.b.pk + a.b.woff[rid];
    const int last_word = L > 0 ? (L - 1) >> 4 : 0;
    int32_t* track = a.track + off;
    const int pad = 64 * R - S;
    const int mm4 = (a.mismatch & 255) * 0x01010101;
    int hprev[R], P[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      int i = lane * R + k - pad;
      int code = (i >= 0) ? (int)sp[i] : 5;              // padding rows (and splint N) never match
      T[k * 64 + lane] = code < 4 ? (mm4 & ~(255 << (8 * code))) | ((a.match & 255) << (8 * code)) : mm4;
      hprev[k] = 0; P[k] = 0;
    }
    int W = 0, up_prev = 0;
    int pen = a.penalty;
    asm volatile("" : "+v"(pen));                      // a VALU instruction with an SGPR operand issues at half rate
    int smax = -1, sarg = 0; long long ssum = 0;          // SCAN accumulators (meaningful in lane 63)
    // lane 63 finishes diagonal d = t - c0 at step t
    const int c0 = 63 * (R + 1) + (R - 1) - pad;
    const int T_end = L + c0;                 // last useful step is L-1+c0
    for (int t0 = 0; t0 < T_end; t0 += 16) {
      const bool fast = (t0 >= 63) && (t0 + 15 < L);
      // 16 read bases of this lane's columns jb .. jb+15 (word indices clamped: columns outside the read are killed below)
      const int jb = t0 - lane;
      const int wi = jb >> 4;
      const uint32_t w0 = pk[min(max(wi, 0), last_word)], w1 = pk[min(max(wi + 1, 0), last_word)];
      const uint32_t x = __builtin_amdgcn_alignbit(w1, w0, (jb & 15) * 2);
      const int d0 = t0 - c0;
      int o[16];
      if (fast) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          conk_step<R, false>(tbase + ((x >> (2 * s)) & 3), false, hprev, P, W, up_prev, pen);
          o[s] = P[R - 1];
          if (SCAN) { int d = d0 + s; if (d >= 0 && d < L) { ssum += P[R - 1]; if (P[R - 1] > smax) { smax = P[R - 1]; sarg = d; } } }
        }
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const bool oob = (unsigned)(jb + s) >= (unsigned)L;
          conk_step<R, true>(tbase + ((x >> (2 * s)) & 3), oob, hprev, P, W, up_prev, pen);
          o[s] = P[R - 1];
          if (SCAN) { int d = d0 + s; if (d >= 0 && d < L) { ssum += P[R - 1]; if (P[R - 1] > smax) { smax = P[R - 1]; sarg = d; } } }
        }
      }
      // the 16 finished diagonals of this block leave lane 63 together: one predicate per block, no per-step branch
      if (!SCAN && lane == 63) {
        if (d0 >= 0 && d0 + 15 < L) {
#pragma unroll
          for (int s = 0; s < 16; ++s) track[d0 + s] = o[s];
        } else {
#pragma unroll
          for (int s = 0; s < 16; ++s) if (d0 + s >= 0 && d0 + s < L) track[d0 + s] = o[s];
        }
      }
    }
    if (SCAN) {
      // broadcast lane 63's accumulators and store from EVERY lane: a divergent `if (lane == 63)` right before
      // the back edge of the persistent loop live-locks (same hipcc pitfall as in k_prep)
      const int mean = (int)(ssum / (L > 0 ? L : 1));
      int4 v = make_int4(__builtin_amdgcn_readlane(smax, 63), __builtin_amdgcn_readlane(sarg, 63),
                         __builtin_amdgcn_readlane(mean, 63), L);
      *reinterpret_cast<int4*>(a.scan + (size_t)item * 4) = v;
    }
  }
}

extern "C" void c3k_launch_conk(const ConkArgs* a, int max_splint, int grid, int scan, hipStream_t stream) {
  int R = (max_splint + 63) / 64;
  dim3 g(grid), b(256);
#define LC(r) if (scan) hipLaunchKernelGGL((k_conk<r, true>), g, b, 0, stream, *a); else hipLaunchKernelGGL((k_conk<r, false>), g, b, 0, stream, *a); break;
  switch (R) {
    case 1: LC(1) case 2: LC(2) case 3: LC(3) case 4: LC(4) case 5: LC(5) case 6: LC(6) case 7: LC(7)
    default: LC(8)
  }
#undef LC
}
