// k_zero.hip -- zero-repeat rescue: overlap of the two dangling pieces of a read with a single splint.
//
// Replaces bin/determine_consensus.py:106-136 (zero_repeats; mappy overlap + 2-row abPOA + pairwise merge).
// Spec: DESIGN.md 4.7, restated by oracle/c3o_zero.c (bit-exact).  k_zero finds the best forward local
// alignment of d1 = read[tail_beg:] (rows) against d0 = read[:front_end] (columns) with affine gaps and
// turns the read into a 2-"subread" POA job (the two overlap slices); k_zero_finish stitches
// d1[:q_st] + overlap consensus + d0[r_en:] after k_poa.  Rare path: one wave per read, previous
// H/E row in LDS (front pieces of up to ZW columns; longer ones keep the two rows in global memory behind the
// block's direction bytes), direction bytes in global memory, scalar traceback.
#include "c3_dev.h"
#include "c3_args.h"

#define ZW 4096           // columns kept in the LDS row buffers
#define WSYNC() __syncthreads()

__global__ __launch_bounds__(64) void k_zero(ZeroArgs a) {
  __shared__ int Hlds[ZW + 1];
  __shared__ int Elds[ZW + 1];
  const int lane = wave_lane();
  const int go = a.p.zr_gapo, ge = a.p.zr_gape, ma = a.p.zr_match, mb = -a.p.zr_mismatch;
  const int NEGZ = INT32_MIN / 2;
  for (int wi = blockIdx.x; wi < a.n_work; wi += gridDim.x) {
    const int rid = a.work[wi];
    C3Info* info = &a.info[rid];
    const int64_t off = a.b.off[rid];
    const int L = (int)(a.b.off[rid + 1] - off);
    const uint32_t* pk = a.b.pk + a.b.woff[rid];
    const int n0 = info->front_end, t0 = info->tail_beg, n1 = L - t0;
    uint8_t* D = a.D + (size_t)blockIdx.x * a.dstride;
    int4 z; z.x = z.y = z.z = z.w = 0;
    bool ok = n0 > 0 && n1 > 0 && (n0 <= ZW || n0 <= a.rowcap) && (long long)n0 * n1 <= a.p.zr_max_cells &&
              (long long)(n0 + 1) * (n1 + 1) <= a.dcap;
    if (ok) {
      const int W = n0 + 1;
      int* Hrow = Hlds; int* Erow = Elds;          // (generic pointers: flat loads serve both homes)
      if (n0 > ZW) { Hrow = (int*)(D + a.dcap); Erow = Hrow + a.rowcap + 1; }
      for (int j = lane; j <= n0; j += 64) { Hrow[j] = 0; Erow[j] = NEGZ; }
      int best = 0, bi = 0, bj = 0;
      for (int i = 1; i <= n1; ++i) {
        const int qc = c3_code_at(pk, t0 + i - 1);
        int carry_old = 0;            // H[i-1][c0]   (column left of the chunk; column 0 is always 0)
        int carry_f = NEGZ;           // max over previous chunks of Ht[k] + ge*k
        int carry_h = 0;              // H[i][c0]      (full H of the column left of the chunk)
        for (int c0 = 0; c0 < n0; c0 += 64) {
          const int j = c0 + lane + 1;
          const bool act = j <= n0;
          const int hpj = act ? Hrow[j] : 0;
          const int epj = act ? Erow[j] : NEGZ;
          const int hpm = wave_shr1(hpj, carry_old);
          carry_old = wave_bcast(hpj, 63);
          const int eo = hpj - go - ge, ee = epj - ge;
          const int ex = ee > eo;
          const int e = ex ? ee : eo;
          const int rc = act ? c3_code_at(pk, j - 1) : 0;
          const int dg = hpm + (qc == rc ? ma : mb);
          int ht = 0, src = 0;
          if (dg > ht) { ht = dg; src = 1; }
          if (e > ht) { ht = e; src = 2; }
          // F[j] = max_{k<j} Ht[k] - go - ge*(j-k)   (gap opened after an F cell is dominated)
          const int x = act ? ht + ge * j : NEGZ;
          const int sc = wave_scan_max(x);
          const int px = max(wave_shr1(sc, NEGZ), carry_f);
          // column 0 of the row (H = 0) can also open a gap
          const int f = max(px, 0 + ge * 0) - go - ge * j;
          carry_f = max(carry_f, wave_bcast(sc, 63));
          int h = ht;
          if (f > h) { h = f; src = 3; }
          // F extended?  oracle: fe > fo with fo = H[i][j-1] - go - ge  <=>  F != fo
          const int hleft = wave_shr1(act ? h : 0, carry_h);
          carry_h = wave_bcast(act ? h : 0, 63);
          const int fx = f != hleft - go - ge;
          if (act) {
            Hrow[j] = h; Erow[j] = e;
            D[(size_t)i * W + j] = (uint8_t)(src | (ex << 2) | (fx << 3));
            if (h > best) { best = h; bi = i; bj = j; }
          }
        }
      }
      WSYNC();
      const int gb = wave_max(best);
      const int gi = wave_min(best == gb ? bi : INT32_MAX / 2);
      const int gj = wave_min((best == gb && bi == gi) ? bj : INT32_MAX / 2);
      if (gb >= a.p.zr_min_score) {
        int i = gi, j = gj, st = 0;
        for (;;) {
          if (i == 0 || j == 0) break;                 // border cells are 0 and never stored
          const int d = D[(size_t)i * W + j];                 // border cells are 0 and never stored
          if (st == 0) { const int src = d & 3; if (src == 0) break; if (src == 1) { --i; --j; } else st = src; }
          else if (st == 2) { st = (d & 4) ? 2 : 0; --i; }
          else { st = (d & 8) ? 3 : 0; --j; }
        }
        z.x = j; z.y = gj; z.z = i; z.w = gi;           // r_st, r_en, q_st, q_en
        ok = gj > j && gi > i;
      } else ok = false;
      if (lane == 0) atomicAdd((unsigned long long*)(a.counter + 2), (unsigned long long)n0 * n1);
    }
    if (lane == 0) {
      a.zinfo[rid] = z;
      a.zflag[rid] = ok ? 1 : 0;
      if (ok) {       // hand the two overlap slices to k_poa as a 2-subread job
        info->n_sub = 2; info->status = C3_ST_OK;
        info->sub_beg[0] = z.x; info->sub_end[0] = z.y;
        info->sub_beg[1] = t0 + z.z; info->sub_end[1] = t0 + z.w;
      }
    }
    WSYNC();
  }
}

// after k_poa: consensus = d1[:q_st] + overlap consensus + d0[r_en:]; accepted when >= mdistcutoff
__global__ __launch_bounds__(64) void k_zero_finish(ZeroArgs a) {
  const int lane = wave_lane();
  for (int wi = blockIdx.x; wi < a.n_work; wi += gridDim.x) {
    const int rid = a.work[wi];
    if (!a.zflag[rid]) continue;
    C3Info* info = &a.info[rid];
    const int64_t off = a.b.off[rid];
    const uint32_t* pk = a.b.pk + a.b.woff[rid];
    const int4 z = a.zinfo[rid];
    const int n0 = info->front_end, t0 = info->tail_beg;
    const int C = info->draft_len;
    const uint8_t* draft = a.draft + off;
    char* cons = a.cons + off;
    const int total = z.z + C + (n0 - z.y);
    const bool good = info->status == C3_ST_OK && C > 0 && total >= a.p.mdist;
    if (good) {
      for (int k = lane; k < z.z; k += 64) cons[k] = "ACGT"[c3_code_at(pk, t0 + k)];
      for (int k = lane; k < C; k += 64) cons[z.z + k] = "ACGT"[draft[k] & 3];
      for (int k = lane; k < n0 - z.y; k += 64) cons[z.z + C + k] = "ACGT"[c3_code_at(pk, z.y + k)];
    }
    if (lane == 0) {
      info->n_sub = 0;                       // repeats == 0 (determine_consensus.py:18)
      info->draft_len = 0;                   // no polish on this path (:16-18)
      info->cons_len = good ? total : 0;
      info->status = good ? C3_ST_OK : C3_ST_NO_CONSENSUS;
      if (!good) a.zflag[rid] = 0;
    }
  }
}

extern "C" void c3k_launch_zero(const ZeroArgs* a, int grid, hipStream_t s) { hipLaunchKernelGGL(k_zero, dim3(grid), dim3(64), 0, s, *a); }
extern "C" void c3k_launch_zero_finish(const ZeroArgs* a, int grid, hipStream_t s) { hipLaunchKernelGGL(k_zero_finish, dim3(grid), dim3(64), 0, s, *a); }
