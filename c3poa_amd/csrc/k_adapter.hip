// k_adapter.hip -- adapter finder of the post-processing step.
//
// Replaces the blat call of C3POa_postprocessing.py:229-236 (`blat -stepSize=1 -tileSize=6 -minScore=10 -minIdentity=10
// -minMatch=1 -oneOff=1 adapters reads psl`): for every consensus read, every adapter and both strands the best LOCAL
// alignment with affine gaps is found and traced back, which yields exactly the PSL fields parse_blat reads
// (:238-264): matches, qBaseInsert, strand, qStart/qEnd, tStart/tEnd.  Spec: DESIGN.md 4.9, restated bit-exactly by
// oracle/c3o_adapter.c.  Scoring = the map-ont base scoring already used for the zero-repeat overlap (2, -4, 4+2k).
//
// Mapping: work item = (read, adapter, strand), one wave per item from an atomic queue.  Rows = read bases,
// columns = adapter bases (<= 512, the splint table), previous H/E row in LDS, one direction byte per cell in global
// memory, uniform scalar traceback that also counts matches / mismatches / inserted bases.
#include "c3_dev.h"
#include "c3_args.h"

#define WSYNC() __syncthreads()

// every item stores from every lane (same values): see the live-lock note in k_polish.hip (prep_one)
__device__ __forceinline__ void ad_store(int32_t* o, int sc, int qs, int qe, int ts, int te, int ma, int mm, int qi, int ti, int qn, int tn, int L) {
  o[0] = sc; o[1] = qs; o[2] = qe; o[3] = ts; o[4] = te; o[5] = ma; o[6] = mm; o[7] = qi; o[8] = ti; o[9] = qn; o[10] = tn; o[11] = L;
}

__global__ __launch_bounds__(64) void k_adapter(AdapterArgs a) {
  __shared__ int Hrow[C3_SPLINT_MAX + 1];
  __shared__ int Erow[C3_SPLINT_MAX + 1];
  __shared__ unsigned char acode[C3_SPLINT_MAX];
  const int lane = wave_lane();
  const int go = a.p.zr_gapo, ge = a.p.zr_gape, ma = a.p.zr_match, mb = -a.p.zr_mismatch;
  const int NEGZ = INT32_MIN / 2;
  const int n_items = a.b.n * a.n_ad * 2;
  uint8_t* D = a.D + (size_t)blockIdx.x * a.dcap;
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(a.counter, 1);
    item = wave_first(item);
    if (item >= n_items) break;
    const int rid = item / (a.n_ad * 2), aid = (item >> 1) % a.n_ad, rc = item & 1;
    const int64_t off = a.b.off[rid];
    const int L = (int)(a.b.off[rid + 1] - off);
    const uint32_t* pk = a.b.pk + a.b.woff[rid];
    const int m = a.ad_len[aid];
    const uint8_t* ad = a.ad_codes + ((size_t)aid * 2 + rc) * C3_SPLINT_MAX;
    int32_t* out = a.out + (size_t)item * 12;
    const int W = m + 1;
    if (L <= 0 || m <= 0 || (long long)(L + 1) * W > a.dcap) { ad_store(out, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, L); continue; }
    for (int j = lane; j <= m; j += 64) { Hrow[j] = 0; Erow[j] = NEGZ; if (j < m) acode[j] = ad[j]; }
    WSYNC();
    int best = 0, bi = 0, bj = 0;
    for (int i = 1; i <= L; ++i) {
      const int qc = c3_code_at(pk, i - 1);
      int carry_old = 0, carry_f = NEGZ, carry_h = 0;
      for (int c0 = 0; c0 < m; c0 += 64) {
        const int j = c0 + lane + 1;
        const bool act = j <= m;
        const int hpj = act ? Hrow[j] : 0;
        const int epj = act ? Erow[j] : NEGZ;
        const int hpm = wave_shr1(hpj, carry_old);
        carry_old = wave_bcast(hpj, 63);
        const int eo = hpj - go - ge, ee = epj - ge;
        const int ex = ee > eo;
        const int e = ex ? ee : eo;
        const int rcd = act ? (int)acode[j - 1] : 0;
        const int dg = hpm + (qc == rcd ? ma : mb);
        int ht = 0, src = 0;
        if (dg > ht) { ht = dg; src = 1; }
        if (e > ht) { ht = e; src = 2; }
        const int x = act ? ht + ge * j : NEGZ;
        const int sc = wave_scan_max(x);
        const int px = max(wave_shr1(sc, NEGZ), carry_f);
        const int f = max(px, 0) - go - ge * j;
        carry_f = max(carry_f, wave_bcast(sc, 63));
        int h = ht;
        if (f > h) { h = f; src = 3; }
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
    int nma = 0, nmm = 0, qi = 0, ti = 0, qn = 0, tn = 0, i = 0, j = 0;
    if (gb > 0) {
      i = gi; j = gj;
      int st = 0;
      for (;;) {
        if (i == 0 || j == 0) break;                   // border cells are 0 and never stored
        const int d = D[(size_t)i * W + j];
        if (st == 0) {
          const int src = d & 3;
          if (src == 0) break;
          if (src == 1) { if (c3_code_at(pk, i - 1) == (int)acode[j - 1]) ++nma; else ++nmm; --i; --j; }
          else { st = src; if (src == 2) ++qn; else ++tn; }
        } else if (st == 2) { st = (d & 4) ? 2 : 0; ++qi; --i; }
        else { st = (d & 8) ? 3 : 0; ++ti; --j; }
      }
    }
    // target coordinates on the adapter's forward strand (PSL convention); query coordinates are forward already
    const int ts = rc ? m - gj : j, te = rc ? m - j : gj;
    if (gb > 0) ad_store(out, gb, i, gi, ts, te, nma, nmm, qi, ti, qn, tn, L);
    else ad_store(out, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, L);
    WSYNC();
  }
}

extern "C" void c3k_launch_adapter(const AdapterArgs* a, int grid, hipStream_t s) { hipLaunchKernelGGL(k_adapter, dim3(grid), dim3(64), 0, s, *a); }

// ---- oligo-dT index matcher (C3POa_postprocessing.py:266-285, match_index) for a whole batch of pieces -------------
// One lane per piece (the 20-mer cut next to an adapter): sliding Levenshtein distance against every index, the
// reference's quirks included (a slice that is too short for index k ends the loop over indexes at that position;
// stable order; winner needs distance < 2 and a runner-up more than 1 further away).  Same function as the host
// statement c3_match_index (c3_io.cpp), which the golden cases of the reference pin.
#define IDX_MAX 32      // longest index handled on the device
#define PIECE_W 64      // bytes per piece slot
__global__ __launch_bounds__(64) void k_match_index(const char* pieces, const int* lens, int n, int n_idx,
                                                     const char* idx_cat, const long long* idx_off, int* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const char* seq = pieces + (size_t)t * PIECE_W;
  const int L = lens[t];
  int best[16];
  for (int k = 0; k < 16; ++k) best[k] = INT32_MAX;
  for (int p = 0; p < L; ++p) {
    for (int k = 0; k < n_idx; ++k) {
      const int len = (int)(idx_off[k + 1] - idx_off[k]);
      if (p + len > L) break;
      const char* b = idx_cat + idx_off[k];
      int prev[IDX_MAX + 1], cur[IDX_MAX + 1];
      for (int j = 0; j <= len; ++j) prev[j] = j;
      for (int i = 1; i <= len; ++i) {
        cur[0] = i;
        const char ca = seq[p + i - 1];
        for (int j = 1; j <= len; ++j) {
          int v = prev[j - 1] + (ca != b[j - 1]);
          v = min(v, prev[j] + 1); v = min(v, cur[j - 1] + 1);
          cur[j] = v;
        }
        for (int j = 0; j <= len; ++j) prev[j] = cur[j];
      }
      best[k] = min(best[k], prev[len]);
    }
  }
  int i0 = -1, i1 = -1, res = -1; bool bad = false;
  for (int k = 0; k < n_idx; ++k) {
    if (best[k] == INT32_MAX) { bad = true; break; }
    if (i0 < 0 || best[k] < best[i0]) { i1 = i0; i0 = k; }
    else if (i1 < 0 || best[k] < best[i1]) i1 = k;
  }
  if (!bad && i0 >= 0 && i1 >= 0 && best[i0] < 2 && best[i1] - best[i0] > 1) res = i0;
  out[t] = res;
}

extern "C" void c3k_launch_match_index(const char* pieces, const int* lens, int n, int n_idx, const char* idx_cat,
                                       const long long* idx_off, int* out, hipStream_t s) {
  hipLaunchKernelGGL(k_match_index, dim3((n + 63) / 64), dim3(64), 0, s, pieces, lens, n, n_idx, idx_cat, idx_off, out);
}
