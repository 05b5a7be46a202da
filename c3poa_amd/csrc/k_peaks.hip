// k_peaks.hip -- K2: Savitzky-Golay smoothing x3, median gate, find_peaks, shift/clip, subread split.
//
// Replaces, per read (paths relative to /root/reference):
//   bin/call_peaks.py:8-16      call_peaks(scores, min_dist, 3, 41, 2)
//   bin/savitzky_golay.py:17-38 (float64 FIR, mirrored-difference padding)
//   scipy.signal.find_peaks(x, distance=, height=)  (strict local maxima, plateau midpoint,
//                               inclusive height, highest-first distance suppression)
//   C3POa.py:106-108,127-155    rounding, peak shift/clip, 0.8-1.2x median subread filter
// Bit-exact (fp64 included) with oracle/c3o_signal.c: same coefficients (computed on the host
// with the same expression), same fma order.
//
// Mapping: one 256-thread workgroup per read; the track is smoothed through two fp64 scratch
// rows owned by the workgroup (L2-resident), the median is an 8-pass radix select over
// order-preserving 64-bit keys with an LDS histogram, peak suppression is an argmax loop.
#include "c3_dev.h"
#include "c3_args.h"

#define PK_T 256
#ifndef SG_T
#define SG_T 896         /* outputs per LDS tile: 22 592 bytes of LDS with the rest = 18 granules of 1 280, seven workgroups per CU (1 024: six) */
#endif
#ifndef C3_PK_WAVES
#define C3_PK_WAVES 7     /* waves per SIMD the register allocation is held to (<= 72 VGPRs; the compiler's own choice is 82 = five workgroups per CU) */
#endif
#ifndef PK_CR
#define PK_CR 2          /* candidates per thread that the suppression loop keeps in registers (more candidates: the loop through memory) */
#endif
#define SG_H 64          /* largest halo (iters * half) the tiled path supports; 3 * 20 = 60 for the reference's settings */


__device__ __forceinline__ double sg_pad(const double* y, int n, int half, int x) {
  // bin/savitzky_golay.py:33-35
  if (x < half) return y[0] - fabs(y[half - x] - y[0]);
  if (x >= n + half) return y[n - 1] + fabs(y[n - 2 - (x - n - half)] - y[n - 1]);
  return y[x - half];
}

__device__ __forceinline__ uint64_t dkey(double v) {
  uint64_t u = (uint64_t)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(uint64_t k) {
  uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

// exclusive prefix sum of one value per thread over the 256-thread workgroup; *total receives the sum.  sh4: 8 ints of LDS
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, int* sh4, unsigned* total) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned inc = (unsigned)wave_scan_add((int)v);
  if (lane == 63) sh4[wv] = (int)inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < PK_T / 64; ++w) { const unsigned t = (unsigned)sh4[w]; if (w < wv) base += t; tot += t; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// k-th smallest (0-based) of x[0..n): 8-bit radix select, all threads return the key.  The histogram of a pass is built with
// wave-aggregated LDS atomics for the bins that dominate a wave (in the leading passes every key falls into one or two bins:
// 64 lanes hammering one LDS word serialise), and the bin holding the k-th key is found with a workgroup prefix sum of the
// 256 bins instead of a serial walk by one thread.
__device__ uint64_t block_select(const double* x, int n, int k, unsigned* hist, int* sh_i, uint64_t* keys /* [PK_T] */) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint64_t prefix = 0, mask = 0;
  for (int pass = 7; pass >= 0; --pass) {
    const int shift = pass * 8;
    hist[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 4 * PK_T) {
      // four independent loads per thread first (one memory round trip for four keys), then the histogram updates
      double xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * PK_T + tid; xv[u] = i < n ? x[i] : 0.0; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * PK_T + tid;
        const uint64_t key = dkey(xv[u]);
        const bool match = i < n && (key & mask) == prefix;
        const unsigned bin = (unsigned)(key >> shift) & 255u;
        unsigned long long todo = __ballot(match);
        for (int it = 0; it < 2 && todo; ++it) {               // peel the bins of the first two lanes still waiting
          const int l = __builtin_ctzll(todo);
          const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, l);
          const unsigned long long m = __ballot(match && bin == b) & todo;
          if (lane == l) atomicAdd(&hist[b], (unsigned)__popcll(m));
          todo &= ~m;
        }
        if (match && ((todo >> lane) & 1ull)) atomicAdd(&hist[bin], 1u);
      }
    }
    __syncthreads();
    {
      const unsigned c = hist[tid];
      unsigned tot;
      const unsigned cum = block_excl_scan(c, sh_i + 2, &tot);
      // the bin of the k-th key: cum <= k < cum + c; if k is beyond the total (cannot happen) the last bin as before
      if (cum <= (unsigned)k && (unsigned)k < cum + c) { sh_i[0] = tid; sh_i[1] = (int)cum; sh_i[6] = (int)c; }
      if (tid == PK_T - 1 && (unsigned)k >= tot) { sh_i[0] = 255; sh_i[1] = (int)tot; sh_i[6] = INT32_MAX; }
    }
    __syncthreads();
    prefix |= (uint64_t)sh_i[0] << shift;
    mask |= (uint64_t)0xff << shift;
    k -= sh_i[1];
    const int cb = sh_i[6];
    __syncthreads();
    // The bin of the k-th key rarely holds more than a few dozen keys after the third pass (sign + exponent + the top mantissa
    // bits): as soon as they fit one key per thread they are gathered into LDS and ranked there -- one more sweep over the
    // track instead of up to five (the select was a quarter of k_peaks).  The result is a VALUE: the gather order cannot matter.
    if (pass > 0 && cb <= PK_T) {
      if (tid == 0) sh_i[7] = 0;
      __syncthreads();
      for (int i0 = 0; i0 < n; i0 += 4 * PK_T) {
        double xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * PK_T + tid; xv[u] = i < n ? x[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * PK_T + tid;
          const uint64_t key = dkey(xv[u]);
          if (i < n && (key & mask) == prefix) keys[atomicAdd(&sh_i[7], 1)] = key;
        }
      }
      __syncthreads();
      const uint64_t mine = tid < cb ? keys[tid] : ~0ull;
      int rank = 0;
      for (int j = 0; j < cb; ++j) { const uint64_t o = keys[j]; rank += (o < mine) || (o == mine && j < tid); }
      if (tid < cb && rank == k) *(uint64_t*)hist = mine;
      __syncthreads();
      const uint64_t r = *(const uint64_t*)hist;
      __syncthreads();
      return r;
    }
  }
  return prefix;
}

// C3POa.py:106-108 (Python round = half to even; exact in integers)
__device__ __forceinline__ int c3_rounding(int x, int base) {
  int q = x / base, r = x % base;
  if (2 * r < base) return q * base;
  if (2 * r > base) return (q + 1) * base;
  return ((q & 1) ? q + 1 : q) * base;
}

__attribute__((amdgpu_waves_per_eu(C3_PK_WAVES, C3_PK_WAVES)))
__global__ __launch_bounds__(PK_T) void k_peaks(PeaksArgs a) {
  __shared__ __attribute__((aligned(8))) unsigned hist[256];
  __shared__ int sh_i[8];
  __shared__ double sh_d[PK_T];
  __shared__ int sh_idx[PK_T];
  __shared__ int sh_cnt[PK_T + 1];
  __shared__ int kept[C3_MAX_PEAKS + 1];
  __shared__ double tb0[SG_T + 2 * SG_H], tb1[SG_T + 2 * SG_H];
  const int tid = threadIdx.x;
  double* A = a.bufA + (size_t)blockIdx.x * a.maxL;
  double* B = a.bufB + (size_t)blockIdx.x * a.maxL;
  int32_t* cand = a.cand + (size_t)blockIdx.x * (a.maxL / 2 + 2);
  uint8_t* cst = a.cstate + (size_t)blockIdx.x * (a.maxL / 2 + 2);
  const int half = (a.window - 1) / 2;

  // reads come off a queue: the grid holds exactly the workgroups that are resident at once (c3k_peaks_blocks_per_cu), so no
  // CU waits for a second, thinner round of blocks and long reads do not pile up on one block (a static stride over
  // 8 blocks per CU, of which 6 fit, cost a fifth of the kernel)
  __shared__ int s_rid;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_rid = atomicAdd(a.queue, 1);
    __syncthreads();
    const int rid = s_rid;
    if (rid >= a.b.n) break;
    C3Info* info = &a.info[rid];
    if (info->status == C3_ST_NOT_ASSIGNED) { if (tid == 0) a.n_raw[rid] = 0; continue; }
    const int64_t off = a.b.off[rid];
    const int n = (int)(a.b.off[rid + 1] - off);
    if (n < half + 1 || n < 2) { if (tid == 0) { info->status = C3_ST_TOO_SHORT; a.n_raw[rid] = 0; } continue; }
    const int32_t* tr = a.track + off;
    // ---- smoothing passes.  Tiled through LDS: a tile of SG_T outputs plus a halo of iters*half points on each side is
    // loaded once, all passes run LDS -> LDS (each pass needs half fewer halo points than the one before), and only the
    // last pass writes the smoothed track to memory.  Per point the arithmetic (and its order) is the one of the
    // untiled loop below, so the result is bit-identical.  Read ends: the mirrored-difference padding
    // (bin/savitzky_golay.py:33-35) is evaluated on the fly from that pass's own input, which the first / last tile holds.
    double* src = A; double* dst = B;
    const int H = a.iters * half;
    if (a.iters >= 1 && H <= SG_H) {
      double* xout = (a.iters & 1) ? B : A;
      for (int t0 = 0; t0 < n; t0 += SG_T) {
        const int base = t0 - H;                                   // global index of LDS slot 0
        const int T = min(SG_T, n - t0);
        for (int x = tid; x < T + 2 * H; x += PK_T) { const int g = base + x; if (g >= 0 && g < n) tb0[x] = (double)tr[g]; }
        __syncthreads();
        double* sb = tb0; double* db = tb1;
        for (int it = 0; it < a.iters; ++it) {
          const int ext = (a.iters - 1 - it) * half;               // outputs still needed beyond the core
          const int glo = max(0, t0 - ext), ghi = min(n, t0 + T + ext);
          const bool last = it == a.iters - 1;
          const double* yv = sb - base;                            // yv[g] = input of this pass at global index g
          for (int g = glo + tid; g < ghi; g += PK_T) {
            double acc = 0.0;
            if (g >= half && g + half < n) {
              const double* y = yv + (g - half);
              if (half == 20) {
#pragma unroll
                for (int k = 0; k < 20; ++k) acc = __builtin_fma(a.coef[k], y[k] + y[40 - k], acc);
                acc = __builtin_fma(a.coef[20], y[20], acc);
              } else {
                for (int k = 0; k < half; ++k) acc = __builtin_fma(a.coef[k], y[k] + y[2 * half - k], acc);
                acc = __builtin_fma(a.coef[half], y[half], acc);
              }
            } else {
              for (int k = 0; k < half; ++k)
                acc = __builtin_fma(a.coef[k], sg_pad(yv, n, half, g + k) + sg_pad(yv, n, half, g + 2 * half - k), acc);
              acc = __builtin_fma(a.coef[half], sg_pad(yv, n, half, g + half), acc);
            }
            if (last) xout[g] = acc; else db[g - base] = acc;
          }
          __syncthreads();
          double* t = sb; sb = db; db = t;
        }
      }
      src = xout;
      __syncthreads();
    } else {
    for (int i = tid; i < n; i += PK_T) A[i] = (double)tr[i];
    __syncthreads();
    for (int it = 0; it < a.iters; ++it) {
      for (int i = tid; i < n; i += PK_T) {
        double acc = 0.0;
        if (i >= half && i + half < n) {
          const double* y = src + (i - half);
          for (int k = 0; k < half; ++k) acc = __builtin_fma(a.coef[k], y[k] + y[2 * half - k], acc);
          acc = __builtin_fma(a.coef[half], y[half], acc);
        } else {
          for (int k = 0; k < half; ++k)
            acc = __builtin_fma(a.coef[k], sg_pad(src, n, half, i + k) + sg_pad(src, n, half, i + 2 * half - k), acc);
          acc = __builtin_fma(a.coef[half], sg_pad(src, n, half, i + half), acc);
        }
        dst[i] = acc;
      }
      __syncthreads();
      double* t = src; src = dst; dst = t;
    }
    }
    const double* x = src;    // smoothed track
    // ---- np.median
    double med;
    {
#ifdef C3_EXP_X2_SEL
      { volatile uint64_t sink_ = block_select(x, n, (n - 1) / 3, hist, sh_i, (uint64_t*)sh_d); (void)sink_; }
#endif
      uint64_t k1 = block_select(x, n, (n - 1) / 2, hist, sh_i, (uint64_t*)sh_d);
      double v1 = dunkey(k1);
      if (n & 1) med = v1;
      else {
        // next order statistic: v1 again if duplicated, else the smallest value above it
        int cnt_le = 0; uint64_t mn = ~0ull;
        for (int i = tid; i < n; i += PK_T) {
          uint64_t key = dkey(x[i]);
          if (key <= k1) ++cnt_le; else if (key < mn) mn = key;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          cnt_le += __shfl_xor(cnt_le, o);
          const unsigned long long om = __shfl_xor((unsigned long long)mn, o);
          if (om < mn) mn = om;
        }
        if ((tid & 63) == 0) { sh_cnt[tid >> 6] = cnt_le; sh_d[tid >> 6] = __longlong_as_double((long long)mn); }
        __syncthreads();
        int tot_le = 0; uint64_t mnk = ~0ull;
#pragma unroll
        for (int q = 0; q < PK_T / 64; ++q) { tot_le += sh_cnt[q]; const uint64_t o = (uint64_t)__double_as_longlong(sh_d[q]); if (o < mnk) mnk = o; }
        __syncthreads();
        double v2 = (tot_le > n / 2) ? v1 : dunkey(mnk);
        med = (v1 + v2) / 2.0;
      }
    }
    // ---- max (wave shuffles, then the four wave results: two barriers instead of nine)
    double mx = -1.0e308;
    for (int i = tid; i < n; i += PK_T) mx = fmax(mx, x[i]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) sh_d[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(sh_d[0], sh_d[1]), fmax(sh_d[2], sh_d[3]));
    __syncthreads();
    int n_kept = 0;
    if (!(mx < 6 * med)) {
      const double height = med * 3;
      // ---- strict local maxima (plateau midpoint) with inclusive height, order preserving.  Every wave takes a contiguous quarter of the
      // track in strips of 64 consecutive points (lane = point: the three loads of a strip are coalesced; a thread walking its own 20
      // consecutive points made every load instruction touch 64 cache lines), counts its candidates with ballots, and after the four
      // totals are known writes them in order.
      const int lane = tid & 63, wv = tid >> 6;
      const int seg = ((n + 4 * 64 - 1) / (4 * 64)) * 64;
      const int wlo = wv * seg, whi = min(n - 1, wlo + seg);
      int wcnt = 0;
      for (int pass = 0; pass < 2; ++pass) {
        int w = (pass == 1) ? sh_cnt[wv] : 0;
        for (int b = wlo; b < whi; b += 64) {
          const int i = b + lane;
          bool is = false; int mid = 0;
          if (i >= 1 && i < whi) {
            const double xi = x[i];
            if (x[i - 1] < xi) {
              int ia = i + 1;
              while (ia < n - 1 && x[ia] == xi) ++ia;
              if (x[ia] < xi && xi >= height) { is = true; mid = (i + ia - 1) / 2; }
            }
          }
          const unsigned long long m = __ballot(is);
          if (pass == 1 && is) { const int k = w + __popcll(m & ((1ull << lane) - 1ull)); cand[k] = mid; cst[k] = 1; }
          w += __popcll(m);
        }
        if (pass == 0) {
          wcnt = w;
          if (lane == 0) sh_i[2 + wv] = wcnt;
          __syncthreads();
          if (tid == 0) { int acc = 0; for (int q = 0; q < PK_T / 64; ++q) { sh_cnt[q] = acc; acc += sh_i[2 + q]; } sh_cnt[PK_T] = acc; }
          __syncthreads();
        }
      }
      __syncthreads();
      const int nc = sh_cnt[PK_T];
      __syncthreads();
      // ---- _select_by_peak_distance: highest first, equal priority -> later index first
      const int dist = a.min_dist < 1 ? 1 : a.min_dist;
      if (nc <= PK_T * PK_CR) {
        // the usual read: every candidate (position, value) lives in a register of one thread for the whole loop -- no memory access in it; the
        // winner of a round is found by wave shuffles and one barrier (the four wave results alternate between two LDS rows).  Candidate
        // positions are distinct and ascending in the candidate index, so "equal value: the later index" is "the larger position".
        double cv[PK_CR]; int cp[PK_CR];
#pragma unroll
        for (int u = 0; u < PK_CR; ++u) { const int c = tid + u * PK_T; cp[u] = c < nc ? cand[c] : -1; cv[u] = c < nc ? x[max(cp[u], 0)] : -1.0e308; }
        for (int round = 0;; ++round) {
          double bv = -1.0e308; int bp = -1;
#pragma unroll
          for (int u = 0; u < PK_CR; ++u) if (cp[u] >= 0 && (bp < 0 || cv[u] > bv || (cv[u] == bv && cp[u] > bp))) { bv = cv[u]; bp = cp[u]; }
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) {
            const double ov = __shfl_xor(bv, o); const int op = __shfl_xor(bp, o);
            if (op >= 0 && (bp < 0 || ov > bv || (ov == bv && op > bp))) { bv = ov; bp = op; }
          }
          const int row = (round & 1) * 4;
          if (lane == 0) { sh_d[row + wv] = bv; sh_idx[row + wv] = bp; }
          __syncthreads();
          bv = sh_d[row]; bp = sh_idx[row];
#pragma unroll
          for (int q = 1; q < PK_T / 64; ++q) { const double ov = sh_d[row + q]; const int op = sh_idx[row + q]; if (op >= 0 && (bp < 0 || ov > bv || (ov == bv && op > bp))) { bv = ov; bp = op; } }
          if (bp < 0) break;
          if (tid == 0 && n_kept < C3_MAX_PEAKS) kept[n_kept] = bp;
          ++n_kept;
#pragma unroll
          for (int u = 0; u < PK_CR; ++u) { int dlt = cp[u] - bp; if (dlt < 0) dlt = -dlt; if (cp[u] >= 0 && dlt < dist) cp[u] = -1; }      // (the winner itself: distance 0)
        }
        __syncthreads();
      } else
      for (;;) {
        double bv = -1.0e308; int bi = -1;
        for (int c = tid; c < nc; c += PK_T)
          if (cst[c] == 1) { double v = x[cand[c]]; if (v > bv || (v == bv && c > bi)) { bv = v; bi = c; } }
        sh_d[tid] = bv; sh_idx[tid] = bi;
        __syncthreads();
        for (int s = PK_T / 2; s > 0; s >>= 1) {
          if (tid < s) {
            double ov = sh_d[tid + s]; int oi = sh_idx[tid + s];
            if (oi >= 0 && (sh_idx[tid] < 0 || ov > sh_d[tid] || (ov == sh_d[tid] && oi > sh_idx[tid]))) { sh_d[tid] = ov; sh_idx[tid] = oi; }
          }
          __syncthreads();
        }
        const int win = sh_idx[0];
        __syncthreads();
        if (win < 0) break;
        const int wp = cand[win];
        if (tid == 0) { cst[win] = 2; if (n_kept < C3_MAX_PEAKS) kept[n_kept] = wp; }
        ++n_kept;
        for (int c = tid; c < nc; c += PK_T)
          if (cst[c] == 1) { int dlt = cand[c] - wp; if (dlt < 0) dlt = -dlt; if (dlt < dist) cst[c] = 0; }
        __syncthreads();
      }
    }
    // ---- thread 0: order the kept peaks, shift, clip, split (C3POa.py:127-155)
    if (tid == 0) {
      int status = C3_ST_OK;
      if (n_kept > C3_MAX_PEAKS - 1) { status = C3_ST_LIMIT; n_kept = 0; }
      for (int i = 1; i < n_kept; ++i) { int v = kept[i], k = i - 1; while (k >= 0 && kept[k] > v) { kept[k + 1] = kept[k]; --k; } kept[k + 1] = v; }
      a.n_raw[rid] = n_kept;
      for (int i = 0; i < n_kept; ++i) a.raw_peaks[(size_t)rid * C3_MAX_PEAKS + i] = kept[i];
      const int S = a.sp_len[a.b.splint_id[rid]];
      int np = 0;
      for (int i = 0; i < n_kept; ++i) { int p = kept[i] + S / 2; if (p < n) info->peaks[np++] = p; }
      info->n_peaks = np;
      int ns = 0, hf = 0, ht = 0, fe = 0, tb = 0;
      if (status == C3_ST_OK && np == 0) status = C3_ST_NO_PEAKS;
      if (np > 1) {
        const int nl = np - 1;
        int* r = sh_idx;                       // reuse LDS: nl <= 255
        int* srt = sh_cnt;
        for (int i = 0; i < nl; ++i) srt[i] = r[i] = c3_rounding(info->peaks[i + 1] - info->peaks[i], 50);
        for (int i = 1; i < nl; ++i) { int v = srt[i], k = i - 1; while (k >= 0 && srt[k] > v) { srt[k + 1] = srt[k]; --k; } srt[k + 1] = v; }
        const double med2 = (nl & 1) ? (double)srt[nl / 2] : ((double)srt[nl / 2 - 1] + (double)srt[nl / 2]) / 2.0;
        const double lo8 = med2 * 0.8, hi12 = med2 * 1.2;
        for (int i = 0; i < nl; ++i)
          if (lo8 <= (double)r[i] && (double)r[i] <= hi12) { info->sub_beg[ns] = info->peaks[i]; info->sub_end[ns] = info->peaks[i + 1]; ++ns; }
        if (info->peaks[0] > 100) { hf = 1; fe = info->peaks[0]; }
        if (n - info->peaks[np - 1] > 100) { ht = 1; tb = info->peaks[np - 1]; }
      } else if (np == 1) {
        hf = 1; fe = info->peaks[0]; ht = 1; tb = info->peaks[0];
      }
      info->n_sub = ns; info->has_front = hf; info->has_tail = ht; info->front_end = fe; info->tail_beg = tb;
      if (status == C3_ST_OK && ns == 0) status = C3_ST_NO_CONSENSUS;   // repeats == 0 (zero-repeat rescue: DESIGN.md 6)
      if (status == C3_ST_OK && ns > C3_MAX_SUB) status = C3_ST_LIMIT;
      info->status = status;
    }
    __syncthreads();
  }
}

extern "C" int c3k_peaks_blocks_per_cu(void) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_peaks, PK_T, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 4; }
  return nb;
}
extern "C" void c3k_launch_peaks(const PeaksArgs* a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL(k_peaks, dim3(grid), dim3(PK_T), 0, stream, *a);
}
