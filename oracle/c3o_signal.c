/*
 * c3o_signal.c -- ORACLE (test infrastructure, never shipped): splint score track,
 * Savitzky-Golay smoothing, peak calling and subread split.
 *
 * Follows (paths relative to /root/reference):
 *   C3POa.py:123               conk.conk(splint, seq, penalty=20)     [parity unpinned: conk is
 *                               an un-vendored Cython dependency, rvolden/conk @ HEAD]
 *   bin/savitzky_golay.py:17-38
 *   bin/call_peaks.py:8-16
 *   scipy/signal/_peak_finding.py find_peaks / _local_maxima_1d / _select_by_peak_distance
 *   C3POa.py:106-108 (rounding), 127-155 (shift, clip, split)
 */
#include "c3o.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "c3o_mem.h"

__thread c3o_arena c3o_tls_arena;

void c3o_default_params(c3o_params* p) {
  p->conk_match = 5; p->conk_mismatch = -4; p->conk_penalty = 20;
  p->sg_iters = 3; p->sg_window = 41; p->sg_order = 2;
  p->mdistcutoff = 500;
  p->poa_match = 5; p->poa_mismatch = 4;
  p->poa_o1 = 4; p->poa_e1 = 2; p->poa_o2 = 24; p->poa_e2 = 1;
  p->poa_band_b = 10; p->poa_band_f = 0.01;
  p->pol_match = 3; p->pol_mismatch = -5; p->pol_gap = -4;
  p->pol_window = 500; p->pol_q = 5;
  p->dang_band = 128;
  p->zero = 1; p->zr_match = 2; p->zr_mismatch = 4; p->zr_gapo = 4; p->zr_gape = 2;
  p->zr_min_score = 80; p->zr_max_cells = 16 << 20;
}

int c3o_code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 0;
  }
}

/* conk (C3POa.py:123).  Spec frozen in DESIGN.md 4.1:
 *   H[i][j] = max(0, H[i-1][j-1] + (a_i==b_j ? match : mismatch),
 *                    H[i-1][j] - penalty, H[i][j-1] - penalty),   H[-1][*]=H[*][-1]=0
 *   track[d] = sum over cells with j - i == d, for d in [0, L)
 * splint indexes rows (i), read indexes columns (j). */
static int64_t c3o_conk_impl(const char* splint, int S, const char* read, int L,
                 int match, int mismatch, int penalty, int32_t* track) {
  int32_t* prev = (int32_t*)calloc((size_t)L + 1, sizeof(int32_t));
  int32_t* cur = (int32_t*)calloc((size_t)L + 1, sizeof(int32_t));
  uint8_t* rc = (uint8_t*)malloc((size_t)L + 1);
  for (int j = 0; j < L; ++j) rc[j] = (uint8_t)c3o_code(read[j]);
  memset(track, 0, sizeof(int32_t) * (size_t)L);
  for (int i = 0; i < S; ++i) {
    int a = c3o_code(splint[i]);
    cur[0] = 0;
    for (int j = 0; j < L; ++j) {
      int32_t dg = prev[j] + (a == rc[j] ? match : mismatch);
      int32_t up = prev[j + 1] - penalty;
      int32_t lf = cur[j] - penalty;
      int32_t h = dg > up ? dg : up;
      if (lf > h) h = lf;
      if (h < 0) h = 0;
      cur[j + 1] = h;
      if (j >= i) track[j - i] += h;
    }
    int32_t* t = prev; prev = cur; cur = t;
  }
  free(prev); free(cur); free(rc);
  return (int64_t)S * L;
}

/* savitzky_golay coefficients (bin/savitzky_golay.py:30-31): row 0 of pinv(Vandermonde).
 * For deriv=0 and order 2 (also 3) the least-squares smoother has the closed form
 *   c_k = 3(3m^2+3m-1-5k^2) / ((2m-1)(2m+1)(2m+3)),  k in [-m, m]. */
int c3o_savgol_coeffs(int window, int order, double* c) {
  if (window < 1 || (window & 1) == 0) return -1;
  if (order != 2 && order != 3) return -1;
  if (window < order + 2) return -1;
  int m = (window - 1) / 2;
  double den = (double)(2 * m - 1) * (double)(2 * m + 1) * (double)(2 * m + 3);
  for (int k = -m; k <= m; ++k)
    c[k + m] = 3.0 * (double)(3 * m * m + 3 * m - 1 - 5 * k * k) / den;
  return 0;
}

/* one smoothing pass (bin/savitzky_golay.py:33-36).  Summation order is part of the spec
 * (DESIGN.md 4.2): the coefficients are symmetric, so mirrored taps are added first,
 *   acc = 0; for k in 0..half-1: acc = fma(c[k], ypad[i+k] + ypad[i+window-1-k], acc);
 *   acc = fma(c[half], ypad[i+half], acc)
 * which keeps exactly mirror-symmetric inputs exactly symmetric (plateau midpoints as scipy). */
static int c3o_savgol_impl(const double* y, int n, int window, int order, double* out) {
  double c[257];
  if (window > 257 || c3o_savgol_coeffs(window, order, c)) return -1;
  int half = (window - 1) / 2;
  if (n < half + 1) return -1;
  double* pad = (double*)malloc(sizeof(double) * ((size_t)n + 2 * half));
  for (int k = 0; k < half; ++k) {
    pad[k] = y[0] - fabs(y[half - k] - y[0]);                 /* firstvals */
    pad[half + n + k] = y[n - 1] + fabs(y[n - 2 - k] - y[n - 1]); /* lastvals */
  }
  memcpy(pad + half, y, sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    double acc = 0.0;
    for (int k = 0; k < half; ++k) acc = fma(c[k], pad[i + k] + pad[i + window - 1 - k], acc);
    acc = fma(c[half], pad[i + half], acc);
    out[i] = acc;
  }
  free(pad);
  return 0;
}

static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

typedef struct { double h; int idx; } prio_t;
static int cmp_prio(const void* a, const void* b) {
  const prio_t* x = (const prio_t*)a; const prio_t* y = (const prio_t*)b;
  if (x->h < y->h) return -1;
  if (x->h > y->h) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);   /* stable: equal heights keep index order */
}

/* scipy.signal.find_peaks(x, height=height, distance=distance)[0] */
static int c3o_find_peaks_impl(const double* x, int n, double height, int distance,
                   int64_t* peaks, int cap) {
  int* cand = (int*)malloc(sizeof(int) * (size_t)(n / 2 + 2));
  int nc = 0;
  /* _local_maxima_1d: strict rise, plateau midpoint, strict fall; ends excluded */
  int i = 1, imax = n - 1;
  while (i < imax) {
    if (x[i - 1] < x[i]) {
      int ia = i + 1;
      while (ia < imax && x[ia] == x[i]) ++ia;
      if (x[ia] < x[i]) {
        cand[nc++] = (i + ia - 1) / 2;
        i = ia;
      }
    }
    ++i;
  }
  /* height filter (inclusive) */
  int nh = 0;
  for (int k = 0; k < nc; ++k)
    if (x[cand[k]] >= height) cand[nh++] = cand[k];
  nc = nh;
  /* _select_by_peak_distance: highest first; equal priority -> later index first */
  if (distance < 1) distance = 1;
  char* keep = (char*)malloc((size_t)nc + 1);
  prio_t* pr = (prio_t*)malloc(sizeof(prio_t) * (size_t)(nc + 1));
  for (int k = 0; k < nc; ++k) { keep[k] = 1; pr[k].h = x[cand[k]]; pr[k].idx = k; }
  qsort(pr, (size_t)nc, sizeof(prio_t), cmp_prio);
  for (int t = nc - 1; t >= 0; --t) {
    int j = pr[t].idx;
    if (!keep[j]) continue;
    int k = j - 1;
    while (k >= 0 && cand[j] - cand[k] < distance) { keep[k] = 0; --k; }
    k = j + 1;
    while (k < nc && cand[k] - cand[j] < distance) { keep[k] = 0; ++k; }
  }
  int np = 0;
  for (int k = 0; k < nc; ++k)
    if (keep[k]) { if (np < cap) peaks[np] = cand[k]; ++np; }
  free(cand); free(keep); free(pr);
  return np;
}

/* bin/call_peaks.py:8-16 */
static int c3o_call_peaks_impl(const int32_t* scores, int n, int min_dist, int iters,
                   int window, int order, int64_t* peaks, int cap, double* smoothed) {
  int half = (window - 1) / 2;
  if (n < half + 1) return -1;
  double* a = (double*)malloc(sizeof(double) * (size_t)n);
  double* b = (double*)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) a[i] = (double)scores[i];
  for (int it = 0; it < iters; ++it) {
    c3o_savgol(a, n, window, order, b);
    double* t = a; a = b; b = t;
  }
  if (smoothed) memcpy(smoothed, a, sizeof(double) * (size_t)n);
  /* np.median */
  memcpy(b, a, sizeof(double) * (size_t)n);
  qsort(b, (size_t)n, sizeof(double), cmp_double);
  double med = (n & 1) ? b[n / 2] : (b[n / 2 - 1] + b[n / 2]) / 2.0;
  double mx = a[0];
  for (int i = 1; i < n; ++i) if (a[i] > mx) mx = a[i];
  int np = 0;
  if (!(mx < 6 * med)) np = c3o_find_peaks(a, n, med * 3, min_dist, peaks, cap);
  free(a); free(b);
  return np;
}

/* C3POa.py:106-108: int(base * round(float(x)/base)), Python round = half to even.
 * x/base is exactly representable at the tie (r == base/2), so integer arithmetic is exact. */
int c3o_rounding(int x, int base) {
  int q = x / base, r = x % base;
  if (2 * r < base) return q * base;
  if (2 * r > base) return (q + 1) * base;
  return ((q & 1) ? q + 1 : q) * base;
}

/* C3POa.py:127-155 */
static int c3o_split_impl(const int64_t* peaks_in, int n_in, int S, int L,
              int64_t* peaks_out, int* sub_beg, int* sub_end, c3o_split_info* info) {
  memset(info, 0, sizeof(*info));
  int np = 0;
  for (int i = 0; i < n_in; ++i) {
    int64_t p = peaks_in[i] + S / 2;
    if (p < L) peaks_out[np++] = p;
  }
  info->n_peaks = np;
  if (np == 0) return 0;
  if (np > 1) {
    int nl = np - 1;
    int* r = (int*)malloc(sizeof(int) * (size_t)nl);
    int* s = (int*)malloc(sizeof(int) * (size_t)nl);
    for (int i = 0; i < nl; ++i) s[i] = r[i] = c3o_rounding((int)(peaks_out[i + 1] - peaks_out[i]), 50);
    /* np.median of ints -> float */
    for (int i = 1; i < nl; ++i) { int v = s[i], k = i - 1; while (k >= 0 && s[k] > v) { s[k + 1] = s[k]; --k; } s[k + 1] = v; }
    double med = (nl & 1) ? (double)s[nl / 2] : ((double)s[nl / 2 - 1] + (double)s[nl / 2]) / 2.0;
    int ns = 0;
    for (int i = 0; i < nl; ++i) {
      if (med * 0.8 <= (double)r[i] && (double)r[i] <= med * 1.2) {
        sub_beg[ns] = (int)peaks_out[i]; sub_end[ns] = (int)peaks_out[i + 1]; ++ns;
      }
    }
    info->n_sub = ns;
    if (peaks_out[0] > 100) { info->has_front = 1; info->front_end = (int)peaks_out[0]; }
    if (L - peaks_out[np - 1] > 100) { info->has_tail = 1; info->tail_beg = (int)peaks_out[np - 1]; }
    free(r); free(s);
  } else {
    info->n_sub = 0;
    info->has_front = 1; info->front_end = (int)peaks_out[0];
    info->has_tail = 1; info->tail_beg = (int)peaks_out[0];
  }
  return np;
}

int64_t c3o_conk(const char* splint, int S, const char* read, int L,
                 int match, int mismatch, int penalty, int32_t* track) {
  c3o_enter(); int64_t r_ = c3o_conk_impl(splint, S, read, L, match, mismatch, penalty, track); c3o_leave(); return r_;
}

int c3o_savgol(const double* y, int n, int window, int order, double* out) {
  c3o_enter(); int r_ = c3o_savgol_impl(y, n, window, order, out); c3o_leave(); return r_;
}

int c3o_find_peaks(const double* x, int n, double height, int distance,
                   int64_t* peaks, int cap) {
  c3o_enter(); int r_ = c3o_find_peaks_impl(x, n, height, distance, peaks, cap); c3o_leave(); return r_;
}

int c3o_call_peaks(const int32_t* scores, int n, int min_dist, int iters,
                   int window, int order, int64_t* peaks, int cap, double* smoothed) {
  c3o_enter(); int r_ = c3o_call_peaks_impl(scores, n, min_dist, iters, window, order, peaks, cap, smoothed); c3o_leave(); return r_;
}

int c3o_split(const int64_t* peaks_in, int n_in, int S, int L,
              int64_t* peaks_out, int* sub_beg, int* sub_end, c3o_split_info* info) {
  c3o_enter(); int r_ = c3o_split_impl(peaks_in, n_in, S, L, peaks_out, sub_beg, sub_end, info); c3o_leave(); return r_;
}
