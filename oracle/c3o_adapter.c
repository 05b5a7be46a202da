/*
 * c3o_adapter.c -- ORACLE (test infrastructure, never shipped): adapter finder of the post-processing step.
 *
 * Restates the role of the blat call in /root/reference/C3POa_postprocessing.py:229-236 as consumed by parse_blat
 * (:238-264).  blat is an external binary that is not in /root/reference: **parity unpinned**.  Restatement
 * (DESIGN.md 4.9): for one read, one adapter and one strand, the best LOCAL alignment with affine gaps under the
 * map-ont base scoring (match 2, mismatch -4, gap 4 + 2k; the scoring of the zero-repeat overlap),
 * rows = read bases, columns = adapter bases (strand '-': the reverse complement of the adapter), first maximum in
 * row-major order, tie order diagonal > vertical (consumes a read base) > horizontal, gap open before extend; the
 * traceback counts matches, mismatches and inserted bases.  Coordinates follow PSL: query = read (forward),
 * target = adapter (forward: for strand '-' the columns are mirrored back).
 */
#include "c3o.h"
#include "c3o_internal.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "c3o_mem.h"

static int adapter_align_impl(const char* read, int n, const char* adapter, int m, int rc, const c3o_params* P, int32_t* out) {
  memset(out, 0, sizeof(int32_t) * 12);
  out[11] = n;
  if (n <= 0 || m <= 0) return 0;
  const int a = P->zr_match, b = -P->zr_mismatch, go = P->zr_gapo, ge = P->zr_gape;
  uint8_t* q = (uint8_t*)malloc((size_t)n); uint8_t* r = (uint8_t*)malloc((size_t)m);
  for (int i = 0; i < n; ++i) q[i] = (uint8_t)c3o_code(read[i]);
  for (int j = 0; j < m; ++j) r[j] = rc ? (uint8_t)(3 - c3o_code(adapter[m - 1 - j])) : (uint8_t)c3o_code(adapter[j]);
  const int W = m + 1;
  int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)2 * W);
  int32_t* E = (int32_t*)malloc(sizeof(int32_t) * (size_t)2 * W);
  uint8_t* D = (uint8_t*)malloc((size_t)(n + 1) * W);
  for (int j = 0; j <= m; ++j) { H[j] = 0; E[j] = INT_MIN / 2; D[j] = 0; }
  int best = 0, bi = 0, bj = 0;
  for (int i = 1; i <= n; ++i) {
    int32_t* hp = H + (size_t)((i - 1) & 1) * W; int32_t* hc = H + (size_t)(i & 1) * W;
    int32_t* ep = E + (size_t)((i - 1) & 1) * W; int32_t* ec = E + (size_t)(i & 1) * W;
    uint8_t* d = D + (size_t)i * W;
    hc[0] = 0; ec[0] = INT_MIN / 2; d[0] = 0;
    int32_t f = INT_MIN / 2;
    for (int j = 1; j <= m; ++j) {
      int eo = hp[j] - go - ge, ee = ep[j] - ge;
      int ex = ee > eo; int32_t e = ex ? ee : eo;
      int fo = hc[j - 1] - go - ge, fe = f - ge;
      int fx = fe > fo; f = fx ? fe : fo;
      int32_t dg = hp[j - 1] + (q[i - 1] == r[j - 1] ? a : b);
      int32_t h = 0; int src = 0;
      if (dg > h) { h = dg; src = 1; }
      if (e > h) { h = e; src = 2; }
      if (f > h) { h = f; src = 3; }
      hc[j] = h; ec[j] = e;
      d[j] = (uint8_t)(src | (ex << 2) | (fx << 3));
      if (h > best) { best = h; bi = i; bj = j; }
    }
  }
  if (best > 0) {
    int i = bi, j = bj, st = 0, nma = 0, nmm = 0, qi = 0, ti = 0, qn = 0, tn = 0;
    for (;;) {
      if (i == 0 || j == 0) break;
      uint8_t d = D[(size_t)i * W + j];
      if (st == 0) {
        int src = d & 3;
        if (src == 0) break;
        if (src == 1) { if (q[i - 1] == r[j - 1]) ++nma; else ++nmm; --i; --j; }
        else { st = src; if (src == 2) ++qn; else ++tn; }
      } else if (st == 2) { st = (d & 4) ? 2 : 0; ++qi; --i; }
      else { st = (d & 8) ? 3 : 0; ++ti; --j; }
    }
    out[0] = best; out[1] = i; out[2] = bi;
    out[3] = rc ? m - bj : j; out[4] = rc ? m - j : bj;
    out[5] = nma; out[6] = nmm; out[7] = qi; out[8] = ti; out[9] = qn; out[10] = tn;
  }
  free(H); free(E); free(D); free(q); free(r);
  return best;
}

int c3o_adapter_align(const char* read, int n, const char* adapter, int m, int rc, const c3o_params* P, int32_t* out) {
  c3o_enter();
  int r = adapter_align_impl(read, n, adapter, m, rc, P, out);
  c3o_leave();
  return r;
}
