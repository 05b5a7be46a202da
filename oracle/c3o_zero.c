/*
 * c3o_zero.c -- ORACLE (test infrastructure, never shipped): zero-repeat rescue.
 *
 * Restates /root/reference/bin/determine_consensus.py:106-136.  The reference finds the overlap of the
 * two dangling pieces with mappy (minimap2 map-ont, scoring=(20,7,10,5)), which is not vendored:
 * **parity unpinned**.  Restatement: the overlap is the best forward-strand LOCAL
 * alignment of d1 (query) against d0 (reference) with the map-ont base scoring (match 2, mismatch -4,
 * gap 4 + 2k); first maximum in row-major order; tie order diagonal > vertical (consumes d1) >
 * horizontal, gap open before extend.  (minimap2 locates the overlap by minimizer chaining; the
 * call-site scoring (20,7,10,5) only steers its extension.  With that scoring a plain local alignment
 * is in the linear regime -- two RANDOM 1-kb sequences out-score a true 400-nt overlap -- so it cannot
 * be used to find the overlap.)  Accepted when the score is >= 80 (40 matching bases); then
 *   left = d1[:q_st], right = d0[r_en:], overlap consensus = pairwise_consensus of the 2-row abPOA MSA
 * and the result is left + consensus + right (decoded from the 2-bit codes, as everywhere).
 */
#include "c3o.h"
#include "c3o_internal.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "c3o_mem.h"

static const char ACGT[] = "ACGT";

/* local affine alignment of q (rows, d1) vs r (columns, d0); returns best score, fills coords */
static int zr_local(const uint8_t* r, int nr, const uint8_t* q, int nq, const c3o_params* P,
                    int* r_st, int* r_en, int* q_st, int* q_en, int64_t* cells) {
  const int a = P->zr_match, b = -P->zr_mismatch, go = P->zr_gapo, ge = P->zr_gape;
  const int W = nr + 1;
  int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)2 * W);
  int32_t* E = (int32_t*)malloc(sizeof(int32_t) * (size_t)2 * W);   /* vertical gap state (consumes q) */
  uint8_t* D = (uint8_t*)malloc((size_t)(nq + 1) * W);
  /* D bits: 0-1 H source (0 stop, 1 diag, 2 E, 3 F), bit2 E extended, bit3 F extended */
  for (int j = 0; j <= nr; ++j) { H[j] = 0; E[j] = INT_MIN / 2; D[j] = 0; }
  int best = 0, bi = 0, bj = 0;
  for (int i = 1; i <= nq; ++i) {
    int32_t* hp = H + (size_t)((i - 1) & 1) * W; int32_t* hc = H + (size_t)(i & 1) * W;
    int32_t* ep = E + (size_t)((i - 1) & 1) * W; int32_t* ec = E + (size_t)(i & 1) * W;
    uint8_t* d = D + (size_t)i * W;
    hc[0] = 0; ec[0] = INT_MIN / 2; d[0] = 0;
    int32_t f = INT_MIN / 2;
    for (int j = 1; j <= nr; ++j) {
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
  *cells += (int64_t)nq * nr;
  if (best > 0) {
    int i = bi, j = bj, st = 0;     /* 0 H, 2 E, 3 F */
    for (;;) {
      uint8_t d = D[(size_t)i * W + j];
      if (st == 0) {
        int src = d & 3;
        if (src == 0) break;
        if (src == 1) { --i; --j; }
        else st = src;
      } else if (st == 2) { st = (d & 4) ? 2 : 0; --i; }
      else { st = (d & 8) ? 3 : 0; --j; }
    }
    *q_st = i; *r_st = j; *q_en = bi; *r_en = bj;
  }
  free(H); free(E); free(D);
  return best;
}

static int zero_repeats_impl(const char* d0, const char* q0, int n0, const char* d1, const char* q1, int n1,
                             const c3o_params* P, char* out, int cap, int64_t* cells) {
  if (n0 <= 0 || n1 <= 0 || (int64_t)n0 * n1 > P->zr_max_cells) return 0;
  uint8_t* c0 = (uint8_t*)malloc((size_t)n0 + 1); uint8_t* c1 = (uint8_t*)malloc((size_t)n1 + 1);
  for (int i = 0; i < n0; ++i) c0[i] = (uint8_t)c3o_code(d0[i]);
  for (int i = 0; i < n1; ++i) c1[i] = (uint8_t)c3o_code(d1[i]);
  int r_st = 0, r_en = 0, q_st = 0, q_en = 0;
  int sc = zr_local(c0, n0, c1, n1, P, &r_st, &r_en, &q_st, &q_en, cells);
  int olen = 0;
  if (sc >= P->zr_min_score && r_en > r_st && q_en > q_st) {
    /* 2-row MSA of the overlap + quality merge */
    const char* seqs[2] = {d0 + r_st, d1 + q_st};
    int lens[2] = {r_en - r_st, q_en - q_st};
    int tot = lens[0] + lens[1] + 8;
    char* msa = (char*)malloc((size_t)tot * 2);
    char* cons = (char*)malloc((size_t)tot);
    int ml = 0, cl = 0; int64_t pc = 0;
    int rc = c3o_poa_msa(seqs, lens, 2, P, NULL, 0, &cl, msa, (int64_t)tot * 2, &ml, &pc);
    *cells += pc;
    if (rc == 0 && ml > 0) {
      char* sa = (char*)malloc((size_t)lens[0] + 1); char* sb = (char*)malloc((size_t)lens[1] + 1);
      for (int k = 0; k < lens[0]; ++k) sa[k] = ACGT[c0[r_st + k]];
      for (int k = 0; k < lens[1]; ++k) sb[k] = ACGT[c1[q_st + k]];
      int n = c3o_pairwise_consensus(msa, msa + ml, ml, sa, lens[0], q0 + r_st, sb, lens[1], q1 + q_st, cons, tot);
      int need = q_st + n + (n0 - r_en);
      if (n > 0 && need <= cap) {
        for (int k = 0; k < q_st; ++k) out[olen++] = ACGT[c1[k]];           /* left  = d1[:q_st] */
        memcpy(out + olen, cons, (size_t)n); olen += n;
        for (int k = r_en; k < n0; ++k) out[olen++] = ACGT[c0[k]];          /* right = d0[r_en:] */
      }
      free(sa); free(sb);
    }
    free(msa); free(cons);
  }
  free(c0); free(c1);
  return olen;
}

int c3o_zero_repeats(const char* d0, const char* q0, int n0, const char* d1, const char* q1, int n1,
                     const c3o_params* P, char* out, int cap, int64_t* cells) {
  c3o_enter();
  int r = zero_repeats_impl(d0, q0, n0, d1, q1, n1, P, out, cap, cells);
  c3o_leave();
  return r;
}
