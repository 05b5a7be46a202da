/*
 * c3o_pairwise.c -- ORACLE (test infrastructure): quality-aware merge of a 2-row MSA.
 * Restates /root/reference/bin/consensus.py:4-81 (consensus, avgQual, normalizeLen,
 * pairwise_consensus) including its quirks; pinned by tests/golden/pairwise.json, which is
 * generated from the reference's own Python.
 */
#include "c3o.h"
#include "c3o_internal.h"
#include <stdlib.h>
#include <string.h>
#include "c3o_mem.h"

/* bin/consensus.py:50-74 */
int c3o_normalize_len(const char* row, int msa_len, const char* qual, int qlen, char* out) {
  int si = 0, qi = 0, n = 0;
  while (qi < qlen) {
    if (row[si] != '-') { out[n++] = qual[qi]; ++qi; ++si; }
    else if (qi == 0) { out[n++] = qual[qi]; ++si; }
    else { out[n++] = (char)(((int)(unsigned char)qual[qi - 1] + (int)(unsigned char)qual[qi]) / 2); ++si; }
  }
  if (n != msa_len) {
    int gap = 0;
    while (gap < msa_len && row[msa_len - 1 - gap] == '-') { out[n] = out[n - 1]; ++n; ++gap; }
  }
  return n;
}

/* bin/consensus.py:4-44,76-81.  avgQual(A) > avgQual(B) over the same slice and the same
 * divisor is equivalent to comparing the integer sums. */
static int c3o_pairwise_consensus_cols_impl(const char* A, const char* B, int n,
                                const char* subA, int lenA, const char* qualA,
                                const char* subB, int lenB, const char* qualB,
                                char* out, int cap, int* out_col) {
  /* seqDict keyed by subread string: identical subreads share the LATER quality (:77-79) */
  if (lenA == lenB && memcmp(subA, subB, (size_t)lenA) == 0) qualA = qualB;
  char* qa = (char*)malloc((size_t)n + 1);
  char* qb = (char*)malloc((size_t)n + 1);
  c3o_normalize_len(A, n, qualA, lenA, qa);
  c3o_normalize_len(B, n, qualB, lenB, qb);
  int o = 0, i = 0;
  while (i != n) {
    if (A[i] == B[i]) { if (A[i] != '-' && o < cap) { if (out_col) out_col[o] = i; out[o++] = A[i]; } }
    if (A[i] != B[i] && A[i] != '-' && B[i] != '-') {
      char c = ((unsigned char)qa[i] > (unsigned char)qb[i]) ? A[i] : B[i];
      if (o < cap) { if (out_col) out_col[o] = i; out[o++] = c; }
    }
    if (A[i] == '-' || B[i] == '-') {
      int gl = 1;
      const char* gs = (A[i] == '-') ? A : B;
      for (;;) {
        if (i + gl >= n) { gl = 1; break; }   /* IndexError branch (:35-36) */
        if (gs[i + gl] == '-') ++gl; else break;
      }
      long sa = 0, sb = 0;
      for (int k = i; k < i + gl && k < n; ++k) { sa += (unsigned char)qa[k]; sb += (unsigned char)qb[k]; }
      const char* src = (sa > sb) ? A : B;
      for (int k = i; k < i + gl && k < n; ++k)
        if (src[k] != '-' && o < cap) { if (out_col) out_col[o] = k; out[o++] = src[k]; }
      i += gl;
      continue;
    }
    ++i;
  }
  free(qa); free(qb);
  return o;
}

int c3o_pairwise_consensus(const char* A, const char* B, int n,
                           const char* subA, int lenA, const char* qualA,
                           const char* subB, int lenB, const char* qualB,
                           char* out, int cap) {
  return c3o_pairwise_consensus_cols(A, B, n, subA, lenA, qualA, subB, lenB, qualB, out, cap, 0);
}

int c3o_pairwise_consensus_cols(const char* A, const char* B, int n,
                                const char* subA, int lenA, const char* qualA,
                                const char* subB, int lenB, const char* qualB,
                                char* out, int cap, int* out_col) {
  c3o_enter(); int r_ = c3o_pairwise_consensus_cols_impl(A, B, n, subA, lenA, qualA, subB, lenB, qualB, out, cap, out_col); c3o_leave(); return r_;
}
