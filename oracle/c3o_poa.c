/*
 * c3o_poa.c -- ORACLE (test infrastructure, never shipped): adaptive-band partial-order
 * alignment with convex gap cost, heaviest-bundling consensus and row-column MSA.
 *
 * Restates what the reference obtains from pyabpoa==1.0.5 (setup.sh:8) at
 *   bin/determine_consensus.py:30   poa.msa_aligner(match=5)
 *   bin/determine_consensus.py:34   .msa(subreads, out_cons=False, out_msa=True)
 *   bin/determine_consensus.py:43   .msa(subreads, out_cons=True,  out_msa=True)
 * abPOA is NOT vendored in /root/reference and not installed here: this follows the published
 * algorithm (Gao et al. 2021: global mode, mismatch 4, convex gap 4/2 + 24/1, adaptive band
 * w = 10 + 0.01*qlen, sequences fused in input order, heaviest bundling) with tie-break rules
 * frozen in DESIGN.md 4.3.  **parity with the real library: unpinned.**
 */
#include "c3o.h"
#include "c3o_graph.h"
#include "c3o_internal.h"
#include <limits.h>
#include "c3o_mem.h"

/* DP score of the end cell of every sequence-to-graph alignment of the last c3o_poa_msa call on this thread: lets an
 * independent brute-force aligner (tests/test_independent_checkers.py) check optimality without sharing any code */
static __thread int32_t g_scores[256];
static __thread int g_nscores = 0;
int c3o_poa_last_scores(int32_t* out, int cap) { int n = g_nscores < cap ? g_nscores : cap; for (int i = 0; i < n; ++i) out[i] = g_scores[i]; return g_nscores; }

#define SRC 0
#define SNK 1

/* diagnostic row statistics (tools/poa_row_stats.py; single-threaded use only): what kinds of DP rows the alignments of a
 * batch consist of, as the HIP kernel classifies them, and how far real cell values sit below their row's maximum */
int64_t c3o_poa_rowstats[64];
int c3o_poa_rowstats_on = 0;
/* run-length statistics (tools/poa_run_lengths.py; round 6): [0..63] rows that lie in a run of k+1 consecutive fast rows (63 = 64 or
 * more), [64..127] the same for runs of fast rows whose band moved by exactly one column at both ends (what a skewed multi-row
 * step would have to predict), [128..191] fast rows whose band shift / width change was (dbeg, dend) = (1,1) / other, by kind */
int64_t c3o_poa_runstats[192];
static void run_close(int64_t* S, int len) { if (len > 0) S[len > 64 ? 63 : len - 1] += len; }

typedef struct {
  int32_t *H, *E1, *E2; uint32_t* D;
  int64_t ncell, cap;
  int *rbeg, *rend; int64_t* roff;   /* per order index */
} dpmat;

static void dp_reserve(dpmat* m, int64_t need) {
  if (need <= m->cap) return;
  int64_t c = m->cap ? m->cap : 1 << 16;
  while (c < need) c *= 2;
  m->H = (int32_t*)realloc(m->H, sizeof(int32_t) * (size_t)c);
  m->E1 = (int32_t*)realloc(m->E1, sizeof(int32_t) * (size_t)c);
  m->E2 = (int32_t*)realloc(m->E2, sizeof(int32_t) * (size_t)c);
  m->D = (uint32_t*)realloc(m->D, sizeof(uint32_t) * (size_t)c);
  m->cap = c;
}
static inline int32_t rd(const dpmat* m, const int32_t* a, int r, int j) {
  if (j < m->rbeg[r] || j > m->rend[r]) return C3O_NEG;
  return a[m->roff[r] + (j - m->rbeg[r])];
}

/* direction word layout */
#define D_MP(d)   ((d) & 0xff)
#define D_E1P(d)  (((d) >> 8) & 0xff)
#define D_E2P(d)  (((d) >> 16) & 0xff)
#define D_HT(d)   (((d) >> 24) & 3)
#define D_HS(d)   (((d) >> 26) & 3)
#define D_E1X(d)  (((d) >> 28) & 1)
#define D_E2X(d)  (((d) >> 29) & 1)
#define D_F1X(d)  (((d) >> 30) & 1)
#define D_F2X(d)  (((d) >> 31) & 1)

typedef struct { int* node; int* q; int n, cap; } oplist;
static void op_push(oplist* o, int node, int q) {
  if (o->n == o->cap) {
    o->cap = o->cap ? 2 * o->cap : 4096;
    o->node = (int*)realloc(o->node, sizeof(int) * (size_t)o->cap);
    o->q = (int*)realloc(o->q, sizeof(int) * (size_t)o->cap);
  }
  o->node[o->n] = node; o->q[o->n] = q; o->n++;
}

/* remaining length along the heaviest out-edge (first max in out-list order) */
static void compute_remain(const c3o_graph* g, int* rem) {
  int K = g->K;
  for (int i = g->n - 1; i >= 0; --i) {
    int v = g->order[i];
    if (v == SNK) { rem[v] = -1; continue; }
    int bw = INT_MIN, bt = SNK;
    for (int k = 0; k < g->n_out[v]; ++k)
      if (g->out_w[v * K + k] > bw) { bw = g->out_w[v * K + k]; bt = g->out_to[v * K + k]; }
    rem[v] = rem[bt] + 1;
  }
}

/* banded global alignment of q (codes, length Q) to graph g.  ops returned forward. */
static int poa_align(const c3o_graph* g, const uint8_t* q, int Q, const c3o_params* P,
                     oplist* ops, int64_t* cells) {
  const int K = g->K, n = g->n;
  const int mt = P->poa_match, mm = -P->poa_mismatch;
  const int o1 = P->poa_o1, e1 = P->poa_e1, o2 = P->poa_o2, e2 = P->poa_e2;
  const int oe1 = o1 + e1, oe2 = o2 + e2;
  const int w = P->poa_band_b + (int)(P->poa_band_f * Q);
  int* rem = (int*)malloc(sizeof(int) * (size_t)n);
  int* mpl = (int*)malloc(sizeof(int) * (size_t)n);
  int* mpr = (int*)malloc(sizeof(int) * (size_t)n);
  compute_remain(g, rem);
  for (int v = 0; v < n; ++v) { mpl[v] = INT_MAX / 2; mpr[v] = 0; }
  dpmat m; memset(&m, 0, sizeof(m));
  m.rbeg = (int*)malloc(sizeof(int) * (size_t)n);
  m.rend = (int*)malloc(sizeof(int) * (size_t)n);
  m.roff = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  int32_t* ht = (int32_t*)malloc(sizeof(int32_t) * (size_t)(Q + 2));
  int run_fast = 0, run_diag = 0;

  for (int idx = 0; idx < n; ++idx) {
    int v = g->order[idx];
    if (v == SNK) { m.rbeg[idx] = 0; m.rend[idx] = -1; m.roff[idx] = m.ncell; continue; }
    int beg, end;
    int qr = Q - rem[v];
    if (v == SRC) {
      beg = 0;
      end = (qr > 0 ? qr : 0) + w; if (end > Q) end = Q;
    } else {
      int lo = mpl[v] < qr ? mpl[v] : qr, hi = mpr[v] > qr ? mpr[v] : qr;
      beg = lo - w; if (beg < 0) beg = 0;
      end = hi + w; if (end > Q) end = Q;
      int minb = INT_MAX, maxe = INT_MIN;
      for (int k = 0; k < g->n_in[v]; ++k) {
        int pi = g->index[g->in_from[v * K + k]];
        if (m.rbeg[pi] < minb) minb = m.rbeg[pi];
        if (m.rend[pi] + 1 > maxe) maxe = m.rend[pi] + 1;
      }
      if (beg < minb) beg = minb;
      if (end > maxe) end = maxe;
    }
    if (end < beg) end = beg - 1;           /* empty row */
    int wd = end - beg + 1;
    m.rbeg[idx] = beg; m.rend[idx] = end; m.roff[idx] = m.ncell;
    dp_reserve(&m, m.ncell + wd + 1);
    int32_t* H = m.H + m.ncell; int32_t* E1 = m.E1 + m.ncell; int32_t* E2 = m.E2 + m.ncell;
    uint32_t* D = m.D + m.ncell;
    m.ncell += wd;
    if (wd <= 0) continue;
    /* vertical / diagonal part */
    for (int j = beg; j <= end; ++j) {
      int c = j - beg; uint32_t d = 0;
      if (v == SRC) {
        ht[c] = (j == 0) ? 0 : C3O_NEG; E1[c] = E2[c] = C3O_NEG; D[c] = 0; continue;
      }
      int32_t M = INT_MIN, e1b = INT_MIN, e2b = INT_MIN; int mp = 0, e1p = 0, e2p = 0, e1x = 0, e2x = 0;
      for (int k = 0; k < g->n_in[v]; ++k) {
        int pi = g->index[g->in_from[v * K + k]];
        int32_t hd = (j > 0) ? rd(&m, m.H, pi, j - 1) : C3O_NEG;
        if (hd > M) { M = hd; mp = k; }
        int32_t hp = rd(&m, m.H, pi, j);
        int32_t a = hp - oe1, b = rd(&m, m.E1, pi, j) - e1;
        int32_t cnd = a >= b ? a : b; int x = b > a;
        if (cnd > e1b) { e1b = cnd; e1p = k; e1x = x; }
        a = hp - oe2; b = rd(&m, m.E2, pi, j) - e2;
        cnd = a >= b ? a : b; x = b > a;
        if (cnd > e2b) { e2b = cnd; e2p = k; e2x = x; }
      }
      if (j > 0) M += (g->base[v] == q[j - 1]) ? mt : mm; else M = C3O_NEG;
      int hts; int32_t t;
      if (M >= e1b && M >= e2b) { hts = 0; t = M; }
      else if (e1b >= e2b) { hts = 1; t = e1b; }
      else { hts = 2; t = e2b; }
      ht[c] = t; E1[c] = e1b; E2[c] = e2b;
      d = (uint32_t)mp | ((uint32_t)e1p << 8) | ((uint32_t)e2p << 16) | ((uint32_t)hts << 24)
          | ((uint32_t)e1x << 28) | ((uint32_t)e2x << 29);
      D[c] = d;
    }
    /* horizontal part: F[j] = max_{beg<=k<j} Ht[k] - o - e*(j-k) */
    int32_t f1 = C3O_NEG2, f2 = C3O_NEG2;
    int32_t best = INT_MIN; int left = beg, right = beg;
    for (int c = 0; c < wd; ++c) {
      int f1x = 0, f2x = 0;
      if (c == 0) { f1 = f2 = C3O_NEG2; }
      else {
        int32_t a = ht[c - 1] - oe1;
        if (c == 1) f1 = a; else { int32_t b = f1 - e1; f1x = b > a; f1 = f1x ? b : a; }
        a = ht[c - 1] - oe2;
        if (c == 1) f2 = a; else { int32_t b = f2 - e2; f2x = b > a; f2 = f2x ? b : a; }
      }
      int hs; int32_t h;
      if (ht[c] >= f1 && ht[c] >= f2) { hs = 0; h = ht[c]; }
      else if (f1 >= f2) { hs = 1; h = f1; }
      else { hs = 2; h = f2; }
      H[c] = h;
      D[c] |= ((uint32_t)hs << 26) | ((uint32_t)f1x << 30) | ((uint32_t)f2x << 31);
      if (h > best) { best = h; left = right = beg + c; }
      else if (h == best) right = beg + c;
    }
    if (c3o_poa_rowstats_on && v != SRC) {
      int64_t* S = c3o_poa_rowstats;
      int nin = g->n_in[v], dmax = 0, dmin = 1 << 30, pwmax = 0;
      for (int k = 0; k < nin; ++k) {
        int pi = g->index[g->in_from[v * K + k]];
        if (idx - pi > dmax) dmax = idx - pi;
        if (idx - pi < dmin) dmin = idx - pi;
        if (m.rend[pi] - m.rbeg[pi] > pwmax) pwmax = m.rend[pi] - m.rbeg[pi];
      }
      int pwd = m.rend[idx - 1] - m.rbeg[idx - 1] + 1, sh = beg - m.rbeg[idx - 1];
      S[0]++; S[1] += wd;
      { int f6 = 0, f4 = 0;
        for (int k = 0; k < g->n_out[v]; ++k) { int t = g->out_to[v * K + k]; int dd = (t == SNK) ? 99 : g->index[t] - idx; if (dd > 5) f6 = 1; if (dd > 3) f4 = 1; }
        if (f6) S[56] += wd;
        if (f4) S[57] += wd; }
      if (nin == 1 && dmax == 1 && wd <= 64 && pwd <= 64 && sh < 64 && !(wd + sh <= 64 && (sh >= 1 || pwd <= 63))) S[54]++;     /* fast in round 3, not under the round-4 no-wrap rule */
      { int64_t* R = c3o_poa_runstats;
        const int isfast = nin == 1 && dmax == 1 && wd <= 64 && pwd <= 64 && sh < 64 && wd + sh <= 64 && (sh >= 1 || pwd <= 63);
        const int isdiag = isfast && sh == 1 && end == m.rend[idx - 1] + 1;
        if (isfast) { ++run_fast; R[128 + (isdiag ? 0 : 1)]++; if (!isdiag) { int db = sh, de = end - m.rend[idx - 1]; R[130 + (db == 1 ? 0 : db == 0 ? 1 : db == 2 ? 2 : 3) * 4 + (de == 1 ? 0 : de == 0 ? 1 : de == 2 ? 2 : 3)]++; } }
        else { run_close(R, run_fast); run_fast = 0; }
        if (isdiag) ++run_diag; else { run_close(R + 64, run_diag); run_diag = 0; } }
      if (nin == 1 && dmax == 1 && wd <= 64 && pwd <= 64 && sh < 64) { S[2]++; S[8 + (sh < 0 ? 0 : sh > 3 ? 4 : sh + 0)]++; }   /* fast; shift histogram 8..12 */
      else if (nin <= 4 && dmax < 4 && wd <= 128 && pwmax < 128) {
        S[3]++;
        if (wd > 64) S[5]++;
        if (nin == 1) S[16 + (dmax == 1 ? 0 : dmax == 2 ? 1 : 2)]++;          /* 16: d=1 (wide), 17: d=2, 18: d=3 */
        else if (nin == 2) S[19 + (dmax <= 2 ? 0 : 1)]++;                      /* 19: the two rows above, 20: other */
        else S[21 + (nin - 3)]++;                                              /* 21: three, 22: four */
      } else { S[4]++; if (nin > 4) S[24]++; else if (dmax >= 4) { S[25]++; S[48 + (dmax < 6 ? 0 : dmax < 8 ? 1 : dmax < 12 ? 2 : dmax < 16 ? 3 : 4)]++; } else S[26]++; }
      S[28 + (wd <= 32 ? 0 : wd <= 48 ? 1 : wd <= 64 ? 2 : wd <= 96 ? 3 : wd <= 128 ? 4 : 5)]++;
      if (wd > 128) S[58 + (wd <= 160 ? 0 : wd <= 192 ? 1 : wd <= 256 ? 2 : 3)]++;
      /* lowest real value below the row maximum */
      int lo = 0;
      for (int c = 0; c < wd; ++c) {
        if (H[c] > C3O_NEG / 2 && H[c] - best < lo) lo = H[c] - best;
        if (E1[c] > C3O_NEG / 2 && E1[c] - best < lo) lo = E1[c] - best;
        if (E2[c] > C3O_NEG / 2 && E2[c] - best < lo) lo = E2[c] - best;
      }
      if (lo < S[40]) S[40] = lo;
      S[41 + (lo > -200 ? 0 : lo > -400 ? 1 : lo > -800 ? 2 : lo > -1600 ? 3 : 4)]++;
      /* row maximum against the previous row's (rebasing step) */
      if (idx >= 2 && pwd > 0) {
        int pbest = INT_MIN; const int32_t* PH = m.H + m.roff[idx - 1];
        for (int c = 0; c < pwd; ++c) if (PH[c] > pbest) pbest = PH[c];
        int dd = best - pbest; if (dd < 0) dd = -dd;
        if (dd > S[47]) S[47] = dd;
      }
    }
    /* adaptive band hints for successors */
    for (int k = 0; k < g->n_out[v]; ++k) {
      int t = g->out_to[v * K + k];
      if (right + 1 > mpr[t]) mpr[t] = right + 1;
      if (left + 1 < mpl[t]) mpl[t] = left + 1;
    }
  }
  if (c3o_poa_rowstats_on) { run_close(c3o_poa_runstats, run_fast); run_close(c3o_poa_runstats + 64, run_diag); }
  *cells += m.ncell;

  /* end cell: best predecessor of the sink at column Q (first max in in-edge order) */
  int bi = -1; int32_t bs = INT_MIN;
  for (int k = 0; k < g->n_in[SNK]; ++k) {
    int pi = g->index[g->in_from[SNK * K + k]];
    int32_t h = rd(&m, m.H, pi, Q);
    if (h > bs) { bs = h; bi = pi; }
  }
  if (g_nscores < 256) g_scores[g_nscores++] = bs;       /* checker hook: see c3o_poa_last_scores */
  int rc = 0;
  if (bi < 0 || bs <= C3O_NEG / 2) rc = -1;
  else {
    /* traceback; ops collected backwards then reversed */
    oplist r; memset(&r, 0, sizeof(r));
    int i = bi, j = Q, st = 0; /* 0=H 1=HT 2=E1 3=E2 4=F1 5=F2 */
    while (!(i == 0 && j == 0)) {
      if (j < m.rbeg[i] || j > m.rend[i]) { rc = -2; break; }
      uint32_t d = m.D[m.roff[i] + (j - m.rbeg[i])];
      int v = g->order[i];
      if (st == 0) { int hs = D_HS(d); st = hs == 0 ? 1 : (hs == 1 ? 4 : 5); }
      else if (st == 1) {
        int hts = D_HT(d);
        if (hts == 0) { op_push(&r, v, j - 1); i = g->index[g->in_from[v * K + D_MP(d)]]; --j; st = 0; }
        else st = hts == 1 ? 2 : 3;
      }
      else if (st == 2) { op_push(&r, v, -1); i = g->index[g->in_from[v * K + D_E1P(d)]]; st = D_E1X(d) ? 2 : 0; }
      else if (st == 3) { op_push(&r, v, -1); i = g->index[g->in_from[v * K + D_E2P(d)]]; st = D_E2X(d) ? 3 : 0; }
      else if (st == 4) { op_push(&r, -1, j - 1); st = D_F1X(d) ? 4 : 1; --j; }
      else { op_push(&r, -1, j - 1); st = D_F2X(d) ? 5 : 1; --j; }
    }
    for (int k = r.n - 1; k >= 0; --k) op_push(ops, r.node[k], r.q[k]);
    free(r.node); free(r.q);
  }
  free(rem); free(mpl); free(mpr); free(ht);
  free(m.H); free(m.E1); free(m.E2); free(m.D); free(m.rbeg); free(m.rend); free(m.roff);
  return rc;
}

/* fuse an aligned sequence into the graph; path[qpos] = node.  weight +1 per traversed edge */
static void poa_fuse(c3o_graph* g, const oplist* ops, const uint8_t* q, int Q, int* path) {
  int n_old = g->n;
  int* anchor = (int*)malloc(sizeof(int) * (size_t)(Q + 1));
  int n_new = 0;
  int prev = SRC, cur_anchor = 0 /* order index of SRC */;
  for (int k = 0; k < ops->n; ++k) {
    int v = ops->node[k], qp = ops->q[k];
    if (qp < 0) continue;                     /* deletion: graph node skipped */
    int c = q[qp], t;
    if (v >= 0) {                             /* (mis)match against row node v */
      int r = g->grp[v];
      cur_anchor = g->glast[r];
      t = -1;
      if (g->base[v] == c) t = v;
      else
        for (int i = g->gfirst[r]; i <= g->glast[r]; ++i)
          if (g->base[g->order[i]] == c) { t = g->order[i]; break; }
      if (t < 0) { t = c3o_graph_new_node(g, c); g->grp[t] = r; anchor[n_new++] = cur_anchor; }
    } else {                                  /* insertion */
      t = c3o_graph_new_node(g, c); anchor[n_new++] = cur_anchor;
    }
    c3o_graph_add_edge(g, prev, t, 1);
    g->ncov[t]++;
    path[qp] = t;
    prev = t;
  }
  c3o_graph_add_edge(g, prev, SNK, 1);
  c3o_graph_reorder(g, n_old, anchor);
  free(anchor);
}

/* abPOA heaviest bundling: reverse topological sweep */
static int poa_consensus(const c3o_graph* g, int* cons_nodes) {
  int K = g->K, n = g->n;
  int64_t* score = (int64_t*)calloc((size_t)n, sizeof(int64_t));
  int* nxt = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = n - 1; i >= 0; --i) {
    int v = g->order[i];
    if (v == SNK) { score[v] = 0; nxt[v] = -1; continue; }
    int bw = INT_MIN, bt = -1;
    for (int k = 0; k < g->n_out[v]; ++k) {
      int t = g->out_to[v * K + k], w = g->out_w[v * K + k];
      if (w > bw) { bw = w; bt = t; }
      else if (w == bw && score[bt] <= score[t]) bt = t;
    }
    nxt[v] = bt; score[v] = (int64_t)bw + score[bt];
  }
  int nc = 0;
  for (int v = nxt[SRC]; v != SNK && v >= 0; v = nxt[v]) cons_nodes[nc++] = v;
  free(score); free(nxt);
  return nc;
}

static const char ACGT[] = "ACGT";

int c3o_poa_build(const char* const* seqs, const int* lens, int n, const c3o_params* P,
                  c3o_poa_state* st, int64_t* cells) {
  int total = 2;
  for (int i = 0; i < n; ++i) total += lens[i];
  c3o_graph_init(&st->g, total, n + 1);
  st->n = n; st->lens = (int*)malloc(sizeof(int) * (size_t)n);
  st->path = (int**)calloc((size_t)n, sizeof(int*));
  st->cons_nodes = NULL; st->cons_len = 0;
  c3o_graph* g = &st->g;
  c3o_graph_new_node(g, 0); c3o_graph_new_node(g, 0);
  g->order[0] = SRC; g->order[1] = SNK; g->index[SRC] = 0; g->index[SNK] = 1;
  c3o_graph_blocks(g);
  int rc = 0;
  for (int s = 0; s < n; ++s) {
    int Q = lens[s];
    st->lens[s] = Q;
    uint8_t* q = (uint8_t*)malloc((size_t)Q + 1);
    for (int i = 0; i < Q; ++i) q[i] = (uint8_t)c3o_code(seqs[s][i]);
    st->path[s] = (int*)malloc(sizeof(int) * (size_t)(Q + 1));
    oplist ops; memset(&ops, 0, sizeof(ops));
    if (s == 0) { for (int i = 0; i < Q; ++i) op_push(&ops, -1, i); }
    else rc = poa_align(g, q, Q, P, &ops, cells);
    if (rc == 0) poa_fuse(g, &ops, q, Q, st->path[s]);
    free(ops.node); free(ops.q); free(q);
    if (rc) break;
  }
  return rc;
}
void c3o_poa_free(c3o_poa_state* st) {
  c3o_graph_free(&st->g);
  for (int s = 0; s < st->n; ++s) free(st->path[s]);
  free(st->path); free(st->lens); free(st->cons_nodes);
}
/* column (block rank, 0-based, src/sink excluded) of every node; returns #columns */
int c3o_poa_columns(const c3o_graph* g, int* col) {
  int nc = 0, prev_rep = -1;
  for (int i = 0; i < g->n; ++i) {
    int v = g->order[i];
    if (v == SRC || v == SNK) { col[v] = -1; continue; }
    if (g->grp[v] != prev_rep) { ++nc; prev_rep = g->grp[v]; }
    col[v] = nc - 1;
  }
  return nc;
}
int c3o_poa_make_consensus(c3o_poa_state* st) {
  st->cons_nodes = (int*)malloc(sizeof(int) * (size_t)st->g.n);
  st->cons_len = poa_consensus(&st->g, st->cons_nodes);
  return st->cons_len;
}

static int c3o_poa_msa_impl(const char* const* seqs, const int* lens, int n, const c3o_params* P,
                char* cons, int cons_cap, int* cons_len,
                char* msa, int64_t msa_cap, int* msa_len, int64_t* cells) {
  int64_t cl = 0;
  if (cons_len) *cons_len = 0;
  if (msa_len) *msa_len = 0;
  if (n <= 0) return 0;                       /* msa([]) -> empty result */
  g_nscores = 0;
  c3o_poa_state st;
  int rc = c3o_poa_build(seqs, lens, n, P, &st, &cl);
  if (cells) *cells += cl;
  if (rc == 0 && cons) {
    int nc = c3o_poa_make_consensus(&st);
    if (nc > cons_cap) rc = -3;
    else { for (int i = 0; i < nc; ++i) cons[i] = ACGT[st.g.base[st.cons_nodes[i]]]; *cons_len = nc; }
  }
  if (rc == 0 && msa) {
    int* col = (int*)malloc(sizeof(int) * (size_t)st.g.n);
    int ncol = c3o_poa_columns(&st.g, col);
    if ((int64_t)ncol * n > msa_cap) rc = -3;
    else {
      memset(msa, '-', (size_t)ncol * n);
      for (int s = 0; s < n; ++s)
        for (int i = 0; i < lens[s]; ++i)
          msa[(int64_t)s * ncol + col[st.path[s][i]]] = ACGT[st.g.base[st.path[s][i]]];
      *msa_len = ncol;
    }
    free(col);
  }
  c3o_poa_free(&st);
  return rc;
}

int c3o_poa_msa(const char* const* seqs, const int* lens, int n, const c3o_params* P,
                char* cons, int cons_cap, int* cons_len,
                char* msa, int64_t msa_cap, int* msa_len, int64_t* cells) {
  c3o_enter(); int r_ = c3o_poa_msa_impl(seqs, lens, n, P, cons, cons_cap, cons_len, msa, msa_cap, msa_len, cells); c3o_leave(); return r_;
}
