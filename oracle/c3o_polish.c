/*
 * c3o_polish.c -- ORACLE (test infrastructure, never shipped): draft construction and
 * windowed quality-weighted POA polish.
 *
 * Restates /root/reference/bin/determine_consensus.py:29-99:
 *   :31-47  draft = subread (1 repeat) | pairwise_consensus of the 2-row abPOA MSA | abPOA consensus
 *   :56-82  mappy overlaps of every kept + dangling subread against the draft (PAF)
 *   :92     racon <subreads.fastq> <overlaps.paf> <draft.fasta> -q 5 -t 1
 * mappy/minimap2 and racon (spoa + edlib inside) are NOT vendored in /root/reference and not
 * installed here.  The restatement follows racon's published pipeline (window 500, layers cut at
 * window boundaries from the base-level alignment, -q 5 mean-quality filter, >=3 sequences per
 * window, global linear-gap POA 3/-5/-4 with quality weights, sub-graph alignment for layers that
 * do not span the window, heaviest-bundle consensus, TGS coverage trim).  Overlap coordinates come
 * from the POA path of each kept subread and from a banded anchored extension
 * alignment for the dangling pieces.  **parity with mappy+racon: unpinned.**
 */
#include "c3o.h"
#include "c3o_graph.h"
#include "c3o_internal.h"
#include <limits.h>
#include <stdio.h>
#include "c3o_mem.h"

static const char ACGT[] = "ACGT";

/* ---------- checker hook (tests/test_independent_checkers.py, tools/band_model.py) ----------
 * When capture is on, every window alignment of this thread appends one record of int32 words (system heap, not the
 * arena):  [W_MAGIC, window serial, layer serial in the window, blen, begin, end, full, n, Q, score, end node, n_ops]
 * then base[n], grp[n] (aligned-block representative), order[n], per node (n_in, in_from...), mask[n] (all 1 when full), query codes[Q], ops as (node, q) pairs. */
#undef malloc
#undef realloc
#undef free
static __thread int32_t* g_cap = NULL;
static __thread int64_t g_cap_n = 0, g_cap_cap = 0;
static __thread int g_cap_on = 0, g_cap_win = 0, g_cap_layer = 0, g_cap_blen = 0, g_cap_begin = 0, g_cap_end = 0;
static void cap_push(int32_t v) {
  if (g_cap_n == g_cap_cap) { g_cap_cap = g_cap_cap ? 2 * g_cap_cap : (1 << 16); g_cap = (int32_t*)realloc(g_cap, sizeof(int32_t) * (size_t)g_cap_cap); }
  g_cap[g_cap_n++] = v;
}
void c3o_win_capture(int on) { g_cap_on = on; g_cap_n = 0; g_cap_win = 0; if (!on) { free(g_cap); g_cap = NULL; g_cap_cap = 0; } }
int64_t c3o_win_capture_get(int32_t* out, int64_t cap) {
  if (out) for (int64_t i = 0; i < g_cap_n && i < cap; ++i) out[i] = g_cap[i];
  return g_cap_n;
}
#define malloc(n) c3o_alloc(n)
#define realloc(p, n) c3o_regrow(p, n)
#define free(p) c3o_release(p)

typedef struct { int* node; int* q; int n, cap; } oplist;
static void op_push(oplist* o, int node, int q) {
  if (o->n == o->cap) {
    o->cap = o->cap ? 2 * o->cap : 2048;
    o->node = (int*)realloc(o->node, sizeof(int) * (size_t)o->cap);
    o->q = (int*)realloc(o->q, sizeof(int) * (size_t)o->cap);
  }
  o->node[o->n] = node; o->q[o->n] = q; o->n++;
}

/* ---------- dangling piece -> draft: banded anchored extension ---------- */
/* piece p[0..n) is aligned from the corner (0,0) against d[0..C); the alignment ends at the
 * best-scoring cell (first max in row-major order).  tpos[k] = aligned draft position or -1. */
static int64_t extend_align(const uint8_t* p, int n, const uint8_t* d, int C, const c3o_params* P,
                            int* tpos) {
  const int W = P->dang_band, mt = P->pol_match, mm = P->pol_mismatch, g = P->pol_gap;
  const int bw = 2 * W + 1;
  for (int k = 0; k < n; ++k) tpos[k] = -1;
  int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1) * bw);
  uint8_t* D = (uint8_t*)malloc((size_t)(n + 1) * bw);
#define HB(i, j) H[(size_t)(i) * bw + ((j) - (i) + W)]
#define DB(i, j) D[(size_t)(i) * bw + ((j) - (i) + W)]
#define INB(i, j) ((j) >= 0 && (j) <= C && (j) - (i) >= -W && (j) - (i) <= W)
  int64_t cells = 0;
  int32_t best = 0; int bi = 0, bj = 0;
  for (int i = 0; i <= n; ++i) {
    int jlo = i - W < 0 ? 0 : i - W, jhi = i + W > C ? C : i + W;
    for (int j = jlo; j <= jhi; ++j) {
      ++cells;
      if (i == 0) { HB(i, j) = j * g; DB(i, j) = 2; continue; }
      int32_t b = INT_MIN; int dir = 0;
      if (j > 0 && INB(i - 1, j - 1)) { b = HB(i - 1, j - 1) + (p[i - 1] == d[j - 1] ? mt : mm); dir = 0; }
      if (INB(i - 1, j)) { int32_t u = HB(i - 1, j) + g; if (u > b) { b = u; dir = 1; } }
      if (j > jlo) { int32_t l = HB(i, j - 1) + g; if (l > b) { b = l; dir = 2; } }
      HB(i, j) = b; DB(i, j) = (uint8_t)dir;
      if (j > 0 && b > best) { best = b; bi = i; bj = j; }
    }
  }
  if (best > 0) {
    int i = bi, j = bj;
    while (i > 0 || j > 0) {
      int dir = DB(i, j);
      if (dir == 0) { tpos[i - 1] = j - 1; --i; --j; }
      else if (dir == 1) --i;
      else --j;
    }
  }
  free(H); free(D);
#undef HB
#undef DB
#undef INB
  return cells;
}

/* ---------- window POA (spoa-style, linear gap, global) ---------- */
typedef struct {
  const uint8_t* seq; const char* qual; int len;   /* segment */
  int begin, end;                                  /* backbone positions (inclusive) */
} wlayer;

/* mask of the sub-graph between backbone nodes b..e (spoa Graph::subgraph semantics) */
static void subgraph_mask(const c3o_graph* g, int b, int e, char* mask) {
  int K = g->K;
  memset(mask, 0, (size_t)g->n);
  int i = g->glast[g->grp[e]];
  while (i >= 0) {
    int r = g->grp[g->order[i]];
    int f = g->gfirst[r], l = g->glast[r];
    int seed = 0;
    for (int t = f; t <= l && !seed; ++t) {
      int x = g->order[t];
      if (x < b) continue;
      if (x == e) { seed = 1; break; }
      for (int k = 0; k < g->n_out[x]; ++k) if (mask[g->out_to[x * K + k]]) { seed = 1; break; }
    }
    if (seed) for (int t = f; t <= l; ++t) { int x = g->order[t]; if (x >= b) mask[x] = 1; }
    i = f - 1;
  }
}

static void win_align(const c3o_graph* g, const char* mask, const uint8_t* q, int Q,
                      const c3o_params* P, oplist* ops, int64_t* cells) {
  const int K = g->K, n = g->n, W = Q + 1;
  const int mt = P->pol_match, mm = P->pol_mismatch, gp = P->pol_gap;
  int* rowof = (int*)malloc(sizeof(int) * (size_t)n);
  int* rows = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  int R = 0;
  for (int i = 0; i < n; ++i) {
    int v = g->order[i];
    if (!mask || mask[v]) { rows[++R] = v; rowof[v] = R; } else rowof[v] = -1;
  }
  int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R + 1) * W);
  uint16_t* D = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)(R + 1) * W);
  for (int j = 0; j <= Q; ++j) { H[j] = j * gp; D[j] = 2; }
  *cells += (int64_t)(R + 1) * W;
  int pr[264], pk[264];
  for (int r = 1; r <= R; ++r) {
    int v = rows[r];
    int np = 0;
    for (int k = 0; k < g->n_in[v]; ++k) {
      int u = g->in_from[v * K + k];
      if (rowof[u] >= 0) { pr[np] = rowof[u]; pk[np] = k; ++np; }
    }
    if (np == 0) { pr[0] = 0; pk[0] = 0x3fff; np = 1; }
    int32_t* h = H + (size_t)r * W; uint16_t* d = D + (size_t)r * W;
    for (int j = 0; j <= Q; ++j) {
      int32_t b = INT_MIN; int dir = 0;
      if (j > 0) {
        int s = (g->base[v] == q[j - 1]) ? mt : mm;
        for (int t = 0; t < np; ++t) {
          int32_t c = H[(size_t)pr[t] * W + j - 1] + s;
          if (c > b) { b = c; dir = 0 | (pk[t] << 2); }
        }
      }
      for (int t = 0; t < np; ++t) {
        int32_t c = H[(size_t)pr[t] * W + j] + gp;
        if (c > b) { b = c; dir = 1 | (pk[t] << 2); }
      }
      if (j > 0) { int32_t c = h[j - 1] + gp; if (c > b) { b = c; dir = 2; } }
      h[j] = b; d[j] = (uint16_t)dir;
    }
  }
  /* end row: masked nodes without masked successors, first max in order */
  int br = -1; int32_t bs = INT_MIN;
  for (int r = 1; r <= R; ++r) {
    int v = rows[r], has = 0;
    for (int k = 0; k < g->n_out[v]; ++k) if (rowof[g->out_to[v * K + k]] >= 0) { has = 1; break; }
    if (has) continue;
    int32_t s = H[(size_t)r * W + Q];
    if (s > bs) { bs = s; br = r; }
  }
  oplist rv; memset(&rv, 0, sizeof(rv));
  int r = br < 0 ? 0 : br, j = Q;
  while (r > 0 || j > 0) {
    int dir = D[(size_t)r * W + j], ty = dir & 3, k = dir >> 2;
    if (ty == 2) { op_push(&rv, -1, j - 1); --j; continue; }
    int v = rows[r];
    int nr = (k == 0x3fff) ? 0 : rowof[g->in_from[v * K + k]];
    if (ty == 0) { op_push(&rv, v, j - 1); --j; } else op_push(&rv, v, -1);
    r = nr;
  }
  for (int k = rv.n - 1; k >= 0; --k) op_push(ops, rv.node[k], rv.q[k]);
  if (g_cap_on) {
    cap_push(0x57494e44); cap_push(g_cap_win); cap_push(g_cap_layer); cap_push(g_cap_blen); cap_push(g_cap_begin); cap_push(g_cap_end);
    cap_push(mask ? 0 : 1); cap_push(n); cap_push(Q); cap_push(br < 0 ? INT_MIN : bs); cap_push(br < 0 ? -1 : rows[br]); cap_push(rv.n);
    for (int v = 0; v < n; ++v) cap_push(g->base[v]);
    for (int v = 0; v < n; ++v) cap_push(g->grp[v]);
    for (int i = 0; i < n; ++i) cap_push(g->order[i]);
    for (int v = 0; v < n; ++v) { cap_push(g->n_in[v]); for (int k = 0; k < g->n_in[v]; ++k) cap_push(g->in_from[v * K + k]); }
    for (int v = 0; v < n; ++v) cap_push(mask ? mask[v] : 1);
    for (int j = 0; j < Q; ++j) cap_push(q[j]);
    for (int k = rv.n - 1; k >= 0; --k) { cap_push(rv.node[k]); cap_push(rv.q[k]); }
  }
  free(rv.node); free(rv.q); free(rowof); free(rows); free(H); free(D);
}

static void win_fuse(c3o_graph* g, const oplist* ops, const uint8_t* q, const char* qual, int Q) {
  int n_old = g->n;
  int* anchor = (int*)malloc(sizeof(int) * (size_t)(Q + 1));
  int n_new = 0, prev = -1, prev_w = 0, cur_anchor = -1;
  for (int k = 0; k < ops->n; ++k) {
    int v = ops->node[k], qp = ops->q[k];
    if (qp < 0) continue;
    int c = q[qp], t, w = qual ? (int)(unsigned char)qual[qp] - 33 : 0;
    if (v >= 0) {
      int r = g->grp[v];
      cur_anchor = g->glast[r];
      t = -1;
      if (g->base[v] == c) t = v;
      else
        for (int i = g->gfirst[r]; i <= g->glast[r]; ++i)
          if (g->base[g->order[i]] == c) { t = g->order[i]; break; }
      if (t < 0) { t = c3o_graph_new_node(g, c); g->grp[t] = r; anchor[n_new++] = cur_anchor; }
    } else { t = c3o_graph_new_node(g, c); anchor[n_new++] = cur_anchor; }
    if (prev >= 0) c3o_graph_add_edge(g, prev, t, prev_w + w);
    g->ncov[t]++;
    prev = t; prev_w = w;
  }
  c3o_graph_reorder(g, n_old, anchor);
  free(anchor);
}

/* spoa Graph::traverse_heaviest_bundle + branch_completion */
static int win_consensus(const c3o_graph* g, int* cons) {
  const int K = g->K, n = g->n;
  int64_t* score = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  int* pred = (int*)malloc(sizeof(int) * (size_t)n);
  for (int v = 0; v < n; ++v) { score[v] = -1; pred[v] = -1; }
  int max_id = 0;
  for (int i = 0; i < n; ++i) {
    int v = g->order[i];
    for (int k = 0; k < g->n_in[v]; ++k) {
      int u = g->in_from[v * K + k]; int64_t w = g->in_w[v * K + k];
      if (score[v] < w || (score[v] == w && score[pred[v]] <= score[u])) { score[v] = w; pred[v] = u; }
    }
    if (pred[v] != -1) score[v] += score[pred[v]];
    if (score[max_id] < score[v]) max_id = v;
  }
  while (g->n_out[max_id] > 0) {            /* branch completion */
    int v = max_id;
    for (int k = 0; k < g->n_out[v]; ++k) {
      int t = g->out_to[v * K + k];
      for (int e = 0; e < g->n_in[t]; ++e) { int u = g->in_from[t * K + e]; if (u != v) score[u] = -1; }
    }
    int64_t ms = 0; int mid = -1;
    for (int i = g->index[v] + 1; i < n; ++i) {
      int x = g->order[i];
      score[x] = -1; pred[x] = -1;
      for (int k = 0; k < g->n_in[x]; ++k) {
        int u = g->in_from[x * K + k]; int64_t w = g->in_w[x * K + k];
        if (score[u] == -1) continue;
        if (score[x] < w || (score[x] == w && score[pred[x]] <= score[u])) { score[x] = w; pred[x] = u; }
      }
      if (pred[x] != -1) score[x] += score[pred[x]];
      if (ms < score[x]) { ms = score[x]; mid = x; }
    }
    if (mid < 0) break;                     /* guard: no positive continuation */
    max_id = mid;
  }
  int nc = 0;
  for (int v = max_id; v != -1; v = pred[v]) cons[nc++] = v;
  for (int a = 0, b = nc - 1; a < b; ++a, --b) { int t = cons[a]; cons[a] = cons[b]; cons[b] = t; }
  free(score); free(pred);
  return nc;
}

/* racon Window::generate_consensus.  returns consensus length; *polished = 0/1 */
static int window_polish(const uint8_t* bb, int blen, const wlayer* L, int nl, int tgs,
                         const c3o_params* P, char* out, int* polished, int64_t* cells) {
  *polished = 0;
  if (nl + 1 < 3) { for (int i = 0; i < blen; ++i) out[i] = ACGT[bb[i]]; return blen; }
  int cap = blen;
  for (int i = 0; i < nl; ++i) cap += L[i].len;
  c3o_graph g; c3o_graph_init(&g, cap + 1, nl + 2);
  for (int i = 0; i < blen; ++i) {
    int v = c3o_graph_new_node(&g, bb[i]);
    g.order[i] = v; g.index[v] = i; g.ncov[v] = 1;
    if (i) c3o_graph_add_edge(&g, i - 1, i, 0);
  }
  c3o_graph_blocks(&g);
  /* stable sort of layers by begin position */
  int* rank = (int*)malloc(sizeof(int) * (size_t)nl);
  for (int i = 0; i < nl; ++i) rank[i] = i;
  for (int i = 1; i < nl; ++i) { int v = rank[i], k = i - 1; while (k >= 0 && L[rank[k]].begin > L[v].begin) { rank[k + 1] = rank[k]; --k; } rank[k + 1] = v; }
  int offset = (int)(0.01 * blen);
  char* mask = (char*)malloc((size_t)cap + 1);
  for (int t = 0; t < nl; ++t) {
    const wlayer* l = &L[rank[t]];
    oplist ops; memset(&ops, 0, sizeof(ops));
    g_cap_layer = t; g_cap_blen = blen; g_cap_begin = l->begin; g_cap_end = l->end;
    if (l->begin < offset && l->end > blen - offset) win_align(&g, NULL, l->seq, l->len, P, &ops, cells);
    else { subgraph_mask(&g, l->begin, l->end, mask); win_align(&g, mask, l->seq, l->len, P, &ops, cells); }
    win_fuse(&g, &ops, l->seq, l->qual, l->len);
    free(ops.node); free(ops.q);
  }
  ++g_cap_win;
  int* cons = (int*)malloc(sizeof(int) * (size_t)g.n);
  int nc = win_consensus(&g, cons);
  int b = 0, e = nc - 1;
  if (tgs) {
    int avg = nl / 2;                       /* (sequences.size() - 1) / 2 */
    for (; b < nc; ++b) if (g.ncov[cons[b]] >= avg) break;
    for (; e >= 0; --e) if (g.ncov[cons[e]] >= avg) break;
    if (b >= e) { b = 0; e = nc - 1; }      /* "might be chimeric": left untrimmed */
  }
  int o = 0;
  for (int i = b; i <= e; ++i) out[o++] = ACGT[g.base[cons[i]]];
  *polished = 1;
  free(cons); free(mask); free(rank); c3o_graph_free(&g);
  return o;
}

/* ---------- determine_consensus (repeats >= 1) ---------- */
typedef struct { const char* seq; const char* qual; int len; int* tpos; uint8_t* code; } layer;

static int c3o_determine_consensus_impl(const char* const* subs, const char* const* quals,
                            const int* lens, int n,
                            const char* front, const char* front_q, int front_len,
                            const char* tail, const char* tail_q, int tail_len,
                            const c3o_params* P, char* out, int cap,
                            char* draft_out, int draft_cap, int* draft_len, int64_t* cells) {
  int64_t cl_poa = 0, cl_pol = 0;
  if (draft_len) *draft_len = 0;
  if (n < 1) return 0;
  int maxlen = 0, total = 0;
  for (int i = 0; i < n; ++i) { if (lens[i] > maxlen) maxlen = lens[i]; total += lens[i]; }
  char* draft = (char*)malloc((size_t)total + 16);
  int C = 0;
  int nl = n + (front ? 1 : 0) + (tail ? 1 : 0);
  layer* Ls = (layer*)calloc((size_t)nl, sizeof(layer));
  for (int i = 0; i < n; ++i) { Ls[i].seq = subs[i]; Ls[i].qual = quals[i]; Ls[i].len = lens[i]; }
  int li = n;
  int fi = -1, ti = -1;
  if (front) { fi = li; Ls[li].seq = front; Ls[li].qual = front_q; Ls[li].len = front_len; ++li; }
  if (tail) { ti = li; Ls[li].seq = tail; Ls[li].qual = tail_q; Ls[li].len = tail_len; ++li; }
  for (int i = 0; i < nl; ++i) {
    Ls[i].tpos = (int*)malloc(sizeof(int) * (size_t)(Ls[i].len + 1));
    Ls[i].code = (uint8_t*)malloc((size_t)Ls[i].len + 1);
    for (int k = 0; k < Ls[i].len; ++k) { Ls[i].tpos[k] = -1; Ls[i].code[k] = (uint8_t)c3o_code(Ls[i].seq[k]); }
  }
  int rc = 0;
  /* --- draft + kept-subread coordinates --- */
  if (n == 1) {
    C = lens[0];
    for (int k = 0; k < C; ++k) { draft[k] = ACGT[Ls[0].code[k]]; Ls[0].tpos[k] = k; }
  } else {
    c3o_poa_state st;
    rc = c3o_poa_build(subs, lens, n, P, &st, &cl_poa);
    if (rc == 0) {
      int* col = (int*)malloc(sizeof(int) * (size_t)st.g.n);
      int ncol = c3o_poa_columns(&st.g, col);
      int* col2t = (int*)malloc(sizeof(int) * (size_t)(ncol + 1));
      for (int c = 0; c < ncol; ++c) col2t[c] = -1;
      if (n == 2) {
        char* rows = (char*)malloc((size_t)ncol * 2 + 2);
        memset(rows, '-', (size_t)ncol * 2);
        for (int s = 0; s < 2; ++s)
          for (int k = 0; k < lens[s]; ++k)
            rows[(size_t)s * ncol + col[st.path[s][k]]] = ACGT[st.g.base[st.path[s][k]]];
        int* ocol = (int*)malloc(sizeof(int) * (size_t)(ncol + 1));
        /* the reference feeds the ORIGINAL subread strings as dict keys (consensus.py:13); rows
         * minus gaps equal the code-normalised subreads */
        char* sa = (char*)malloc((size_t)lens[0] + 1); char* sb = (char*)malloc((size_t)lens[1] + 1);
        for (int k = 0; k < lens[0]; ++k) sa[k] = ACGT[Ls[0].code[k]];
        for (int k = 0; k < lens[1]; ++k) sb[k] = ACGT[Ls[1].code[k]];
        C = c3o_pairwise_consensus_cols(rows, rows + ncol, ncol, sa, lens[0], quals[0], sb, lens[1], quals[1],
                                        draft, total, ocol);
        for (int t = 0; t < C; ++t) col2t[ocol[t]] = t;
        free(rows); free(ocol); free(sa); free(sb);
      } else {
        C = c3o_poa_make_consensus(&st);
        for (int t = 0; t < C; ++t) { draft[t] = ACGT[st.g.base[st.cons_nodes[t]]]; col2t[col[st.cons_nodes[t]]] = t; }
      }
      for (int s = 0; s < n; ++s)
        for (int k = 0; k < lens[s]; ++k) Ls[s].tpos[k] = col2t[col[st.path[s][k]]];
      free(col); free(col2t);
    }
    c3o_poa_free(&st);
  }
  int olen = 0;
  if (rc == 0 && C > 0) {
    if (draft_out && C <= draft_cap) { memcpy(draft_out, draft, (size_t)C); if (draft_len) *draft_len = C; }
    uint8_t* dc = (uint8_t*)malloc((size_t)C + 1);
    for (int t = 0; t < C; ++t) dc[t] = (uint8_t)c3o_code(draft[t]);
    /* --- dangling pieces --- */
    if (ti >= 0) cl_pol += extend_align(Ls[ti].code, Ls[ti].len, dc, C, P, Ls[ti].tpos);
    if (fi >= 0) {
      int fl = Ls[fi].len;
      uint8_t* rp = (uint8_t*)malloc((size_t)fl + 1); uint8_t* rd_ = (uint8_t*)malloc((size_t)C + 1);
      int* rt = (int*)malloc(sizeof(int) * (size_t)(fl + 1));
      for (int k = 0; k < fl; ++k) rp[k] = Ls[fi].code[fl - 1 - k];
      for (int t = 0; t < C; ++t) rd_[t] = dc[C - 1 - t];
      cl_pol += extend_align(rp, fl, rd_, C, P, rt);
      for (int k = 0; k < fl; ++k) Ls[fi].tpos[fl - 1 - k] = rt[k] < 0 ? -1 : C - 1 - rt[k];
      free(rp); free(rd_); free(rt);
    }
    /* --- window type (racon: mean read length <= 1000 -> NGS) --- */
    long tl = 0;
    for (int i = 0; i < nl; ++i) tl += Ls[i].len;
    int tgs = tl > 1000L * nl;
    /* --- cut layers at window boundaries --- */
    const int WL = P->pol_window;
    int nwin = (C + WL - 1) / WL;
    wlayer* wl = (wlayer*)malloc(sizeof(wlayer) * (size_t)nwin * nl);
    int* wn = (int*)calloc((size_t)nwin, sizeof(int));
    for (int i = 0; i < nl; ++i) {
      const layer* l = &Ls[i];
      int qf = -1, ql = -1, tf = -1, tlast = -1;
      for (int k = 0; k < l->len; ++k) if (l->tpos[k] >= 0) { if (qf < 0) { qf = k; tf = l->tpos[k]; } ql = k; tlast = l->tpos[k]; }
      if (qf < 0) continue;
      int qs = ql + 1 - qf, ts = tlast + 1 - tf;
      double err = 1.0 - (double)(qs < ts ? qs : ts) / (double)(qs > ts ? qs : ts);
      if (err > 0.3) continue;               /* racon error threshold */
      int k = qf;
      while (k <= ql) {
        while (k <= ql && l->tpos[k] < 0) ++k;
        if (k > ql) break;
        int w = l->tpos[k] / WL;
        int f_t = l->tpos[k], f_q = k, l_t = f_t, l_q = k;
        int kk = k;
        for (; kk <= ql; ++kk) {
          if (l->tpos[kk] < 0) continue;
          if (l->tpos[kk] / WL != w) break;
          l_t = l->tpos[kk]; l_q = kk;
        }
        k = kk;
        int seglen = l_q + 1 - f_q;
        if ((double)seglen < 0.02 * WL) continue;
        long qsum = 0;
        for (int x = f_q; x <= l_q; ++x) qsum += (int)(unsigned char)l->qual[x] - 33;
        if (qsum < (long)P->pol_q * seglen) continue;
        wlayer* dst = &wl[(size_t)w * nl + wn[w]++];
        dst->seq = l->code + f_q; dst->qual = l->qual + f_q; dst->len = seglen;
        dst->begin = f_t - w * WL; dst->end = l_t - w * WL;
      }
    }
    /* --- per-window consensus, concatenated --- */
    int any = 0;
    char* wout = (char*)malloc((size_t)(WL + maxlen + total + 16));
    for (int w = 0; w < nwin; ++w) {
      int ws = w * WL, blen = (ws + WL <= C) ? WL : C - ws, pol = 0;
      int wc = window_polish(dc + ws, blen, &wl[(size_t)w * nl], wn[w], tgs, P, wout, &pol, &cl_pol);
      any |= pol;
      if (olen + wc <= cap) memcpy(out + olen, wout, (size_t)wc);
      olen += wc;
    }
    if (!any || olen > cap) olen = 0;        /* racon drops unpolished targets (no -u) */
    free(wout); free(wl); free(wn); free(dc);
  }
  for (int i = 0; i < nl; ++i) { free(Ls[i].tpos); free(Ls[i].code); }
  free(Ls); free(draft);
  if (cells) { cells[0] += cl_poa; cells[1] += cl_pol; }
  return rc ? 0 : olen;
}

int c3o_determine_consensus(const char* const* subs, const char* const* quals,
                            const int* lens, int n,
                            const char* front, const char* front_q, int front_len,
                            const char* tail, const char* tail_q, int tail_len,
                            const c3o_params* P, char* out, int cap,
                            char* draft_out, int draft_cap, int* draft_len, int64_t* cells) {
  c3o_enter(); int r_ = c3o_determine_consensus_impl(subs, quals, lens, n, front, front_q, front_len, tail, tail_q, tail_len, P, out, cap, draft_out, draft_cap, draft_len, cells); c3o_leave(); return r_;
}
