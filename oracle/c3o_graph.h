/*
 * c3o_graph.h -- ORACLE (test infrastructure): partial-order graph shared by the
 * abPOA-style MSA stage (c3o_poa.c) and the racon-style window polish (c3o_polish.c).
 *
 * Topological order is maintained incrementally: aligned ("mismatch
 * sibling") nodes form a contiguous block in the order; a block is one MSA column.
 */
#ifndef C3O_GRAPH_H
#define C3O_GRAPH_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "c3o_mem.h"

#define C3O_NEG  (-(1 << 28))   /* out-of-band / unreachable score */
#define C3O_NEG2 (-(1 << 30))   /* "no left neighbour" for the horizontal-gap state */

typedef struct {
  int n, cap, K;              /* nodes, node capacity, edge slots per node */
  uint8_t* base;
  int *n_in, *in_from, *in_w;
  int *n_out, *out_to, *out_w;
  int* grp;                   /* group representative (first node of the aligned block) */
  int* ncov;                  /* sequences passing through the node */
  int *order, *index;         /* order[idx]=node, index[node]=idx; n entries */
  int *gfirst, *glast;        /* per representative: first/last order index of its block */
} c3o_graph;

static inline void c3o_graph_init(c3o_graph* g, int cap, int K) {
  memset(g, 0, sizeof(*g));
  g->cap = cap; g->K = K;
  g->base = (uint8_t*)calloc((size_t)cap, 1);
  g->n_in = (int*)calloc((size_t)cap, sizeof(int));
  g->n_out = (int*)calloc((size_t)cap, sizeof(int));
  g->in_from = (int*)malloc(sizeof(int) * (size_t)cap * K);
  g->in_w = (int*)malloc(sizeof(int) * (size_t)cap * K);
  g->out_to = (int*)malloc(sizeof(int) * (size_t)cap * K);
  g->out_w = (int*)malloc(sizeof(int) * (size_t)cap * K);
  g->grp = (int*)malloc(sizeof(int) * (size_t)cap);
  g->ncov = (int*)calloc((size_t)cap, sizeof(int));
  g->order = (int*)malloc(sizeof(int) * (size_t)cap);
  g->index = (int*)malloc(sizeof(int) * (size_t)cap);
  g->gfirst = (int*)malloc(sizeof(int) * (size_t)cap);
  g->glast = (int*)malloc(sizeof(int) * (size_t)cap);
}
static inline void c3o_graph_free(c3o_graph* g) {
  free(g->base); free(g->n_in); free(g->n_out); free(g->in_from); free(g->in_w);
  free(g->out_to); free(g->out_w); free(g->grp); free(g->ncov); free(g->order);
  free(g->index); free(g->gfirst); free(g->glast);
}
static inline int c3o_graph_new_node(c3o_graph* g, int base) {
  int v = g->n++;
  g->base[v] = (uint8_t)base; g->n_in[v] = g->n_out[v] = 0; g->grp[v] = v; g->ncov[v] = 0;
  return v;
}
/* add weight w to edge u->v (create at the end of both lists if missing) */
static inline void c3o_graph_add_edge(c3o_graph* g, int u, int v, int w) {
  int K = g->K;
  for (int k = 0; k < g->n_out[u]; ++k)
    if (g->out_to[u * K + k] == v) {
      g->out_w[u * K + k] += w;
      for (int t = 0; t < g->n_in[v]; ++t)
        if (g->in_from[v * K + t] == u) { g->in_w[v * K + t] += w; break; }
      return;
    }
  g->out_to[u * K + g->n_out[u]] = v; g->out_w[u * K + g->n_out[u]] = w; g->n_out[u]++;
  g->in_from[v * K + g->n_in[v]] = u; g->in_w[v * K + g->n_in[v]] = w; g->n_in[v]++;
}
/* recompute block extents from order/grp */
static inline void c3o_graph_blocks(c3o_graph* g) {
  for (int i = 0; i < g->n; ++i) { g->gfirst[i] = 1 << 30; g->glast[i] = -1; }
  for (int i = 0; i < g->n; ++i) {
    int r = g->grp[g->order[i]];
    if (i < g->gfirst[r]) g->gfirst[r] = i;
    if (i > g->glast[r]) g->glast[r] = i;
  }
}
/* merge new nodes (ids new0..n-1, creation order, anchor[k] = OLD order index after which
 * node new0+k is placed) into the order.  n_old = number of nodes already ordered. */
static inline void c3o_graph_reorder(c3o_graph* g, int n_old, const int* anchor) {
  int n_new = g->n - n_old;
  int* neworder = (int*)malloc(sizeof(int) * (size_t)g->n);
  int k = 0, out = 0;
  while (k < n_new && anchor[k] < 0) { neworder[out++] = n_old + k; ++k; }  /* before everything */
  for (int i = 0; i < n_old; ++i) {
    neworder[out++] = g->order[i];
    while (k < n_new && anchor[k] == i) { neworder[out++] = n_old + k; ++k; }
  }
  memcpy(g->order, neworder, sizeof(int) * (size_t)g->n);
  for (int i = 0; i < g->n; ++i) g->index[g->order[i]] = i;
  free(neworder);
  c3o_graph_blocks(g);
}
#endif
