/* c3o_internal.h -- ORACLE private declarations shared between oracle translation units. */
#ifndef C3O_INTERNAL_H
#define C3O_INTERNAL_H
#include "c3o.h"
#include "c3o_graph.h"

/* full MSA state of one read (c3o_poa.c) */
typedef struct {
  c3o_graph g; int n; int** path; int* lens;
  int* cons_nodes; int cons_len;
} c3o_poa_state;

int c3o_poa_build(const char* const* seqs, const int* lens, int n, const c3o_params* P,
                  c3o_poa_state* st, int64_t* cells);
void c3o_poa_free(c3o_poa_state* st);
int c3o_poa_columns(const c3o_graph* g, int* col);
int c3o_poa_make_consensus(c3o_poa_state* st);

/* pairwise consensus that also reports the MSA column of every emitted base (c3o_pairwise.c) */
int c3o_pairwise_consensus_cols(const char* A, const char* B, int n,
                                const char* subA, int lenA, const char* qualA,
                                const char* subB, int lenB, const char* qualB,
                                char* out, int cap, int* out_col);
#endif
