/*
 * c3o_pipeline.c -- ORACLE (test infrastructure, never shipped): the per-read driver.
 * Restates /root/reference/C3POa.py:110-173 (analyze_reads) and the dispatcher of
 * /root/reference/bin/determine_consensus.py:10-47.  Zero-repeat rescue (:106-136) is not
 * restated yet: repeats==0 ends in NO_CONSENSUS, which is also what the reference does whenever
 * rescue fails or -z is given.
 */
#include "c3o.h"
#include <stdlib.h>
#include <string.h>
#include <malloc.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "c3o_mem.h"

#define C3O_MAX_SUB 250

static int c3o_process_read_impl(const char* splint, int S, const char* seq, const char* qual, int L,
                     const c3o_params* P, c3o_read_result* r, char* cons, int cons_cap) {
  memset(r, 0, sizeof(*r));
  int half = (P->sg_window - 1) / 2;
  if (L < half + 1 || L < 2) { r->status = C3O_TOO_SHORT; return r->status; }
  int32_t* track = (int32_t*)malloc(sizeof(int32_t) * (size_t)L);
  r->cells_conk = c3o_conk(splint, S, seq, L, P->conk_match, P->conk_mismatch, P->conk_penalty, track);
  int cap = L / (P->mdistcutoff > 0 ? P->mdistcutoff : 1) + 4;
  int64_t* pk = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
  int np = c3o_call_peaks(track, L, P->mdistcutoff, P->sg_iters, P->sg_window, P->sg_order, pk, cap, NULL);
  free(track);
  if (np <= 0) { free(pk); r->status = C3O_NO_PEAKS; return r->status; }
  if (np > 255) { free(pk); r->status = C3O_ERR_LIMIT; return r->status; }
  c3o_split_info si;
  int ok = c3o_split(pk, np, S, L, r->peaks, r->sub_beg, r->sub_end, &si);
  free(pk);
  r->n_peaks = si.n_peaks; r->n_sub = si.n_sub;
  r->has_front = si.has_front; r->has_tail = si.has_tail; r->front_end = si.front_end; r->tail_beg = si.tail_beg;
  if (!ok) { r->status = C3O_NO_PEAKS; return r->status; }
  if (si.n_sub == 0) {
    /* determine_consensus.py:14-18: zero-repeat rescue, accepted when len >= mdistcutoff */
    if (P->zero && si.has_front && si.has_tail) {
      int64_t zc = 0;
      int zl = c3o_zero_repeats(seq, qual, si.front_end, seq + si.tail_beg, qual + si.tail_beg, L - si.tail_beg, P, cons, cons_cap, &zc);
      r->cells_poa += zc;
      if (zl > 0 && zl >= P->mdistcutoff) { r->cons_len = zl; r->status = C3O_OK; return r->status; }
    }
    r->status = C3O_NO_CONSENSUS; return r->status;
  }
  if (si.n_sub > C3O_MAX_SUB) { r->status = C3O_ERR_LIMIT; return r->status; }
  const char* subs[C3O_MAX_SUB]; const char* quals[C3O_MAX_SUB]; int lens[C3O_MAX_SUB];
  for (int i = 0; i < si.n_sub; ++i) { subs[i] = seq + r->sub_beg[i]; quals[i] = qual + r->sub_beg[i]; lens[i] = r->sub_end[i] - r->sub_beg[i]; }
  int64_t cells[2] = {0, 0};
  int cl = c3o_determine_consensus(subs, quals, lens, si.n_sub,
                                   si.has_front ? seq : NULL, qual, si.front_end,
                                   si.has_tail ? seq + si.tail_beg : NULL, qual + si.tail_beg, L - si.tail_beg,
                                   P, cons, cons_cap, NULL, 0, NULL, cells);
  r->cells_poa = cells[0]; r->cells_polish = cells[1];
  r->cons_len = cl;
  r->status = cl > 0 ? C3O_OK : C3O_NO_CONSENSUS;
  return r->status;
}

int c3o_process_batch(const char* splint_fwd, const char* splint_rc, int S,
                      const char* seqs, const char* quals, const int64_t* off, int n,
                      const char* strand, const c3o_params* P, int threads,
                      c3o_read_result* results, char* cons, const int64_t* cons_off) {
  /* keep multi-MB DP buffers on the heap: with the default thresholds every malloc/free of them is
   * an mmap/munmap, which serialises many-core runs on the kernel's address-space lock */
  if (getenv("C3O_MALLOPT")) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_ARENA_MAX, 1024);
  }
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < n; ++i) {
    int L = (int)(off[i + 1] - off[i]);
    if (strand[i] != '+' && strand[i] != '-') { memset(&results[i], 0, sizeof(results[i])); results[i].status = C3O_NOT_ASSIGNED; continue; }
    const char* sp = strand[i] == '-' ? splint_rc : splint_fwd;
    c3o_process_read(sp, S, seqs + off[i], quals + off[i], L, P, &results[i],
                     cons + cons_off[i], (int)(cons_off[i + 1] - cons_off[i]));
  }
  return 0;
}

int c3o_process_read(const char* splint, int S, const char* seq, const char* qual, int L,
                     const c3o_params* P, c3o_read_result* r, char* cons, int cons_cap) {
  c3o_enter(); int r_ = c3o_process_read_impl(splint, S, seq, qual, L, P, r, cons, cons_cap); c3o_leave(); return r_;
}
