/*
 * c3o.h -- CPU ORACLE for the R2C2 consensus hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is the executable specification ("oracle") that the HIP
 * kernels in c3poa_amd/csrc are checked against bit-for-bit.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (c3poa_amd/) never links, imports or calls anything in here.
 *
 * PARITY STATUS:
 *   - savitzky_golay / call_peaks / rounding / subread split / pairwise_consensus /
 *     normalizeLen / record formatting are pinned against golden vectors captured
 *     from the reference's own Python (tests/golden/, generator committed).
 *   - conk, abPOA (pyabpoa 1.0.5), minimap2 (mappy) and racon are third-party
 *     dependencies that are NOT vendored in /root/reference and not installed in
 *     the build image; the reference ships no tests or fixtures for them.  Their
 *     restatements here follow the published algorithms with the parameters of
 *     the reference's call sites:  **parity unpinned** for those four stages.
 *
 * Reference call sites restated (file:line are relative to /root/reference):
 *   c3o_conk               C3POa.py:123            conk.conk(splint, seq, 20)
 *   c3o_savgol             bin/savitzky_golay.py:7-38
 *   c3o_call_peaks         bin/call_peaks.py:8-16  (+ scipy.signal.find_peaks semantics)
 *   c3o_split              C3POa.py:106-108,127-155
 *   c3o_poa_msa            bin/determine_consensus.py:30,34,43 (pyabpoa.msa_aligner(match=5).msa)
 *   c3o_pairwise_consensus bin/consensus.py:4-81
 *   c3o_polish             bin/determine_consensus.py:56-99 (mappy overlaps + racon -q 5 -t 1)
 *   c3o_process_read       C3POa.py:110-173 + bin/determine_consensus.py:10-104
 *   c3o_adapter_align      C3POa_postprocessing.py:229-264 (blat adapters vs consensus reads; parity unpinned)
 */
#ifndef C3O_H
#define C3O_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-read status codes (shared numbering with include/c3poa.h) */
enum {
  C3O_OK = 0,
  C3O_NOT_ASSIGNED = 1,   /* read has no splint assignment (C3POa.py:115) */
  C3O_NO_PEAKS = 2,       /* call_peaks returned [] or all peaks clipped (C3POa.py:125,131) */
  C3O_NO_CONSENSUS = 3,   /* repeats==0, or polish emitted nothing */
  C3O_TOO_SHORT = 4,      /* L < 41: smoothing window does not fit */
  C3O_ERR_LIMIT = 5       /* capacity limit hit (too many subreads, band arena) */
};

/* scoring / algorithm constants of the reference call sites */
typedef struct {
  int conk_match, conk_mismatch, conk_penalty;      /* 5, -4, 20 (C3POa.py:111) */
  int sg_iters, sg_window, sg_order;                /* 3, 41, 2 (C3POa.py:111) */
  int mdistcutoff;                                  /* -d, default 500 */
  int poa_match, poa_mismatch;                      /* 5 (determine_consensus.py:30), 4 */
  int poa_o1, poa_e1, poa_o2, poa_e2;               /* 4,2,24,1 abPOA defaults */
  int poa_band_b; double poa_band_f;                /* 10, 0.01 */
  int pol_match, pol_mismatch, pol_gap;             /* racon: 3,-5,-4 */
  int pol_window; int pol_q;                        /* 500, 5 (-q 5) */
  int dang_band;                                    /* dangling extension half band, 128 */
  int zero;                                         /* args.zero (C3POa.py:48): attempt the zero-repeat rescue */
  int zr_match, zr_mismatch, zr_gapo, zr_gape;      /* overlap finder: map-ont base scoring 2,4,4,2 */
  int zr_min_score, zr_max_cells;                   /* accept threshold (80 = 40 matches), DP size cap (16M cells) */
} c3o_params;

void c3o_default_params(c3o_params* p);

/* ---- stage probes ------------------------------------------------------- */

/* 2-bit code of an ASCII base: A/a=0 C/c=1 G/g=2 T/t/U/u=3, anything else 0 */
int c3o_code(char c);

/* splint x read local-alignment matrix (linear gap), summed per diagonal
 * d = j - i >= 0; track has L entries.  returns number of DP cells. */
int64_t c3o_conk(const char* splint, int S, const char* read, int L,
                 int match, int mismatch, int penalty, int32_t* track);

/* closed-form Savitzky-Golay smoothing coefficients (order 2 or 3), window odd */
int c3o_savgol_coeffs(int window, int order, double* c);
/* one pass; out has n entries; requires n > window/2 */
int c3o_savgol(const double* y, int n, int window, int order, double* out);

/* call_peaks: returns number of peaks (0 if gated), -1 if n too short.
 * smoothed (optional, n doubles) receives the 3x smoothed track. */
int c3o_call_peaks(const int32_t* scores, int n, int min_dist, int iters,
                   int window, int order, int64_t* peaks, int cap, double* smoothed);

/* scipy.signal.find_peaks(x, distance=, height=)[0] */
int c3o_find_peaks(const double* x, int n, double height, int distance,
                   int64_t* peaks, int cap);

int c3o_rounding(int x, int base);

/* subread split.  peaks_in are raw call_peaks indices; S = splint length.
 * outputs: shifted/clipped peaks, kept segments [sub_beg,sub_end), dangling flags.
 * returns number of shifted peaks kept (0 => NO_PEAKS). */
typedef struct {
  int n_peaks;            /* after shift+clip */
  int n_sub;              /* kept subreads */
  int has_front, has_tail;/* dangling pieces (front = read[:p0], tail = read[p_last:]) */
  int front_end, tail_beg;
} c3o_split_info;
int c3o_split(const int64_t* peaks_in, int n_in, int S, int L,
              int64_t* peaks_out, int* sub_beg, int* sub_end, c3o_split_info* info);

/* abPOA-style MSA.  seqs: n ASCII strings.  out_cons/out_msa as pyabpoa.
 * cons (cap bytes) receives consensus, msa receives n rows of msa_len chars each
 * (row-major, no terminators).  returns 0 on success. */
int c3o_poa_msa(const char* const* seqs, const int* lens, int n, const c3o_params* P,
                char* cons, int cons_cap, int* cons_len,
                char* msa, int64_t msa_cap, int* msa_len, int64_t* cells);

/* end-cell DP scores of the alignments of the last c3o_poa_msa call on this thread (checker hook); returns their number */
int c3o_poa_last_scores(int32_t* out, int cap);
/* checker hook of the window polish: capture every window alignment of this thread (graph, mask, query, end score, path);
 * record layout in c3o_polish.c.  _get returns the number of int32 words captured so far (copies up to cap of them). */
void c3o_win_capture(int on);
int64_t c3o_win_capture_get(int32_t* out, int64_t cap);

/* bin/consensus.py pairwise_consensus: rows have msa_len chars */
int c3o_pairwise_consensus(const char* rowA, const char* rowB, int msa_len,
                           const char* subA, int lenA, const char* qualA,
                           const char* subB, int lenB, const char* qualB,
                           char* out, int cap);
/* bin/consensus.py normalizeLen */
int c3o_normalize_len(const char* row, int msa_len, const char* qual, int qlen, char* out);

/* full determine_consensus for repeats>=1 (POA/pairwise draft + polish).
 * subs: kept subreads (seq+qual); front/tail may be NULL.  returns consensus length
 * (0 => nothing emitted). */
int c3o_determine_consensus(const char* const* subs, const char* const* quals,
                            const int* lens, int n,
                            const char* front, const char* front_q, int front_len,
                            const char* tail, const char* tail_q, int tail_len,
                            const c3o_params* P, char* out, int cap,
                            char* draft_out, int draft_cap, int* draft_len,
                            int64_t* cells);

/* zero-repeat rescue (bin/determine_consensus.py:106-136): d0 = read[:front_end], d1 = read[tail_beg:].
 * returns the stitched length (0 = no rescue). */
int c3o_zero_repeats(const char* d0, const char* q0, int n0, const char* d1, const char* q1, int n1,
                     const c3o_params* P, char* out, int cap, int64_t* cells);

/* adapter finder of the post-processing step (C3POa_postprocessing.py:229-264): out[12] = score, qStart, qEnd, tStart,
 * tEnd, matches, misMatches, qBaseInsert, tBaseInsert, qNumInsert, tNumInsert, read length.  rc = 1: strand '-'. */
int c3o_adapter_align(const char* read, int n, const char* adapter, int m, int rc, const c3o_params* P, int32_t* out);

/* whole per-read path.  returns status; on C3O_OK cons/cons_len are set. */
typedef struct {
  int status;
  int n_peaks; int64_t peaks[256];
  int n_sub; int sub_beg[256]; int sub_end[256];
  int has_front, has_tail, front_end, tail_beg;
  int cons_len;
  int64_t cells_conk, cells_poa, cells_polish;
} c3o_read_result;

int c3o_process_read(const char* splint, int S, const char* seq, const char* qual, int L,
                     const c3o_params* P, c3o_read_result* r, char* cons, int cons_cap);

/* batch (OpenMP over reads).  seqs/quals concatenated, off[n+1]; strand[i] '+'/'-'
 * picks splint_fwd / splint_rc.  cons_off[n+1] into cons arena (caller sized:
 * sum over reads of L bytes is always enough). */
int c3o_process_batch(const char* splint_fwd, const char* splint_rc, int S,
                      const char* seqs, const char* quals, const int64_t* off, int n,
                      const char* strand, const c3o_params* P, int threads,
                      c3o_read_result* results, char* cons, const int64_t* cons_off);

#ifdef __cplusplus
}
#endif
#endif
