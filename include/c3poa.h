/*
 * c3poa.h -- C ABI of the MI355X-native R2C2 consensus hot path (libc3poa_hip.so).
 *
 * The reference (rvolden/C3POa v2.2.3) has no FFI layer of its own: its seams are plain Python
 * call sites into un-vendored native dependencies.  Every entry point below names the reference
 * call site(s) it replaces (paths relative to the reference repository root); INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add at each site.
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types; all functions return 0 or a negative c3_err
 *   - one c3_handle per GPU, not thread-safe; create once per worker, never per batch
 *   - a per-read failure is a status code in c3_read_result, never a batch failure
 *   - sequences are ASCII; A/C/G/T/U in either case are coded 0..3, every other byte is coded
 *     as 'A' for alignment purposes (2-bit packing)
 *   - the library FAILS LOUDLY without a GPU: there is no CPU fallback anywhere in it
 */
#ifndef C3POA_H
#define C3POA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C3_MAX_PEAKS 256   /* peaks / kept subreads recorded per read */

typedef enum {
  C3_E_OK = 0,
  C3_E_NO_DEVICE = -1,
  C3_E_HIP = -2,
  C3_E_ARG = -3,
  C3_E_NOMEM = -4,
  C3_E_STATE = -5,
  C3_E_LIMIT = -6
} c3_err;

/* per-read status (same numbering as the oracle's) */
typedef enum {
  C3_ST_OK = 0,
  C3_ST_NOT_ASSIGNED = 1,  /* C3POa.py:115 */
  C3_ST_NO_PEAKS = 2,      /* C3POa.py:125,131 */
  C3_ST_NO_CONSENSUS = 3,  /* repeats == 0, or polish emitted nothing (determine_consensus.py:44-47,97-99) */
  C3_ST_TOO_SHORT = 4,     /* shorter than the smoothing half-window */
  C3_ST_LIMIT = 5          /* capacity limit: more than 250 kept subreads (peaks beyond C3_MAX_PEAKS), a polishing window graph of more than 65 534 nodes;
                              every other scratch of the path is sized from the batch and redone at worst-case size when it overflows */
} c3_status;

/* algorithm constants of the reference call sites; c3_default_config fills them */
typedef struct {
  int device;                                   /* HIP device ordinal */
  int conk_match, conk_mismatch, conk_penalty;  /* conk.conk(splint, seq, 20): C3POa.py:111,123 */
  int sg_iters, sg_window, sg_order;            /* call_peaks(scores, d, 3, 41, 2): C3POa.py:111,124 */
  int mdistcutoff;                              /* -d: C3POa.py:45 */
  int poa_match, poa_mismatch;                  /* poa.msa_aligner(match=5): determine_consensus.py:30 */
  int poa_o1, poa_e1, poa_o2, poa_e2;           /* abPOA defaults 4,2,24,1 */
  int poa_band_b; double poa_band_f;            /* abPOA defaults 10, 0.01 */
  int pol_match, pol_mismatch, pol_gap;         /* racon 3,-5,-4 */
  int pol_window, pol_q;                        /* racon window 500; -q 5: determine_consensus.py:92 */
  int dang_band;                                /* half band of the dangling-piece extension, 128 */
  int slots_poa, slots_win;                     /* resident wave slots (0 = auto) */
  int zero;                                     /* args.zero (C3POa.py:48-49): attempt the zero-repeat rescue (default 1) */
} c3_config;

typedef struct {
  int32_t status;
  int32_t n_peaks;                 /* after shift by len(splint)//2 and clip (C3POa.py:127-132) */
  int32_t n_sub;                   /* kept subreads = "repeats" (determine_consensus.py:12) */
  int32_t has_front, has_tail;     /* dangling pieces read[:front_end], read[tail_beg:] (C3POa.py:145-155) */
  int32_t front_end, tail_beg;
  int32_t cons_len, draft_len;
  int32_t n_win;
  int32_t peaks[C3_MAX_PEAKS];
  int32_t sub_beg[C3_MAX_PEAKS], sub_end[C3_MAX_PEAKS];   /* pure slices of the read (C3POa.py:141-144) */
} c3_read_result;
/* ARRAY TAILS ARE UNSPECIFIED: c3_batch_results copies only the used prefix of peaks / sub_beg / sub_end across PCIe, so in the
 * caller's records peaks[k] for k >= n_peaks and sub_beg[k] / sub_end[k] for k >= n_sub hold whatever the buffer held before
 * (never read them; set C3_FULL_RESULTS=1 in the environment to get whole records, e.g. when diffing raw buffers). */

/* kernel time of the last c3_batch_run, measured with hipEvents on the library's own stream */
typedef struct {
  float ms_pack, ms_conk, ms_peaks, ms_poa, ms_prep, ms_window, ms_stitch, ms_total;
  int64_t n_reads, n_bases, n_windows;
  int64_t cells_conk, cells_poa, cells_polish;
  int64_t n_poa_redo;      /* DISTINCT reads redone by a later POA pass: scratch (sized for the typical alignment) overflowed -> full-size pass, or handed to the last pass (n_poa_redo16 counts the visits of that pass; a read that took both is counted once here) */
  float ms_wall;           /* host wall time of the whole c3_batch_run call; ms_wall - ms_total = time the GPU waited for the host */
  float ms_host_worklist;  /* of which: building + uploading the POA work list on the host (timer starts AFTER the wait for k_conk / k_peaks) */
  float ms_alloc;          /* of which: growing device scratch buffers (only while batch shapes are still growing) */
  float ms_host_gap;       /* ms_wall - ms_total: time the GPU was not running one of the six timed kernels during the call */
  int64_t cells_polish_computed;   /* polish DP cells actually computed: banded layers fill 64*CB columns per row, a layer whose band
                                      certificate failed counts band + full matrix.  cells_polish stays the full-matrix count (= oracle) */
  int64_t n_band_layers, n_band_fallback;   /* window layers aligned in a band and accepted / redone unbanded after a failed certificate */
  int64_t n_band_mismatch;                  /* C3_DEBUG_BAND=verify only: accepted band layers whose traceback differs from the full matrix's (must be 0) */
  int64_t n_win_redo;                       /* windows with a layer beyond the first launch's DP scratch, redone by the full-size second launch of k_window */
  int64_t n_poa_redo16;                     /* of n_poa_redo: reads redone by the LAST POA pass (32-bit cells, a workgroup of eight waves per read) -- a score did not fit the 16-bit cells of the first passes, or the far arena of a pass overflowed (long subreads: a band that blew up goes straight there) */
  float ms_poa_tail;                        /* the last POA pass when it ran on a stream of its own BESIDE k_prep / k_window of the other reads (its reads are polished by a small tail afterwards); 0 when it ran in series (then it is part of ms_poa).  Not part of ms_total */
  float pad_;
} c3_timing;

typedef struct c3_handle c3_handle;

void c3_default_config(c3_config* cfg);
const char* c3_version(void);
/* GPUs visible to the process (0 without a GPU); -n of the CLI is clamped to it (C3POa.py:236 sized a process pool) */
int c3_device_count(void);
/* Creates the HIP context of device `device` on the calling thread and returns (hipSetDevice + an empty hipFree): the CLI calls it
 * on a thread of its own per GPU at start-up so that the workers' c3_create finds the context ready (C3POa.py:236-248 paid the
 * start-up of every worker process instead).  0 or a negative c3_status. */
int c3_warm_device(int device);

/* lifecycle.  Replaces the per-task worker process of C3POa.py:236 (mp.Pool, maxtasksperchild=1). */
int c3_create(const c3_config* cfg, c3_handle** out);
void c3_destroy(c3_handle* h);
const char* c3_last_error(const c3_handle* h);

/* splint table (C3POa.py:231-234: splint_dict[name] = [seq, revcomp(seq)]); the library makes the
 * reverse complements itself.  cat = concatenated ASCII, off[n+1]. */
int c3_set_splints(c3_handle* h, int n, const char* cat, const int64_t* off);

/* batch = the `reads` argument of analyze_reads (C3POa.py:110) in SoA form.
 *   seqs/quals: concatenated ASCII, off[n+1]; splint_id[i] = row of c3_set_splints;
 *   strand[i] = '+' / '-' (adapter_dict[name][1], C3POa.py:117-122), anything else = not assigned.
 * Copies to the device and packs to 2 bit; the caller may free its buffers on return. */
int c3_batch_upload(c3_handle* h, int n, const char* seqs, const char* quals, const int64_t* off,
                    const int16_t* splint_id, const char* strand);

/* double buffering: c3_batch_stage copies (and 2-bit packs) the NEXT batch on a second stream while the resident batch
 * is being processed; c3_batch_commit makes it resident once the results of the previous batch have been fetched.
 * seqs / quals must stay valid until c3_batch_commit returns; c3_batch_upload = stage + commit. */
int c3_batch_stage(c3_handle* h, int n, const char* seqs, const char* quals, const int64_t* off,
                   const int16_t* splint_id, const char* strand);
int c3_batch_commit(c3_handle* h);

/* overwrite the splint row / strand of the resident batch (e.g. with the output of c3_scan_splints) before c3_batch_run */
int c3_batch_assign(c3_handle* h, const int16_t* splint_id, const char* strand);

/* run the resident batch through the hot path.  The call returns when the batch is done: the stage sizes (work lists,
 * window count) are read back on the host between the kernels.  Overlap comes from c3_batch_stage on the second stream.
 * stages: bit0 conk, bit1 peaks+split, bit2 POA/draft, bit3 polish.  C3_STAGES_ALL = whole path
 * = one call of analyze_reads (C3POa.py:110-173) minus file I/O. */
#define C3_STAGE_CONK 1
#define C3_STAGE_PEAKS 2
#define C3_STAGE_POA 4
#define C3_STAGE_POLISH 8
#define C3_STAGES_ALL 15
int c3_batch_run(c3_handle* h, int stages);
int c3_batch_sync(c3_handle* h);

/* results of the resident batch.  cons receives the consensus bytes of read i at cons_off[i]
 * (cons_off[n+1] is written by the call; capacity cons_cap bytes; returns C3_E_LIMIT and the
 * needed size in cons_off[n] if too small).  Pass cons=NULL to fetch only the per-read records.
 * After the POA / polish stages only the used part of every record is copied: peaks[k >= n_peaks] and
 * sub_beg / sub_end[k >= n_sub] of the caller's records are then unspecified (left as they were). */
int c3_batch_results(c3_handle* h, c3_read_result* res, char* cons, int64_t cons_cap, int64_t* cons_off);
/* The same in two halves, so that the device->host copy runs beside the NEXT batch's kernels (the reference overlaps nothing:
 * analyze_reads writes its files before the worker takes the next group, C3POa.py:110-173).
 * c3_batch_results_snapshot (owner thread, after c3_batch_run) freezes the records and the compact consensus bytes in device
 * buffers of their own; afterwards the owner may c3_batch_commit and c3_batch_run the next batch.
 * c3_batch_results_fetch copies the snapshot into the caller's buffers (same arguments and C3_E_LIMIT rule as c3_batch_results)
 * and returns when they have landed; it touches nothing but the snapshot and MAY BE CALLED FROM ANOTHER THREAD while the owner
 * works on the next batch -- the only two calls on one handle that may overlap (it does not set c3_last_error).
 * One snapshot per handle: _snapshot before the previous one was fetched, or _fetch without a snapshot, returns C3_E_STATE. */
int c3_batch_results_snapshot(c3_handle* h);
int c3_batch_results_fetch(c3_handle* h, c3_read_result* res, char* cons, int64_t cons_cap, int64_t* cons_off);
int c3_batch_timing(c3_handle* h, c3_timing* t);

/* ---- stage probes (tests / per-stage shims), all operate on the resident batch ---- */
/* conk.conk(splint, seq, penalty) score track of read i (C3POa.py:123): L int32 */
int c3_fetch_track(c3_handle* h, int read, int32_t* out, int64_t cap);
/* 3x Savitzky-Golay smoothed track (bin/call_peaks.py:10-11): L doubles.  Only valid for the last
 * `slots` reads processed, so tests call it with single-read batches. */
int c3_fetch_smoothed(c3_handle* h, int read, double* out, int64_t cap);
/* raw call_peaks() indices before the shift (bin/call_peaks.py:15) */
int c3_fetch_raw_peaks(c3_handle* h, int read, int32_t* out, int cap);
/* draft consensus before polish (abpoa_cons, determine_consensus.py:32,41,47) */
int c3_fetch_draft(c3_handle* h, int read, char* out, int cap);
/* 2-row MSA of a 2-subread read (res.msa_seq, determine_consensus.py:34); returns msa_len */
int c3_fetch_msa2(c3_handle* h, int read, char* rowA, char* rowB, int cap);

/* stand-alone stage entry points used by the per-stage Python shims.  Each one uploads its
 * arguments, runs the corresponding kernels and returns the result. */
/* call_peaks(scores, min_dist, iters, window, order) (bin/call_peaks.py:8-16; iters/window/order are the
 * handle's sg_* settings): returns the number of peaks written to `peaks` (0 = gated / none);
 * smoothed (optional, n doubles) receives the smoothed track. */
int c3_call_peaks(c3_handle* h, const int32_t* scores, int n, int min_dist, int32_t* peaks, int cap, double* smoothed);
/* pyabpoa.msa_aligner(match=5).msa(seqs, out_cons, out_msa) (determine_consensus.py:30,34,43):
 * msa receives n rows of *msa_len chars (row-major).  quals may be NULL. */
int c3_poa_msa(c3_handle* h, int n, const char* const* seqs, const int* lens,
               char* cons, int cons_cap, int* cons_len, char* msa, int64_t msa_cap, int* msa_len);
/* pairwise_consensus(msa_rows, subreads, quals) (bin/consensus.py:76-81; call site determine_consensus.py:36-40): rowA/rowB
 * are the two MSA rows ('-' = gap), msa_len columns each.  *out_len <= msa_len bases are written to out. */
int c3_pairwise_consensus(c3_handle* h, const char* rowA, const char* rowB, int msa_len,
                          const char* subA, int lenA, const char* qualA, const char* subB, int lenB, const char* qualB,
                          char* out, int cap, int* out_len);
/* determine_consensus for repeats >= 1 (determine_consensus.py:29-99): draft + polish.
 * front/tail may be NULL.  returns consensus length in *out_len (0 = nothing emitted). */
int c3_determine_consensus(c3_handle* h, int n, const char* const* subs, const char* const* quals,
                           const int* lens, const char* front, const char* front_q, int front_len,
                           const char* tail, const char* tail_q, int tail_len,
                           char* out, int cap, int* out_len, char* draft, int draft_cap, int* draft_len);

/* splint / strand assignment of the resident batch (replaces the blat step of bin/preprocess.py:12-45,61-77):
 * every read is scored against every splint on both strands with the conk kernel.
 * out (optional) [n][n_splints][2][4] = {max of the track, its offset, mean of the track, read length};
 * assign_splint[i] = best splint (-1 = none), assign_strand[i] = '+', '-' or '?'; a candidate is accepted when
 * max >= 6 * mean (the contrast call_peaks demands later, bin/call_peaks.py:13) and max >= match*51*52/2
 * (the diagonal sum of a perfect 51-base match: `matches > 50`, bin/preprocess.py:32). */
int c3_scan_splints(c3_handle* h, int32_t* out, int16_t* assign_splint, char* assign_strand);

/* adapter finder of the post-processing step (replaces the blat call of C3POa_postprocessing.py:229-236; fields as
 * read by parse_blat, :238-264): best local affine alignment of every read of the resident batch against every entry
 * of the splint table (load the adapters with c3_set_splints) on both strands.
 * out[(i*n_adapters + a)*2 + rc][12] = score, qStart, qEnd, tStart, tEnd (PSL conventions, forward coordinates),
 * matches, misMatches, qBaseInsert, tBaseInsert, qNumInsert, tNumInsert, read length; score 0 = no alignment. */
int c3_scan_adapters(c3_handle* h, int32_t* out);

/* zero_repeats(name, seq, qual, dangling, qual_dangling, subread_file) (determine_consensus.py:106-136):
 * d0 = first dangling piece, d1 = second; *out_len = 0 when there is no acceptable overlap or the
 * stitched sequence is shorter than min_len (args.mdistcutoff, determine_consensus.py:17). */
int c3_zero_repeats(c3_handle* h, const char* d0, const char* q0, int n0, const char* d1, const char* q1, int n1,
                    int min_len, char* out, int cap, int* out_len);

/* ---- host I/O either side of the path; host code only, works without a GPU ---- */

/* one group of reads in structure-of-arrays form; the buffers belong to the reader (page-locked when a GPU is
 * present, so c3_batch_upload(h, b.n, b.seqs, b.quals, b.off, ...) copies them by DMA) */
typedef struct {
  int32_t n;                 /* reads in the group */
  int64_t n_short;           /* records skipped because they were shorter than min_len */
  const char* names;         /* concatenated names, name_off[n+1] (name = header up to the first blank) */
  const int64_t* name_off;
  const char* seqs;          /* concatenated bases / qualities, off[n+1] */
  const char* quals;         /* FASTA records get '!' */
  const int64_t* off;
} c3_host_batch;

/* host buffers for callers that assemble their own batches: page-locked when a GPU is present (DMA copies in
 * c3_batch_stage / c3_batch_upload), plain memory otherwise.  The reference builds Python lists (C3POa.py:239-244). */
int c3_host_alloc(int64_t bytes, void** out);
void c3_host_free(void* p);

typedef struct c3_reader c3_reader;
/* mm.fastx_read(path, read_comment=False) (C3POa.py:201,239): FASTA or FASTQ, multi-line, plain or .gz.
 * n_sets = how many groups stay valid at once (the buffers of a group are reused n_sets calls later). */
int c3_reader_open(const char* path, int n_sets, c3_reader** out);
/* the same over the byte range [beg, end) of a plain FASTA / 4-line FASTQ file: begins at the first record starting at or
 * after beg, ends before the first record starting at or after end (end < 0: end of file), so ranges that tile the file
 * read every record exactly once -- one reader per GPU worker (C3POa.py:236-256 sharded 1000-read groups over a pool) */
int c3_reader_open_range(const char* path, int n_sets, int64_t beg, int64_t end, c3_reader** out);
/* BGZF (bgzip) input can be cut into ranges as well -- its members are located by their headers without inflating them: the inflated
 * size of the file (-1: not BGZF from end to end), and c3_reader_open_range over [beg, end) in bytes of the INFLATED file.  A plain
 * gzip stream cannot be entered in the middle: c3_reader_open_range refuses it (C3_E_ARG), one c3_reader_open reads it -- with several
 * inflating threads behind its one parser (csrc/c3_gzpar.hpp: block starts by trial decoding, the unknown window resolved afterwards).
 * (C3POa.py:201,239 read .gz through one Python gzip stream; the read sharding is C3POa.py:236-256) */
int64_t c3_bgzf_size(const char* path);
void c3_reader_close(c3_reader* r);
const char* c3_reader_error(const c3_reader* r);
/* records without a quality line seen so far (FASTA).  The reference cannot process them (C3POa.py:167 takes ord() of every
 * quality character; racon runs with -q 5), so the CLI refuses such input */
int64_t c3_reader_noqual(const c3_reader* r);
/* bytes of (page-locked) host buffers the reader holds: sized by what its file / byte range can still deliver, never by max_reads alone */
int64_t c3_reader_reserved_bytes(const c3_reader* r);
/* 1 when c3_reader_open_range found bytes but no 4-line FASTQ / FASTA record start in its range (multi-line FASTQ): the caller
 * must read the file with ONE reader (c3_reader_open), or the records of that range are lost */
int c3_reader_range_lost(const c3_reader* r);
/* names_only != 0: parse but do not store sequences/qualities (first pass of C3POa.py:200-207: names + counts) */
void c3_reader_names_only(c3_reader* r, int names_only);
/* next group: at most max_reads reads of length >= min_len (C3POa.py:202-204,240-241), stops early once max_bases
 * bases are held (0 = no limit).  out->n == 0 at end of file. */
int c3_reader_next(c3_reader* r, int max_reads, int64_t max_bases, int min_len, c3_host_batch* out);
/* the same into buffer set `set` (0 .. n_sets-1) named by the caller instead of round-robin: a pipeline whose consumers
 * finish out of order (several GPUs) keeps a free list of sets and reuses one only after its group has been written */
int c3_reader_next_set(c3_reader* r, int set, int max_reads, int64_t max_bases, int min_len, c3_host_batch* out);

/* file side effects of analyze_reads + determine_consensus for one group (C3POa.py:141-173,
 * bin/determine_consensus.py:57-77,108-114): appends the consensus FASTA records to cons_paths[splint_id[i]] and the
 * subread FASTQ records to sub_paths[splint_id[i]].  cons/cons_off as returned by c3_batch_results; zero = args.zero. */
int c3_write_group(const c3_host_batch* b, const c3_read_result* res, const char* cons, const int64_t* cons_off,
                   const int16_t* splint_id, int n_splints, const char* const* cons_paths,
                   const char* const* sub_paths, int zero);

/* c3_write_group may be called from several threads on the same files (one writer per GPU worker): every call reserves its
 * byte range at the end of each file under a lock.  Reservations are keyed by the file itself (device, inode) and live only
 * while a writer of that file is in flight: a file truncated or replaced between two calls is appended to from its real end.
 * c3_writer_reset forgets all reservations (start of a run; never while c3_write_group calls are in flight). */
void c3_writer_reset(void);

/* -co / --compress_output (C3POa.py:46-47; cat_files writes the final file through gzip.open, C3POa.py:86-99; C3POa_postprocessing.py -co):
 * src is compressed into dst as a gzip file made of independent members of <= 64 KiB in the BGZF layout (each header carries the
 * member's size in a 'BC' extra subfield; the empty end-of-file member closes it) -- every gzip reader inflates it (zcat, Python's
 * gzip, mm.fastx_read), and c3_reader_open inflates its members in parallel.  `threads` deflate at once (0 = the usable host cores),
 * `level` 1..9 (0 = zlib's default 6).  Host code.  C3_E_ARG: a file cannot be opened / written; the partial dst is removed. */
int c3_compress_file(const char* src, const char* dst, int level, int threads);

/* splint assignment from the PSL (bin/preprocess.py:22-45) without per-read host objects: rows with qBaseInsert < 50 and
 * matches > 50 count, per read the row with the most matches wins (the earliest on ties).  Host code. */
typedef struct c3_assign c3_assign;
int c3_assign_open(const char* psl_path, int n_splints, const char* const* splint_names, c3_assign** out);
void c3_assign_close(c3_assign* a);
/* splint row / strand ('+', '-'; -1 / '?' = no counted row) of every read of the group; returns how many are assigned */
int c3_assign_batch(const c3_assign* a, const c3_host_batch* b, int16_t* splint_id, char* strand);
/* adapter_set (bin/preprocess.py:34,43): flags[s] = 1 when a counted row names splint s; *rows_kept = counted rows */
int c3_assign_seen(const c3_assign* a, uint8_t* flags, int64_t* rows_kept);

/* PSL rows of the GPU splint finder, appended to `path` (one 21-column row per assigned read of the group; table /
 * splint_id / strand as returned by c3_scan_splints; the row format is stated by psl_row() in c3poa_amd/preprocess.py) */
int c3_write_splint_psl(const c3_host_batch* b, const int32_t* table, const int16_t* splint_id, const char* strand,
                        int n_splints, const char* const* splint_names, const int32_t* splint_lens, int match,
                        const char* path, int64_t* rows_written);

/* match_index for a whole batch on the GPU (one lane per piece): pieces = n slots of 64 bytes, lens[n] <= 64, at most
 * 16 indexes of at most 32 bases; out[i] = winning index number or -1.  Same function as c3_match_index below. */
int c3_match_index_batch(c3_handle* h, int n, const char* pieces, const int32_t* lens, int n_idx,
                         const char* idx_cat, const int64_t* idx_off, int32_t* out);

/* match_index(seq, seq_to_idx) of C3POa_postprocessing.py:266-285 (oligo-dT demultiplexing): sliding Levenshtein
 * distance of seq against every index (file order, idx_off[n_idx+1] into idx_cat); returns the winning index number or
 * -1 for '-'.  Host code. */
int c3_match_index(const char* seq, int n, int n_idx, const char* idx_cat, const int64_t* idx_off);

#ifdef __cplusplus
}
#endif
#endif
