// c3_api.hip -- C ABI (include/c3poa.h) and host orchestration of the HIP hot path.
// No CPU fallback exists in this library: without a gfx950 device c3_create fails.
#include "c3_dev.h"
#include "c3_args.h"
#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// ---- kernel launchers (one translation unit per kernel family) --------------------------
extern "C" void c3k_launch_conk(const ConkArgs*, int, int, int, hipStream_t);
extern "C" void c3k_launch_adapter(const AdapterArgs*, int, hipStream_t);
extern "C" void c3k_launch_pairwise(const uint8_t*, int, const uint8_t*, int, const uint8_t*, int, uint8_t*, uint8_t*, int*, hipStream_t);
extern "C" void c3k_launch_match_index(const char*, const int*, int, int, const char*, const long long*, int*, hipStream_t);
extern "C" void c3k_launch_peaks(const PeaksArgs*, int, hipStream_t);
extern "C" int c3k_peaks_blocks_per_cu(void);
extern "C" void c3k_launch_poa(const PoaArgs*, int, int, int, hipStream_t);
extern "C" void c3k_launch_prep(const PrepArgs*, int, hipStream_t);
extern "C" void c3k_launch_window(const WinArgs*, int, hipStream_t);
extern "C" void c3k_launch_stitch(const StitchArgs*, int, hipStream_t);
extern "C" void c3k_launch_poa_mw(const PoaArgs*, int slots, hipStream_t);      // the last pass: a workgroup of eight waves per read (k_poa_mw.hip)
extern "C" void c3k_launch_zero(const ZeroArgs*, int, hipStream_t);
extern "C" void c3k_launch_zero_finish(const ZeroArgs*, int, hipStream_t);

// ---- small kernels ----------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_code(uint32_t b) {
  // A/a=0 C/c=1 G/g=2 T/t/U/u=3, every other byte 0 (c3poa.h conventions)
  const uint32_t u = b & 0xDFu;                                // upper case
  const uint32_t c = (u >> 1) & 3u;                            // A0 C1 G3 T2 U2
  const bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T') | (u == 'U');
  return ok ? (c ^ (c >> 1)) : 0u;
}
__global__ __launch_bounds__(256) void k_pack(const uint8_t* ascii, const int64_t* off, const int64_t* woff, int n, uint32_t* pk) {
  // one wave per read (grid-stride); lane l packs word w = 64*it + l from 16 consecutive bytes (one 16-byte load per
  // lane -> a wave reads 1 KiB contiguous).  The tail word is assembled byte by byte.
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < n; r += n_waves) {
    const int64_t L = off[r + 1] - off[r];
    const int64_t nw = woff[r + 1] - woff[r];
    const uint8_t* s = ascii + off[r];
    uint32_t* dst = pk + woff[r];
    for (int64_t w = lane; w < nw; w += 64) {
      uint32_t x = 0;
      if (w * 16 + 16 <= L) {
        uint32_t v[4];
        __builtin_memcpy(v, s + w * 16, 16);
#pragma unroll
        for (int k = 0; k < 16; ++k) x |= pack_code((v[k >> 2] >> (8 * (k & 3))) & 0xFFu) << (2 * k);
      } else {
        for (int k = 0; k < 16; ++k) { int64_t i = w * 16 + k; if (i < L) x |= pack_code(s[i]) << (2 * k); }
      }
      dst[w] = x;
    }
  }
}
// consensus of read r lives at arena[off[r] ..]; compact copies go to out[coff[r] .. coff[r+1]) (one wave per read)
__global__ __launch_bounds__(256) void k_gather_cons(const char* arena, const int64_t* off, const int64_t* coff, int n, char* out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < n; r += n_waves) {
    const int64_t len = coff[r + 1] - coff[r];
    const char* src = arena + off[r]; char* dst = out + coff[r];
    for (int64_t k = lane; k < len; k += 64) dst[k] = src[k];
  }
}
// cons_off[i + 1] = sum over reads <= i of (status OK ? cons_len : 0), on the device (three tiny launches: block sums, scan of the
// block sums by one block, block-local scan + block offset): the host no longer needs the records before it can size the copy
__device__ __forceinline__ long long cons_len_of(const C3Info* p) { return p->status == C3_ST_OK ? (long long)p->cons_len : 0; }
__global__ __launch_bounds__(256) void k_coff_sums(const C3Info* info, int n, long long* part) {
  __shared__ long long sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  sh[threadIdx.x] = i < n ? cons_len_of(info + i) : 0;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(1024) void k_coff_scan(long long* part, int nb, int64_t* coff, int n) {
  __shared__ long long sh[1024];
  long long carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    const long long v = b < nb ? part[b] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { long long t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
    if (b < nb) part[b] = carry + sh[threadIdx.x] - v;               // exclusive
    carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) { coff[0] = 0; coff[n] = carry; }
}
__global__ __launch_bounds__(256) void k_coff_final(const C3Info* info, int n, const long long* part, int64_t* coff) {
  __shared__ long long sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  sh[threadIdx.x] = i < n ? cons_len_of(info + i) : 0;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) { long long t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
  if (i < n) coff[i + 1] = part[blockIdx.x] + sh[threadIdx.x];
}
__global__ void k_init_info(C3Info* info, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { C3Info* p = &info[i]; p->status = C3_ST_OK; p->n_peaks = 0; p->n_sub = 0; p->has_front = p->has_tail = 0;
               p->front_end = p->tail_beg = 0; p->cons_len = 0; p->draft_len = 0; p->n_win = 0; }
}
// after k_poa: the longest draft and the polishing windows of all drafts of the work list -> out[0], out[1] (what k_prep's window tables
// must hold; kept out of k_poa, whose register allocation a two-atomic epilogue cost 1.5-2 %)
__global__ void k_draft_stats(const C3Info* info, const int* work, int nw, int WL, int* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  if (k < nw) { const C3Info* p = &info[work[k]]; c = p->status == C3_ST_OK ? p->draft_len : 0; }
  int mx = c, nwin = c > 0 ? (c + WL - 1) / WL + 1 : 0;
  for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o)); nwin += __shfl_xor(nwin, o); }
  if ((threadIdx.x & 63) == 0 && nwin > 0) { atomicMax(out, mx); atomicAdd(out + 1, nwin); }
}
struct Summary { int status, n_sub, max_sub, sum_sub, max_dang, front, tail, n_peaks; };
__global__ void k_summary(const C3Info* info, const int64_t* off, int n, Summary* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const C3Info* p = &info[i];
  Summary s; s.status = p->status; s.n_sub = p->n_sub; s.max_sub = 0; s.sum_sub = 0; s.max_dang = 0; s.n_peaks = p->n_peaks;
  for (int k = 0; k < p->n_sub; ++k) { int l = p->sub_end[k] - p->sub_beg[k]; s.sum_sub += l; if (l > s.max_sub) s.max_sub = l; }
  int L = (int)(off[i + 1] - off[i]);
  s.front = p->has_front ? p->front_end : 0; s.tail = p->has_tail ? L - p->tail_beg : 0;
  if (p->has_front) s.max_dang = p->front_end;
  if (p->has_tail && L - p->tail_beg > s.max_dang) s.max_dang = L - p->tail_beg;
  out[i] = s;
}

#include <atomic>
#include <chrono>
static inline double dbg_now_ms() { using namespace std::chrono; return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count(); }
// C3_DEBUG=1: progress lines on stderr, each stamped with the host clock (ms) -- shows the host gaps between the stages
#define DBG(...) do { if (getenv("C3_DEBUG")) { fprintf(stderr, "[c3 %.3f] ", dbg_now_ms()); fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)

// ---- handle -----------------------------------------------------------------------------
static thread_local double g_alloc_ms = 0.0;   // host time spent growing device buffers (hipFree synchronises the device)
struct DBuf {
  void* p = nullptr; size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    const double t0 = dbg_now_ms();
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    // test hook: fresh device memory usually reads as zero, which hides reads of cells nobody wrote; C3_DEBUG_POISON fills every
    // new buffer with a pattern instead (tests/test_gpu_band.py runs the pipeline that way)
    if (e == hipSuccess && getenv("C3_DEBUG_POISON")) e = hipMemset(p, 0xA5, want);
    g_alloc_ms += dbg_now_ms() - t0;
    if (getenv("C3_DEBUG_ALLOC")) fprintf(stderr, "c3poa alloc: %.1f MB in %.1f ms\n", want / 1048576.0, dbg_now_ms() - t0);
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return (T*)p; }
};

enum { EV_N = 10 };
struct c3_handle {
  c3_config cfg; std::string err; hipStream_t stream = nullptr; int n_cus = 256; size_t mem_total = 0;
  // staged (next) batch: copied on its own stream while the resident batch is being processed
  hipStream_t stream_up = nullptr; hipEvent_t ev_up[2] = {nullptr, nullptr};
  // results in flight (c3_batch_results_snapshot .. _fetch): snapshot of the records + compact consensus bytes, copied on a third stream
  hipStream_t stream_dn = nullptr; hipEvent_t ev_dn = nullptr; long long* h_tot = nullptr;
  std::atomic<bool> snap_pending{false}; int snap_n = 0, snap_kp = 0; long long snap_tot = 0; bool snap_cons = false;
  DBuf d_info_snap, d_coff_part;
  struct Staged { DBuf d_ascii, d_pk, d_woff, d_qual, d_off, d_strand, d_sid; std::vector<int64_t> off, woff; std::vector<int16_t> sid; std::string strand;
                  int n = 0; int64_t total = 0, words = 0, maxL = 0; bool pending = false; } st;
  hipEvent_t ev[EV_N];
  // splints
  int n_spl = 0, max_spl = 0; std::vector<int> sp_len; DBuf d_sp_codes, d_sp_len;
  // batch
  int n = 0; int64_t total = 0, words = 0, maxL = 0; std::vector<int64_t> off, woff;
  DBuf d_ascii, d_pk, d_woff, d_qual, d_off, d_strand, d_sid, d_info, d_track, d_draft, d_tpos, d_cons, d_counter, d_gather, d_gather_off;
  DBuf d_raw, d_nraw, d_sum, d_work, d_bufA, d_bufB, d_cand, d_cst, d_msa, d_msa_off, d_msa_len;
  DBuf s_poa_i, s_poa_nk, s_poa_cells, s_poa_b, s_poa_sc, s_poa_desc, s_poa_jump, s_poa_path, d_overflow;      // POA scratch
  // the LAST POA pass (a handful of reads whose bands blew up: a workgroup of eight waves each, >100 ms on four CUs) runs on a stream of
  // its own beside k_prep / k_window of every other read; the stragglers are polished by a small tail afterwards (run_tail)
  hipStream_t stream_mw = nullptr; hipEvent_t ev_mw[2] = {nullptr, nullptr}; DBuf d_counter_mw, d_work_main;
  // k_window's full-size launch as a consumer beside the first launch (round 6): "inputs ready" / "consumer done".  It runs on stream_mw: a
  // fifth stream would share one of the runtime's four hardware queues with another stream of this handle, and kernels that share a queue
  // run one after the other (measured: with a stream of its own the consumer ended up on the main stream's queue once stream_mw had taken
  // the fourth, and the last POA pass of cfgL stopped overlapping the polish: +130 ms).  While that pass is in flight there is no consumer
  hipEvent_t ev_w2[2] = {nullptr, nullptr}; int win_consumers = 0;
  std::vector<int> strag, work_main; bool tail_pending = false;
  int n_poa_redo = 0;        // reads of the last run that needed the full-size second POA pass
  int n_poa_redo16 = 0;      // ... of them: because a score left the 16-bit cells
  DBuf s_eH, s_eD, s_lw, d_wrec, d_wlay, d_wbase, d_wout;       // prep / windows
  DBuf s_win_i, s_win_nk, s_win_h, s_win_d, s_win_b, s_win_sc, s_win_desc, s_win_h2, s_win_d2, s_win_i2, s_win_nk2, s_win_b2, s_win_sc2, s_win_desc2, d_wout2, d_wovf;
  int win_out2_cap = 0;
  int poa_max_draft = -1, poa_sum_win = -1;       // from k_poa of THIS batch (-1: the drafts did not come from it)
  DBuf s_zero_d, d_zinfo, d_zflag, d_zwork; std::vector<int> zwork;  // window scratch
  std::vector<Summary> sum; std::vector<int> work;
  int res_prefix = 0;            // entries of peaks[] / sub_beg[] / sub_end[] that any read of the resident batch uses (0: unknown)
  int peaks_grid = 0; bool debug_msa = false; bool injected = false;
  int n_windows = 0;
  c3_timing tm;
  unsigned long long phase_poa[16] = {0}, phase_win[16] = {0};
  int stages_done = 0;
};

static int c3_hip_fail(c3_handle* h, hipError_t e, const char* what, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at c3_api.hip:%d: %s", (int)e, hipGetErrorString(e), line, what);
  if (h) h->err = buf;
  return C3_E_HIP;
}
static int c3_fail(c3_handle* h, int code, const char* msg) { if (h) h->err = msg; return code; }

extern "C" void c3_default_config(c3_config* c) {
  memset(c, 0, sizeof(*c));
  c->device = 0;
  c->conk_match = 5; c->conk_mismatch = -4; c->conk_penalty = 20;
  c->sg_iters = 3; c->sg_window = 41; c->sg_order = 2; c->mdistcutoff = 500;
  c->poa_match = 5; c->poa_mismatch = 4; c->poa_o1 = 4; c->poa_e1 = 2; c->poa_o2 = 24; c->poa_e2 = 1;
  c->poa_band_b = 10; c->poa_band_f = 0.01;
  c->pol_match = 3; c->pol_mismatch = -5; c->pol_gap = -4; c->pol_window = 500; c->pol_q = 5; c->dang_band = 128;
  c->slots_poa = 0; c->slots_win = 0; c->zero = 1;
}
extern "C" const char* c3_version(void) { return "c3poa_amd 0.1 (gfx950)"; }
extern "C" int c3_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

extern "C" int c3_warm_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return C3_E_NO_DEVICE; }
  if (hipSetDevice(device) != hipSuccess || hipFree(nullptr) != hipSuccess) { (void)hipGetLastError(); return C3_E_HIP; }
  return C3_E_OK;
}

static thread_local std::string g_create_err;
extern "C" const char* c3_last_error(const c3_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int c3_create(const c3_config* cfg, c3_handle** out) {
  if (!cfg || !out) return C3_E_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) { g_create_err = "no HIP device: the c3poa HIP backend has no CPU fallback"; return C3_E_NO_DEVICE; }
  if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "bad device ordinal"; return C3_E_ARG; }
  if (cfg->conk_match < -127 || cfg->conk_match > 127 || cfg->conk_mismatch < -127 || cfg->conk_mismatch > 127) {
    g_create_err = "conk_match / conk_mismatch must fit a signed byte"; return C3_E_ARG;                 // k_conk keeps them in byte tables
  }
  if (cfg->sg_order != 2 && cfg->sg_order != 3) { g_create_err = "sg_order must be 2 or 3"; return C3_E_ARG; }
  if (cfg->sg_window < 5 || cfg->sg_window > 127 || !(cfg->sg_window & 1)) { g_create_err = "sg_window must be odd, 5..127"; return C3_E_ARG; }
  c3_handle* h = new c3_handle();
  h->cfg = *cfg;
  if ((e = hipSetDevice(cfg->device)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  // synchronisation points sleep instead of spinning: the stages are milliseconds long, and a spinning waiter per
  // handle eats the CPU quota the reader / writer threads need (refused once the context exists: ignored)
  (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError();
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  h->mem_total = prop.totalGlobalMem;
  if ((e = hipStreamCreate(&h->stream)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  if ((e = hipStreamCreate(&h->stream_up)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  if ((e = hipStreamCreate(&h->stream_dn)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  if ((e = hipStreamCreate(&h->stream_mw)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  for (int i = 0; i < 2; ++i) if ((e = hipEventCreate(&h->ev_mw[i])) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  for (int i = 0; i < 2; ++i) if ((e = hipEventCreateWithFlags(&h->ev_w2[i], hipEventDisableTiming)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  if ((e = hipEventCreateWithFlags(&h->ev_dn, hipEventDisableTiming)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  if ((e = hipHostMalloc((void**)&h->h_tot, 64, hipHostMallocDefault)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  for (int i = 0; i < 2; ++i) if ((e = hipEventCreate(&h->ev_up[i])) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  for (int i = 0; i < EV_N; ++i) if ((e = hipEventCreate(&h->ev[i])) != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return C3_E_HIP; }
  memset(&h->tm, 0, sizeof(h->tm));
  *out = h;
  return C3_E_OK;
}

extern "C" void c3_destroy(c3_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  (void)hipStreamSynchronize(h->stream);
  if (h->stream_dn) { (void)hipStreamSynchronize(h->stream_dn); (void)hipStreamDestroy(h->stream_dn); }
  if (h->stream_mw) { (void)hipStreamSynchronize(h->stream_mw); (void)hipStreamDestroy(h->stream_mw); }
  for (int i = 0; i < 2; ++i) if (h->ev_mw[i]) (void)hipEventDestroy(h->ev_mw[i]);
  for (int i = 0; i < 2; ++i) if (h->ev_w2[i]) (void)hipEventDestroy(h->ev_w2[i]);
  h->d_counter_mw.release(); h->d_work_main.release();
  if (h->ev_dn) (void)hipEventDestroy(h->ev_dn);
  if (h->h_tot) (void)hipHostFree(h->h_tot);
  h->d_info_snap.release(); h->d_coff_part.release(); h->s_win_h2.release(); h->s_win_d2.release(); h->s_win_i2.release(); h->s_win_nk2.release(); h->s_win_b2.release(); h->s_win_sc2.release(); h->s_win_desc2.release(); h->d_wout2.release(); h->d_wovf.release();
  DBuf* all[] = {&h->d_sp_codes, &h->d_sp_len, &h->d_ascii, &h->d_pk, &h->d_woff, &h->d_qual, &h->d_off, &h->d_strand, &h->d_sid,
                 &h->d_info, &h->d_track, &h->d_draft, &h->d_tpos, &h->d_cons, &h->d_counter, &h->d_raw, &h->d_nraw, &h->d_sum,
                 &h->d_work, &h->d_bufA, &h->d_bufB, &h->d_cand, &h->d_cst, &h->d_msa, &h->d_msa_off, &h->d_msa_len,
                 &h->s_poa_i, &h->s_poa_nk, &h->s_poa_cells, &h->s_poa_b, &h->s_poa_sc, &h->s_poa_desc, &h->s_poa_jump, &h->s_poa_path, &h->d_overflow, &h->s_eH, &h->s_eD, &h->s_lw, &h->d_wrec,
                 &h->d_wlay, &h->d_wbase, &h->d_wout, &h->s_win_i, &h->s_win_nk, &h->s_win_h, &h->s_win_d, &h->s_win_b, &h->s_win_sc, &h->s_win_desc, &h->s_zero_d, &h->d_zinfo, &h->d_zflag, &h->d_zwork, &h->d_gather, &h->d_gather_off};
  for (DBuf* b : all) b->release();
  { DBuf* sh[] = {&h->st.d_ascii, &h->st.d_pk, &h->st.d_woff, &h->st.d_qual, &h->st.d_off, &h->st.d_strand, &h->st.d_sid}; for (DBuf* b : sh) b->release(); }
  if (h->stream_up) { (void)hipStreamSynchronize(h->stream_up); (void)hipStreamDestroy(h->stream_up); }
  for (int i = 0; i < 2; ++i) if (h->ev_up[i]) (void)hipEventDestroy(h->ev_up[i]);
  for (int i = 0; i < EV_N; ++i) (void)hipEventDestroy(h->ev[i]);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

static inline int code_of(char c) {
  switch (c) { case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3; default: return 0; }
}

extern "C" int c3_set_splints(c3_handle* h, int n, const char* cat, const int64_t* off) {
  if (!h || n <= 0 || !cat || !off) return C3_E_ARG;
  HIPCHK(hipSetDevice(h->cfg.device));
  std::vector<uint8_t> codes((size_t)n * 2 * C3_SPLINT_MAX, 0);
  h->sp_len.assign(n, 0); h->max_spl = 0;
  for (int i = 0; i < n; ++i) {
    int S = (int)(off[i + 1] - off[i]);
    if (S <= 0 || S > C3_SPLINT_MAX) return c3_fail(h, C3_E_LIMIT, "splint length must be 1..512");
    if ((long long)std::max(h->cfg.conk_match, 0) * S > 32000 || h->cfg.conk_penalty < 0 || h->cfg.conk_penalty > 32000)
      return c3_fail(h, C3_E_LIMIT, "conk_match * splint length and conk_penalty must stay below 32000 (16-bit score cells in k_conk)");
    h->sp_len[i] = S; h->max_spl = std::max(h->max_spl, S);
    for (int k = 0; k < S; ++k) {
      int c = code_of(cat[off[i] + k]);
      codes[((size_t)i * 2 + 0) * C3_SPLINT_MAX + k] = (uint8_t)c;
      codes[((size_t)i * 2 + 1) * C3_SPLINT_MAX + (S - 1 - k)] = (uint8_t)(3 - c);   // reverse complement (C3POa.py:234)
    }
  }
  HIPCHK(h->d_sp_codes.ensure(codes.size()));
  HIPCHK(h->d_sp_len.ensure(sizeof(int) * n));
  HIPCHK(hipMemcpyAsync(h->d_sp_codes.p, codes.data(), codes.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_sp_len.p, h->sp_len.data(), sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->n_spl = n;
  return C3_E_OK;
}

static C3Batch dev_batch(c3_handle* h) {
  C3Batch b; b.n = h->n; b.pk = h->d_pk.as<uint32_t>(); b.woff = h->d_woff.as<int64_t>(); b.qual = h->d_qual.as<uint8_t>();
  b.off = h->d_off.as<int64_t>(); b.strand = h->d_strand.as<uint8_t>(); b.splint_id = h->d_sid.as<int16_t>();
  return b;
}
static C3Params dev_params(const c3_config& c) {
  C3Params p;
  p.conk_match = c.conk_match; p.conk_mismatch = c.conk_mismatch; p.conk_penalty = c.conk_penalty;
  p.sg_iters = c.sg_iters; p.sg_window = c.sg_window; p.sg_order = c.sg_order; p.mdist = c.mdistcutoff;
  p.poa_match = c.poa_match; p.poa_mismatch = c.poa_mismatch; p.o1 = c.poa_o1; p.e1 = c.poa_e1; p.o2 = c.poa_o2; p.e2 = c.poa_e2;
  p.band_b = c.poa_band_b; p.band_f = c.poa_band_f;
  p.pol_match = c.pol_match; p.pol_mismatch = c.pol_mismatch; p.pol_gap = c.pol_gap; p.pol_window = c.pol_window; p.pol_q = c.pol_q;
  p.dang_band = c.dang_band;
  p.zero = c.zero; p.zr_match = 2; p.zr_mismatch = 4; p.zr_gapo = 4; p.zr_gape = 2; p.zr_min_score = 80; p.zr_max_cells = 16 << 20;
  return p;
}

// Stage the NEXT batch: validation, H2D copies and the 2-bit pack run on a second stream, so they overlap the kernels of
// the resident batch.  seqs / quals must stay valid until c3_batch_commit returns (page-locked buffers make the copies
// truly asynchronous); off / splint_id / strand are copied before the call returns.
extern "C" int c3_batch_stage(c3_handle* h, int n, const char* seqs, const char* quals, const int64_t* off,
                              const int16_t* splint_id, const char* strand) {
  if (!h || n <= 0 || !seqs || !quals || !off || !strand) return C3_E_ARG;
  if (h->n_spl <= 0) return c3_fail(h, C3_E_STATE, "c3_set_splints must be called first");
  if (h->st.pending) return c3_fail(h, C3_E_STATE, "a staged batch is waiting for c3_batch_commit");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (off[0] != 0) return c3_fail(h, C3_E_ARG, "off[0] must be 0");
  c3_handle::Staged& t = h->st;
  t.n = n; t.total = off[n] - off[0]; t.off.assign(off, off + n + 1); t.woff.assign(n + 1, 0); t.maxL = 0;
  t.sid.assign((size_t)n, 0); t.strand.assign(strand, (size_t)n);
  for (int i = 0; i < n; ++i) {
    int64_t L = off[i + 1] - off[i];
    if (L < 0 || L > (1 << 30)) return c3_fail(h, C3_E_ARG, "bad read length");
    t.maxL = std::max(t.maxL, L);
    t.woff[i + 1] = t.woff[i] + (L + 15) / 16 + 2;          // +2 words: aligned-window overread
    if (splint_id) { if (splint_id[i] < 0 || splint_id[i] >= h->n_spl) return c3_fail(h, C3_E_ARG, "splint_id out of range"); t.sid[i] = splint_id[i]; }
  }
  t.words = t.woff[n];
  const size_t T = (size_t)t.total;
  HIPCHK(t.d_ascii.ensure(T + 16)); HIPCHK(t.d_pk.ensure(sizeof(uint32_t) * (size_t)t.words + 64));
  HIPCHK(t.d_qual.ensure(T + 16)); HIPCHK(t.d_off.ensure(sizeof(int64_t) * (n + 1))); HIPCHK(t.d_woff.ensure(sizeof(int64_t) * (n + 1)));
  HIPCHK(t.d_strand.ensure(n)); HIPCHK(t.d_sid.ensure(sizeof(int16_t) * n));
  hipStream_t su = h->stream_up;
  HIPCHK(hipEventRecord(h->ev_up[0], su));
  HIPCHK(hipMemcpyAsync(t.d_ascii.p, seqs, T, hipMemcpyHostToDevice, su));
  HIPCHK(hipMemcpyAsync(t.d_qual.p, quals, T, hipMemcpyHostToDevice, su));
  HIPCHK(hipMemcpyAsync(t.d_off.p, t.off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, su));
  HIPCHK(hipMemcpyAsync(t.d_woff.p, t.woff.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, su));
  HIPCHK(hipMemcpyAsync(t.d_strand.p, t.strand.data(), n, hipMemcpyHostToDevice, su));
  HIPCHK(hipMemcpyAsync(t.d_sid.p, t.sid.data(), sizeof(int16_t) * n, hipMemcpyHostToDevice, su));
  dim3 g((unsigned)std::min((n + 3) / 4, h->n_cus * 32));
  hipLaunchKernelGGL(k_pack, g, dim3(256), 0, su, t.d_ascii.as<uint8_t>(), t.d_off.as<int64_t>(), t.d_woff.as<int64_t>(), n, t.d_pk.as<uint32_t>());
  HIPCHK(hipEventRecord(h->ev_up[1], su));
  HIPCHK(hipGetLastError());
  t.pending = true;
  return C3_E_OK;
}

// Make the staged batch the resident one (after the results of the previous batch have been fetched).
extern "C" int c3_batch_commit(c3_handle* h) {
  if (!h) return C3_E_ARG;
  if (!h->st.pending) return c3_fail(h, C3_E_STATE, "no staged batch");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));            // the previous batch is completely done
  HIPCHK(hipStreamSynchronize(h->stream_up));         // the staged copies and the pack have landed
  c3_handle::Staged& t = h->st;
  std::swap(h->d_ascii, t.d_ascii); std::swap(h->d_pk, t.d_pk); std::swap(h->d_woff, t.d_woff); std::swap(h->d_qual, t.d_qual);
  std::swap(h->d_off, t.d_off); std::swap(h->d_strand, t.d_strand); std::swap(h->d_sid, t.d_sid);
  h->off.swap(t.off); h->woff.swap(t.woff);
  h->n = t.n; h->total = t.total; h->words = t.words; h->maxL = t.maxL;
  t.pending = false;
  const int n = h->n;
  HIPCHK(h->d_info.ensure(sizeof(C3Info) * (size_t)n));
  HIPCHK(h->d_counter.ensure(256));
  HIPCHK(hipMemsetAsync(h->d_info.p, 0, sizeof(C3Info) * (size_t)n, h->stream));   // the unused tails of peaks[] / sub_*[] read as 0
  hipLaunchKernelGGL(k_init_info, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_info.as<C3Info>(), n);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, h->ev_up[0], h->ev_up[1]));
  memset(&h->tm, 0, sizeof(h->tm)); h->tm.ms_pack = ms; h->tm.n_reads = n; h->tm.n_bases = h->total;
  h->stages_done = 0; h->injected = false; h->n_windows = 0; h->res_prefix = 0; h->poa_max_draft = h->poa_sum_win = -1;
  return C3_E_OK;
}

// Overwrite the splint row / strand of every read of the resident batch (after c3_scan_splints, before c3_batch_run):
// strand[i] = '+' / '-', anything else = not assigned; splint_id[i] < 0 is stored as 0 for such reads.
extern "C" int c3_batch_assign(c3_handle* h, const int16_t* splint_id, const char* strand) {
  if (!h || h->n <= 0 || !splint_id || !strand) return C3_E_ARG;
  HIPCHK(hipSetDevice(h->cfg.device));
  std::vector<int16_t> sid((size_t)h->n);
  for (int i = 0; i < h->n; ++i) {
    const bool on = strand[i] == '+' || strand[i] == '-';
    if (on && (splint_id[i] < 0 || splint_id[i] >= h->n_spl)) return c3_fail(h, C3_E_ARG, "splint_id out of range");
    sid[(size_t)i] = on ? splint_id[i] : (int16_t)0;
  }
  HIPCHK(hipMemcpyAsync(h->d_strand.p, strand, (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_sid.p, sid.data(), sizeof(int16_t) * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
  // a read that was unassigned in an earlier run carries C3_ST_NOT_ASSIGNED: every record starts over
  hipLaunchKernelGGL(k_init_info, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->d_info.as<C3Info>(), h->n);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  h->stages_done = 0; h->poa_max_draft = h->poa_sum_win = -1;
  return C3_E_OK;
}

// upload = stage + commit (nothing to overlap with)
extern "C" int c3_batch_upload(c3_handle* h, int n, const char* seqs, const char* quals, const int64_t* off,
                               const int16_t* splint_id, const char* strand) {
  if (h && h->st.pending) { (void)hipStreamSynchronize(h->stream_up); h->st.pending = false; }     // an abandoned staged batch is dropped
  int rc = c3_batch_stage(h, n, seqs, quals, off, splint_id, strand);
  if (rc != C3_E_OK) return rc;
  return c3_batch_commit(h);
}

static int auto_slots(c3_handle* h, int want, size_t per_slot_bytes, int n_items, int waves_per_cu) {
  int s = want > 0 ? want : h->n_cus * waves_per_cu;
  size_t budget = h->mem_total ? h->mem_total / 3 : ((size_t)64 << 30);
  if (per_slot_bytes > 0) { size_t mx = budget / per_slot_bytes; if ((size_t)s > mx) s = (int)std::max<size_t>(mx, 1); }
  if (s > n_items) s = std::max(n_items, 1);
  return s;
}

static int run_conk(c3_handle* h) {
  HIPCHK(h->d_track.ensure(sizeof(int32_t) * (size_t)h->total + 64));
  HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 64, h->stream));
  ConkArgs a; a.b = dev_batch(h); a.sp_codes = h->d_sp_codes.as<uint8_t>(); a.sp_len = h->d_sp_len.as<int>();
  a.track = h->d_track.as<int32_t>(); a.info = h->d_info.as<C3Info>(); a.counter = h->d_counter.as<int>();
  a.match = h->cfg.conk_match; a.mismatch = h->cfg.conk_mismatch; a.penalty = h->cfg.conk_penalty; a.n_spl = h->n_spl; a.scan = nullptr;
  int waves = std::min(h->n, h->n_cus * 32);
  c3k_launch_conk(&a, h->max_spl, (waves + 3) / 4, 0, h->stream);
  HIPCHK(hipGetLastError());
  return 0;
}

// closed-form Savitzky-Golay coefficients: the same expression as the oracle, evaluated on the host
static void savgol_coeffs(int window, double* c) {
  int m = (window - 1) / 2;
  double den = (double)(2 * m - 1) * (double)(2 * m + 1) * (double)(2 * m + 3);
  for (int k = -m; k <= 0; ++k) c[k + m] = 3.0 * (double)(3 * m * m + 3 * m - 1 - 5 * k * k) / den;      // (symmetric: k_peaks reads c[0 .. m]; PeaksArgs::coef holds 64 = the window limit of c3_create)
}

static int run_peaks(c3_handle* h) {
  static const int blocks_per_cu = c3k_peaks_blocks_per_cu();
  const int grid = std::min(h->n, h->n_cus * blocks_per_cu);       // = the workgroups resident at once; reads come off a queue
  h->peaks_grid = grid;
  const size_t mL = (size_t)h->maxL + 8;
  HIPCHK(h->d_bufA.ensure(sizeof(double) * mL * grid)); HIPCHK(h->d_bufB.ensure(sizeof(double) * mL * grid));
  HIPCHK(h->d_cand.ensure(sizeof(int32_t) * (mL / 2 + 2) * grid)); HIPCHK(h->d_cst.ensure((mL / 2 + 2) * grid));
  HIPCHK(h->d_raw.ensure(sizeof(int32_t) * (size_t)h->n * C3_MAX_PEAKS)); HIPCHK(h->d_nraw.ensure(sizeof(int32_t) * (size_t)h->n));
  PeaksArgs a; memset(&a, 0, sizeof(a));
  a.b = dev_batch(h); a.track = h->d_track.as<int32_t>(); a.info = h->d_info.as<C3Info>();
  a.bufA = h->d_bufA.as<double>(); a.bufB = h->d_bufB.as<double>(); a.cand = h->d_cand.as<int32_t>(); a.cstate = h->d_cst.as<uint8_t>();
  a.raw_peaks = h->d_raw.as<int32_t>(); a.n_raw = h->d_nraw.as<int32_t>(); a.sp_len = h->d_sp_len.as<int>();
  savgol_coeffs(h->cfg.sg_window, a.coef);
  a.maxL = (int64_t)mL; a.window = h->cfg.sg_window; a.iters = h->cfg.sg_iters; a.min_dist = h->cfg.mdistcutoff;
  a.queue = h->d_counter.as<int>() + 60;
  HIPCHK(hipMemsetAsync(a.queue, 0, sizeof(int), h->stream));
  c3k_launch_peaks(&a, grid, h->stream);
  HIPCHK(hipGetLastError());
  return 0;
}

// summary of the split -> work list + capacities
static int copy_summary(c3_handle* h) {
  const int n = h->n;
  HIPCHK(h->d_sum.ensure(sizeof(Summary) * (size_t)n));
  hipLaunchKernelGGL(k_summary, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_info.as<C3Info>(), h->d_off.as<int64_t>(), n, h->d_sum.as<Summary>());
  h->sum.resize(n);
  HIPCHK(hipMemcpyAsync(h->sum.data(), h->d_sum.p, sizeof(Summary) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  // This wait (k_conk + k_peaks, ~15 % of a batch) is followed by the only host section the GPU waits for: the work list.  A
  // thread that slept through it (blocking sync) wakes up on a core that has dropped its clock, and the section then takes twice
  // as long (measured: 2.0 ms against 0.9 per 100 000 reads); so THIS wait polls.
  if (!getenv("C3_NO_SPIN")) {
    hipError_t q;
    while ((q = hipStreamQuery(h->stream)) == hipErrorNotReady) { for (int k_ = 0; k_ < 64; ++k_) __builtin_ia32_pause(); }
    if (q != hipSuccess) HIPCHK(q);
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  // longest used prefix of the per-read arrays (peaks / kept subreads are final after k_peaks; the zero-repeat rescue adds two)
  int k = 2;
  for (int i = 0; i < n; ++i) k = std::max(k, std::max(h->sum[i].n_peaks, h->sum[i].n_sub));
  h->res_prefix = std::min((k + 7) & ~7, (int)C3_MAX_PEAKS);
  return 0;
}

static void fill_zero_args(c3_handle* h, ZeroArgs& z, int nz) {
  memset(&z, 0, sizeof(z));
  z.b = dev_batch(h); z.info = h->d_info.as<C3Info>(); z.p = dev_params(h->cfg); z.counter = h->d_counter.as<int>();
  z.work = h->d_zwork.as<int>(); z.n_work = nz;
  z.D = h->s_zero_d.as<uint8_t>(); z.zinfo = h->d_zinfo.as<int4>(); z.zflag = h->d_zflag.as<uint8_t>();
  z.draft = h->d_draft.as<uint8_t>(); z.cons = h->d_cons.as<char>();
}

// zero-repeat rescue, first half (bin/determine_consensus.py:14-18,106-128): reads whose split kept no
// subread but has both dangling pieces get their overlap located and become 2-subread POA jobs
static int run_zero(c3_handle* h) {
  h->zwork.clear();
  HIPCHK(h->d_zflag.ensure((size_t)h->n + 16));
  HIPCHK(hipMemsetAsync(h->d_zflag.p, 0, (size_t)h->n, h->stream));
  if (!h->cfg.zero || h->injected) return 0;
  long long dmax = 0; int fmax = 0;
  for (int i = 0; i < h->n; ++i) {
    const Summary& s = h->sum[i];
    if (s.status == C3_ST_NO_CONSENSUS && s.n_sub == 0 && s.front > 0 && s.tail > 0 &&
        (long long)s.front * s.tail <= (16 << 20)) {                       // (oracle/c3o_zero.c: zr_max_cells)
      h->zwork.push_back(i);
      dmax = std::max(dmax, (long long)(s.front + 1) * (s.tail + 1));
      fmax = std::max(fmax, s.front);
    }
  }
  const int nz = (int)h->zwork.size();
  if (nz == 0) return 0;
  const int grid = std::min(nz, 512);
  HIPCHK(h->d_zwork.ensure(sizeof(int) * (size_t)nz)); HIPCHK(h->d_zinfo.ensure(sizeof(int4) * (size_t)h->n));
  dmax = (dmax + 15) & ~15LL;
  const int rowcap = fmax > 4096 ? fmax : 0;                                // k_zero: ZW columns live in LDS
  const long long dstride = dmax + (rowcap ? (((long long)(rowcap + 1) * 8 + 15) & ~15LL) : 0);
  HIPCHK(h->s_zero_d.ensure((size_t)dstride * grid + 64));
  HIPCHK(hipMemcpyAsync(h->d_zwork.p, h->zwork.data(), sizeof(int) * (size_t)nz, hipMemcpyHostToDevice, h->stream));
  ZeroArgs z; fill_zero_args(h, z, nz); z.dcap = dmax; z.dstride = dstride; z.rowcap = rowcap;
  DBG("zero: nz=%d grid=%d dmax=%lld\n", nz, grid, dmax);
  c3k_launch_zero(&z, grid, h->stream);
  HIPCHK(hipGetLastError());
  { int r_ = copy_summary(h); DBG("zero done\n"); return r_; }              // the rescued reads now carry 2 pseudo-subreads
}

static int fetch_summary(c3_handle* h) {
  int rc = copy_summary(h);                       // (waits for k_conk + k_peaks: not host time)
  DBG("summary copied\n");
  if (rc) return rc;
  const auto wl0 = std::chrono::steady_clock::now();
  struct WlTimer { c3_handle* h; std::chrono::steady_clock::time_point t0; ~WlTimer() { h->tm.ms_host_worklist = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } wl_timer_{h, wl0};
  HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 64, h->stream));
  if ((rc = run_zero(h))) return rc;
  const int n = h->n;
  h->work.clear();
  for (int i = 0; i < n; ++i) if (h->sum[i].status == C3_ST_OK && h->sum[i].n_sub >= 1) h->work.push_back(i);
  // longest first: better tail behaviour of the dynamic work queues.  Same order as a stable sort by descending cost
  // (ties: read order), done on packed (inverted cost, read) keys -- the GPU is idle while this runs
  if (n < (1 << 24)) {
    // stable LSD radix sort of (inverted cost, read) keys, 4 passes of 10 bits over the 40 cost bits (the reads are already in
    // index order, so stability gives the tie order for free): ~0.5 ms per 100 000 reads where std::sort took ~4 ms
    const size_t m = h->work.size();
    std::vector<uint64_t> keys(m), tmp(m);
    for (size_t k = 0; k < m; ++k) {
      const int i = h->work[k];
      const uint64_t cost = (uint64_t)h->sum[i].sum_sub * (uint64_t)h->sum[i].n_sub;          // < 2^40
      keys[k] = ((((uint64_t)1 << 40) - 1 - cost) << 24) | (uint64_t)i;
    }
    uint64_t* src = keys.data(); uint64_t* dst = tmp.data();
    for (int pass = 0; pass < 4; ++pass) {
      const int sh = 24 + 10 * pass;
      size_t cnt[1025] = {0};
      for (size_t k = 0; k < m; ++k) ++cnt[((src[k] >> sh) & 1023) + 1];
      for (int b = 0; b < 1024; ++b) cnt[b + 1] += cnt[b];
      for (size_t k = 0; k < m; ++k) dst[cnt[(src[k] >> sh) & 1023]++] = src[k];
      std::swap(src, dst);
    }
    for (size_t k = 0; k < m; ++k) h->work[k] = (int)(src[k] & 0xffffff);
  } else {
    std::stable_sort(h->work.begin(), h->work.end(), [&](int x, int y) {
      long cx = (long)h->sum[x].sum_sub * h->sum[x].n_sub, cy = (long)h->sum[y].sum_sub * h->sum[y].n_sub; return cx > cy; });
  }
  HIPCHK(h->d_work.ensure(sizeof(int) * std::max<size_t>(h->work.size(), 1)));
  if (!h->work.empty()) HIPCHK(hipMemcpyAsync(h->d_work.p, h->work.data(), sizeof(int) * h->work.size(), hipMemcpyHostToDevice, h->stream));
  return 0;
}

// one launch of k_poa over `nw` reads of `d_work` with the given capacities
static int launch_poa(c3_handle* h, const int* d_work, int nw, int Ncap, int K, int Pcap, long long cells, int* d_overflow, int* d_overflow16, int waves_per_cu, int wide_ring,
                      DBuf* cnt_buf = nullptr, hipStream_t st = nullptr) {
  if (!cnt_buf) cnt_buf = &h->d_counter;          // (the overlapped last pass counts in a buffer of its own: k_prep / k_window use d_counter meanwhile)
  if (!st) st = h->stream;
  const size_t N = (size_t)Ncap;
  const int NI = 19;      // int arrays of N (c3_args.h)
  cells = (cells + 15) & ~15LL;                   // every per-slot arena starts 16-byte aligned
  cells = (cells + 63) & ~63LL;
  // far arena (32-bit cells of rows with a successor beyond the LDS ring, rows wider than a ring slot, rows with > 4 predecessors):
  // a quarter of the cells in the first pass (a few per cent are used), all of them in the 32-bit pass, where every row is far
  const bool w32 = d_overflow16 == nullptr || getenv("C3_DEBUG_POA32");      // the pass that takes the reads beyond 16 bits (test hook: every pass)
  const long long far = w32 ? cells : cells >> 2;
  const size_t per_slot = N * (NI * 4 + 8 + 5 + 32 + 4 * C3_JUMP_LEVELS) + N * K * 12 + (size_t)cells * 2 + (size_t)far * 16 + (size_t)Pcap * 4;
  // the pass nothing may outgrow runs as a workgroup of eight waves per read (the reads that reach it are the ones with bands as wide
  // as the subread: hundreds of ms on one wave); C3_DEBUG_POA_MW=0: the single-wave 32-bit instance instead, C3_DEBUG_POA32=2: every pass (test hooks)
  const char* e32 = getenv("C3_DEBUG_POA32"); const char* emw = getenv("C3_DEBUG_POA_MW");
  const bool mw = w32 && ((d_overflow == nullptr && d_overflow16 == nullptr && !(emw && atoi(emw) == 0)) || (e32 && atoi(e32) == 2));
  const int slots = auto_slots(h, h->cfg.slots_poa, per_slot, nw, mw ? std::max(1, waves_per_cu / 8) : waves_per_cu);
  HIPCHK(h->s_poa_i.ensure(sizeof(int) * N * NI * slots)); HIPCHK(h->s_poa_nk.ensure(sizeof(int) * N * K * 3 * slots));
  HIPCHK(h->s_poa_cells.ensure(((size_t)cells * 2 + (size_t)far * 16) * slots + 256)); HIPCHK(h->s_poa_b.ensure(N * 5 * slots)); HIPCHK(h->s_poa_sc.ensure(sizeof(long long) * N * slots));
  HIPCHK(h->s_poa_desc.ensure(sizeof(uint4) * 2 * N * slots));
  HIPCHK(h->s_poa_jump.ensure(sizeof(int) * C3_JUMP_LEVELS * N * slots));
  HIPCHK(h->s_poa_path.ensure(sizeof(int) * (size_t)Pcap * slots));
  PoaArgs a; memset(&a, 0, sizeof(a));
  a.b = dev_batch(h); a.info = h->d_info.as<C3Info>(); a.p = dev_params(h->cfg);
  a.counter = cnt_buf->as<int>(); a.work = d_work; a.n_work = nw;
  a.ibase = h->s_poa_i.as<int>(); a.ebase = h->s_poa_nk.as<int>(); a.cellsb = h->s_poa_cells.as<char>();
  a.bbase = h->s_poa_b.as<uint8_t>(); a.score = h->s_poa_sc.as<long long>();
  a.Ncap = Ncap; a.K = K; a.Pcap = Pcap; a.cells_cap = (int)cells; a.desc = h->s_poa_desc.as<uint4>(); a.jump = h->s_poa_jump.as<int>();
  a.pbase = h->s_poa_path.as<int>(); a.overflow = d_overflow; a.overflow16 = d_overflow16;
  if (const char* e = getenv("C3_DEBUG_POA_RBSPAN")) a.rb_span = std::max(3300, atoi(e));      // (>= the 400 units below the bias + a row's growth)
  a.draft = h->d_draft.as<uint8_t>(); a.tpos = h->d_tpos.as<int32_t>();
  a.msa_dbg = nullptr; a.msa_off = nullptr; a.msa_len = nullptr;
  if (h->debug_msa) { a.msa_dbg = h->d_msa.as<uint8_t>(); a.msa_off = h->d_msa_off.as<int64_t>(); a.msa_len = h->d_msa_len.as<int>(); }
  a.phases = (unsigned long long*)(cnt_buf->as<char>() + 64);
  DBG("poa: nw=%d Ncap=%d K=%d cells=%lld slots=%d (%.1f MB per slot)%s\n", nw, Ncap, K, cells, slots, per_slot / 1048576.0, d_overflow ? "" : (d_overflow16 ? " [full-size pass]" : (mw ? " [32-bit pass, eight waves per read]" : " [32-bit pass]")));
  // the pass with an overflow list runs the 16-bit rows; the final pass (no list) the 32-bit rows only (C3_DEBUG_POA32: test hook, first pass too)
  if (mw) c3k_launch_poa_mw(&a, slots, st);
  else c3k_launch_poa(&a, slots, w32 ? 1 : 0, wide_ring, st);
  HIPCHK(hipGetLastError());
  return 0;
}

// K3 over the work list.  The per-slot scratch (graph arrays, DP cells) is sized for the TYPICAL alignment of the batch --
// small slots mean more resident waves, and the DP kernels live on resident waves -- and the few reads that overflow it
// (ragged subread lengths widen the adaptive band; long insertions add nodes) are queued by the kernel and redone by a
// second launch with worst-case scratch, so no read is ever lost to the smaller first-pass capacity.
static int run_poa(c3_handle* h, bool polish_follows) {
  h->tail_pending = false; h->strag.clear();
  const int nw = (int)h->work.size();
  HIPCHK(h->d_draft.ensure((size_t)h->total + 64)); HIPCHK(h->d_tpos.ensure(sizeof(int32_t) * (size_t)h->total + 64));
  HIPCHK(h->d_cons.ensure((size_t)h->total + 64));
  HIPCHK(hipMemsetAsync(h->d_tpos.p, 0xff, sizeof(int32_t) * (size_t)h->total, h->stream));
  if (nw == 0) return 0;
  int max_sum = 0, max_ns = 0, max_q = 0;
  for (int i : h->work) { max_sum = std::max(max_sum, h->sum[i].sum_sub); max_ns = std::max(max_ns, h->sum[i].n_sub); max_q = std::max(max_q, h->sum[i].max_sub); }
  const int Ncap_full = max_sum + 8, K = max_ns + 1, Pcap = max_sum + 8;
  const int w = h->cfg.poa_band_b + (int)(h->cfg.poa_band_f * max_q);
  long long cells_full = (long long)(2 * max_q + 2) * (2 * w + 1 + max_q / 5);
  if (max_ns < 2) cells_full = 64;
  // (every cell capacity stays max_q + 512 below INT_MAX: the kernel's `used + width > capacity` tests add a row -- at most a subread
  // wide -- or up to 256 cells of head room to a 32-bit count that never exceeds the capacity, and must not wrap)
  const long long cells_max = 0x7fffffffLL - (long long)max_q - 512;
  if (cells_full > cells_max) cells_full = cells_max;
  // the LAST pass must hold any alignment the reference would finish: every node a row (a graph never has more nodes than
  // bases went into it), every row as wide as the subread -- per read, not from the batch maxima.  (With match << mismatch an
  // alignment prefers gaps to mismatches and nearly every base becomes a node of its own: tools/fuzz_parity3.py seeds 55, 93, 111
  // ended such reads as LIMIT while the oracle finished them.)
  long long cells_worst = cells_full;
  for (int i : h->work) if (h->sum[i].n_sub >= 2) cells_worst = std::max(cells_worst, (long long)(h->sum[i].sum_sub + 8) * (h->sum[i].max_sub + 2));
  // ... clamped to what ONE slot of the last pass may allocate (18 bytes per cell of the memory budget auto_slots() works with; the row
  // loops add up to 256 cells of head room to an int: stay clear of INT_MAX).  A read beyond it -- a 150 kb read with a missed peak --
  // ends as C3_ST_LIMIT by the kernel's own capacity check; it must not turn into an allocation failure for the whole batch
  {
    const long long budget = (long long)((h->mem_total ? h->mem_total / 3 : ((size_t)64 << 30)) / 18) - (long long)(Ncap_full + Pcap) * 16;
    cells_worst = std::min(cells_worst, std::max(budget, cells_full));
    if (cells_worst > cells_max) cells_worst = cells_max;
  }
  // typical need: every further subread adds ~12 % nodes (mismatch siblings + insertions) to a graph of max_q nodes; a row
  // holds 2w+1 cells plus the drift between the row's nominal column and the argmax of its predecessors
  const double nodes_typ = (double)max_q * (1.0 + 0.15 * std::max(0, max_ns - 1));
  int Ncap = (int)std::min<double>(Ncap_full, 1.3 * nodes_typ + 256);
  long long cells = std::min<long long>(cells_full, (long long)(1.5 * nodes_typ * (2 * w + 12)) + 4096);
  if (const char* e_ = getenv("C3_DEBUG_POA_SMALL")) { Ncap = std::min(Ncap_full, std::max(64, atoi(e_))); cells = std::min<long long>(cells_full, 16LL * Ncap); }   // test hook: forces the second pass
  if (h->debug_msa) {
    std::vector<int64_t> mo(h->n + 1, 0);
    for (int i = 0; i < h->n; ++i) mo[i + 1] = mo[i] + (int64_t)h->sum[i].n_sub * (h->sum[i].sum_sub + 2);
    HIPCHK(h->d_msa.ensure((size_t)mo[h->n] + 64)); HIPCHK(h->d_msa_off.ensure(sizeof(int64_t) * (h->n + 1))); HIPCHK(h->d_msa_len.ensure(sizeof(int) * h->n));
    HIPCHK(hipMemcpyAsync(h->d_msa_off.p, mo.data(), sizeof(int64_t) * (h->n + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->d_msa_len.p, 0, sizeof(int) * h->n, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  HIPCHK(h->d_overflow.ensure(sizeof(int) * 2 * (size_t)nw));
  HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 4, h->stream));                       // work queue only: [2..3] already holds the zero-repeat cells
  HIPCHK(hipMemsetAsync(h->d_counter.as<char>() + 16, 0, 240, h->stream));       // [4] overflow count, phase counters
  // ring geometry of the first pass: subreads beyond the LDS query copy (1792 bases) or with bands beyond two 64-column chunks
  // (w = band_b + band_f * Q; a row holds 2w+1 columns + the drift of its predecessors' maxima) take the WIDE instance
  // (4 ring rows of 192 cells, sliding query window); C3_DEBUG_POA_WIDE = 0 / 1 forces one (test hook)
  int wide_ring = (max_q > 1792 || 2 * w + 1 + 24 > 128) ? 1 : 0;
  if (const char* e_ = getenv("C3_DEBUG_POA_WIDE")) wide_ring = atoi(e_) ? 1 : 0;
  // pass 1: scratch for the typical alignment.  Its two lists: reads the scratch was too small for -> pass 2 (the same kernel, worst-case
  // scratch); reads with a score beyond the 16-bit cells (from either pass) -> pass 3 (the 32-bit instance, worst-case scratch)
  int* ovA = h->d_overflow.as<int>(); int* ovB = h->d_overflow.as<int>() + nw;
  int rc = launch_poa(h, h->d_work.as<int>(), nw, Ncap, K, Pcap, cells, ovA, ovB, 24, wide_ring);
  if (rc) return rc;
  h->n_poa_redo = 0; h->n_poa_redo16 = 0;
  {
    int cnt[8];
    HIPCHK(hipMemcpyAsync(cnt, h->d_counter.p, 32, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int beyond_p1 = cnt[5];                   // reads pass 1 handed straight to the last pass (cnt[5] keeps counting through pass 2)
    if (cnt[4] > 0) {
      h->n_poa_redo = cnt[4];
      HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 4, h->stream));
      if ((rc = launch_poa(h, ovA, cnt[4], Ncap_full, K, Pcap, cells_full, nullptr, ovB, 24, wide_ring))) return rc;
      HIPCHK(hipMemcpyAsync(cnt, h->d_counter.p, 32, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (const char* e = getenv("C3_DEBUG_POA_PUNT_MOD")) {
      // test hook (host only -- nothing in the kernels' row loops pays for it): every read with rid % k == 0 and two or more subreads joins
      // the last pass's list as if one of its scores had left the 16-bit cells; the last pass recomputes and overwrites its draft
      const int k_ = std::max(1, atoi(e));
      std::vector<int> lst(cnt[5] > 0 ? cnt[5] : 0);
      if (cnt[5] > 0) { HIPCHK(hipMemcpyAsync(lst.data(), ovB, sizeof(int) * (size_t)cnt[5], hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); }
      std::vector<char> in(h->n, 0);
      for (int r : lst) in[r] = 1;
      for (int r : h->work) if (r % k_ == 0 && h->sum[r].n_sub >= 2 && !in[r]) lst.push_back(r);
      if (!lst.empty()) { HIPCHK(hipMemcpyAsync(ovB, lst.data(), sizeof(int) * lst.size(), hipMemcpyHostToDevice, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); }
      cnt[5] = (int)lst.size();
    }
    if (cnt[5] > 0) {
      h->n_poa_redo += beyond_p1; h->n_poa_redo16 = cnt[5];      // distinct reads redone: a read pass 2 sent on is in cnt[4] already
      // The last pass: a handful of reads on a handful of CUs for 100+ ms (cfgL: four reads, 132 ms of a 1.1 s batch).  When the polish
      // follows in this call it runs on a stream of its own BESIDE k_prep / k_window / k_stitch of all the other reads; the stragglers are
      // polished by a small tail (run_tail).  Not when a straggler is a zero-repeat rescue (k_zero_finish needs its draft first), and
      // C3_NO_TAIL_OVERLAP=1 keeps everything in series (A/B and test hook).
      h->strag.resize(cnt[5]);
      HIPCHK(hipMemcpyAsync(h->strag.data(), ovB, sizeof(int) * (size_t)cnt[5], hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      bool overlap = polish_follows && !getenv("C3_NO_TAIL_OVERLAP") && cnt[5] < nw;
      if (overlap && !h->zwork.empty()) {
        std::vector<char> isz(h->n, 0);
        for (int z : h->zwork) isz[z] = 1;
        for (int r : h->strag) if (isz[r]) { overlap = false; break; }
      }
      if (overlap) {
        HIPCHK(h->d_counter_mw.ensure(512));
        HIPCHK(hipMemsetAsync(h->d_counter_mw.p, 0, 512, h->stream_mw));       // (the main stream is idle: its passes were waited for above)
        HIPCHK(hipEventRecord(h->ev_mw[0], h->stream_mw));
        h->tail_pending = true;                 // (set BEFORE the launch: every error return from here on waits for stream_mw -- TailGuard in c3_batch_run)
        if ((rc = launch_poa(h, ovB, cnt[5], Ncap_full, K, Pcap, cells_worst, nullptr, nullptr, 24, 0, &h->d_counter_mw, h->stream_mw))) return rc;
        HIPCHK(hipEventRecord(h->ev_mw[1], h->stream_mw));
        // the batch without the stragglers (order kept: longest first): what the polish beside the last pass works on, and what the
        // draft statistics below may read -- the stragglers' C3Info is being written by the last pass right now
        std::vector<char> iss(h->n, 0);
        for (int r : h->strag) iss[r] = 1;
        h->work_main.clear();
        for (int r : h->work) if (!iss[r]) h->work_main.push_back(r);
        HIPCHK(h->d_work_main.ensure(sizeof(int) * std::max<size_t>(h->work_main.size(), 1)));
        if (!h->work_main.empty()) HIPCHK(hipMemcpyAsync(h->d_work_main.p, h->work_main.data(), sizeof(int) * h->work_main.size(), hipMemcpyHostToDevice, h->stream));
      } else {
        h->strag.clear();
        HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 4, h->stream));
        if ((rc = launch_poa(h, ovB, cnt[5], Ncap_full, K, Pcap, cells_worst, nullptr, nullptr, 24, 0))) return rc;
      }
    }
  }
  if (!h->zwork.empty()) {             // zero-repeat rescue, second half: stitch left + overlap consensus + right
    ZeroArgs z; fill_zero_args(h, z, (int)h->zwork.size());
    c3k_launch_zero_finish(&z, std::min((int)h->zwork.size(), 512), h->stream);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(h->phase_poa, h->d_counter.as<char>() + 64, 128, hipMemcpyDeviceToHost, h->stream));
  {
    const int* dw_ = h->tail_pending ? h->d_work_main.as<int>() : h->d_work.as<int>();
    const int nw_ = h->tail_pending ? (int)h->work_main.size() : nw;
    if (nw_ > 0) hipLaunchKernelGGL(k_draft_stats, dim3((nw_ + 255) / 256), dim3(256), 0, h->stream, h->d_info.as<C3Info>(), dw_, nw_, h->cfg.pol_window, h->d_counter.as<int>() + 6);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// k_prep -> k_window (two launches) -> k_stitch over one work list (the whole batch, or -- when the last POA pass runs beside it -- the
// batch without the stragglers and then the stragglers alone); times and counters ADD to h->tm (c3_batch_run zeroes them)
// max_draft / sum_win: longest draft and window count of THIS work list as k_draft_stats measured them (-1: the drafts did not come from
// k_poa of this batch -- the tables are sized from bounds).  Arguments, not handle state: the tail's sizing must not replace the batch's
static int run_polish(c3_handle* h, const std::vector<int>& work, const int* d_work, int max_draft, long long sum_win, float* ms_prep, float* ms_win, float* ms_st) {
  const int nw = (int)work.size();
  DBG("polish: nw=%d\n", nw);
  HIPCHK(h->d_cons.ensure((size_t)h->total + 64));
  if (nw == 0) return 0;
  const int WL = h->cfg.pol_window;
  int max_ns = 0, max_q = 0, max_dang = 0; long long wcap = 0;
  for (int i : work) {
    max_ns = std::max(max_ns, h->sum[i].n_sub); max_q = std::max(max_q, h->sum[i].max_sub); max_dang = std::max(max_dang, h->sum[i].max_dang);
    wcap += (2 * h->sum[i].max_sub + WL - 1) / WL + 1;
  }
  int NLcap = max_ns + 2, NWcap = (2 * max_q + WL - 1) / WL + 1;
  if (max_draft >= 0) {       // the drafts exist: the window tables are sized for them, not for a bound (a draft is a path of the graph and can be longer than twice the longest subread)
    NWcap = (max_draft + WL - 1) / WL + 1;
    wcap = sum_win + 8;
  }
  const int64_t ecap = ((int64_t)(max_dang + 2) / 3 + 2) * 256;    // 2-bit directions: one dword per lane and three piece rows
  const size_t per_slot_prep = (size_t)ecap + (size_t)NLcap * NWcap * 8;
  const int slots_p = auto_slots(h, h->cfg.slots_poa, per_slot_prep, nw, getenv("C3_DEBUG_PREP_WPC") ? atoi(getenv("C3_DEBUG_PREP_WPC")) : 20);
  HIPCHK(h->s_eD.ensure((size_t)ecap * slots_p));
  HIPCHK(h->s_lw.ensure(sizeof(int) * (size_t)NLcap * NWcap * 2 * slots_p));
  HIPCHK(h->d_wrec.ensure(sizeof(WinRec) * (size_t)wcap)); HIPCHK(h->d_wlay.ensure(sizeof(WLayer) * (size_t)wcap * NLcap));
  HIPCHK(h->d_wbase.ensure(sizeof(int) * (size_t)h->n));
  PrepArgs p; memset(&p, 0, sizeof(p));
  p.b = dev_batch(h); p.info = h->d_info.as<C3Info>(); p.p = dev_params(h->cfg);
  p.counter = h->d_counter.as<int>(); p.work = d_work; p.n_work = nw;
  p.draft = h->d_draft.as<uint8_t>(); p.tpos = h->d_tpos.as<int32_t>();
  p.eH = h->s_eH.as<int32_t>(); p.eD = h->s_eD.as<uint8_t>(); p.ecap = ecap;
  p.lw_first = h->s_lw.as<int>(); p.lw_last = p.lw_first + (size_t)NLcap * NWcap * slots_p; p.NLcap = NLcap; p.NWcap = NWcap;
  p.wrec = h->d_wrec.as<WinRec>(); p.wlay = h->d_wlay.as<WLayer>(); p.win_base = h->d_wbase.as<int>();
  p.n_windows = h->d_counter.as<int>() + 8; p.wcap = (int)std::min<long long>(wcap, 0x7fffffff);
  HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 64, h->stream));
  HIPCHK(hipEventRecord(h->ev[5], h->stream));
  DBG("prep: slots=%d ecap=%lld NL=%d NW=%d wcap=%lld\n", slots_p, (long long)ecap, NLcap, NWcap, (long long)wcap);
  c3k_launch_prep(&p, slots_p, h->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev[6], h->stream));
  int cnt[16];
  HIPCHK(hipMemcpyAsync(cnt, h->d_counter.p, 64, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->tm.cells_polish += *(long long*)(cnt + 2); h->tm.cells_polish_computed += *(long long*)(cnt + 2);       // dangling-piece extensions
  // k_prep reserves windows with an atomicAdd BEFORE its capacity check: after an overflow the counter exceeds wcap, and
  // the records past wcap were never written (the reads that overflowed carry C3_ST_LIMIT and n_win = 0)
  const int n_win = (int)std::min<long long>(cnt[8], std::min<long long>(wcap, 0x7fffffff));
  h->n_windows += n_win;
  DBG("prep done: n_win=%d\n", n_win);
  const int wout_cap = 3 * WL + 64;
  HIPCHK(hipEventRecord(h->ev[7], h->stream));
  if (n_win > 0) {
    HIPCHK(h->d_wout.ensure((size_t)n_win * wout_cap));
    const int Ncap = 3 * WL + 40 * NLcap, K = NLcap + 2;   // cfg2: 1700 nodes -> 10.2 KB of LDS per wave, 16 waves per CU
    // DP scratch per slot, in cells (4 bytes of H + 1 byte of D each).  Worst case: every node a row, 704+ columns.  The FIRST
    // launch gets what the usual layer needs -- banded rows or a matrix of at most 256 columns over a graph of a window and a
    // quarter plus the branches its layers add: 256 bytes of direction words per row (+ the index rows and some head room) -- which is
    // a fifth of the worst case; a window with a layer beyond that is queued on the device and redone by a SECOND launch with
    // worst-case scratch on a few slots (no host round trip: it reads the count from device memory).  40 GB -> 10 GB of scratch
    // at cfg2 / cfg5 on 256 CUs: that much less to allocate and to touch for the first time in a fresh process.
    const long long hcap_full = (long long)(Ncap + 1) * 64 * 12;
    const int R_typ = std::min(Ncap, WL + WL / 4 + 30 * NLcap + 64);
    long long hcap = std::min(hcap_full, (long long)(R_typ + R_typ / 2 + 4) * 256);      // (a window in the second launch runs alone on an idle device, ~1.5 ms: sized so that a usual batch has none -- at + R_typ / 4 one cfg2 window in 400 000 took it)
    if (const char* e = getenv("C3_DEBUG_HCAP_DIV")) hcap = std::max(4096LL, hcap_full / std::max(1, atoi(e)) / 64 * 64);       // test hook: smaller first-launch scratch (more windows take the second launch)
    const size_t N = (size_t)Ncap;
    const int NI = 19;      // W_INTS of k_polish.hip
    const size_t per_slot = N * (NI * 4 + 8 + 2) + N * K * 16 + (size_t)hcap * 5;
    const int slots = auto_slots(h, h->cfg.slots_win, per_slot, n_win, 24);       // six waves per SIMD (80 VGPRs, five LDS granules of 1 280 bytes)
    // the second launch holds ANY window: graph arrays for every base of every layer becoming a node, DP rows as wide as the longest
    // layer (both maxima come back from k_prep with the window count); a handful of slots when that is large
    const int Ncap2 = (int)std::min<long long>(65534, std::max<long long>(Ncap, (long long)cnt[10] + 8));
    const long long rs2 = std::max<long long>(768, (((long long)cnt[11] + 1 + 63) / 64) * 64);
    const long long hcap2 = std::max(hcap_full, (long long)(Ncap2 + 1) * rs2);
    const size_t N2 = (size_t)Ncap2;
    const size_t per_slot2 = N2 * (NI * 4 + 8 + 2) + N2 * K * 16 + sizeof(uint4) * (N2 + 1) + (size_t)hcap2 * 5;
    const int slots2 = (int)std::max<long long>(1, std::min<long long>(std::min(slots, 1024), (4LL << 30) / (long long)per_slot2));      // (up to four waves per CU: since round 5 the second launch also takes the layers that need wide unbanded rows -- noisy reads can send a few per cent of the windows there)
    HIPCHK(h->s_win_i.ensure(sizeof(int) * N * NI * slots + 64)); HIPCHK(h->s_win_nk.ensure(sizeof(int) * N * K * 4 * slots));
    HIPCHK(h->s_win_h.ensure(sizeof(int32_t) * (size_t)hcap * slots)); HIPCHK(h->s_win_d.ensure((size_t)hcap * slots + 256));
    HIPCHK(h->s_win_b.ensure(N * 2 * slots)); HIPCHK(h->s_win_sc.ensure(sizeof(long long) * N * slots));
    HIPCHK(h->s_win_desc.ensure(sizeof(uint4) * (N + 1) * slots));
    {
      HIPCHK(h->s_win_h2.ensure(sizeof(int32_t) * (size_t)hcap2 * slots2)); HIPCHK(h->s_win_d2.ensure((size_t)hcap2 * slots2 + 256));
      HIPCHK(h->s_win_i2.ensure(sizeof(int) * N2 * NI * slots2 + 64)); HIPCHK(h->s_win_nk2.ensure(sizeof(int) * N2 * K * 4 * slots2));
      HIPCHK(h->s_win_b2.ensure(N2 * 2 * slots2)); HIPCHK(h->s_win_sc2.ensure(sizeof(long long) * N2 * slots2));
      HIPCHK(h->s_win_desc2.ensure(sizeof(uint4) * (N2 + 1) * slots2));
      HIPCHK(h->d_wovf.ensure(sizeof(int) * (size_t)n_win));
    }
    // output slots of the second launch: a consensus is a path of the graph, at most Ncap2 bases; up to 1 GB of them
    const int wout2_cap = (Ncap2 + 63) & ~63;
    const int wout2_n = (int)std::max<long long>(std::min<long long>(n_win, 16), std::min<long long>(n_win, (1LL << 30) / wout2_cap));
    HIPCHK(h->d_wout2.ensure((size_t)wout2_cap * wout2_n + 64)); h->win_out2_cap = wout2_cap;
    DBG("window: Ncap=%d hcap=%lld slots=%d | full-size launch: Ncap=%d hcap=%lld slots=%d (%.1f MB per slot)\n", Ncap, hcap, slots, Ncap2, hcap2, slots2, per_slot2 / 1048576.0);
    WinArgs a; memset(&a, 0, sizeof(a));
    a.b = dev_batch(h); a.p = dev_params(h->cfg); a.counter = h->d_counter.as<int>(); a.n_win = n_win;
    a.wrec_in = h->d_wrec.as<WinRec>(); a.wrec = h->d_wrec.as<WinRec>(); a.wlay = h->d_wlay.as<WLayer>(); a.NLcap = NLcap;
    a.draft = h->d_draft.as<uint8_t>();
    a.ibase = h->s_win_i.as<int>(); a.ebase = h->s_win_nk.as<int>();
    a.base = h->s_win_b.as<uint8_t>(); a.score = h->s_win_sc.as<long long>();
    a.H = h->s_win_h.as<int32_t>(); a.D = h->s_win_d.as<uint16_t>(); a.rdesc = h->s_win_desc.as<uint4>(); a.Ncap = Ncap; a.K = K; a.hcap = hcap; a.Lcap = std::min(std::min(Ncap, 2 * WL + 30 * NLcap), ((getenv("C3_DEBUG_WIN_LDS") ? atoi(getenv("C3_DEBUG_WIN_LDS")) : 6400) - 16) / 6);       // (LDS per wave capped at FIVE allocation granules of 1 280 bytes, 24 waves per CU: at cfg4 the uncapped sweep arrays took 8.5 KB and k_window ran 7 % slower; larger graphs use the global-scratch sweep)
    if (const char* e = getenv("C3_DEBUG_WIN_LCAP")) a.Lcap = std::max(64, std::min(Ncap, atoi(e)));   // test hook: forces the global-scratch consensus path
    a.wout = h->d_wout.as<uint8_t>(); a.wout_cap = wout_cap; a.wout2 = h->d_wout2.as<uint8_t>(); a.wout2_cap = wout2_cap; a.wout2_n = wout2_n;
    HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 256, h->stream));
    a.phases = (unsigned long long*)(h->d_counter.as<char>() + 64);
    if (const char* e = getenv("C3_DEBUG_BAND")) a.band_mode = !strcmp(e, "off") ? 1 : !strcmp(e, "fail") ? 2 : !strcmp(e, "verify") ? 3 : 0;    // test hook (tests/test_gpu_band.py)
    a.ovf_list = h->d_wovf.as<int>();
    if (const char* e = getenv("C3_DEBUG_OVF_DELAY_MS")) a.dbg_ovf_delay = atoi(e);      // test hook (tests/test_gpu_band.py): an overflow entry that appears long after its index was taken
    // Three launches (round 6).  (1) on stream_mw, FIRST, so that its few waves are resident before the first launch fills the device: the
    // full-size kernel as a consumer of the overflow list (every entry -1, the flag 0), a few waves, each of which costs one SIMD one
    // of its six first-launch waves (twice the number of windows the previous run handed over, 16..256: enough to take every window at once
    // when batches resemble each other, cheap when there are none).  (2) the first launch; behind it, on its stream, the flag.  (3) the
    // full-size kernel once more, in list + count mode, for whatever the consumers left (they stop taking tickets at the flag) -- when nothing
    // is left, its waves take one ticket each and exit.  C3_NO_WIN_CONSUMER=1: launches (2) and (3) only, the order of round 5
    WinArgs a2 = a;
    a2.ibase = h->s_win_i2.as<int>(); a2.ebase = h->s_win_nk2.as<int>(); a2.base = h->s_win_b2.as<uint8_t>(); a2.score = h->s_win_sc2.as<long long>();
    a2.rdesc = h->s_win_desc2.as<uint4>(); a2.Ncap = Ncap2; a2.Lcap = std::min(a.Lcap, Ncap2);
    a2.H = h->s_win_h2.as<int32_t>(); a2.D = h->s_win_d2.as<uint16_t>(); a2.hcap = hcap2;
    a2.wlist = h->d_wovf.as<int>(); a2.n_win_dev = h->d_counter.as<int>() + W_CNT_OVF; a2.ovf_list = nullptr; a2.done_flag = nullptr;
    const bool consumer = !getenv("C3_NO_WIN_CONSUMER") && !getenv("C3_DEBUG_SYNC") && !h->tail_pending;
    HIPCHK(hipMemsetAsync(h->d_wovf.p, 0xff, sizeof(int) * (size_t)n_win, h->stream));
    if (consumer) {
      int nc = h->win_consumers > 0 ? h->win_consumers : 32;
      if (const char* e = getenv("C3_DEBUG_WIN_CONSUMERS")) nc = atoi(e);
      nc = std::max(1, std::min(std::min(nc, 256), slots2));
      WinArgs ac = a2; ac.done_flag = h->d_counter.as<int>() + W_CNT_DONE;
      HIPCHK(hipEventRecord(h->ev_w2[0], h->stream));
      HIPCHK(hipStreamWaitEvent(h->stream_mw, h->ev_w2[0], 0));
      c3k_launch_window(&ac, nc, h->stream_mw);
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventRecord(h->ev_w2[1], h->stream_mw));
    }
    c3k_launch_window(&a, slots, h->stream);
    { const hipError_t le = hipGetLastError();
      // (whatever happened to the first launch: the flag goes out, or the consumers wait for their whole bounded spin)
      (void)hipMemsetAsync(h->d_counter.as<int>() + W_CNT_DONE, 0x01, sizeof(int), h->stream);
      if (consumer) (void)hipStreamWaitEvent(h->stream, h->ev_w2[1], 0);
      HIPCHK(le); }
    if (getenv("C3_DEBUG_SYNC")) { HIPCHK(hipStreamSynchronize(h->stream)); DBG("window: first launch done\n"); }
    c3k_launch_window(&a2, slots2, h->stream);
    HIPCHK(hipGetLastError());
    if (getenv("C3_DEBUG_SYNC")) { HIPCHK(hipStreamSynchronize(h->stream)); DBG("window: full-size launch done\n"); }
    HIPCHK(hipMemcpyAsync(h->phase_win, h->d_counter.as<char>() + 64, 128, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(hipEventRecord(h->ev[8], h->stream));
  StitchArgs s; memset(&s, 0, sizeof(s));
  s.b = dev_batch(h); s.info = h->d_info.as<C3Info>(); s.work = d_work; s.n_work = nw;
  s.wrec = h->d_wrec.as<WinRec>(); s.win_base = h->d_wbase.as<int>(); s.wout = h->d_wout.as<uint8_t>(); s.wout_cap = wout_cap;
  s.cons = h->d_cons.as<char>(); s.zflag = h->d_zflag.as<uint8_t>(); s.wout2 = h->d_wout2.as<uint8_t>(); s.wout2_cap = h->win_out2_cap;
  DBG("stitch\n");
  c3k_launch_stitch(&s, std::min(nw, h->n_cus * 16), h->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev[9], h->stream));
  int cnt_all[64];
  HIPCHK(hipMemcpyAsync(cnt_all, h->d_counter.p, 256, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  memcpy(cnt, cnt_all, 64);
  if (n_win > 0) { h->tm.n_win_redo += cnt_all[W_CNT_OVF]; h->win_consumers = std::max(16, std::min(256, 2 * cnt_all[W_CNT_OVF])); }
  if (n_win > 0) DBG("window: second launch %d; given up: backbone %d scratch %d nodes %d consensus %d\n", cnt_all[W_CNT_OVF], cnt_all[W_CNT_WHY], cnt_all[W_CNT_WHY + 1], cnt_all[W_CNT_WHY + 2], cnt_all[W_CNT_WHY + 3]);
  if (n_win > 0) { h->tm.cells_polish += *(long long*)(cnt + 2); h->tm.cells_polish_computed += *(long long*)(cnt + 4); h->tm.n_band_layers += cnt[6]; h->tm.n_band_fallback += cnt[7]; h->tm.n_band_mismatch += cnt[8]; if (cnt[8]) fprintf(stderr, "c3poa: band verify mismatch in window %d layer %d (R = %d): last differing base q = %d, band row %d, full row %d, row of q+1 = %d\n", cnt[9], cnt[10], cnt[11], cnt[12], cnt[13], cnt[14], cnt[15]); }
  { float a_, b_, c_;
    HIPCHK(hipEventElapsedTime(&a_, h->ev[5], h->ev[6])); HIPCHK(hipEventElapsedTime(&b_, h->ev[7], h->ev[8])); HIPCHK(hipEventElapsedTime(&c_, h->ev[8], h->ev[9]));
    *ms_prep += a_; *ms_win += b_; *ms_st += c_; }
  h->tm.n_windows += n_win;
  return 0;
}

extern "C" int c3_batch_run(c3_handle* h, int stages) {
  if (!h || h->n <= 0) return C3_E_STATE;
  HIPCHK(hipSetDevice(h->cfg.device));
  int rc;
  float ms;
  hipEvent_t t0 = h->ev[0], t1 = h->ev[1], t2 = h->ev[2], t3 = h->ev[3], t4 = h->ev[4];
  // any return while the last POA pass is still running on stream_mw (a failing run_polish, a HIPCHK) waits for it first: the next call
  // may free or reuse s_poa_* / d_overflow / d_info under the running kernel otherwise
  struct TailGuard { c3_handle* h; ~TailGuard() { if (h->tail_pending) { (void)hipStreamSynchronize(h->stream_mw); h->tail_pending = false; h->strag.clear(); } } } tail_guard{h};
  DBG("run: start n=%d\n", h->n);
  const auto wall0 = std::chrono::steady_clock::now();
  const double alloc0 = g_alloc_ms;
  // per-run figures start from zero: repeated runs of one resident batch (bench.py, tools/) must not accumulate
  if (stages & C3_STAGE_CONK) { h->tm.ms_conk = 0; h->tm.cells_conk = 0; }
  if (stages & C3_STAGE_PEAKS) h->tm.ms_peaks = 0;
  if (stages & C3_STAGE_POA) { h->tm.ms_poa = 0; h->tm.cells_poa = 0; h->tm.ms_poa_tail = 0; }
  if (stages & C3_STAGE_POLISH) { h->tm.ms_prep = h->tm.ms_window = h->tm.ms_stitch = 0; h->tm.cells_polish = 0; h->tm.cells_polish_computed = 0; h->tm.n_band_layers = h->tm.n_band_fallback = h->tm.n_band_mismatch = 0; h->tm.n_windows = 0; h->tm.n_win_redo = 0; }
  HIPCHK(hipEventRecord(t0, h->stream));
  if (stages & C3_STAGE_CONK) { if ((rc = run_conk(h))) return rc; h->tm.cells_conk = 0; for (int i = 0; i < h->n; ++i) h->tm.cells_conk += (h->off[i + 1] - h->off[i]) * (int64_t)h->max_spl; }
  HIPCHK(hipEventRecord(t1, h->stream));
  if (stages & C3_STAGE_PEAKS) { if ((rc = run_peaks(h))) return rc; }
  HIPCHK(hipEventRecord(t2, h->stream));
  float ms_prep = 0, ms_win = 0, ms_st = 0;
  if (stages & (C3_STAGE_POA | C3_STAGE_POLISH)) {
    DBG("run: conk+peaks launched\n");
    if ((rc = fetch_summary(h))) return rc;
    DBG("run: work list ready (%zu reads)\n", h->work.size());
    HIPCHK(hipEventRecord(t3, h->stream));
    if (stages & C3_STAGE_POA) {
      if ((rc = run_poa(h, (stages & C3_STAGE_POLISH) != 0))) return rc;
    }
    HIPCHK(hipEventRecord(t4, h->stream));
    if (stages & C3_STAGE_POA) {
      int cnt[16];
      HIPCHK(hipMemcpyAsync(cnt, h->d_counter.p, 64, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      if (!h->work.empty()) h->tm.cells_poa = *(long long*)(cnt + 2);
      h->poa_max_draft = h->work.empty() ? 0 : cnt[6]; h->poa_sum_win = h->work.empty() ? 0 : cnt[7];
      h->tm.n_poa_redo = h->n_poa_redo; h->tm.n_poa_redo16 = h->n_poa_redo16;
      DBG("run: poa done\n");
      HIPCHK(hipEventElapsedTime(&ms, t3, t4)); h->tm.ms_poa = ms;
    }
    h->n_windows = 0;
    if ((stages & C3_STAGE_POLISH) && !h->tail_pending) { if ((rc = run_polish(h, h->work, h->d_work.as<int>(), h->poa_max_draft, h->poa_sum_win, &ms_prep, &ms_win, &ms_st))) return rc; }
    if (h->tail_pending) {
      // the last POA pass is running on stream_mw: polish everything else beside it (run_poa built work_main and measured ITS drafts:
      // poa_max_draft / poa_sum_win hold the main list's figures at this point), then the stragglers
      const int main_draft = h->poa_max_draft; const long long main_win = h->poa_sum_win;
      if ((rc = run_polish(h, h->work_main, h->d_work_main.as<int>(), main_draft, main_win, &ms_prep, &ms_win, &ms_st))) return rc;
      // ... the stragglers: their drafts exist once the last pass is done
      HIPCHK(hipStreamWaitEvent(h->stream, h->ev_mw[1], 0));
      const int ns_ = (int)h->strag.size();
      const int* d_strag = h->d_overflow.as<int>() + (int)h->work.size();     // (= the last pass's work list, still in place)
      HIPCHK(hipMemsetAsync(h->d_counter.as<int>() + 6, 0, 8, h->stream));
      hipLaunchKernelGGL(k_draft_stats, dim3((ns_ + 255) / 256), dim3(256), 0, h->stream, h->d_info.as<C3Info>(), d_strag, ns_, h->cfg.pol_window, h->d_counter.as<int>() + 6);
      HIPCHK(hipGetLastError());
      int cnt[16], cmw[4];
      HIPCHK(hipMemcpyAsync(cnt, h->d_counter.p, 64, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipMemcpyAsync(cmw, h->d_counter_mw.p, 16, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      h->tm.cells_poa += *(long long*)(cmw + 2);
      // the handle keeps the figures of the WHOLE batch (a later c3_batch_run(C3_STAGE_POLISH) on this resident batch sizes its window
      // tables from them); the stragglers' own figures only size the tail below
      const int strag_draft = cnt[6]; const long long strag_win = cnt[7];
      h->poa_max_draft = std::max(main_draft, strag_draft); h->poa_sum_win = (int)std::min<long long>(main_win + strag_win, 0x7fffffff);
      { float t_ = 0; HIPCHK(hipEventElapsedTime(&t_, h->ev_mw[0], h->ev_mw[1])); h->tm.ms_poa_tail = t_; }
      h->tail_pending = false;
      if ((rc = run_polish(h, h->strag, d_strag, strag_draft, strag_win, &ms_prep, &ms_win, &ms_st))) return rc;
    }
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  DBG("run: done\n");
  HIPCHK(hipGetLastError());
  if (stages & C3_STAGE_CONK) { HIPCHK(hipEventElapsedTime(&ms, t0, t1)); h->tm.ms_conk = ms; }
  if (stages & C3_STAGE_PEAKS) { HIPCHK(hipEventElapsedTime(&ms, t1, t2)); h->tm.ms_peaks = ms; }
  if (stages & C3_STAGE_POLISH) { h->tm.ms_prep = ms_prep; h->tm.ms_window = ms_win; h->tm.ms_stitch = ms_st; }
  h->tm.ms_total = h->tm.ms_conk + h->tm.ms_peaks + h->tm.ms_poa + h->tm.ms_prep + h->tm.ms_window + h->tm.ms_stitch;
  h->tm.ms_wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  h->tm.ms_alloc = (float)(g_alloc_ms - alloc0);
  h->tm.ms_host_gap = h->tm.ms_wall - h->tm.ms_total;
  h->stages_done |= stages;
  return C3_E_OK;
}

extern "C" int c3_batch_sync(c3_handle* h) {
  if (!h) return C3_E_ARG;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize(h->stream));
  return C3_E_OK;
}

// Results of the resident batch, in two halves so that the copy can run beside the NEXT batch's kernels:
//   c3_batch_results_snapshot  (owner thread, after c3_batch_run) freezes the records and the compact consensus bytes in device
//                              buffers of their own: offsets by a device scan (one 8-byte read back for the total), one gather
//                              kernel, three strided device copies -- ~0.3 ms;
//   c3_batch_results_fetch     copies the snapshot into the caller's buffers on the handle's third stream and waits for it.  It
//                              touches nothing but the snapshot, so ANOTHER thread may call it while the owner commits and runs
//                              the next batch (the only pair of calls on one handle that may overlap).
// One snapshot exists per handle: a second _snapshot before the _fetch returns C3_E_STATE.  c3_batch_results = both, back to back.
extern "C" int c3_batch_results_snapshot(c3_handle* h) {
  if (!h || h->n <= 0) return C3_E_ARG;
  if (h->snap_pending.load(std::memory_order_acquire)) return c3_fail(h, C3_E_STATE, "c3_batch_results_snapshot: the previous snapshot has not been fetched");
  HIPCHK(hipSetDevice(h->cfg.device));
  const int n = h->n;
  DBuf& d_coff = h->d_gather_off; DBuf& d_out = h->d_gather;
  const int nb = (n + 255) / 256;
  HIPCHK(d_coff.ensure(sizeof(int64_t) * (size_t)(n + 1))); HIPCHK(h->d_coff_part.ensure(sizeof(long long) * (size_t)nb));
  hipLaunchKernelGGL(k_coff_sums, dim3(nb), dim3(256), 0, h->stream, h->d_info.as<C3Info>(), n, h->d_coff_part.as<long long>());
  hipLaunchKernelGGL(k_coff_scan, dim3(1), dim3(1024), 0, h->stream, h->d_coff_part.as<long long>(), nb, d_coff.as<int64_t>(), n);
  hipLaunchKernelGGL(k_coff_final, dim3(nb), dim3(256), 0, h->stream, h->d_info.as<C3Info>(), n, h->d_coff_part.as<long long>(), d_coff.as<int64_t>());
  long long tot = 0;
  const bool have_cons = (h->stages_done & C3_STAGE_POLISH) != 0;
  if (have_cons) {
    HIPCHK(hipMemcpyAsync(h->h_tot, d_coff.as<int64_t>() + n, sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    tot = *h->h_tot;
  }
  // A record is 3 kB of which a typical read uses ~150 bytes: the header plus the first n_peaks / n_sub entries of three
  // arrays.  When the batch's longest prefix is known (after the POA / polish stages) only those bytes are kept and cross PCIe, as
  // three strided copies; array entries past a read's n_peaks / n_sub are then UNSPECIFIED in the caller's records.
  const int kp = h->res_prefix;
  const bool prefix = kp > 0 && kp * 4 < C3_MAX_PEAKS && (h->stages_done & (C3_STAGE_POA | C3_STAGE_POLISH)) && !getenv("C3_FULL_RESULTS");
  const size_t pitch = sizeof(C3Info), head = offsetof(C3Info, peaks);
  const size_t o_sb = offsetof(C3Info, sub_beg), o_se = offsetof(C3Info, sub_end);
  HIPCHK(h->d_info_snap.ensure(pitch * (size_t)n));
  const char* src = h->d_info.as<char>(); char* snap = h->d_info_snap.as<char>();
  if (prefix) {
    HIPCHK(hipMemcpy2DAsync(snap, pitch, src, pitch, head + 4 * (size_t)kp, (size_t)n, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpy2DAsync(snap + o_sb, pitch, src + o_sb, pitch, 4 * (size_t)kp, (size_t)n, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpy2DAsync(snap + o_se, pitch, src + o_se, pitch, 4 * (size_t)kp, (size_t)n, hipMemcpyDeviceToDevice, h->stream));
  } else {
    HIPCHK(hipMemcpyAsync(snap, src, pitch * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
  }
  if (have_cons && tot > 0) {
    // gather on the device (one wave per read): the fetch is then ONE device->host copy of the compact bytes
    HIPCHK(d_out.ensure((size_t)tot + 64));
    hipLaunchKernelGGL(k_gather_cons, dim3((unsigned)std::min((n + 3) / 4, h->n_cus * 32)), dim3(256), 0, h->stream,
                       h->d_cons.as<char>(), h->d_off.as<int64_t>(), d_coff.as<int64_t>(), n, d_out.as<char>());
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(h->ev_dn, h->stream));
  h->snap_n = n; h->snap_kp = prefix ? kp : 0; h->snap_tot = tot; h->snap_cons = have_cons;
  h->snap_pending.store(true, std::memory_order_release);
  return C3_E_OK;
}

extern "C" int c3_batch_results_fetch(c3_handle* h, c3_read_result* res, char* cons, int64_t cons_cap, int64_t* cons_off) {
  if (!h || !res) return C3_E_ARG;
  if (!h->snap_pending.load(std::memory_order_acquire)) return C3_E_STATE;          // (h->err belongs to the owner thread: not touched here)
  hipError_t e;
#define DNCHK(x) do { if ((e = (x)) != hipSuccess) { h->snap_pending.store(false, std::memory_order_release); return C3_E_HIP; } } while (0)
  DNCHK(hipSetDevice(h->cfg.device));
  const int n = h->snap_n, kp = h->snap_kp;
  const size_t pitch = sizeof(C3Info), head = offsetof(C3Info, peaks);
  const size_t o_sb = offsetof(C3Info, sub_beg), o_se = offsetof(C3Info, sub_end);
  const char* snap = h->d_info_snap.as<char>(); char* dst = (char*)res;
  hipStream_t dn = h->stream_dn;
  DNCHK(hipStreamWaitEvent(dn, h->ev_dn, 0));
  if (kp > 0) {
    DNCHK(hipMemcpy2DAsync(dst, pitch, snap, pitch, head + 4 * (size_t)kp, (size_t)n, hipMemcpyDeviceToHost, dn));
    DNCHK(hipMemcpy2DAsync(dst + o_sb, pitch, snap + o_sb, pitch, 4 * (size_t)kp, (size_t)n, hipMemcpyDeviceToHost, dn));
    DNCHK(hipMemcpy2DAsync(dst + o_se, pitch, snap + o_se, pitch, 4 * (size_t)kp, (size_t)n, hipMemcpyDeviceToHost, dn));
  } else {
    DNCHK(hipMemcpyAsync(dst, snap, pitch * (size_t)n, hipMemcpyDeviceToHost, dn));
  }
  if (cons_off) DNCHK(hipMemcpyAsync(cons_off, h->d_gather_off.p, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToHost, dn));
  const bool fits = !(cons_off && cons) || h->snap_tot <= cons_cap;
  if (cons_off && cons && fits && h->snap_cons && h->snap_tot > 0) DNCHK(hipMemcpyAsync(cons, h->d_gather.p, (size_t)h->snap_tot, hipMemcpyDeviceToHost, dn));
  DNCHK(hipStreamSynchronize(dn));
#undef DNCHK
  h->snap_pending.store(false, std::memory_order_release);
  return fits ? C3_E_OK : C3_E_LIMIT;                 // too small: the records and the offsets (needed size = cons_off[n]) were still delivered
}

extern "C" int c3_batch_results(c3_handle* h, c3_read_result* res, char* cons, int64_t cons_cap, int64_t* cons_off) {
  if (!h || h->n <= 0 || !res) return C3_E_ARG;
  int rc = c3_batch_results_snapshot(h);
  if (rc) return rc;
  rc = c3_batch_results_fetch(h, res, cons, cons_cap, cons_off);
  if (rc == C3_E_LIMIT) return c3_fail(h, C3_E_LIMIT, "consensus buffer too small");
  if (rc == C3_E_HIP) return c3_fail(h, C3_E_HIP, "HIP error while copying the results");
  return rc;
}

// PMC calibration: read `bytes` with one dword per lane, write `bytes` with one dword per
// lane -- the access width the DP kernels use -- so FETCH_SIZE / WRITE_SIZE can be checked against a
// known byte count on this device before they are trusted for k_window / k_poa.
__global__ void k_calib_rw(const uint32_t* in, uint32_t* out, size_t nwords) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (size_t k = i; k < nwords; k += st) acc += in[k];
  for (size_t k = i; k < nwords; k += st) out[k] = acc + (uint32_t)k;
}
extern "C" int c3_debug_calibrate(c3_handle* h, long long bytes) {
  if (!h || bytes < 4096) return C3_E_ARG;
  HIPCHK(hipSetDevice(h->cfg.device));
  uint32_t *a = nullptr, *b = nullptr;
  HIPCHK(hipMalloc(&a, (size_t)bytes)); HIPCHK(hipMalloc(&b, (size_t)bytes));
  HIPCHK(hipMemsetAsync(a, 1, (size_t)bytes, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  hipLaunchKernelGGL(k_calib_rw, dim3(h->n_cus * 8), dim3(256), 0, h->stream, a, b, (size_t)bytes / 4);
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipFree(a)); HIPCHK(hipFree(b));
  return C3_E_OK;
}

// diagnostic builds (-DC3_PHASE_PROF) only: per-phase cycle sums of k_poa (which=0) / k_window (which=1)
extern "C" int c3_debug_phases(c3_handle* h, int which, unsigned long long* out) {
  if (!h || !out) return C3_E_ARG;
  memcpy(out, which ? h->phase_win : h->phase_poa, 128);
  return C3_E_OK;
}

extern "C" int c3_batch_timing(c3_handle* h, c3_timing* t) { if (!h || !t) return C3_E_ARG; *t = h->tm; return C3_E_OK; }

// ---- probes -----------------------------------------------------------------------------
extern "C" int c3_fetch_track(c3_handle* h, int read, int32_t* out, int64_t cap) {
  if (!h || read < 0 || read >= h->n || !out) return C3_E_ARG;
  if (!(h->stages_done & C3_STAGE_CONK)) return c3_fail(h, C3_E_STATE, "conk stage not run");
  int64_t L = h->off[read + 1] - h->off[read];
  if (cap < L) return C3_E_LIMIT;
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipMemcpy(out, h->d_track.as<int32_t>() + h->off[read], sizeof(int32_t) * (size_t)L, hipMemcpyDeviceToHost));
  return (int)L;
}
extern "C" int c3_fetch_smoothed(c3_handle* h, int read, double* out, int64_t cap) {
  if (!h || read < 0 || read >= h->n || !out) return C3_E_ARG;
  if (!(h->stages_done & C3_STAGE_PEAKS)) return c3_fail(h, C3_E_STATE, "peaks stage not run");
  if (h->n > h->peaks_grid) return c3_fail(h, C3_E_STATE, "smoothed tracks are only retained when the batch fits the peaks grid");
  int64_t L = h->off[read + 1] - h->off[read];
  if (cap < L) return C3_E_LIMIT;
  HIPCHK(hipSetDevice(h->cfg.device));
  // result buffer after `iters` ping-pong passes: A if iters is even, B if odd
  const double* src = ((h->cfg.sg_iters & 1) ? h->d_bufB.as<double>() : h->d_bufA.as<double>()) + (size_t)read * ((size_t)h->maxL + 8);
  HIPCHK(hipMemcpy(out, src, sizeof(double) * (size_t)L, hipMemcpyDeviceToHost));
  return (int)L;
}
extern "C" int c3_fetch_raw_peaks(c3_handle* h, int read, int32_t* out, int cap) {
  if (!h || read < 0 || read >= h->n || !out) return C3_E_ARG;
  if (!(h->stages_done & C3_STAGE_PEAKS)) return c3_fail(h, C3_E_STATE, "peaks stage not run");
  HIPCHK(hipSetDevice(h->cfg.device));
  int n = 0;
  HIPCHK(hipMemcpy(&n, h->d_nraw.as<int32_t>() + read, sizeof(int), hipMemcpyDeviceToHost));
  if (n > cap) return C3_E_LIMIT;
  if (n > 0) HIPCHK(hipMemcpy(out, h->d_raw.as<int32_t>() + (size_t)read * C3_MAX_PEAKS, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  return n;
}
extern "C" int c3_fetch_draft(c3_handle* h, int read, char* out, int cap) {
  if (!h || read < 0 || read >= h->n || !out) return C3_E_ARG;
  if (!(h->stages_done & C3_STAGE_POA)) return c3_fail(h, C3_E_STATE, "POA stage not run");
  HIPCHK(hipSetDevice(h->cfg.device));
  C3Info inf;
  HIPCHK(hipMemcpy(&inf, h->d_info.as<C3Info>() + read, sizeof(C3Info), hipMemcpyDeviceToHost));
  int C = inf.draft_len;
  if (C > cap) return C3_E_LIMIT;
  if (C > 0) {
    std::vector<uint8_t> tmp(C);
    HIPCHK(hipMemcpy(tmp.data(), h->d_draft.as<uint8_t>() + h->off[read], C, hipMemcpyDeviceToHost));
    for (int i = 0; i < C; ++i) out[i] = "ACGT"[tmp[i] & 3];
  }
  return C;
}
static int fetch_msa_rows(c3_handle* h, int read, int nrows, char* out, int64_t cap, int* msa_len) {
  if (!h->debug_msa || !h->d_msa.p) return c3_fail(h, C3_E_STATE, "MSA rows are only kept by c3_poa_msa / debug batches");
  int ml = 0;
  HIPCHK(hipMemcpy(&ml, h->d_msa_len.as<int>() + read, sizeof(int), hipMemcpyDeviceToHost));
  *msa_len = ml;
  if (ml <= 0) return 0;
  if ((int64_t)ml * nrows > cap) return C3_E_LIMIT;
  std::vector<int64_t> mo(2);
  HIPCHK(hipMemcpy(mo.data(), h->d_msa_off.as<int64_t>() + read, sizeof(int64_t), hipMemcpyDeviceToHost));
  std::vector<uint8_t> tmp((size_t)ml * nrows);
  HIPCHK(hipMemcpy(tmp.data(), h->d_msa.as<uint8_t>() + mo[0], tmp.size(), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < tmp.size(); ++i) out[i] = tmp[i] > 3 ? '-' : "ACGT"[tmp[i]];
  return 0;
}
extern "C" int c3_fetch_msa2(c3_handle* h, int read, char* rowA, char* rowB, int cap) {
  if (!h || read < 0 || read >= h->n || !rowA || !rowB) return C3_E_ARG;
  HIPCHK(hipSetDevice(h->cfg.device));
  std::vector<char> tmp((size_t)cap * 2 + 2);
  int ml = 0;
  int rc = fetch_msa_rows(h, read, 2, tmp.data(), (int64_t)cap * 2, &ml);
  if (rc) return rc;
  memcpy(rowA, tmp.data(), ml); memcpy(rowB, tmp.data() + ml, ml);
  return ml;
}

// inject a pre-split "read": [front][sub0]...[subn-1][tail]; skips conk/peaks
static int inject(c3_handle* h, int n, const char* const* subs, const char* const* quals, const int* lens,
                  const char* front, const char* front_q, int front_len, const char* tail, const char* tail_q, int tail_len) {
  if (n < 1 || n > C3_MAX_SUB) return c3_fail(h, C3_E_LIMIT, "1..250 subreads");
  if (h->n_spl <= 0) { const char sp[] = "ACGT"; int64_t o[2] = {0, 4}; int rc = c3_set_splints(h, 1, sp, o); if (rc) return rc; }
  std::string seq, ql;
  c3_read_result r; memset(&r, 0, sizeof(r));
  if (front && front_len > 0) { seq.append(front, front_len); if (front_q) ql.append(front_q, front_len); else ql.append(front_len, 'I'); r.has_front = 1; r.front_end = front_len; }
  for (int i = 0; i < n; ++i) {
    r.sub_beg[i] = (int)seq.size(); seq.append(subs[i], lens[i]);
    if (quals && quals[i]) ql.append(quals[i], lens[i]); else ql.append(lens[i], 'I');
    r.sub_end[i] = (int)seq.size();
  }
  r.n_sub = n; r.n_peaks = n + 1; r.status = C3_ST_OK;
  if (tail && tail_len > 0) { r.has_tail = 1; r.tail_beg = (int)seq.size(); seq.append(tail, tail_len); if (tail_q) ql.append(tail_q, tail_len); else ql.append(tail_len, 'I'); }
  int64_t off[2] = {0, (int64_t)seq.size()};
  int16_t sid = 0; char st = '+';
  int rc = c3_batch_upload(h, 1, seq.data(), ql.data(), off, &sid, &st);
  if (rc) return rc;
  HIPCHK(hipMemcpy(h->d_info.p, &r, sizeof(r), hipMemcpyHostToDevice));
  h->injected = true; h->poa_max_draft = h->poa_sum_win = -1;
  return 0;
}

// splint / strand finder (replaces the blat step of bin/preprocess.py:12-45,61-77): every read of the resident
// batch is scored against every splint on both strands with the conk kernel; out[i*n_spl*2 + s*2 + rc] =
// {max of the track, its offset, mean of the track, read length}.  assign_* picks the best candidate and
// accepts it when max >= 6 * mean (the same contrast test call_peaks applies later, bin/call_peaks.py:13) and
// max >= match*51*52/2 (a perfect 51-base match: the `matches > 50` filter of bin/preprocess.py:32).
extern "C" int c3_scan_splints(c3_handle* h, int32_t* out /* [n][n_spl][2][4] */, int16_t* assign_splint, char* assign_strand) {
  if (!h || h->n <= 0 || h->n_spl <= 0) return C3_E_STATE;
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t items = (size_t)h->n * h->n_spl * 2;
  DBuf scan;
  HIPCHK(scan.ensure(sizeof(int32_t) * 4 * items));
  HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 64, h->stream));
  ConkArgs a; memset(&a, 0, sizeof(a));
  a.b = dev_batch(h); a.sp_codes = h->d_sp_codes.as<uint8_t>(); a.sp_len = h->d_sp_len.as<int>();
  a.track = nullptr; a.info = h->d_info.as<C3Info>(); a.counter = h->d_counter.as<int>();
  a.match = h->cfg.conk_match; a.mismatch = h->cfg.conk_mismatch; a.penalty = h->cfg.conk_penalty;
  a.n_spl = h->n_spl; a.scan = scan.as<int32_t>();
  const int waves = (int)std::min<size_t>(items, (size_t)h->n_cus * 32);
  c3k_launch_conk(&a, h->max_spl, (waves + 3) / 4, 1, h->stream);
  HIPCHK(hipGetLastError());
  std::vector<int32_t> tmp(4 * items);
  HIPCHK(hipMemcpyAsync(tmp.data(), scan.p, sizeof(int32_t) * 4 * items, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  scan.release();
  if (out) memcpy(out, tmp.data(), sizeof(int32_t) * 4 * items);
  for (int i = 0; i < h->n; ++i) {
    int best = -1, bs = -1;
    for (int c = 0; c < h->n_spl * 2; ++c) { const int32_t* e = &tmp[((size_t)i * h->n_spl * 2 + c) * 4]; if (e[0] > bs) { bs = e[0]; best = c; } }
    // matches > 50 of bin/preprocess.py:32, as the diagonal sum of a perfect 51-base match
    const long long floor51 = (long long)h->cfg.conk_match * 51 * 52 / 2;
    bool ok = best >= 0 && bs >= floor51 && (long long)bs >= 6LL * tmp[((size_t)i * h->n_spl * 2 + best) * 4 + 2];
    if (assign_splint) assign_splint[i] = ok ? (int16_t)(best >> 1) : (int16_t)-1;
    if (assign_strand) assign_strand[i] = ok ? ((best & 1) ? '-' : '+') : '?';
  }
  return C3_E_OK;
}

// adapter finder of the post-processing step (replaces the blat call of C3POa_postprocessing.py:229-236): best local
// affine alignment of every read of the resident batch against every entry of the splint table (= the adapters,
// c3_set_splints) on both strands, traced back.  out[(i*n_ad + a)*2 + rc][12] = score, qStart, qEnd, tStart, tEnd
// (PSL conventions: query = read, forward coordinates; target = adapter, forward coordinates), matches, mismatches,
// qBaseInsert, tBaseInsert, qNumInsert, tNumInsert, read length.  score 0 = no alignment.
extern "C" int c3_scan_adapters(c3_handle* h, int32_t* out) {
  if (!h || h->n <= 0 || h->n_spl <= 0 || !out) return C3_E_STATE;
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t items = (size_t)h->n * h->n_spl * 2;
  const long long dcap = (long long)(h->maxL + 1) * (h->max_spl + 1) + 64;
  const int grid = (int)std::min<size_t>(items, (size_t)h->n_cus * 16);
  DBuf res, dd;
  HIPCHK(res.ensure(sizeof(int32_t) * 12 * items)); HIPCHK(dd.ensure((size_t)dcap * grid));
  HIPCHK(hipMemsetAsync(h->d_counter.p, 0, 64, h->stream));
  AdapterArgs a; memset(&a, 0, sizeof(a));
  a.b = dev_batch(h); a.p = dev_params(h->cfg); a.counter = h->d_counter.as<int>();
  a.ad_codes = h->d_sp_codes.as<uint8_t>(); a.ad_len = h->d_sp_len.as<int>(); a.n_ad = h->n_spl;
  a.D = dd.as<uint8_t>(); a.dcap = dcap; a.out = res.as<int32_t>();
  c3k_launch_adapter(&a, grid, h->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, res.p, sizeof(int32_t) * 12 * items, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  res.release(); dd.release();
  return C3_E_OK;
}

// match_index for a batch of pieces (C3POa_postprocessing.py:266-285): pieces = n slots of 64 bytes, lens[n] (<= 64);
// at most 16 indexes of at most 32 bases (idx_off[n_idx+1] into idx_cat); out[i] = winning index number or -1.
extern "C" int c3_match_index_batch(c3_handle* h, int n, const char* pieces, const int32_t* lens, int n_idx,
                                    const char* idx_cat, const int64_t* idx_off, int32_t* out) {
  if (!h || n <= 0 || !pieces || !lens || !idx_cat || !idx_off || !out) return C3_E_ARG;
  if (n_idx < 2) { for (int i = 0; i < n; ++i) out[i] = -1; return C3_E_OK; }      // the reference needs a runner-up
  if (n_idx > 16) return c3_fail(h, C3_E_LIMIT, "more than 16 indexes");
  for (int k = 0; k < n_idx; ++k) if (idx_off[k + 1] - idx_off[k] > 32 || idx_off[k + 1] < idx_off[k]) return c3_fail(h, C3_E_LIMIT, "index longer than 32 bases");
  for (int i = 0; i < n; ++i) if (lens[i] < 0 || lens[i] > 64) return c3_fail(h, C3_E_ARG, "piece longer than 64 bases");
  HIPCHK(hipSetDevice(h->cfg.device));
  DBuf dp, dl, di, doff, dout;
  const size_t ib = (size_t)idx_off[n_idx];
  HIPCHK(dp.ensure((size_t)n * 64)); HIPCHK(dl.ensure(sizeof(int) * (size_t)n)); HIPCHK(di.ensure(ib + 16));
  HIPCHK(doff.ensure(sizeof(int64_t) * (size_t)(n_idx + 1))); HIPCHK(dout.ensure(sizeof(int) * (size_t)n));
  HIPCHK(hipMemcpyAsync(dp.p, pieces, (size_t)n * 64, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dl.p, lens, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(di.p, idx_cat, ib, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(doff.p, idx_off, sizeof(int64_t) * (size_t)(n_idx + 1), hipMemcpyHostToDevice, h->stream));
  c3k_launch_match_index(dp.as<char>(), dl.as<int>(), n, n_idx, di.as<char>(), doff.as<long long>(), dout.as<int>(), h->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, dout.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  dp.release(); dl.release(); di.release(); doff.release(); dout.release();
  return C3_E_OK;
}

// pairwise_consensus(msa_rows, subreads, quals) (bin/consensus.py:76-81; call site determine_consensus.py:36-40): rows are
// the two MSA rows ('-' = gap, msa_len columns), subA/subB the ungapped subreads with their qualities.  Identical
// subreads share the later quality (the seqDict collision of consensus.py:77-79).
extern "C" int c3_pairwise_consensus(c3_handle* h, const char* rowA, const char* rowB, int msa_len,
                                     const char* subA, int lenA, const char* qualA, const char* subB, int lenB, const char* qualB,
                                     char* out, int cap, int* out_len) {
  if (!h || !rowA || !rowB || msa_len < 0 || !subA || !subB || !qualA || !qualB || !out || !out_len) return C3_E_ARG;
  *out_len = 0;
  if (msa_len == 0) return C3_E_OK;
  if (cap < msa_len) return C3_E_LIMIT;
  HIPCHK(hipSetDevice(h->cfg.device));
  auto code = [](char ch) -> uint8_t { switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                                                       case 'T': case 't': case 'U': case 'u': return 3; case '-': return 4; default: return 0; } };
  std::vector<uint8_t> rows((size_t)2 * msa_len);
  int na = 0, nb = 0;
  for (int i = 0; i < msa_len; ++i) { rows[i] = code(rowA[i]); rows[(size_t)msa_len + i] = code(rowB[i]); na += rows[i] != 4; nb += rows[(size_t)msa_len + i] != 4; }
  if (na != lenA || nb != lenB) return c3_fail(h, C3_E_ARG, "MSA rows do not spell the subreads");
  const bool same = lenA == lenB && memcmp(subA, subB, (size_t)lenA) == 0;
  DBuf d_rows, d_qa, d_qb, d_scr, d_out, d_len;
  HIPCHK(d_rows.ensure((size_t)2 * msa_len)); HIPCHK(d_qa.ensure((size_t)lenA + 16)); HIPCHK(d_qb.ensure((size_t)lenB + 16));
  HIPCHK(d_scr.ensure((size_t)2 * msa_len + 16)); HIPCHK(d_out.ensure((size_t)msa_len + 16)); HIPCHK(d_len.ensure(16));
  HIPCHK(hipMemcpyAsync(d_rows.p, rows.data(), rows.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d_qa.p, same ? qualB : qualA, (size_t)lenA, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(d_qb.p, qualB, (size_t)lenB, hipMemcpyHostToDevice, h->stream));
  c3k_launch_pairwise(d_rows.as<uint8_t>(), msa_len, d_qa.as<uint8_t>(), lenA, d_qb.as<uint8_t>(), lenB, d_scr.as<uint8_t>(), d_out.as<uint8_t>(), d_len.as<int>(), h->stream);
  HIPCHK(hipGetLastError());
  std::vector<uint8_t> codes((size_t)msa_len);
  int n = 0;
  HIPCHK(hipMemcpyAsync(&n, d_len.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(codes.data(), d_out.p, (size_t)msa_len, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) out[i] = "ACGT"[codes[(size_t)i] & 3];
  *out_len = n;
  d_rows.release(); d_qa.release(); d_qb.release(); d_scr.release(); d_out.release(); d_len.release();
  return C3_E_OK;
}

extern "C" int c3_call_peaks(c3_handle* h, const int32_t* scores, int n, int min_dist, int32_t* peaks, int cap, double* smoothed) {
  if (!h || !scores || n <= 0 || !peaks) return C3_E_ARG;
  if (h->n_spl <= 0) { const char sp[] = "ACGT"; int64_t o[2] = {0, 4}; int rc = c3_set_splints(h, 1, sp, o); if (rc) return rc; }
  std::string seq((size_t)n, 'A'), ql((size_t)n, 'I');
  int64_t off[2] = {0, n};
  int16_t sid = 0; char st = '+';
  int rc = c3_batch_upload(h, 1, seq.data(), ql.data(), off, &sid, &st);
  if (rc) return rc;
  HIPCHK(h->d_track.ensure(sizeof(int32_t) * (size_t)n + 64));
  HIPCHK(hipMemcpy(h->d_track.p, scores, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
  h->stages_done |= C3_STAGE_CONK;
  const int md0 = h->cfg.mdistcutoff;
  h->cfg.mdistcutoff = min_dist;
  rc = c3_batch_run(h, C3_STAGE_PEAKS);
  h->cfg.mdistcutoff = md0;
  if (rc) return rc;
  int np = c3_fetch_raw_peaks(h, 0, peaks, cap);
  if (np < 0) return np;
  if (smoothed) { int r2 = c3_fetch_smoothed(h, 0, smoothed, n); if (r2 < 0) return r2; }
  return np;
}

extern "C" int c3_poa_msa(c3_handle* h, int n, const char* const* seqs, const int* lens,
                          char* cons, int cons_cap, int* cons_len, char* msa, int64_t msa_cap, int* msa_len) {
  if (!h) return C3_E_ARG;
  if (cons_len) *cons_len = 0;
  if (msa_len) *msa_len = 0;
  if (n == 0) return C3_E_OK;                       // msa([]) -> empty result (determine_consensus.py:43-47 with repeats==0)
  if (!seqs || !lens) return C3_E_ARG;
  int rc = inject(h, n, seqs, nullptr, lens, nullptr, nullptr, 0, nullptr, nullptr, 0);
  if (rc) return rc;
  const bool dbg0 = h->debug_msa; h->debug_msa = (msa != nullptr);
  rc = c3_batch_run(h, C3_STAGE_POA);
  if (rc == 0 && cons) {
    // pyabpoa semantics: the consensus is the heaviest bundle also for n == 2; the batch path gives the
    // pairwise-merged draft there, so only n != 2 is served from the draft
    if (n == 2) { h->debug_msa = dbg0; return c3_fail(h, C3_E_ARG, "out_cons with exactly 2 sequences is not a reference call shape"); }
    int C = c3_fetch_draft(h, 0, cons, cons_cap);
    if (C < 0) rc = C; else if (cons_len) *cons_len = C;
  }
  if (rc == 0 && msa) { int ml = 0; rc = fetch_msa_rows(h, 0, n, msa, msa_cap, &ml); if (rc == 0 && msa_len) *msa_len = ml; }
  h->debug_msa = dbg0;
  return rc;
}

extern "C" int c3_zero_repeats(c3_handle* h, const char* d0, const char* q0, int n0, const char* d1, const char* q1, int n1,
                               int min_len, char* out, int cap, int* out_len) {
  if (!h || !d0 || !d1 || n0 <= 0 || n1 <= 0 || !out || !out_len) return C3_E_ARG;
  *out_len = 0;
  if (h->n_spl <= 0) { const char sp[] = "ACGT"; int64_t o[2] = {0, 4}; int rc = c3_set_splints(h, 1, sp, o); if (rc) return rc; }
  std::string seq(d0, n0), ql;
  seq.append(d1, n1);
  if (q0) ql.append(q0, n0); else ql.append(n0, 'I');
  if (q1) ql.append(q1, n1); else ql.append(n1, 'I');
  int64_t off[2] = {0, (int64_t)seq.size()};
  int16_t sid = 0; char st = '+';
  int rc = c3_batch_upload(h, 1, seq.data(), ql.data(), off, &sid, &st);
  if (rc) return rc;
  c3_read_result r; memset(&r, 0, sizeof(r));
  r.status = C3_ST_NO_CONSENSUS; r.n_peaks = 1; r.has_front = 1; r.has_tail = 1; r.front_end = n0; r.tail_beg = n0;
  HIPCHK(hipMemcpy(h->d_info.p, &r, sizeof(r), hipMemcpyHostToDevice));
  const int md0 = h->cfg.mdistcutoff, z0 = h->cfg.zero;
  h->cfg.mdistcutoff = min_len; h->cfg.zero = 1;
  rc = c3_batch_run(h, C3_STAGE_POA | C3_STAGE_POLISH);
  h->cfg.mdistcutoff = md0; h->cfg.zero = z0;
  if (rc) return rc;
  std::vector<c3_read_result> res(1);
  int64_t co[2];
  rc = c3_batch_results(h, res.data(), out, cap, co);
  if (rc) return rc;
  *out_len = (res[0].status == C3_ST_OK) ? res[0].cons_len : 0;
  return C3_E_OK;
}

extern "C" int c3_determine_consensus(c3_handle* h, int n, const char* const* subs, const char* const* quals,
                                      const int* lens, const char* front, const char* front_q, int front_len,
                                      const char* tail, const char* tail_q, int tail_len,
                                      char* out, int cap, int* out_len, char* draft, int draft_cap, int* draft_len) {
  if (!h || !subs || !lens || !out || !out_len) return C3_E_ARG;
  *out_len = 0; if (draft_len) *draft_len = 0;
  int rc = inject(h, n, subs, quals, lens, front, front_q, front_len, tail, tail_q, tail_len);
  if (rc) return rc;
  rc = c3_batch_run(h, C3_STAGE_POA | C3_STAGE_POLISH);
  if (rc) return rc;
  if (draft) { int C = c3_fetch_draft(h, 0, draft, draft_cap); if (C < 0) return C; if (draft_len) *draft_len = C; }
  std::vector<c3_read_result> res(1);
  int64_t co[2];
  rc = c3_batch_results(h, res.data(), out, cap, co);
  if (rc) return rc;
  *out_len = (res[0].status == C3_ST_OK) ? res[0].cons_len : 0;
  return C3_E_OK;
}
