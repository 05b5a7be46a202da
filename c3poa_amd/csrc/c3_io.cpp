// c3_io.cpp -- native host I/O either side of the hot path: streaming FASTA/FASTQ(.gz) reader that
// fills structure-of-arrays host batches in page-locked memory (so c3_batch_upload copies them by DMA), and the writer
// of the reference's two per-group output files.  Host code only: no kernels here, nothing here touches the oracle.
//
//   c3_reader_*      replaces mm.fastx_read(args.reads, read_comment=False)   (C3POa.py:201,239; kseq.h semantics)
//   c3_write_group   replaces the file side effects of analyze_reads + determine_consensus
//                    (C3POa.py:141-173, bin/determine_consensus.py:57-77,108-114)
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <sched.h>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <memory>
#include <mutex>
#include <thread>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unordered_map>
#include <vector>

#include "../../include/c3poa.h"
#include "c3_inflate.hpp"
#include "c3_gzpar.hpp"
#include <sys/mman.h>

namespace {

// ---- growable host buffer, page-locked when a GPU runtime is present (plain malloc otherwise: pinning is a transfer
//      optimisation, not a compute path) -------------------------------------------------------------------------
// ---- process-wide cap on the native host threads that RUN at the same time.  Every GPU worker of stream.py brings two parser
// threads and a writer that fans out into formatter threads; eight workers used to put ~100 runnable threads on a 16-core cgroup
// quota, the quota was spent in the first sixth of every scheduler period and the whole process slept through the rest of it (8 workers
// ran at 0.6x the rate of one: profiles/r04_host_ceiling_*).  Parsing a group, formatting a slice and writing a file now each hold one
// of host_cores() slots while they run: the usable cores (affinity mask and cgroup quota; C3_HOST_THREADS overrides).
int host_cores() {
  static const int n = [] {
    if (const char* e = getenv("C3_HOST_THREADS")) return std::max(1, atoi(e));
    long c = sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) c = std::min<long>(c, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64]; long long per = 0;
      if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) c = std::min<long>(c, std::max<long>(1, (long)(atoll(q) / per)));
      fclose(f);
    } else {
      // cgroup v1 hosts: the CFS quota of the cpu controller (-1 = none)
      long long quota = -1, per = 0;
      for (const char* dir : {"/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"}) {
        char path[128];
        snprintf(path, sizeof(path), "%s/cpu.cfs_quota_us", dir);
        FILE* fq = fopen(path, "r");
        if (!fq) continue;
        const bool okq = fscanf(fq, "%lld", &quota) == 1; fclose(fq);
        snprintf(path, sizeof(path), "%s/cpu.cfs_period_us", dir);
        FILE* fp = fopen(path, "r");
        const bool okp = fp && fscanf(fp, "%lld", &per) == 1; if (fp) fclose(fp);
        if (okq && okp && quota > 0 && per > 0) c = std::min<long>(c, std::max<long>(1, (long)(quota / per)));
        break;
      }
    }
    return (int)std::max<long>(1, c);
  }();
  return n;
}
struct CpuSlots {
  std::mutex mu; std::condition_variable cv; int free_ = host_cores();
  void acquire() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return free_ > 0; }); --free_; }
  void release() { { std::lock_guard<std::mutex> lk(mu); ++free_; } cv.notify_one(); }
};
CpuSlots& cpu_slots() { static CpuSlots s; return s; }
struct CpuSlot {                                   // (never nested: a holder that waits for other slot holders could starve them)
  CpuSlot() { cpu_slots().acquire(); }
  ~CpuSlot() { cpu_slots().release(); }
  CpuSlot(const CpuSlot&) = delete; CpuSlot& operator=(const CpuSlot&) = delete;
};

struct HostBuf {
  char* p = nullptr; size_t cap = 0; bool pinned = false;
  ~HostBuf() { release(); }
  void release() {
    if (!p) return;
    if (pinned) (void)hipHostFree(p); else free(p);
    p = nullptr; cap = 0;
  }
  bool want_pin = true;       // page-lock new allocations (readers of small inputs switch it off: see c3_reader::pin_policy)
  bool reserve(size_t need, size_t keep) {
    if (need <= cap) return true;
    size_t ncap = cap ? cap : (size_t)1 << 20;
    while (ncap < need) ncap += ncap / 2;
    char* q = nullptr; bool pin = false;
    static int can_pin = -1;
    if (can_pin < 0) { int n = 0; can_pin = (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? 1 : 0; (void)hipGetLastError(); }
    if (can_pin && want_pin && hipHostMalloc((void**)&q, ncap, hipHostMallocDefault) == hipSuccess) pin = true;
    else { (void)hipGetLastError(); q = (char*)malloc(ncap); }
    if (getenv("C3_DEBUG")) fprintf(stderr, "[c3_io] host buffer %zu MiB %s\n", ncap >> 20, pin ? "page-locked" : "pageable");
    if (!q) return false;
    if (keep) memcpy(q, p, keep);
    release();
    p = q; cap = ncap; pinned = pin;
    return true;
  }
};

struct BatchSet {
  HostBuf names, seqs, quals;
  std::vector<int64_t> name_off, off;
  size_t n_names = 0, n_bases = 0;
};

}  // namespace

// BGZF input (bgzip / htslib): a gzip file made of independent members of <= 64 KiB whose header carries the member's size
// (extra subfield 'B','C'), so the members of a stretch of the file can be located WITHOUT inflating them and inflated by several
// threads at once -- a plain gzip stream is one zlib state and inflates at ~275 MB/s whatever the machine (27 k reads/s:
// profiles/r04_host_ceiling_gz.txt).  Plain gzip: one stream, inflated by the own decoder on a thread beside the parser (GzFast below).
struct BgzfStretch {
  std::vector<unsigned char> comp;          // compressed members of the stretch, back to back
  std::vector<size_t> coff, csz, doff;      // per member: offset / size in comp, offset of its data in dec
  std::vector<char> dec;                    // inflated stretch
  size_t dend = 0;
};
struct Bgzf {
  FILE* fp = nullptr;
  int threads = 1;
  BgzfStretch st[2];                        // the stretch being parsed and the one being read + inflated beside it
  int cur = 0;
  size_t dpos = 0;                          // unread part of st[cur].dec starts here
  std::thread pre; bool pre_on = false, pre_ok = false;
  double waited = 0;                        // seconds the parser waited for the stretch inflated beside it (C3_STREAM_STATS)
  std::atomic<bool> eof{false}, bad{false};  // (written by the prefetch thread, read by the parser)
  size_t skip = 0;                          // inflated bytes to drop before the first one handed out (a reader that starts inside a member)
  std::vector<int64_t> icoff, idoff;        // member table, built on demand (range readers): offset in the file / in the inflated stream
  int64_t itotal = -1;                      // inflated size of the whole file once the table is complete
};

// plain gzip input read with the own decoder (c3_inflate.hpp): see gzfast_chunk
struct GzFast {
  static constexpr size_t W = 32768, CHUNK = (size_t)4 << 20;      // (constexpr: std::min takes W by reference)
  int fd = -1; const uint8_t* map = nullptr; size_t size = 0, at = 0;
  c3inf::Inflater inf; bool in_member = false; size_t hist = 0; int members = 0;
  std::vector<uint8_t> win;
  std::thread th; std::mutex mu; std::condition_variable cv;
  std::vector<char> ready[2]; bool full[2] = {false, false}; int prod = 0, cons = 0; size_t cpos = 0;
  struct End { size_t off; uint32_t crc, isize; };
  std::vector<End> ends[2];                 // members that end inside the chunk: offset of the end, the trailer's CRC-32 and length
  uint32_t run_crc = 0; uint64_t run_len = 0; bool run_init = false;      // the parser's side: CRC of the member so far (checked there: the inflating thread is the slow one)
  bool done = false, bad = false, stop = false, started = false;
};

// plain gzip input inflated by several threads (c3_gzpar.hpp, round 6): a producer thread runs the rounds of the parallel decoder one ahead
// of the parser, which copies the finished chunks out in order
struct GzParReader {
  int fd = -1; const uint8_t* map = nullptr; size_t size = 0;
  c3inf::GzPar par;
  std::thread th; std::mutex mu; std::condition_variable cv;
  struct Out { std::vector<char> v; size_t len; };          // a chunk: `len` bytes behind par.head free ones
  std::vector<Out> ready[2]; bool full[2] = {false, false}; int prod = 0, cons = 0; size_t ci = 0, cpos = 0;
  bool done = false, bad = false, stop = false, started = false;
  // buffers the parser has read come back to the inflating threads (GzPar::take_buf)
  std::mutex pmu; std::vector<std::vector<char>> pool;
  double waited = 0;                                        // seconds the parser waited for a finished round (C3_STREAM_STATS)
  void give_back(std::vector<char>&& v) {
    if (v.size() < par.head + par.chunk * 4) return;         // (the reader's own first buffer, a chunk that had to grow oddly: freed)
    std::lock_guard<std::mutex> lk(pmu);
    if (pool.size() < 3 * (size_t)std::max(par.per_round, par.T)) pool.emplace_back(std::move(v));
  }
};

struct c3_reader {
  FILE* fp = nullptr; gzFile gz = nullptr; Bgzf* bz = nullptr; GzFast* gzf = nullptr; GzParReader* gzp = nullptr;
  std::vector<char> buf; size_t beg = 0, end = 0; bool eof = false; bool gz_bad = false;
  size_t pf = 0; int pf_dist = -1;             // software prefetch cursor of next_line (inflated input only: the bytes were written by other cores)
  std::vector<BatchSet> sets; int cur = -1;
  std::string err;
  bool have_line = false; const char* lp = nullptr; size_t ll = 0;   // one line of look-ahead
  int64_t n_records = 0, n_noqual = 0;
  bool names_only = false;
  bool range_lost = false;      // a byte range with bytes in it held no recognisable record start (e.g. multi-line FASTQ)
  size_t file_bytes = 0, hint_bases = 0;
  int64_t buf_off = 0;          // file offset of buf[0] (plain files)
  int64_t range_end = -1;       // byte range readers stop at the first record that starts at or after this offset
};

namespace {

// header of a BGZF member at p (n bytes available): total member size, or 0 when it is not one
size_t bgzf_member_size(const unsigned char* p, size_t n) {
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
  const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
  if (12 + xlen > n) return 0;
  for (size_t q = 12; q + 4 <= 12 + xlen;) {
    const size_t slen = (size_t)p[q + 2] | ((size_t)p[q + 3] << 8);
    if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) return ((size_t)p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
    q += 4 + slen;
  }
  return 0;
}

// inflate one member (raw deflate between the header and the 8-byte trailer), checking its CRC and size
bool bgzf_inflate(const unsigned char* m, size_t msz, char* out, size_t osz) {
  const size_t xlen = (size_t)m[10] | ((size_t)m[11] << 8);
  const size_t hdr = 12 + xlen;
  if (msz < hdr + 8) return false;
  static const bool use_zlib = getenv("C3_GZ_ZLIB") != nullptr;      // (the zlib path stays for comparison)
  if (use_zlib) {
    z_stream z; memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) return false;
    z.next_in = const_cast<unsigned char*>(m + hdr); z.avail_in = (unsigned)(msz - hdr - 8);
    z.next_out = (unsigned char*)out; z.avail_out = (unsigned)osz;
    const int rc = inflate(&z, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && z.avail_out == 0;
    inflateEnd(&z);
    if (!ok) return false;
  } else {
    // own decoder (c3_inflate.hpp): the member's size is known, so it runs to exactly osz bytes and must then find the end of the stream
    static thread_local c3inf::Inflater inf;
    inf.reset(m + hdr, m + msz - 8);
    size_t pos = 0;
    if (inf.run((uint8_t*)out, &pos, osz + 1, osz, 0) != 1 || pos != osz) return false;
  }
  const unsigned char* t = m + msz - 8;
  const unsigned long crc = (unsigned long)t[0] | ((unsigned long)t[1] << 8) | ((unsigned long)t[2] << 16) | ((unsigned long)t[3] << 24);
  return c3crc::crc32_fast(0u, (const unsigned char*)out, osz) == crc;      // (c3_crc32.hpp: zlib's table CRC was a third of a BGZF member's inflating time)
}

// ---- plain gzip input with the own decoder: the file is mapped, a thread of its own inflates it chunk by chunk (32 KiB of history in
// front of every chunk) and checks every member's CRC-32 and length; the parser copies finished chunks.  Concatenated members are one
// stream, as for gzread; bytes after the last member that are not a gzip header end the input (zlib: "trailing garbage ignored").
// header of the member at g->at: false when it is not a gzip member
bool gzfast_header(GzFast* g) {
  const uint8_t* p = g->map + g->at; const size_t n = g->size - g->at;
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return false;
  const int flg = p[3]; size_t q = 10;
  if (flg & 4) { if (q + 2 > n) return false; q += 2 + ((size_t)p[q] | ((size_t)p[q + 1] << 8)); }
  if (flg & 8) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 16) { while (q < n && p[q]) ++q; ++q; }
  if (flg & 2) q += 2;
  if (q + 8 > n) return false;
  g->inf.reset(p + q, g->map + g->size); g->in_member = true; g->hist = 0;
  return true;
}
// one chunk into out (at most CHUNK bytes): bytes produced, 0 at the end of the input, -1 on a damaged stream
long gzfast_chunk(GzFast* g, std::vector<char>& out, std::vector<GzFast::End>& ends) {
  out.clear(); ends.clear();
  while (out.size() < GzFast::CHUNK / 2) {
    if (!g->in_member) {
      if (g->at >= g->size) break;
      if (!gzfast_header(g)) { if (g->members == 0) return -1; break; }      // (trailing bytes that are not a member)
    }
    uint8_t* base = g->win.data() + GzFast::W;
    size_t pos = 0;
    const int rc = g->inf.run(base, &pos, GzFast::CHUNK - out.size(), GzFast::CHUNK + 512, g->hist);
    if (rc < 0) return -1;
    out.insert(out.end(), (const char*)base, (const char*)base + pos);
    // the last 32 KiB become the history in front of the next chunk
    if (pos >= GzFast::W) { memcpy(g->win.data(), base + pos - GzFast::W, GzFast::W); g->hist = GzFast::W; }
    else { memmove(g->win.data(), g->win.data() + pos, GzFast::W); g->hist = std::min(GzFast::W, g->hist + pos); }
    if (rc == 1) {
      const uint8_t* t = g->inf.byte_pos();
      if (t + 8 > g->map + g->size) return -1;
      const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
      const uint32_t isz = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
      ends.push_back({out.size(), crc, isz});                               // (checked by the parser's side: gzfast_verify)
      g->at = (size_t)(t + 8 - g->map); g->in_member = false; ++g->members;
    }
  }
  return (long)out.size();
}
void gzfast_thread(GzFast* g) {
  for (;;) {
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [g] { return g->stop || !g->full[g->prod]; });
    if (g->stop) return;
    lk.unlock();
    const long n = gzfast_chunk(g, g->ready[g->prod], g->ends[g->prod]);
    lk.lock();
    if (n < 0) { g->bad = true; g->done = true; g->cv.notify_all(); return; }
    if (n == 0) { g->done = true; g->cv.notify_all(); return; }
    g->full[g->prod] = true; g->prod ^= 1; g->cv.notify_all();
  }
}
// CRC-32 and length of every member that ends in this chunk, and the running CRC of the one that goes on
bool gzfast_verify(GzFast* g, const std::vector<char>& c, const std::vector<GzFast::End>& ends) {
  if (!g->run_init) { g->run_crc = (uint32_t)crc32(0L, Z_NULL, 0); g->run_len = 0; g->run_init = true; }
  size_t s0 = 0;
  auto feed = [&](size_t to) {
    g->run_crc = c3crc::crc32_fast(g->run_crc, (const unsigned char*)c.data() + s0, to - s0);
    g->run_len += to - s0; s0 = to;
  };
  for (const GzFast::End& e : ends) {
    feed(e.off);
    if (g->run_crc != e.crc || (uint32_t)g->run_len != e.isize) return false;
    g->run_crc = (uint32_t)crc32(0L, Z_NULL, 0); g->run_len = 0;
  }
  feed(c.size());
  return true;
}
long gzfast_read(GzFast* g, char* dst, size_t room) {
  if (!g->started) { g->started = true; g->win.resize(GzFast::W + GzFast::CHUNK + 1024); g->th = std::thread(gzfast_thread, g); }
  std::unique_lock<std::mutex> lk(g->mu);
  g->cv.wait(lk, [g] { return g->full[g->cons] || g->done; });
  if (!g->full[g->cons]) return g->bad ? -1 : 0;
  lk.unlock();
  std::vector<char>& c = g->ready[g->cons];
  if (g->cpos == 0 && !gzfast_verify(g, c, g->ends[g->cons])) { g->bad = true; return -1; }      // (before a byte of the chunk is handed out)
  const size_t k = std::min(room, c.size() - g->cpos);
  memcpy(dst, c.data() + g->cpos, k); g->cpos += k;
  if (g->cpos == c.size()) { lk.lock(); g->full[g->cons] = false; g->cons ^= 1; g->cpos = 0; g->cv.notify_all(); }
  return (long)k;
}
void gzfast_close(GzFast* g) {
  if (g->started) { { std::lock_guard<std::mutex> lk(g->mu); g->stop = true; } g->cv.notify_all(); g->th.join(); }
  if (g->map) munmap((void*)g->map, g->size);
  if (g->fd >= 0) close(g->fd);
  delete g;
}

void gzpar_thread(GzParReader* g) {
  for (;;) {
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [g] { return g->stop || !g->full[g->prod]; });
    if (g->stop) return;
    lk.unlock();
    const bool ok = g->par.next_round();
    std::vector<GzParReader::Out>& dst = g->ready[g->prod];
    dst.clear();
    if (ok) for (c3inf::ParChunk& c : g->par.chunks) if (c.start != (size_t)-1 && c.cb.len > 0) dst.push_back(GzParReader::Out{std::move(c.cb.out), c.cb.len});
    lk.lock();
    if (g->par.bad) { g->bad = true; g->done = true; g->cv.notify_all(); return; }
    if (!dst.empty()) { g->full[g->prod] = true; g->prod ^= 1; }
    if (!ok || g->par.done) { g->done = true; g->cv.notify_all(); return; }
    g->cv.notify_all();
  }
}
long gzpar_read(GzParReader* g, char* dst, size_t room) {
  if (!g->started) {
    g->started = true;
    g->par.with_slot = [](std::function<void()> f) { CpuSlot s_; f(); };
    g->par.take_buf = [g](std::vector<char>& v) { std::lock_guard<std::mutex> lk(g->pmu); if (!g->pool.empty()) { v = std::move(g->pool.back()); g->pool.pop_back(); } };
    if (!g->par.open()) { g->bad = true; return -1; }
    g->th = std::thread(gzpar_thread, g);
  }
  for (;;) {
    std::unique_lock<std::mutex> lk(g->mu);
    if (!(g->full[g->cons] || g->done)) { const auto w0 = std::chrono::steady_clock::now(); g->cv.wait(lk, [g] { return g->full[g->cons] || g->done; }); g->waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(); }
    if (g->bad) return -1;                                          // (a damaged stream: nothing of the failing round is handed out)
    if (!g->full[g->cons]) return 0;
    lk.unlock();
    std::vector<GzParReader::Out>& cs = g->ready[g->cons];
    if (g->ci < cs.size()) {
      GzParReader::Out& c = cs[g->ci];
      const size_t hd = g->par.head;                                 // (the chunk's bytes sit behind `head` free ones: gzpar_swap)
      const size_t k = std::min(room, c.len - g->cpos);
      memcpy(dst, c.v.data() + hd + g->cpos, k); g->cpos += k;
      if (g->cpos == c.len) { g->give_back(std::move(c.v)); std::vector<char>().swap(c.v); ++g->ci; g->cpos = 0; }
      if (k) return (long)k;
    }
    if (g->ci >= cs.size()) { lk.lock(); g->full[g->cons] = false; g->cons ^= 1; g->ci = 0; g->cpos = 0; g->cv.notify_all(); }
  }
}
// The next chunk WITHOUT a copy: the reader's buffer and the chunk's buffer change places, the unread rest of the reader's buffer (a partial
// line: `*end - *beg` bytes) is moved into the free bytes in front of the chunk first.  1 = done, 0 = end of the input, -1 = damaged stream,
// -2 = not applicable right now (a chunk partly handed out by gzpar_read, or a rest longer than the free space): the caller copies instead.
// (One pass over the inflated bytes less on the one thread every byte of the file goes through: the parser.)
int gzpar_swap(GzParReader* g, std::vector<char>& buf, size_t* beg, size_t* end) {
  if (!g->started) return -2;
  for (;;) {
    std::unique_lock<std::mutex> lk(g->mu);
    if (!(g->full[g->cons] || g->done)) { const auto w0 = std::chrono::steady_clock::now(); g->cv.wait(lk, [g] { return g->full[g->cons] || g->done; }); g->waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(); }
    if (g->bad) return -1;
    if (!g->full[g->cons]) return 0;
    lk.unlock();
    std::vector<GzParReader::Out>& cs = g->ready[g->cons];
    if (g->ci < cs.size()) {
      GzParReader::Out& c = cs[g->ci];
      const size_t hd = g->par.head, left = *end - *beg;
      if (g->cpos != 0 || left > hd) return -2;
      memcpy(c.v.data() + hd - left, buf.data() + *beg, left);
      buf.swap(c.v);
      *beg = hd - left; *end = hd + c.len;                           // (the vector is longer than the chunk: its size is its capacity for the next chunk it will hold)
      g->give_back(std::move(c.v)); std::vector<char>().swap(c.v); ++g->ci;
      if (g->ci >= cs.size()) { lk.lock(); g->full[g->cons] = false; g->cons ^= 1; g->ci = 0; g->cpos = 0; g->cv.notify_all(); }
      return 1;
    }
    lk.lock(); g->full[g->cons] = false; g->cons ^= 1; g->ci = 0; g->cpos = 0; g->cv.notify_all();
  }
}
void gzpar_close(GzParReader* g) {
  if (g->started && g->th.joinable()) { { std::lock_guard<std::mutex> lk(g->mu); g->stop = true; } g->cv.notify_all(); g->th.join(); }
  if (g->map) munmap((void*)g->map, g->size);
  if (g->fd >= 0) close(g->fd);
  delete g;
}

// next stretch of the file: up to `max_members` members read, located by their headers, inflated by b->threads threads
bool bgzf_next_stretch(Bgzf* bz, BgzfStretch* b, size_t max_members = 512) {
  b->comp.clear(); b->coff.clear(); b->csz.clear(); b->doff.clear();
  b->dend = 0;
  size_t dtot = 0;
  unsigned char hd[18];
  while (b->coff.size() < max_members && !bz->eof) {
    const size_t got = fread(hd, 1, 18, bz->fp);
    if (got == 0) { bz->eof = true; break; }
    if (got < 18) { bz->bad = true; return false; }
    // the fixed part of the header holds the BC subfield when it comes first (every writer puts it there); otherwise read on
    size_t msz = bgzf_member_size(hd, 18);
    const size_t at = b->comp.size();
    if (!msz) {
      const size_t xlen = (size_t)hd[10] | ((size_t)hd[11] << 8);
      if (hd[0] != 0x1f || hd[1] != 0x8b || !(hd[3] & 4) || xlen > 65535) { bz->bad = true; return false; }
      b->comp.resize(at + 12 + xlen);
      memcpy(b->comp.data() + at, hd, 18);
      if (12 + xlen > 18 && fread(b->comp.data() + at + 18, 1, 12 + xlen - 18, bz->fp) != 12 + xlen - 18) { bz->bad = true; return false; }
      msz = bgzf_member_size(b->comp.data() + at, 12 + xlen);
      if (!msz || msz < 12 + xlen + 8) { bz->bad = true; return false; }
      const size_t have = 12 + xlen;
      b->comp.resize(at + msz);
      if (fread(b->comp.data() + at + have, 1, msz - have, bz->fp) != msz - have) { bz->bad = true; return false; }
    } else {
      if (msz < 26) { bz->bad = true; return false; }
      b->comp.resize(at + msz);
      memcpy(b->comp.data() + at, hd, 18);
      if (fread(b->comp.data() + at + 18, 1, msz - 18, bz->fp) != msz - 18) { bz->bad = true; return false; }
    }
    const unsigned char* t = b->comp.data() + at + msz - 4;
    const size_t isz = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
    if (isz > (1u << 16)) { bz->bad = true; return false; }
    if (isz == 0) {
      // the empty end-of-file member (or an empty one in between).  It is checked like any other member -- an empty deflate stream and
      // the CRC of no bytes -- so a damaged member whose ISIZE field reads zero is an error, not a silently dropped block
      char none;
      if (!bgzf_inflate(b->comp.data() + at, msz, &none, 0)) { bz->bad = true; return false; }
      b->comp.resize(at);
      continue;
    }
    b->coff.push_back(at); b->csz.push_back(msz); b->doff.push_back(dtot);
    dtot += isz;
  }
  if (b->coff.empty()) return false;
  b->dec.resize(dtot);
  const size_t nm = b->coff.size();
  auto work = [&](size_t t0, size_t step, std::atomic<bool>* ok) {
    for (size_t i = t0; i < nm; i += step) {
      const size_t osz = (i + 1 < nm ? b->doff[i + 1] : dtot) - b->doff[i];
      if (!bgzf_inflate(b->comp.data() + b->coff[i], b->csz[i], b->dec.data() + b->doff[i], osz)) { *ok = false; return; }
    }
  };
  const size_t nt = std::min<size_t>((size_t)std::max(1, bz->threads), nm);
  std::unique_ptr<std::atomic<bool>[]> oks(new std::atomic<bool>[nt]);
  for (size_t t = 0; t < nt; ++t) oks[t] = true;
  std::vector<std::thread> th;
  for (size_t t = 1; t < nt; ++t) th.emplace_back(work, t, nt, &oks[t]);
  work(0, nt, &oks[0]);
  for (auto& x : th) x.join();
  for (size_t t = 0; t < nt; ++t) if (!oks[t]) { bz->bad = true; return false; }
  b->dend = dtot;
  return true;
}

// bytes of the inflated file, in order.  While the parser works through one stretch the next one is read and inflated on a
// thread of its own (which fans out to b->threads inflaters): the reader thread only waits when inflating is the slower side
// Member table of a BGZF file WITHOUT inflating anything: every member's header gives its size, its last four bytes the size of its data.
// Complete table -> bz->itotal; false when a member is not BGZF (or the file is cut short).
bool bgzf_index(Bgzf* bz) {
  if (bz->itotal >= 0) return true;
  const int fd = fileno(bz->fp);
  struct stat sb; if (fstat(fd, &sb) != 0) return false;
  const int64_t fsz = (int64_t)sb.st_size;
  bz->icoff.clear(); bz->idoff.clear();
  int64_t off = 0, d = 0;
  std::vector<unsigned char> hd(12 + 65536);
  while (off < fsz) {
    ssize_t got = pread(fd, hd.data(), 18, (off_t)off);
    if (got < 18) return false;
    size_t msz = bgzf_member_size(hd.data(), 18);
    if (!msz) {                                                       // (the BC subfield is not the first one)
      const size_t xlen = (size_t)hd[10] | ((size_t)hd[11] << 8);
      if (hd[0] != 0x1f || hd[1] != 0x8b || !(hd[3] & 4)) return false;
      got = pread(fd, hd.data(), 12 + xlen, (off_t)off);
      if (got < (ssize_t)(12 + xlen)) return false;
      msz = bgzf_member_size(hd.data(), 12 + xlen);
      if (!msz) return false;
    }
    if (msz < 26 || off + (int64_t)msz > fsz) return false;
    unsigned char t[4];
    if (pread(fd, t, 4, (off_t)(off + (int64_t)msz - 4)) != 4) return false;
    const int64_t isz = (int64_t)t[0] | ((int64_t)t[1] << 8) | ((int64_t)t[2] << 16) | ((int64_t)t[3] << 24);
    if (isz > 65536) return false;
    bz->icoff.push_back(off); bz->idoff.push_back(d);
    d += isz; off += (int64_t)msz;
  }
  bz->itotal = d;
  return true;
}
// continue at byte `doff` of the inflated stream: the member that holds it is inflated again from its start, the bytes before are dropped
bool bgzf_seek(Bgzf* bz, int64_t doff) {
  if (!bgzf_index(bz)) return false;
  if (bz->pre_on) { bz->pre.join(); bz->pre_on = false; }
  bz->st[0].dend = bz->st[1].dend = 0; bz->cur = 0; bz->dpos = 0; bz->eof = false; bz->bad = false;
  if (doff >= bz->itotal) { bz->eof = true; bz->skip = 0; return fseek(bz->fp, 0, SEEK_END) == 0; }
  // last member whose data starts at or before doff
  size_t lo = 0, hi = bz->idoff.size();
  while (hi - lo > 1) { const size_t m = (lo + hi) / 2; if (bz->idoff[m] <= doff) lo = m; else hi = m; }
  bz->skip = (size_t)(doff - bz->idoff[lo]);
  return fseeko(bz->fp, (off_t)bz->icoff[lo], SEEK_SET) == 0;
}

long bgzf_read(Bgzf* b, char* dst, size_t room) {
  if (b->dpos == b->st[b->cur].dend) {
    if (b->pre_on) { const auto w0 = std::chrono::steady_clock::now(); b->pre.join(); b->waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(); b->pre_on = false; if (b->pre_ok) { b->cur ^= 1; b->dpos = 0; } else b->st[b->cur].dend = b->dpos = 0; }
    else { if (b->bad || !bgzf_next_stretch(b, &b->st[b->cur])) return b->bad ? -1 : 0; b->dpos = 0; }
    if (b->bad) return -1;
    if (b->st[b->cur].dend == 0) return 0;
    if (!b->eof) { BgzfStretch* nx = &b->st[b->cur ^ 1]; b->pre_on = true; b->pre = std::thread([b, nx]() { b->pre_ok = bgzf_next_stretch(b, nx); }); }
  }
  BgzfStretch& c = b->st[b->cur];
  if (b->skip) {                                                    // (after bgzf_seek: the part of the first member before the target)
    const size_t d = std::min(b->skip, c.dend - b->dpos);
    b->dpos += d; b->skip -= d;
    if (b->dpos == c.dend) return bgzf_read(b, dst, room);
  }
  const size_t k = std::min(room, c.dend - b->dpos);
  memcpy(dst, c.dec.data() + b->dpos, k);
  b->dpos += k;
  return (long)k;
}

bool refill(c3_reader* r) {
  if (r->eof) return false;
  r->pf = 0;                                        // (positions in buf change below: the prefetch cursor starts again at beg)
  if (r->gzp && r->gzp->started) {
    // plain gzip by several threads: take the next chunk's buffer as it is (the unread rest of this one moves in front of it)
    const int rc = gzpar_swap(r->gzp, r->buf, &r->beg, &r->end);
    if (rc == 1) return true;
    if (rc == -1) { r->gz_bad = true; r->eof = true; return false; }
    if (rc == 0) { r->eof = true; return false; }
  }
  if (r->beg > 0) { memmove(r->buf.data(), r->buf.data() + r->beg, r->end - r->beg); r->end -= r->beg; r->buf_off += (int64_t)r->beg; r->beg = 0; }
  if (r->end == r->buf.size()) r->buf.resize(r->buf.size() * 2);
  size_t room = r->buf.size() - r->end;
  long got = r->bz ? bgzf_read(r->bz, r->buf.data() + r->end, room)
           : r->gzp ? gzpar_read(r->gzp, r->buf.data() + r->end, room)
           : r->gzf ? gzfast_read(r->gzf, r->buf.data() + r->end, room)
           : r->gz ? (long)gzread(r->gz, r->buf.data() + r->end, (unsigned)std::min<size_t>(room, 1u << 30))
                   : (long)fread(r->buf.data() + r->end, 1, room, r->fp);
  if (got < 0 && r->bz) r->err = "BGZF input: a member is damaged (size, inflate or CRC)";
  if (got < 0 && !r->bz && (r->gz || r->gzf || r->gzp)) r->gz_bad = true;                  // (reported by c3_reader_next: never a silently shorter file)
  if (got <= 0) { r->eof = true; return false; }
  r->end += (size_t)got;
  return true;
}

// next line without its terminator ('\n', optional '\r'); pointer valid until the next call
bool next_line(c3_reader* r, const char** p, size_t* len) {
  if (r->have_line) { r->have_line = false; *p = r->lp; *len = r->ll; return true; }
  size_t scanned = 0;                             // bytes from beg already known to hold no newline
  for (;;) {
    char* base = r->buf.data() + r->beg;
    const size_t avail = r->end - r->beg;
    // Inflated input (plain gzip by several threads, BGZF): the bytes were written by OTHER cores a moment ago, and the parser -- one thread --
    // reads them line by line at cache-to-cache latency (1.9 GB/s against 3.6 on a plain file, which fread has just copied into this core's
    // cache).  A prefetch cursor runs C3_PARSE_PREFETCH bytes (default 16 KiB) ahead of the line being returned: one prefetch per cache line.
    if (r->pf_dist != 0 && (r->gzp || r->bz)) {
      if (r->pf_dist < 0) { const char* e = getenv("C3_PARSE_PREFETCH"); r->pf_dist = e ? std::max(0, atoi(e)) : 16384; }
      if (r->pf_dist > 0) {
        size_t from = std::max(r->pf, r->beg), to = std::min(r->end, r->beg + (size_t)r->pf_dist);
        const char* b0 = r->buf.data();
        for (size_t q = from & ~(size_t)63; q < to; q += 64) __builtin_prefetch(b0 + q, 0, 3);
        if (to > from) r->pf = to;
      }
    }
    const char* nl = (const char*)memchr(base + scanned, '\n', avail - scanned);
    if (nl) {
      size_t l = (size_t)(nl - base);
      *p = base; r->beg += l + 1;
      if (l && base[l - 1] == '\r') --l;
      *len = l;
      return true;
    }
    scanned = avail;
    if (!refill(r)) {                             // refill keeps [beg,end) intact (moved to the front)
      if (r->end > r->beg) {                      // last line without terminator
        base = r->buf.data() + r->beg;
        size_t l = r->end - r->beg;
        *p = base; r->beg = r->end;
        if (l && base[l - 1] == '\r') --l;
        *len = l;
        return true;
      }
      return false;
    }
  }
}
void unget_line(c3_reader* r, const char* p, size_t len) { r->have_line = true; r->lp = p; r->ll = len; }

int fail(c3_reader* r, const char* msg) { r->err = msg; return C3_E_ARG; }

}  // namespace

// page-locked host memory for callers that build their own batches (bench.py, tests): the same kind of buffer the reader
// hands out, so c3_batch_stage / c3_batch_upload copy it by DMA.  Plain malloc when no GPU runtime is present.
extern "C" int c3_host_alloc(int64_t bytes, void** out) {
  if (!out || bytes <= 0) return C3_E_ARG;
  HostBuf b;                                      // 64-byte header in front of the user pointer: byte 0 = page-locked?
  if (!b.reserve((size_t)bytes + 64, 0)) return C3_E_NOMEM;
  b.p[0] = b.pinned ? 1 : 0;
  *out = b.p + 64;
  b.p = nullptr; b.cap = 0;                       // ownership moves to the caller (c3_host_free)
  return C3_E_OK;
}
extern "C" void c3_host_free(void* p) {
  if (!p) return;
  char* base = (char*)p - 64;
  if (base[0]) (void)hipHostFree(base); else free(base);
}

extern "C" int c3_reader_open(const char* path, int n_sets, c3_reader** out) {
  if (!path || !out) return C3_E_ARG;
  c3_reader* r = new c3_reader();
  size_t n = strlen(path);
  if (n > 3 && strcmp(path + n - 3, ".gz") == 0) {
    // BGZF (bgzip): members located by their headers and inflated in parallel; any other gzip file: one zlib stream
    unsigned char hd[18]; size_t got = 0;
    FILE* f = fopen(path, "rb");
    if (f) { got = fread(hd, 1, 18, f); }
    if (f && got == 18 && bgzf_member_size(hd, 18) && !getenv("C3_NO_BGZF")) {
      rewind(f);
      r->bz = new Bgzf(); r->bz->fp = f;
      const char* e = getenv("C3_GZ_THREADS");
      r->bz->threads = e ? std::max(1, atoi(e)) : std::min(16, host_cores());      // (eight until round 6: the parser waited 3.2 of its 5.5 s per million reads for them; command line on 2 M reads 167 -> 223-244 k reads/s with sixteen)
    } else {
      if (f) fclose(f);
      // any other gzip file: the own decoder over the mapped file (C3_GZ_ZLIB: zlib's gzread, one stream at ~275 MB/s)
      // (a file that is not gzip at all despite its name stays with gzread, which passes such files through)
      if (!getenv("C3_GZ_ZLIB") && got >= 3 && hd[0] == 0x1f && hd[1] == 0x8b && hd[2] == 8) {
        const int fd = open(path, O_RDONLY);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && sb.st_size >= 18) {
          void* mp = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
          if (mp != MAP_FAILED) {
            (void)madvise(mp, (size_t)sb.st_size, MADV_SEQUENTIAL);
            // several threads on the one stream (c3_gzpar.hpp) when the file is large enough to cut up and there are cores for it;
            // C3_GZ_SERIAL=1 / C3_GZ_THREADS=1: the single inflating thread of round 5
            const char* et = getenv("C3_GZ_THREADS");
            const int T = et ? std::max(1, atoi(et)) : std::min(16, host_cores());
            const char* ec = getenv("C3_GZ_CHUNK");
            const size_t chunk = ec ? (size_t)std::max(1024, atoi(ec)) : (size_t)2 << 20;      // (16 threads, 1 GB of FASTQ: 3.4 / 4.3 / 3.7 / 1.9 GB/s at 1 / 2 / 4 / 8 MiB, profiles/r06_gzip_parallel_decoder_throughput.txt)
            if (T > 1 && !getenv("C3_GZ_SERIAL") && (size_t)sb.st_size >= 2 * chunk) {
              r->gzp = new GzParReader(); r->gzp->fd = fd; r->gzp->map = (const uint8_t*)mp; r->gzp->size = (size_t)sb.st_size;
              r->gzp->par.map = r->gzp->map; r->gzp->par.size = r->gzp->size; r->gzp->par.T = T; r->gzp->par.chunk = chunk;
              // chunks per round: two per thread -- the threads take chunks as they get free, and a round is a barrier the whole pool waits at
              // (command line, 2 M reads, one gzip -6 member, same box alternating: 16 chunks per round 175-181 k reads/s, 32: 197-200 k)
              r->gzp->par.per_round = 2 * T;
              if (const char* er = getenv("C3_GZ_ROUND")) r->gzp->par.per_round = std::max(1, atoi(er));
              r->gzp->par.head = std::min<size_t>((size_t)512 << 10, std::max<size_t>(chunk / 2, 4096));       // (room for a partial line in front of every chunk: see gzpar_swap)
            } else {
              r->gzf = new GzFast(); r->gzf->fd = fd; r->gzf->map = (const uint8_t*)mp; r->gzf->size = (size_t)sb.st_size;
            }
          }
        }
        if (!r->gzf && !r->gzp && fd >= 0) close(fd);
      }
      if (!r->gzf && !r->gzp) { r->gz = gzopen(path, "rb"); if (r->gz) gzbuffer(r->gz, 1 << 20); }
    }
  }
  else r->fp = fopen(path, "rb");
  if (!r->gz && !r->fp && !r->bz && !r->gzf && !r->gzp) { delete r; return C3_E_ARG; }
  { FILE* f = fopen(path, "rb"); if (f) { fseek(f, 0, SEEK_END); long z = ftell(f); r->file_bytes = z > 0 ? (size_t)z : 0; fclose(f); } }
  r->buf.resize((size_t)16 << 20);
  r->sets.resize((size_t)std::max(1, n_sets));
  *out = r;
  return C3_E_OK;
}

namespace {

// Is `p` (a line start inside [buf, buf+n)) the header of a record?  FASTA: '>' is unambiguous.  FASTQ: '@' also opens
// quality lines, so the 4-line shape is checked: '@' line, sequence, '+' line, quality of the same length as the sequence.
// Returns 1 yes, 0 no, -1 cannot tell (window too short).
int record_starts_at(const char* buf, size_t n, size_t p, bool fastq, bool at_eof = false) {
  if (p >= n) return -1;
  if (!fastq) return buf[p] == '>' ? 1 : 0;            // ('>' is also a quality character: the file's first byte decides)
  if (buf[p] != '@') return 0;
  size_t ls[5]; ls[0] = p;
  for (int k = 1; k < 5; ++k) {
    const char* nl = (const char*)memchr(buf + ls[k - 1], '\n', n - ls[k - 1]);
    if (!nl) {
      if (k == 4 && at_eof && ls[3] < n) { ls[4] = n + 1; break; }        // the file's last record without a trailing newline
      return -1;
    }
    ls[k] = (size_t)(nl - buf) + 1;
    if (k < 4 && ls[k] >= n) return -1;
  }
  if (buf[ls[2]] != '+') return 0;
  auto len = [&](int k) { size_t l = ls[k + 1] - ls[k] - 1; if (l && buf[ls[k] + l - 1] == '\r') --l; return l; };
  return len(1) == len(3) && len(1) > 0 ? 1 : 0;
}

}  // namespace

// Reader over the byte range [beg, end) of a plain (not gzip) FASTA / 4-line FASTQ file: starts at the first record that
// begins at or after `beg`, stops before the first record that begins at or after `end` -- ranges that tile the file read
// every record exactly once.  Lets every GPU worker parse its own contiguous part of the input (the reference hands 1000-read
// groups to a process pool, C3POa.py:236-256).  end < 0 = end of file.
extern "C" int c3_reader_open_range(const char* path, int n_sets, int64_t beg, int64_t end, c3_reader** out) {
  if (!path || !out || beg < 0) return C3_E_ARG;
  size_t n = strlen(path);
  const bool dotgz = n > 3 && strcmp(path + n - 3, ".gz") == 0;
  int rc = c3_reader_open(path, n_sets, out);
  if (rc != C3_E_OK) return rc;
  c3_reader* r = *out;
  // a plain gzip stream cannot be entered in the middle; a BGZF file can: its members are located without inflating them, and the
  // range is given in bytes of the INFLATED file (c3_bgzf_size), so everything below works on one linear coordinate as for plain files
  if (dotgz && !r->bz) { c3_reader_close(r); *out = nullptr; return C3_E_ARG; }
  r->range_end = end;
  if (r->bz) {
    Bgzf* bz = r->bz;
    if (!bgzf_index(bz)) { c3_reader_close(r); *out = nullptr; return C3_E_ARG; }
    r->file_bytes = (size_t)bz->itotal;                             // (inflated size: the coordinate of the ranges)
    if (!getenv("C3_GZ_THREADS")) bz->threads = std::min(bz->threads, 3);      // (one of several readers: three inflating threads keep one parser busy)
    if (beg == 0) return C3_E_OK;
    auto read_at = [&](int64_t at, char* dst, size_t want) -> long {  // up to `want` inflated bytes from offset `at`
      if (!bgzf_seek(bz, at)) return -1;
      size_t have = 0;
      while (have < want) { const long g = bgzf_read(bz, dst + have, want - have); if (g < 0) return -1; if (g == 0) break; have += (size_t)g; }
      return (long)have;
    };
    char c0 = 0;
    if (read_at(0, &c0, 1) < 0) { c3_reader_close(r); *out = nullptr; return C3_E_ARG; }
    const bool fastq = c0 == '@';
    std::vector<char> win((size_t)8 << 20);
    int64_t at = beg - 1;
    for (;;) {
      const long g = read_at(at, win.data(), win.size());
      if (g < 0) { c3_reader_close(r); *out = nullptr; return C3_E_ARG; }
      const size_t got = (size_t)g;
      if (got == 0) { r->eof = true; r->buf_off = at; return C3_E_OK; }
      size_t p = 0; bool found = false, grow = false;
      while (p < got) {
        const char* nl = (const char*)memchr(win.data() + p, '\n', got - p);
        if (!nl) break;
        p = (size_t)(nl - win.data()) + 1;
        if (p >= got) break;
        const int st = record_starts_at(win.data(), got, p, fastq, got < win.size());
        if (st == 1) { found = true; break; }
        if (st < 0) { grow = got == win.size(); break; }
      }
      if (found) {
        at += (int64_t)p;
        if (!bgzf_seek(bz, at)) { c3_reader_close(r); *out = nullptr; return C3_E_ARG; }
        r->buf_off = at;
        if (end >= 0 && at >= end) r->eof = true;
        return C3_E_OK;
      }
      if (grow && win.size() < ((size_t)1 << 30)) { win.resize(win.size() * 4); continue; }
      if (got < win.size()) { r->eof = true; r->buf_off = at; r->range_lost = beg < bz->itotal - 1 && got > 1; return C3_E_OK; }
      at += (int64_t)got - 1;
    }
  }
  if (beg == 0) return C3_E_OK;
  int c0 = fgetc(r->fp);
  const bool fastq = c0 == '@';
  // resync: scan forward from beg-1 (so a record starting exactly at beg is found) for a line start that opens a record
  std::vector<char> win((size_t)8 << 20);
  int64_t at = beg - 1;
  for (;;) {
    if (fseek(r->fp, (long)at, SEEK_SET) != 0) { c3_reader_close(r); *out = nullptr; return C3_E_ARG; }
    const size_t got = fread(win.data(), 1, win.size(), r->fp);
    if (got == 0) { r->eof = true; r->buf_off = at; return C3_E_OK; }            // nothing left: an empty range
    size_t p = 0; bool found = false, grow = false;
    while (p < got) {
      const char* nl = (const char*)memchr(win.data() + p, '\n', got - p);
      if (!nl) break;
      p = (size_t)(nl - win.data()) + 1;
      if (p >= got) break;
      const int st = record_starts_at(win.data(), got, p, fastq, got < win.size());
      if (st == 1) { found = true; break; }
      if (st < 0) { grow = got == win.size(); break; }
    }
    if (found) {
      at += (int64_t)p;
      if (fseek(r->fp, (long)at, SEEK_SET) != 0) { c3_reader_close(r); *out = nullptr; return C3_E_ARG; }
      r->buf_off = at;
      if (end >= 0 && at >= end) r->eof = true;                                    // the range holds no record start
      return C3_E_OK;
    }
    if (grow && win.size() < ((size_t)1 << 30)) { win.resize(win.size() * 4); continue; }     // a record longer than the window
    if (got < win.size()) {                                                         // reached the end of the file without a record start
      // bytes but no record: not a 4-line FASTQ (multi-line records?) -- the caller must fall back to one whole-file reader
      r->eof = true; r->buf_off = at; r->range_lost = (int64_t)beg < (int64_t)r->file_bytes - 1 && got > 1; return C3_E_OK;
    }
    at += (int64_t)got - 1;                                                         // no line start decided: keep scanning
  }
}

// test hook (tests/test_inflate.py): raw DEFLATE stream `in` decoded by the own decoder into out[0..cap), through a window of `chunk`
// bytes with 32 KiB of history in front of it -- the way gzfast_chunk drives it (chunk = 0: in one piece, the way bgzf_inflate does).
// Returns the size, -1 on an error, -2 when cap is too small.
extern "C" long c3_debug_inflate(const unsigned char* in, size_t n, unsigned char* out, size_t cap, size_t chunk) {
  static thread_local c3inf::Inflater inf;
  inf.reset(in, in + n);
  if (chunk == 0) { size_t pos = 0; const int rc = inf.run(out, &pos, cap + 1, cap, 0); return rc == 1 ? (long)pos : (rc == 0 ? -2 : -1); }
  const size_t W = 32768;
  std::vector<uint8_t> win(W + chunk + 1024);
  size_t total = 0, hist = 0;
  for (;;) {
    uint8_t* base = win.data() + W; size_t pos = 0;
    const int rc = inf.run(base, &pos, chunk, chunk + 512, hist);
    if (rc < 0) return -1;
    if (total + pos > cap) return -2;
    memcpy(out + total, base, pos); total += pos;
    if (pos >= W) { memcpy(win.data(), base + pos - W, W); hist = W; } else { memmove(win.data(), win.data() + pos, W); hist = std::min(W, hist + pos); }
    if (rc == 1) return (long)total;
  }
}

// test hook (tests/test_inflate.py): the reader's CRC-32 (c3_crc32.hpp) continued from `crc`, to be held against zlib's
extern "C" unsigned c3_debug_crc32(unsigned crc, const unsigned char* p, size_t n) { return c3crc::crc32_fast(crc, p, n); }
// test hook (tests/test_inflate.py): a whole gzip FILE image `in` through the parallel decoder (threads, chunk bytes); the size, -1 on a
// damaged stream, -2 when cap is too small, -3 when `in` is no gzip member
extern "C" long c3_debug_gunzip_par(const unsigned char* in, size_t n, int threads, size_t chunk, unsigned char* out, size_t cap) {
  c3inf::GzPar par; par.map = in; par.size = n; par.T = std::max(1, threads); par.chunk = std::max<size_t>(chunk, 64);
  if (!par.open()) return -3;
  size_t total = 0;
  for (;;) {
    const bool ok = par.next_round();
    if (par.bad) return -1;
    if (ok) for (c3inf::ParChunk& c : par.chunks) if (c.start != (size_t)-1) {
      if (total + c.cb.len > cap) return -2;
      memcpy(out + total, c.cb.out.data() + par.head, c.cb.len); total += c.cb.len;
    }
    if (!ok || par.done) break;
  }
  return (long)total;
}

extern "C" void c3_reader_close(c3_reader* r) {
  if (!r) return;
  if (getenv("C3_STREAM_STATS") && (r->gzp || r->bz))
    fprintf(stderr, "reader: %lld records, waited %.3f s for inflated input (%s)\n", (long long)r->n_records, r->gzp ? r->gzp->waited : r->bz->waited, r->gzp ? "plain gzip, several threads" : "BGZF");
  if (r->gz) gzclose(r->gz);
  if (r->gzf) gzfast_close(r->gzf);
  if (r->gzp) gzpar_close(r->gzp);
  if (r->fp) fclose(r->fp);
  if (r->bz) { if (r->bz->pre_on) r->bz->pre.join(); if (r->bz->fp) fclose(r->bz->fp); delete r->bz; }
  delete r;
}

// names_only != 0: sequences and qualities are parsed (lengths, offsets and the short-read count stay exact) but not
// stored -- the first pass of C3POa.py:200-207 only needs names
extern "C" void c3_reader_names_only(c3_reader* r, int names_only) { if (r) r->names_only = names_only != 0; }

extern "C" const char* c3_reader_error(const c3_reader* r) { return r ? r->err.c_str() : "null reader"; }

// records without a quality line seen so far (FASTA): the reference cannot process them (C3POa.py:167 takes ord() of every
// quality character) and racon's -q 5 filter would drop every layer, so the CLI refuses such input
extern "C" int64_t c3_reader_noqual(const c3_reader* r) { return r ? r->n_noqual : 0; }
// bytes of host buffers this reader holds (diagnostic: tests/test_host_io.py checks that a tiny input stays tiny)
extern "C" int64_t c3_reader_reserved_bytes(const c3_reader* r) {
  if (!r) return 0;
  int64_t tot = 0;
  for (const BatchSet& b : r->sets) tot += (int64_t)(b.names.cap + b.seqs.cap + b.quals.cap);
  return tot;
}
// 1 when c3_reader_open_range found bytes but no record start in its range (multi-line FASTQ cannot be entered in the middle):
// the records of that range would be lost, so the caller has to read the file with ONE reader instead
extern "C" int c3_reader_range_lost(const c3_reader* r) { return r && r->range_lost ? 1 : 0; }
// Inflated size of a BGZF file (every member located by its header, none inflated), -1 when the file is not BGZF from end to end: the
// caller cuts [0, size) into ranges for c3_reader_open_range, as it cuts a plain file by its byte size
extern "C" int64_t c3_bgzf_size(const char* path) {
  if (!path) return -1;
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  unsigned char hd[18];
  if (fread(hd, 1, 18, f) != 18 || !bgzf_member_size(hd, 18)) { fclose(f); return -1; }
  Bgzf bz; bz.fp = f;
  const bool ok = bgzf_index(&bz);
  fclose(f);
  return ok ? bz.itotal : -1;
}

// One group of reads.  Records shorter than min_len are skipped and counted in out->n_short (C3POa.py:202-204,240-241).
// Stops after max_reads kept reads or once max_bases kept bases are exceeded (0 = no limit).  out->n == 0 at end of file.
extern "C" int c3_reader_next(c3_reader* r, int max_reads, int64_t max_bases, int min_len, c3_host_batch* out) {
  if (!r) return C3_E_ARG;
  return c3_reader_next_set(r, (r->cur + 1) % (int)r->sets.size(), max_reads, max_bases, min_len, out);
}

// The same, into buffer set `set` (0 .. n_sets-1) chosen by the caller: with several consumers finishing out of order the
// caller keeps a free list of sets and hands one back only after its group has been written.
extern "C" int c3_reader_next_set(c3_reader* r, int set, int max_reads, int64_t max_bases, int min_len, c3_host_batch* out) {
  if (!r || !out || max_reads <= 0 || set < 0 || set >= (int)r->sets.size()) return C3_E_ARG;
  CpuSlot slot_;                                   // one of the host's cores while this group is parsed
  r->cur = set;
  BatchSet& s = r->sets[(size_t)r->cur];
  s.name_off.assign(1, 0); s.off.assign(1, 0);
  if (!r->names_only && r->hint_bases) {
    // later sets are allocated once, with the size the previous groups needed (growth by copying only for the first)
    size_t want = r->hint_bases + r->hint_bases / 8 + 4096;
    if (!r->gz && !r->gzf && !r->gzp && !r->bz && r->file_bytes) {       // ... but never more than this reader can still deliver (a set taken for the tail of a range)
      const int64_t stop = r->range_end >= 0 ? std::min<int64_t>(r->range_end + 65536, (int64_t)r->file_bytes) : (int64_t)r->file_bytes;
      const int64_t here = r->buf_off + (int64_t)r->beg;
      want = std::min(want, (size_t)std::max<int64_t>(0, stop - here) / 2 + 65536);
    }
    if (!s.seqs.reserve(want, 0) || !s.quals.reserve(want, 0)) return C3_E_NOMEM;
  }
  size_t nn = 0, nb = 0; int n = 0; int64_t n_short = 0;
  const char* p; size_t l;
  while (n < max_reads && (max_bases <= 0 || (int64_t)nb < max_bases)) {
    if (!next_line(r, &p, &l)) break;
    if (l == 0) continue;
    if (p[0] != '>' && p[0] != '@') return fail(r, "not FASTA/FASTQ: record does not start with '>' or '@'");
    if (r->range_end >= 0 && r->buf_off + (int64_t)(p - r->buf.data()) >= r->range_end) { r->eof = true; r->beg = r->end; break; }   // next range's record
    // name = header up to the first blank (read_comment=False)
    size_t nl = 1; while (nl < l && p[nl] != ' ' && p[nl] != '\t') ++nl;
    if (!s.names.reserve(nn + nl, nn)) return C3_E_NOMEM;
    memcpy(s.names.p + nn, p + 1, nl - 1);
    const size_t name_len = nl - 1;
    // sequence lines until a line that starts with '>', '@' or '+' (kseq.h)
    const size_t sb = nb; size_t sl = 0; bool plus = false;
    while (next_line(r, &p, &l)) {
      if (l && (p[0] == '>' || p[0] == '@')) { unget_line(r, p, l); break; }
      if (l && p[0] == '+') { plus = true; break; }
      if (!r->names_only) {
        if (!s.seqs.reserve(sb + sl + l + 16, sb + sl)) return C3_E_NOMEM;
        memcpy(s.seqs.p + sb + sl, p, l);
      }
      sl += l;
    }
    if (!r->names_only && !s.quals.reserve(sb + sl + 16, sb)) return C3_E_NOMEM;
    if (plus) {
      size_t ql = 0;
      while (ql < sl && next_line(r, &p, &l)) {
        if (ql + l > sl) return fail(r, "quality string longer than the sequence");
        if (!r->names_only) memcpy(s.quals.p + sb + ql, p, l);
        ql += l;
      }
      if (ql != sl) return fail(r, "truncated quality string");
    } else {
      ++r->n_noqual;
      if (!r->names_only) memset(s.quals.p + sb, '!', sl);       // FASTA record: no qualities -> Phred 0
    }
    ++r->n_records;
    if ((int64_t)sl < (int64_t)min_len) { ++n_short; continue; }     // dropped: buffers are simply overwritten
    if (n == 0 && !r->names_only && !r->hint_bases && sl > 0) {
      // first kept record of the very first group: size the buffers for max_reads records of this length in ONE page-locked
      // allocation instead of a dozen grow-and-copy steps (page-locking is slow and serialised by the driver)
      size_t want = (size_t)max_reads * (sl + sl / 4) + 4096;
      if (max_bases > 0) want = std::min(want, (size_t)max_bases + sl + 4096);
      want = std::min(want, (size_t)2 << 30);
      // ... but never more than this reader can still deliver: sequence bytes are at most half of the bytes left in its
      // file / byte range (the other half are qualities), and only as many buffer sets as those bytes can fill are page-locked
      // now (a 100-read input used to pin ten buffers of max_reads reads each)
      size_t deliver = (size_t)-1;
      if (!r->gz && !r->gzf && !r->gzp && !r->bz && r->file_bytes) {
        const int64_t stop = r->range_end >= 0 ? std::min<int64_t>(r->range_end + (int64_t)(4 * sl + 4096), (int64_t)r->file_bytes) : (int64_t)r->file_bytes;
        const int64_t here = r->buf_off + (int64_t)r->beg;
        deliver = (size_t)std::max<int64_t>(0, stop - here) / 2 + sb + sl + 4096;
      } else if ((r->gz || r->gzf || r->gzp || r->bz) && r->file_bytes) deliver = r->file_bytes * 16 + sb + sl + 4096;        // (compressed size: a generous bound)
      want = std::min(want, std::max(deliver, sb + sl + 4096));
      // Page-locking costs ~0.15 s per GB and as much again to undo, i.e. about what THREE copies of the buffer from pageable memory lose
      // against DMA: it pays when a buffer set is refilled several times, not when the whole input of this reader passes through its sets
      // once or twice (eight workers over 2 M reads used to pin 50 GB for 20 GB of input: 7.8 s before the first batch, 5 s to let go --
      // profiles/r05_host_ceiling_*).  Readers whose file / byte range holds less than C3_PIN_MIN_REFILLS (default 3) fillings of their
      // sets use plain memory; C3_PIN_MIN_REFILLS=0 pins always.
      {
        const char* e = getenv("C3_PIN_MIN_REFILLS");
        const double refills = e ? atof(e) : 3.0;
        const double fill_all = (double)want * (double)r->sets.size();
        const bool pin = refills <= 0 || deliver == (size_t)-1 || (double)deliver >= refills * fill_all;
        for (BatchSet& o : r->sets) { o.seqs.want_pin = pin; o.quals.want_pin = pin; o.names.want_pin = pin; }
      }
      if (!s.seqs.reserve(want, sb + sl) || !s.quals.reserve(want, sb + sl)) return C3_E_NOMEM;
      // ... and the other buffer sets of this reader right away: a page-lock issued later, while the GPU is busy, stalls the
      // running kernels for its whole duration (some 50 ms per 400 MiB)
      size_t left = deliver > want ? deliver - want : 0;
      for (BatchSet& o : r->sets) if (&o != &s) {
        if (left == 0) break;                                              // nothing left to fill it with: grows lazily if ever needed
        const size_t w2 = std::min(want, left + 4096);
        if (!o.seqs.reserve(w2, 0) || !o.quals.reserve(w2, 0)) return C3_E_NOMEM;
        left -= std::min(left, w2);
      }
    }
    nn += name_len; nb += sl; ++n;
    s.name_off.push_back((int64_t)nn); s.off.push_back((int64_t)nb);
  }
  if (r->bz && r->bz->bad) return fail(r, "BGZF input: a member is damaged (header, size, inflate or CRC)");      // never a silently short file
  if (r->gz_bad) return fail(r, "gzip input: the stream is damaged (inflate, CRC or length check; with several inflating threads also: 2 MiB of input that inflate to more than 1 GiB -- C3_GZ_SERIAL=1 reads such a file)");
  s.n_names = nn; s.n_bases = nb;
  r->hint_bases = std::max(r->hint_bases, nb);
  out->n = n; out->n_short = n_short;
  out->names = s.names.p; out->name_off = s.name_off.data();
  out->seqs = s.seqs.p; out->quals = s.quals.p; out->off = s.off.data();
  return C3_E_OK;
}

// ---- writer ------------------------------------------------------------------------------------------------------
namespace {

// Output bytes are produced straight into slices of one pooled arena per call (kept across calls: after the first group no
// page of it is faulted in again, nothing grows by copying).  A slice is sized from an upper bound of its records, so the
// appender needs no capacity checks.
struct Out {
  char* p;
  inline void put(char c) { *p++ = c; }
  inline void app(const char* s, size_t n) { memcpy(p, s, n); p += n; }
  inline void num(long long v) {                 // decimal, no allocation
    char t[24]; int k = 0;
    unsigned long long u = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
    do { t[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (k) *p++ = t[--k];
  }
};

// str(round(sum / n, 2)) of Python: correctly rounded to 2 decimals, then the shortest repr (trailing zeros dropped,
// one decimal kept) -- C3POa.py:168
void avg_qual_text(const char* q, int64_t n, Out& out) {
  unsigned long long sum = 0;
  const unsigned char* u = (const unsigned char*)q;
  for (int64_t i = 0; i < n; ++i) sum += u[i];                 // (vectorises; the -33 per base comes off afterwards)
  const long long tot = (long long)sum - 33LL * n;
  char tmp[64];
  int k = snprintf(tmp, sizeof(tmp), "%.2f", (double)tot / (double)n);
  while (k > 0 && tmp[k - 1] == '0' && k >= 2 && tmp[k - 2] != '.') --k;
  out.app(tmp, (size_t)k);
}

inline void fastq(Out& o, const char* name, size_t nl, long idx, const char* s, const char* q, int64_t b, int64_t e) {
  o.put('@'); o.app(name, nl); o.put('_'); o.num(idx); o.put('\n');
  o.app(s + b, (size_t)(e - b)); o.app("\n+\n", 3); o.app(q + b, (size_t)(e - b)); o.put('\n');
}
inline size_t fastq_bound(size_t nl, int64_t len) { return nl + 2 * (size_t)std::max<int64_t>(len, 0) + 32; }

// which records does read i produce?  (one place for the rules of C3POa.py:115-132 / determine_consensus.py:57-77,108-114)
struct Emit { bool any, cons; int ns; };
inline Emit emit_of(const c3_read_result& r, int s, int n_splints, int zero, int64_t clen) {
  Emit e = {false, false, r.n_sub};
  if (s < 0 || s >= n_splints) return e;
  if (r.status == C3_ST_NOT_ASSIGNED || r.status == C3_ST_NO_PEAKS || r.status == C3_ST_TOO_SHORT) return e;   // C3POa.py:115,125,131
  const int nd = (r.has_front ? 1 : 0) + (r.has_tail ? 1 : 0);
  if (r.n_sub == 0) { if (!(zero && nd == 2)) return e; }
  else if (r.status == C3_ST_LIMIT) return e;
  e.any = true; e.cons = r.status == C3_ST_OK && clen > 0;
  return e;
}

}  // namespace

namespace {

// upper bound of the bytes reads [i0, i1) append to the consensus / subread file of every splint
void bound_range(const c3_host_batch* b, const c3_read_result* res, const char* cons, const int64_t* cons_off,
                 const int16_t* splint_id, int n_splints, int zero, int i0, int i1, size_t* bc, size_t* bs) {
  for (int i = i0; i < i1; ++i) {
    const c3_read_result& r = res[i];
    const int s = splint_id[i];
    const int64_t clen = cons ? cons_off[i + 1] - cons_off[i] : 0;
    const Emit e = emit_of(r, s, n_splints, zero, clen);
    if (!e.any) continue;
    const size_t nl = (size_t)(b->name_off[i + 1] - b->name_off[i]);
    const int64_t L = b->off[i + 1] - b->off[i];
    size_t q = 0;
    if (e.ns == 0) q = fastq_bound(nl, r.front_end) + fastq_bound(nl, L - r.tail_beg);
    else {
      for (int k = 0; k < e.ns; ++k) q += fastq_bound(nl, (int64_t)r.sub_end[k] - r.sub_beg[k]);
      if (r.has_front) q += fastq_bound(nl, r.front_end);
      if (r.has_tail) q += fastq_bound(nl, L - r.tail_beg);
    }
    bs[s] += q;
    if (e.cons) bc[s] += nl + (size_t)clen + 96;
  }
}

// records of reads [i0, i1) appended to oc[s] / os[s] (one slice pair per splint)
void format_range(const c3_host_batch* b, const c3_read_result* res, const char* cons, const int64_t* cons_off,
                  const int16_t* splint_id, int n_splints, int zero, int i0, int i1, Out* oc, Out* os) {
  for (int i = i0; i < i1; ++i) {
    const c3_read_result& r = res[i];
    const int s = splint_id[i];
    const int64_t clen = cons ? cons_off[i + 1] - cons_off[i] : 0;
    const Emit e = emit_of(r, s, n_splints, zero, clen);
    if (!e.any) continue;
    const char* name = b->names + b->name_off[i]; const size_t nl = (size_t)(b->name_off[i + 1] - b->name_off[i]);
    const char* seq = b->seqs + b->off[i]; const char* qual = b->quals + b->off[i];
    const int64_t L = b->off[i + 1] - b->off[i];
    const int ns = e.ns;
    Out& fq = os[s];
    if (ns == 0) {
      fastq(fq, name, nl, 0, seq, qual, 0, r.front_end);
      fastq(fq, name, nl, 1, seq, qual, r.tail_beg, L);
    } else {
      for (int k = 0; k < ns; ++k) fastq(fq, name, nl, k + 1, seq, qual, r.sub_beg[k], r.sub_end[k]);
      int j = 0;
      if (r.has_front) { fastq(fq, name, nl, 0, seq, qual, 0, r.front_end); ++j; }
      if (r.has_tail) fastq(fq, name, nl, j == 0 ? 0 : ns + 1, seq, qual, r.tail_beg, L);
    }
    if (e.cons) {
      Out& fa = oc[s];
      fa.put('>'); fa.app(name, nl); fa.put('_');
      avg_qual_text(qual, L, fa);
      fa.put('_'); fa.num((long long)L); fa.put('_'); fa.num(ns);
      fa.put('_'); fa.num((long long)clen); fa.put('\n');
      fa.app(cons + cons_off[i], (size_t)clen); fa.put('\n');
    }
  }
}

// pooled arenas: one per c3_write_group call in flight, grow-only, reused by later calls
struct Arena { char* p = nullptr; size_t cap = 0; };
std::mutex g_arena_mu;
std::vector<Arena> g_arena_free;
Arena arena_get(size_t need) {
  Arena a;
  {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    size_t best = (size_t)-1;
    for (size_t k = 0; k < g_arena_free.size(); ++k)
      if (g_arena_free[k].cap >= need && (best == (size_t)-1 || g_arena_free[k].cap < g_arena_free[best].cap)) best = k;
    if (best == (size_t)-1 && !g_arena_free.empty()) best = 0;                 // too small: take one and grow it
    if (best != (size_t)-1) { a = g_arena_free[best]; g_arena_free.erase(g_arena_free.begin() + (long)best); }
  }
  if (a.cap < need) { free(a.p); a.cap = need + need / 8 + 4096; a.p = (char*)malloc(a.cap); if (!a.p) a.cap = 0; }
  return a;
}
void arena_put(Arena a) { if (a.p) { std::lock_guard<std::mutex> lk(g_arena_mu); g_arena_free.push_back(a); } }

}  // namespace

// Appends the records of one group to <cons_paths[s]> / <sub_paths[s]> (s = splint_id[i]; reads with splint_id < 0 or a
// status that produces no output are skipped).  Record formats and the subread naming asymmetry follow the reference:
// kept subreads _1.._n, first dangling piece _0, second _<n+1> (determine_consensus.py:57-62,69-77); zero-repeat pieces
// _0,_1 are written before the rescue is tried (:108-114); consensus header name_avgQ_rawLen_repeats_consLen (C3POa.py:168-171).
// The group is cut into contiguous read ranges that are formatted and written (pwrite at precomputed offsets) by a few
// threads: record order in the files is read order, exactly as with one thread.
namespace {
// Append reservations per output file: a group's bytes go to [at, at + total) of the file, where `at` is handed out under a
// lock, so the writer threads of several GPU workers can format and pwrite into ONE file concurrently (no part files, no
// merge pass).  c3_writer_reset forgets the table (the caller truncates its files at the start of a run).
std::mutex g_res_mu;
struct ResEntry { off_t next = 0; int inflight = 0; };
struct ResKey { dev_t dev; ino_t ino; bool operator==(const ResKey& o) const { return dev == o.dev && ino == o.ino; } };
struct ResKeyHash { size_t operator()(const ResKey& k) const { return std::hash<unsigned long long>()((unsigned long long)k.dev * 1000003ull ^ (unsigned long long)k.ino); } };
// keyed by the FILE (device, inode), not by the path string: two spellings of one path share one entry, a replaced file
// gets a new one.  An entry only lives while reservations are in flight: once the last writer of a file has finished, the
// next group starts again from the file's real end, so a file truncated between calls is not extended with a hole.
std::unordered_map<ResKey, ResEntry, ResKeyHash> g_res;
bool res_key(int fd, ResKey* k) { struct stat st; if (fstat(fd, &st) != 0) return false; k->dev = st.st_dev; k->ino = st.st_ino; return true; }
off_t reserve_append(int fd, size_t total) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  const off_t end = lseek(fd, 0, SEEK_END);
  ResKey k;
  if (!res_key(fd, &k)) return end;
  ResEntry& e = g_res[k];
  const off_t at = e.inflight > 0 ? std::max(end, e.next) : end;
  e.next = at + (off_t)total; e.inflight++;
  return at;
}
// writers of ONE file take turns in user space (the writer threads of several GPU workers would otherwise queue on the file's
// inode lock inside the kernel, burning their cores: two writers of a file are slower than one)
std::mutex g_file_mu[64];
std::mutex& file_mutex(int fd) { ResKey k; if (!res_key(fd, &k)) return g_file_mu[0]; return g_file_mu[ResKeyHash()(k) % 64]; }
void release_append(int fd) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  ResKey k;
  if (!res_key(fd, &k)) return;
  auto it = g_res.find(k);
  if (it != g_res.end() && --it->second.inflight <= 0) g_res.erase(it);
}
}  // namespace

extern "C" void c3_writer_reset(void) { std::lock_guard<std::mutex> lk(g_res_mu); g_res.clear(); }

extern "C" int c3_write_group(const c3_host_batch* b, const c3_read_result* res, const char* cons, const int64_t* cons_off,
                              const int16_t* splint_id, int n_splints, const char* const* cons_paths,
                              const char* const* sub_paths, int zero) {
  if (!b || !res || !cons_off || !splint_id || n_splints <= 0 || !cons_paths || !sub_paths) return C3_E_ARG;
  int T = 1;
  if (b->n >= 4096) { T = std::min(8, host_cores()); if (const char* e = getenv("C3_WRITER_THREADS")) T = std::max(1, std::min(64, atoi(e))); }
  const size_t NS = (size_t)n_splints;
  auto range = [&](int k, int* i0, int* i1) { *i0 = (int)((int64_t)b->n * k / T); *i1 = (int)((int64_t)b->n * (k + 1) / T); };
  auto run_all = [&](auto&& fn) {
    auto slotted = [&](int k) { CpuSlot slot_; fn(k); };
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k) { if (k + 1 < T) th.emplace_back(slotted, k); else slotted(k); }
    for (auto& x : th) x.join();
  };
  // phase 0: size of every (thread, kind, splint) slice from an upper bound of its records; one pooled arena holds them all
  std::vector<size_t> bc((size_t)T * NS, 0), bs((size_t)T * NS, 0);
  run_all([&](int k) { int i0, i1; range(k, &i0, &i1); bound_range(b, res, cons, cons_off, splint_id, n_splints, zero, i0, i1, &bc[(size_t)k * NS], &bs[(size_t)k * NS]); });
  size_t need = 64;
  for (size_t x : bc) need += x;
  for (size_t x : bs) need += x;
  Arena ar = arena_get(need);
  if (!ar.p) return C3_E_NOMEM;
  std::vector<char*> sc((size_t)T * NS), ss((size_t)T * NS);             // slice starts
  { char* p = ar.p; for (size_t x = 0; x < (size_t)T * NS; ++x) { sc[x] = p; p += bc[x]; ss[x] = p; p += bs[x]; } }
  std::vector<Out> oc((size_t)T * NS), os((size_t)T * NS);
  for (size_t x = 0; x < (size_t)T * NS; ++x) { oc[x].p = sc[x]; os[x].p = ss[x]; }
  // phase 1: format (parallel), straight into the slices
  run_all([&](int k) { int i0, i1; range(k, &i0, &i1); format_range(b, res, cons, cons_off, splint_id, n_splints, zero, i0, i1, &oc[(size_t)k * NS], &os[(size_t)k * NS]); });
  if (getenv("C3_WRITER_NO_IO")) { arena_put(ar); return C3_E_OK; }      // diagnostic (tools/formatter_throughput.py): the formatter alone
  // phase 2: ONE writer per file.  The slices of a file are consecutive in the file (offsets from the reservation) and are written
  // by one thread with pwritev -- several threads writing into one file only queue on its inode lock: measured on the GPU box's
  // tmpfs (tools/experiments/tmpfs_write_bench.cpp) one thread 6.3 GB/s, two 4.2, eight or sixteen 4.0 GB/s into ONE file, 27-42
  // GB/s into one file per thread.  The files of a group (consensus / subreads of every splint) are written side by side.
  struct FileJob { int fd; off_t at; std::vector<struct iovec> iov; };
  std::vector<FileJob> fjobs;
  std::vector<int> fds;
  bool ok = true;
  for (int s = 0; s < n_splints && ok; ++s) {
    for (int kind = 0; kind < 2 && ok; ++kind) {
      const char* path = kind ? sub_paths[s] : cons_paths[s];
      size_t total = 0;
      for (int k = 0; k < T; ++k) { const size_t x = (size_t)k * NS + (size_t)s; total += kind ? (size_t)(os[x].p - ss[x]) : (size_t)(oc[x].p - sc[x]); }
      if (!total || !path) continue;
      int fd = open(path, O_WRONLY | O_CREAT, 0644);
      if (fd < 0) { ok = false; break; }
      fds.push_back(fd);
      FileJob fj; fj.fd = fd;
      fj.at = reserve_append(fd, total);                    // several writer threads (one per GPU worker) append to one file
      for (int k = 0; k < T; ++k) {
        const size_t x = (size_t)k * NS + (size_t)s;
        char* t0 = kind ? ss[x] : sc[x];
        const size_t len = kind ? (size_t)(os[x].p - ss[x]) : (size_t)(oc[x].p - sc[x]);
        if (len) { struct iovec v; v.iov_base = t0; v.iov_len = len; fj.iov.push_back(v); }
      }
      fjobs.push_back(std::move(fj));
    }
  }
  if (ok) {
    std::vector<char> good(fjobs.size(), 1);
    auto write_file = [&](size_t f) {
      FileJob& j = fjobs[f];
      std::lock_guard<std::mutex> turn(file_mutex(j.fd));
      CpuSlot slot_;                               // (taken AFTER the file's turn: a writer that waits for its turn holds no core)
      off_t at = j.at; size_t i = 0;
      while (i < j.iov.size()) {
        const int cnt = (int)std::min<size_t>(j.iov.size() - i, 512);
        ssize_t w = pwritev(j.fd, &j.iov[i], cnt, at);
        if (w < 0) { if (errno == EINTR) continue; good[f] = 0; return; }
        at += w;
        size_t left = (size_t)w;                            // a short write: drop what went out, retry the rest
        while (i < j.iov.size() && left >= j.iov[i].iov_len) { left -= j.iov[i].iov_len; ++i; }
        if (left) { j.iov[i].iov_base = (char*)j.iov[i].iov_base + left; j.iov[i].iov_len -= left; }
      }
    };
    std::vector<std::thread> th;
    for (size_t f = 1; f < fjobs.size(); ++f) th.emplace_back(write_file, f);
    if (!fjobs.empty()) write_file(0);
    for (auto& x : th) x.join();
    for (char g : good) ok = ok && g;
  }
  arena_put(ar);
  for (int fd : fds) { release_append(fd); if (close(fd) != 0) ok = false; }
  return ok ? C3_E_OK : C3_E_ARG;
}

// ---- -co: the output files compressed by every core instead of one Python thread (C3POa.py:86-99 writes the final file through
// gzip.open line by line).  BGZF layout: independent members of BGZF_BLOCK input bytes, so the file is an ordinary multi-member gzip file
// for every other reader and a parallel one for c3_reader_open.
namespace {
const size_t BGZF_BLOCK = 0xff00;             // input bytes per member (bgzip's choice: the deflated member stays below 64 KiB)
// one member: header with the 'BC' subfield, raw deflate, CRC32 + ISIZE.  Returns its size (0: failed).
size_t bgzf_deflate(const unsigned char* in, size_t n, int level, unsigned char* out, size_t cap) {
  if (cap < 26 + n + n / 1000 + 64) return 0;
  z_stream z; memset(&z, 0, sizeof(z));
  if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
  z.next_in = const_cast<unsigned char*>(in); z.avail_in = (unsigned)n;
  z.next_out = out + 18; z.avail_out = (unsigned)(cap - 26);
  const int rc = deflate(&z, Z_FINISH);
  const size_t clen = z.total_out;
  deflateEnd(&z);
  if (rc != Z_STREAM_END) return 0;
  const size_t total = 18 + clen + 8;
  if (total > 65536) return 0;                 // (cannot happen at BGZF_BLOCK input bytes: deflate expands by < 0.1 % + 5 bytes per 16 KiB)
  static const unsigned char hd[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
  memcpy(out, hd, 12);
  out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
  out[16] = (unsigned char)((total - 1) & 255); out[17] = (unsigned char)((total - 1) >> 8);
  const unsigned long crc = c3crc::crc32_fast(0u, in, n);
  unsigned char* t = out + 18 + clen;
  for (int k = 0; k < 4; ++k) { t[k] = (unsigned char)(crc >> (8 * k)); t[4 + k] = (unsigned char)(n >> (8 * k)); }
  return total;
}
}  // namespace

extern "C" int c3_compress_file(const char* src, const char* dst, int level, int threads) {
  if (!src || !dst || level < 0 || level > 9) return C3_E_ARG;
  if (level == 0) level = 6;
  const int T = threads > 0 ? std::min(threads, 256) : host_cores();
  FILE* fi = fopen(src, "rb");
  if (!fi) return C3_E_ARG;
  FILE* fo = fopen(dst, "wb");
  if (!fo) { fclose(fi); return C3_E_ARG; }
  // a stretch = T x 64 members: read, deflated by T threads (members dealt round-robin), written in order; the next stretch is read while
  // this one is written
  // (capped at 32 x 16 members -- 67 MB of buffers -- whatever T: T x 64 members were 1.1 GB of transient host memory on a 128-core host,
  // beside the page-locked reader sets at the end of a run; the threads beyond 32 still share the stretch round-robin)
  const size_t per = T > 8 ? 16 : 64, nb = (size_t)std::min(T, 32) * per < (size_t)T ? (size_t)T : (size_t)std::min(T, 32) * per, in_cap = nb * BGZF_BLOCK, out_slot = 65536 + 64;
  std::vector<unsigned char> in(in_cap), out(nb * out_slot);
  std::vector<size_t> osz(nb);
  bool ok = true;
  for (;;) {
    const size_t got = fread(in.data(), 1, in_cap, fi);
    if (got == 0) { if (ferror(fi)) ok = false; break; }
    const size_t nblk = (got + BGZF_BLOCK - 1) / BGZF_BLOCK;
    std::atomic<bool> good{true};
    auto work = [&](size_t t0) {
      CpuSlot slot_;
      for (size_t b = t0; b < nblk; b += (size_t)T) {
        const size_t o = b * BGZF_BLOCK, n = std::min(BGZF_BLOCK, got - o);
        osz[b] = bgzf_deflate(in.data() + o, n, level, out.data() + b * out_slot, out_slot);
        if (!osz[b]) { good = false; return; }
      }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < (size_t)T && t < nblk; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (!good) { ok = false; break; }
    for (size_t b = 0; b < nblk && ok; ++b) ok = fwrite(out.data() + b * out_slot, 1, osz[b], fo) == osz[b];
    if (!ok || got < in_cap) break;
  }
  static const unsigned char eof_member[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (ok) ok = fwrite(eof_member, 1, 28, fo) == 28;
  fclose(fi);
  if (fclose(fo) != 0) ok = false;
  if (!ok) { unlink(dst); return C3_E_ARG; }
  return C3_E_OK;
}

// ---- oligo-dT index matcher of the post-processing step (C3POa_postprocessing.py:266-285, match_index) ----------
// seq is slid over every index (file order); the Levenshtein distance of seq[p : p+len(idx)] to idx is taken for every
// position where the slice has the full length (at a position where it is too short for index k the reference breaks
// out of the loop over indexes, so k+1.. are skipped there as well); per index the minimum counts.  Stable sort by
// distance; the best index wins when its distance is < 2 and the runner-up is more than 1 further away.  Returns the
// winning index number, -1 for '-'.  (The reference raises on an index that never fits and on fewer than two indexes;
// both cases return -1 here.)
namespace {
int edit_distance(const char* a, const char* b, int n) {
  int prev[256], cur[256];
  if (n > 255) return n;
  for (int j = 0; j <= n; ++j) prev[j] = j;
  for (int i = 1; i <= n; ++i) {
    cur[0] = i;
    for (int j = 1; j <= n; ++j) {
      int v = prev[j - 1] + (a[i - 1] != b[j - 1]);
      v = std::min(v, prev[j] + 1); v = std::min(v, cur[j - 1] + 1);
      cur[j] = v;
    }
    memcpy(prev, cur, sizeof(int) * (size_t)(n + 1));
  }
  return prev[n];
}
}  // namespace

extern "C" int c3_match_index(const char* seq, int n, int n_idx, const char* idx_cat, const int64_t* idx_off) {
  if (!seq || n_idx < 2 || !idx_cat || !idx_off) return -1;
  std::vector<int> best((size_t)n_idx, INT32_MAX);
  for (int p = 0; p < n; ++p) {
    for (int k = 0; k < n_idx; ++k) {
      const int len = (int)(idx_off[k + 1] - idx_off[k]);
      if (p + len > n) break;                                   // the reference's `break` (not `continue`)
      best[(size_t)k] = std::min(best[(size_t)k], edit_distance(seq + p, idx_cat + idx_off[k], len));
    }
  }
  int i0 = -1, i1 = -1;                                         // first and second entry of the stable sort
  for (int k = 0; k < n_idx; ++k) {
    if (best[(size_t)k] == INT32_MAX) return -1;
    if (i0 < 0 || best[(size_t)k] < best[(size_t)i0]) { i1 = i0; i0 = k; }
    else if (i1 < 0 || best[(size_t)k] < best[(size_t)i1]) i1 = k;
  }
  if (best[(size_t)i0] < 2 && best[(size_t)i1] - best[(size_t)i0] > 1) return i0;
  return -1;
}

// ---- splint assignment from the PSL (bin/preprocess.py:22-45) without per-read Python objects ---------------------
// Rows with qBaseInsert (col 5) < 50 and matches (col 0) > 50 count; per read the row with the most matches wins, the
// earliest row on ties (Python's stable sort with reverse=True keeps the first of equal keys); every splint named by a
// counted row is "seen" (adapter_set).  Rows naming a splint that is not in the splint file are ignored.
#include <unordered_map>

struct c3_assign {
  // open-addressing table keyed by the read name: names live in one arena (no per-row allocation), rows are parsed by several
  // threads and inserted in file order (3 M rows: 2.1 s through std::unordered_map<std::string, ...>, ~0.4 s like this)
  struct Slot { uint64_t hash; uint32_t name_off, name_len; float matches; int16_t splint; char strand; char used; };
  std::vector<Slot> slots; size_t mask = 0, n_used = 0;
  std::vector<char> names;
  std::unordered_map<std::string, int> splint_of;
  std::vector<uint8_t> seen;
  int64_t rows_kept = 0;
  static uint64_t hash_of(const char* p, size_t n) { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; } return h | 1; }
  void grow() {
    std::vector<Slot> old; old.swap(slots);
    slots.assign(old.empty() ? (size_t)1 << 16 : old.size() * 2, Slot{0, 0, 0, 0.f, 0, 0, 0});
    mask = slots.size() - 1;
    for (const Slot& o : old) if (o.used) { size_t i = (size_t)o.hash & mask; while (slots[i].used) i = (i + 1) & mask; slots[i] = o; }
  }
  Slot* find(const char* p, size_t n, uint64_t h) {
    if (slots.empty()) return nullptr;
    for (size_t i = (size_t)h & mask;; i = (i + 1) & mask) {
      Slot& s = slots[i];
      if (!s.used) return nullptr;
      if (s.hash == h && s.name_len == n && memcmp(names.data() + s.name_off, p, n) == 0) return &s;
    }
  }
  void upsert(const char* p, size_t n, float matches, int16_t splint, char strand) {
    const uint64_t h = hash_of(p, n);
    if (Slot* s = find(p, n, h)) { if (matches > s->matches) { s->matches = matches; s->splint = splint; s->strand = strand; } return; }   // earliest row wins ties
    if ((n_used + 1) * 10 > slots.size() * 6) grow();
    size_t i = (size_t)h & mask;
    while (slots[i].used) i = (i + 1) & mask;
    slots[i] = Slot{h, (uint32_t)names.size(), (uint32_t)n, matches, splint, strand, 1};
    names.insert(names.end(), p, p + n);
    ++n_used;
  }
};

namespace {
struct PslRow { const char* name; uint32_t name_len; float matches; int16_t splint; char strand; };
// rows of [p, e) (whole lines) that count (bin/preprocess.py:32: qBaseInsert < 50 and matches > 50) and name a known splint
void parse_psl_chunk(const char* p, const char* e, const std::unordered_map<std::string, int>& splint_of, std::vector<PslRow>& out) {
  std::string key;
  while (p < e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    const char* le = nl ? nl : e;
    const char* lend = le;
    while (lend > p && (lend[-1] == '\r' || lend[-1] == '\n')) --lend;
    if (lend > p) {
      const char* col[21]; size_t len[21]; int nc = 0;
      const char* q = p;
      while (nc < 21) {
        const char* t = (const char*)memchr(q, '\t', (size_t)(lend - q));
        col[nc] = q; len[nc] = t ? (size_t)(t - q) : (size_t)(lend - q); ++nc;
        if (!t) break;
        q = t + 1;
      }
      if (nc >= 14) {
        char tmp[64];
        auto num = [&](int c) { size_t l = std::min(len[c], sizeof(tmp) - 1); memcpy(tmp, col[c], l); tmp[l] = 0; return strtod(tmp, nullptr); };
        const double matches = num(0), gaps = num(5);
        if (gaps < 50 && matches > 50) {
          key.assign(col[13], len[13]);
          auto sp = splint_of.find(key);
          if (sp != splint_of.end()) out.push_back(PslRow{col[9], (uint32_t)len[9], (float)matches, (int16_t)sp->second, len[8] ? col[8][0] : '?'});
        }
      }
    }
    p = nl ? nl + 1 : e;
  }
}
}  // namespace

extern "C" int c3_assign_open(const char* psl_path, int n_splints, const char* const* splint_names, c3_assign** out) {
  if (!psl_path || n_splints <= 0 || !splint_names || !out) return C3_E_ARG;
  FILE* f = fopen(psl_path, "rb");
  if (!f) return C3_E_ARG;
  c3_assign* a = new c3_assign();
  for (int i = 0; i < n_splints; ++i) a->splint_of.emplace(splint_names[i], i);
  a->seen.assign((size_t)n_splints, 0);
  a->grow();
  // the file goes through in slabs of whole lines; every slab is cut at line boundaries into one chunk per thread
  const size_t SLAB = (size_t)256 << 20;
  std::vector<char> buf(SLAB + 1);
  size_t have = 0;
  int T = 8; if (const char* e = getenv("C3_PSL_THREADS")) T = std::max(1, std::min(64, atoi(e)));
  for (;;) {
    const size_t got = fread(buf.data() + have, 1, SLAB - have, f);
    const bool last = got == 0 || have + got < SLAB;
    size_t n = have + got;
    size_t use = n;
    if (!last) { while (use > 0 && buf[use - 1] != '\n') --use; if (use == 0) { buf.resize(buf.size() * 2); have = n; continue; } }   // (a line longer than the slab)
    if (use > 0) {
      std::vector<std::vector<PslRow>> rows((size_t)T);
      std::vector<size_t> cut((size_t)T + 1, use);
      cut[0] = 0;
      for (int k = 1; k < T; ++k) { size_t c = use * (size_t)k / (size_t)T; while (c < use && c > 0 && buf[c - 1] != '\n') ++c; cut[(size_t)k] = std::max(c, cut[(size_t)k - 1]); }
      std::vector<std::thread> th;
      for (int k = 0; k < T; ++k) {
        auto fn = [&, k]() { parse_psl_chunk(buf.data() + cut[(size_t)k], buf.data() + cut[(size_t)k + 1], a->splint_of, rows[(size_t)k]); };
        if (k + 1 < T) th.emplace_back(fn); else fn();
      }
      for (auto& x : th) x.join();
      for (int k = 0; k < T; ++k)                                            // file order: the earliest row keeps a tie
        for (const PslRow& r : rows[(size_t)k]) { a->seen[(size_t)r.splint] = 1; ++a->rows_kept; a->upsert(r.name, r.name_len, r.matches, r.splint, r.strand); }
    }
    if (last) break;
    memmove(buf.data(), buf.data() + use, n - use);
    have = n - use;
  }
  fclose(f);
  *out = a;
  return C3_E_OK;
}

extern "C" void c3_assign_close(c3_assign* a) { delete a; }

// splint row / strand of every read of the group: -1 / '?' when the read has no counted row.  Returns the number of
// assigned reads (>= 0) or a negative c3_err.
extern "C" int c3_assign_batch(const c3_assign* a, const c3_host_batch* b, int16_t* splint_id, char* strand) {
  if (!a || !b || !splint_id || !strand) return C3_E_ARG;
  int n_ok = 0;
  c3_assign* m = const_cast<c3_assign*>(a);                                 // (find() does not modify)
  for (int i = 0; i < b->n; ++i) {
    const char* p = b->names + b->name_off[i]; const size_t n = (size_t)(b->name_off[i + 1] - b->name_off[i]);
    const c3_assign::Slot* s = m->find(p, n, c3_assign::hash_of(p, n));
    if (!s) { splint_id[i] = -1; strand[i] = '?'; }
    else { splint_id[i] = s->splint; strand[i] = s->strand == '-' ? '-' : '+'; ++n_ok; }
  }
  return n_ok;
}

// adapter_set of bin/preprocess.py:34,43: flags[s] = 1 when a counted row names splint s
extern "C" int c3_assign_seen(const c3_assign* a, uint8_t* flags, int64_t* rows_kept) {
  if (!a || !flags) return C3_E_ARG;
  memcpy(flags, a->seen.data(), a->seen.size());
  if (rows_kept) *rows_kept = a->rows_kept;
  return C3_E_OK;
}

// ---- PSL rows of the GPU splint finder (c3_scan_splints) written natively ------------------------------------------
// One 21-column row per assigned read of the group, appended to `path`: col 0 = equivalent perfect-match length m of the
// track maximum (match*m*(m+1)/2 = max, capped at the splint length), col 5 = 0, strand, read name, read length, query
// span, splint name / length -- the row psl_row() of c3poa_amd/preprocess.py states in Python.
extern "C" int c3_write_splint_psl(const c3_host_batch* b, const int32_t* table, const int16_t* splint_id, const char* strand,
                                   int n_splints, const char* const* splint_names, const int32_t* splint_lens, int match,
                                   const char* path, int64_t* rows_written) {
  if (!b || !table || !splint_id || !strand || n_splints <= 0 || !splint_names || !splint_lens || match <= 0 || !path) return C3_E_ARG;
  std::string out;
  out.reserve((size_t)b->n * 96);
  int64_t rows = 0;
  char tmp[256];
  for (int i = 0; i < b->n; ++i) {
    const int s = splint_id[i];
    if (s < 0 || s >= n_splints) continue;
    const int rc = strand[i] == '-' ? 1 : 0;
    const int32_t* e = table + (((size_t)i * n_splints + s) * 2 + rc) * 4;
    const long long L = b->off[i + 1] - b->off[i];
    const int S = splint_lens[s];
    const long long score = e[0] > 0 ? e[0] : 0;
    long long m = (long long)((std::sqrt(1.0 + 8.0 * (double)score / (double)match) - 1.0) / 2.0);
    if (m > S) m = S;
    long long q0 = e[1]; if (q0 < 0) q0 = 0; if (q0 > L) q0 = L;
    long long q1 = q0 + S; if (q1 > L) q1 = L;
    int k = snprintf(tmp, sizeof(tmp), "%lld\t%lld\t0\t0\t0\t0\t0\t0\t%c\t", m, (long long)S - m, rc ? '-' : '+');
    out.append(tmp, (size_t)k);
    out.append(b->names + b->name_off[i], (size_t)(b->name_off[i + 1] - b->name_off[i]));
    k = snprintf(tmp, sizeof(tmp), "\t%lld\t%lld\t%lld\t", L, q0, q1);
    out.append(tmp, (size_t)k);
    out.append(splint_names[s]);
    k = snprintf(tmp, sizeof(tmp), "\t%d\t0\t%d\t1\t%d,\t%lld,\t0,\n", S, S, S, q0);
    out.append(tmp, (size_t)k);
    ++rows;
  }
  if (!out.empty()) {
    FILE* f = fopen(path, "ab");
    if (!f) return C3_E_ARG;
    size_t w = fwrite(out.data(), 1, out.size(), f);
    if (fclose(f) != 0 || w != out.size()) return C3_E_ARG;
  }
  if (rows_written) *rows_written = rows;
  return C3_E_OK;
}
