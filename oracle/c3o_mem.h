/*
 * c3o_mem.h -- ORACLE (test infrastructure): per-thread bump arena for all scratch of one API call.
 * Many-core runs of the oracle otherwise serialise on the kernel's address-space lock (every multi-MB
 * DP matrix was an mmap/munmap); with the arena the pages stay mapped and the CPU baseline timed by
 * bench.py reflects the arithmetic, not the allocator.  Everything allocated during a public call is
 * released when the outermost call returns.
 */
#ifndef C3O_MEM_H
#define C3O_MEM_H
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
typedef struct { char* base; size_t cap, off; int depth; } c3o_arena;
extern __thread c3o_arena c3o_tls_arena;
#ifdef C3O_NO_ARENA
/* sanitizer builds (`make asan`): every allocation is the heap's own, so AddressSanitizer sees each buffer's real bounds */
static inline void c3o_enter(void) {}
static inline void c3o_leave(void) {}
static inline void* c3o_alloc(size_t n) { return malloc(n); }
static inline void* c3o_zalloc(size_t n, size_t sz) { return calloc(n, sz); }
static inline void* c3o_regrow(void* p, size_t n) { return realloc(p, n); }
static inline void c3o_release(void* p) { free(p); }
#else

static inline void c3o_enter(void) {
  c3o_arena* a = &c3o_tls_arena;
  if (a->depth++ == 0) {
    if (!a->base) { a->cap = (size_t)1 << 30; a->base = (char*)malloc(a->cap); /* virtual; touched lazily */ }
    a->off = 0;
  }
}
static inline void c3o_leave(void) { c3o_tls_arena.depth--; }

static inline void* c3o_alloc(size_t n) {
  c3o_arena* a = &c3o_tls_arena;
  if (a->depth == 0 || !a->base) return malloc(n);     /* outside a public call: plain heap */
  size_t need = (n + 16 + 63) & ~(size_t)63;
  if (a->off + need > a->cap) return malloc(n);        /* arena exhausted: plain heap, released by c3o_release */
  char* p = a->base + a->off;
  *(size_t*)p = n;
  a->off += need;
  return p + 16;
}
static inline int c3o_in_arena(const void* p) {
  const c3o_arena* a = &c3o_tls_arena;
  return a->base && (const char*)p >= a->base && (const char*)p < a->base + a->cap;
}
static inline void c3o_release(void* p) { if (p && !c3o_in_arena(p)) free(p); }
static inline void* c3o_zalloc(size_t n, size_t sz) { void* p = c3o_alloc(n * sz); if (p) memset(p, 0, n * sz); return p; }
static inline void* c3o_regrow(void* p, size_t n) {
  if (!p) return c3o_alloc(n);
  if (!c3o_in_arena(p)) return realloc(p, n);
  size_t old = *(size_t*)((char*)p - 16);
  if (n <= old) return p;
  void* q = c3o_alloc(n);
  if (q) memcpy(q, p, old);
  return q;
}
#define malloc(n) c3o_alloc(n)
#define calloc(n, s) c3o_zalloc(n, s)
#define realloc(p, n) c3o_regrow(p, n)
#define free(p) c3o_release(p)
#endif /* C3O_NO_ARENA */
#endif
