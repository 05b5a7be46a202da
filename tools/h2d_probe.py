"""H2D bandwidth probe: pageable vs hipHostMalloc'd memory through plain hipMemcpy (diagnostic)"""
import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so")
n = 1 << 30
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), C.c_size_t(n)) == 0
a = np.ones(n, dtype=np.uint8)
p = C.c_void_p(); assert hip.hipHostMalloc(C.byref(p), C.c_size_t(n), 0) == 0
C.memset(p, 1, n)
for name, src in (("pageable", a.ctypes.data), ("pinned", p.value)):
    for it in range(3):
        t = time.perf_counter(); rc = hip.hipMemcpy(d, C.c_void_p(src), C.c_size_t(n), 1); hip.hipDeviceSynchronize(); dt = time.perf_counter() - t
        print(name, rc, "%.1f GB/s" % (n / dt / 1e9))
t = time.perf_counter(); q = C.c_void_p(); hip.hipHostMalloc(C.byref(q), C.c_size_t(n), 0); print("hipHostMalloc 1 GiB: %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); C.memset(q, 1, n); print("first touch: %.3f s" % (time.perf_counter() - t))
