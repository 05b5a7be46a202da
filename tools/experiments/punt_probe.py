"""which capacity ends reads in k_poa (needs the -DC3_DEBUG_PUNT build): C3POA_LIB=.../libc3poa_hip_punt.so python tools/experiments/punt_probe.py SEED"""
import ctypes as C, importlib.util, os, sys
import numpy as np
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fz3", "tools/fuzz_parity3.py"); fz3 = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz3)
from c3poa_amd import _lib
seed = int(sys.argv[1])
cfg = fz3.random_config(np.random.default_rng(10_000 + seed))
splint, _md, reads, strands = fz3.fz.generate(100, 50_000 + seed)
keep = [i for i, r in enumerate(reads) if len(r[0]) < 40_000]
reads = [reads[i] for i in keep]; strands = [strands[i] for i in keep]
h = _lib.Handle(**cfg); h.set_splints([splint]); h.upload([r[0] for r in reads], [r[1] for r in reads], strands); h.run()
res, cons = h.results()
out = (C.c_uint64 * 16)(); h.lib.c3_debug_phases.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; h.lib.c3_debug_phases(h.h, 0, out)
print("statuses", np.bincount(res["status"], minlength=6).tolist(), "cells overflow", out[12] & 0xffffffff, "far overflow", out[13], "punt", out[14], "guard", out[15])
