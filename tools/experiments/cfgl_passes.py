"""why reads of a cfgL batch leave the first k_poa pass (needs the -DC3_DEBUG_PUNT build): python tools/experiments/cfgl_passes.py N"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from c3poa_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
recs = list(synth.generate("cfgL", n_reads=n))
h = _lib.Handle(mdistcutoff=synth.CONFIGS["cfgL"]["mdist"]); h.set_splints([synth.SPLINT1])
h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs]); h.run()
t = h.timing()
out = (C.c_uint64 * 16)(); h.lib.c3_debug_phases.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; h.lib.c3_debug_phases(h.h, 0, out)
print("ms_poa %.1f redo %d redo16 %d | cells overflow %d far overflow %d (max fo/cap %d/%d) punt %d guard %d" % (
    t["ms_poa"], t["n_poa_redo"], t["n_poa_redo16"], out[12] & 0xffffffff, out[13], out[11] >> 32, out[11] & 0xffffffff, out[14], out[15]))
