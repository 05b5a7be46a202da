"""Self-check of the eight-wave rows of the last k_poa pass against the single-wave loop, on the device: needs a library whose
k_poa_mw.hip was compiled with -DC3_MW_CHECK (every wide row is then computed twice and compared before the single-wave result
overwrites it):  C3POA_LIB=.../libc3poa_hip_mwchk.so python tools/experiments/mw_check.py [n_reads]"""
import ctypes as C, os, sys
sys.path.insert(0, ".")
os.environ["C3_DEBUG_POA32"] = "2"
from c3poa_amd import _lib, synth
recs = list(synth.generate("cfgL", n_reads=int(sys.argv[1]) if len(sys.argv) > 1 else 64))
h = _lib.Handle(mdistcutoff=synth.CONFIGS["cfgL"]["mdist"]); h.set_splints([synth.SPLINT1])
h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs]); h.run()
out = (C.c_uint64 * 16)(); h.lib.c3_debug_phases.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; h.lib.c3_debug_phases(h.h, 0, out)
names = ["D word", "D8", "P8", "H", "E1", "E2", "row maximum", "argmax span", "scan 1 aggregate", "scan 2 aggregate", "last Ht"]
print("cells / rows / chunks that differ:", {names[i]: int(out[i]) for i in range(11)})
