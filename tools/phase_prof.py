#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares inside k_poa / k_window (needs the -DC3_PHASE_PROF build:
C3POA_LIB=c3poa_amd/lib/libc3poa_hip_prof.so python tools/phase_prof.py [n_reads] [cfg])."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
recs = list(synth.generate(cfg, n_reads=min(n, 2048)))
recs = (recs * (n // len(recs) + 1))[:n]
h = _lib.Handle(mdistcutoff=synth.CONFIGS[cfg]["mdist"])
h.set_splints([synth.SPLINT1])
h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
best = None
for _rep in range(int(os.environ.get("C3_REPS", "4"))):                 # min over repetitions: clocks / power state add +-5 %
    h.run(); h.results(with_consensus=False)
    t = h.timing()
    best = t if best is None else {k: (min(v, best[k]) if isinstance(v, float) else v) for k, v in t.items()}
print({k: round(v, 2) if isinstance(v, float) else v for k, v in best.items()})
names = [["remain/init", "DP rows", "traceback", "fuse", "reorder", "columns", "consensus/pairwise", "tpos", "-", "between"],
         ["backbone", "sort+mask", "compaction", "DP rows", "end select", "traceback", "fuse", "reorder", "consensus", "queue/other"]]
h.lib.c3_debug_phases.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for which, kn in ((0, "k_poa"), (1, "k_window")):
    out = (C.c_uint64 * 16)()
    h.lib.c3_debug_phases(h.h, which, out)
    tot = float(sum(out[:8]) + out[9] + (out[8] if which else 0)) or 1.0
    print(kn, " ".join("%s=%.1f%%" % (names[which][i], 100 * out[i] / tot) for i in range(10) if out[i]), "| raw[8..11] =", out[8], out[9], out[10], out[11])
    if which == 1 and out[11]:
        rows = out[11] & 0xffffffff; gen = out[10] >> 32; kept = out[10] & 0xffffffff; multi = out[11] >> 32
        print("   rows %d: fast %.1f%%, general %.1f%% = predecessor r-1 but kept for later %.1f%% + several predecessors %.1f%% + one far predecessor %.1f%%" % (
            rows, 100.0 * (rows - gen) / rows, 100.0 * gen / rows, 100.0 * kept / rows, 100.0 * multi / rows, 100.0 * (gen - kept - multi) / rows))
        far = gen - kept - multi
        print("   cycles per general row: several predecessors %.0f, one far predecessor %.0f, r-1 kept %.0f; all DP-row cycles / all rows = %.0f" % (
            out[12] / max(multi, 1), out[13] / max(far, 1), out[14] / max(kept, 1), out[3] / max(rows, 1)))
    if which == 1:
        if out[12]:
            print("   band traceback: %d blocks, load+sync %.0f cycles per block (%.1f%% of the wave time), %.1f steps per block, %.0f cycles per step (%.1f%%)" % (
                out[12], out[13] / out[12], 100.0 * out[13] / tot, out[14] / out[12], out[15] / max(out[14], 1), 100.0 * out[15] / tot))
    if which == 0 and out[1]:
        rc = max(out[8] + out[10] + out[11] + out[12], 1)
        print("   DP rows by kind (cycles): steady %.1f%%, fast %.1f%%, near %.1f%%, general %.1f%% of the row loop" % tuple(100.0 * out[i] / rc for i in (12, 8, 10, 11)))
        if out[13] or out[14]:
            print("   steady rows %d (%.0f cycles each), fast rows %d (%.0f cycles each), ring flushes after a steady run %d" % (
                out[13], out[12] / max(out[13], 1), out[14], out[8] / max(out[14], 1), out[15]))
