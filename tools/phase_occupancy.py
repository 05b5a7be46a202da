#!/usr/bin/env python3
"""Diagnostic: the phases of k_poa / k_window against the number of resident waves per SIMD.

The -DC3_PHASE_PROF build (libc3poa_hip_prof.so) counts wave cycles per phase; run at 3, 4, 5 and 6 resident waves per SIMD
(slots = 256 CUs x 4 SIMDs x w) the product share x kernel time gives every phase its own T(w), fitted as a + b / w:
a = the part that is instruction issue (does not shrink with more waves), b / w = the part that is waiting.  The fit
extrapolated to 8 waves per SIMD -- the most a gfx950 SIMD holds -- is what a separate high-occupancy kernel for the graph
phases could reach at best (the review's item 1b), before the cost of passing the slot state through memory.

    C3POA_LIB=c3poa_amd/lib/libc3poa_hip_prof.so python tools/phase_occupancy.py [n_reads] [cfg]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = [["init", "DP rows", "traceback", "fuse", "reorder", "columns", "consensus", "tpos"],
         ["backbone", "sort+mask", "compaction", "DP rows", "end select", "traceback", "fuse", "reorder", "consensus", "queue/other"]]
GROUPS = [{"rows": ["DP rows"], "graph": ["init", "traceback", "fuse", "reorder", "columns", "consensus", "tpos"]},
          {"rows": ["DP rows", "end select"], "graph": ["backbone", "sort+mask", "compaction", "traceback", "fuse", "reorder", "consensus", "queue/other"]}]

CHILD = r'''
import ctypes as C, json, sys
sys.path.insert(0, %r)
from c3poa_amd import _lib, synth
n = int(sys.argv[1]); cfg = sys.argv[2]; which = int(sys.argv[3]); slots = int(sys.argv[4])
recs = list(synth.generate(cfg, n_reads=min(n, 2048)))
recs = (recs * (n // len(recs) + 1))[:n]
kw = {"slots_poa": slots} if which == 0 else {"slots_win": slots}
h = _lib.Handle(mdistcutoff=synth.CONFIGS[cfg]["mdist"], **kw)
h.set_splints([synth.SPLINT1]); h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
best = None
for _ in range(3):
    h.run(); h.results(with_consensus=False); t = h.timing()
    ms = t["ms_poa"] if which == 0 else t["ms_window"]
    best = ms if best is None else min(best, ms)
h.lib.c3_debug_phases.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
out = (C.c_uint64 * 16)()
h.lib.c3_debug_phases(h.h, which, out)
print(json.dumps({"ms": best, "raw": list(out)}))
''' % ROOT


def fit(ws, ts):
    """least squares of t = a + b / w"""
    xs = [1.0 / w for w in ws]
    n = len(ws); sx = sum(xs); sy = sum(ts); sxx = sum(x * x for x in xs); sxy = sum(x * y for x, y in zip(xs, ts))
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    return (sy - b * sx) / n, b


def main():
    import json
    n = sys.argv[1] if len(sys.argv) > 1 else "16384"
    cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    waves = [3, 4, 5, 6]
    for which, kn in ((0, "k_poa"), (1, "k_window")):
        per = {}
        for w in waves:
            r = subprocess.run([sys.executable, "-c", CHILD, n, cfg, str(which), str(1024 * w)], capture_output=True, text=True)
            if r.returncode:
                print(kn, w, r.stderr[-400:]); return 1
            d = json.loads(r.stdout.strip().splitlines()[-1])
            raw = d["raw"]
            cyc = {nm: float(raw[i]) for i, nm in enumerate(NAMES[which])}
            tot = sum(cyc.values()) or 1.0
            per[w] = {nm: d["ms"] * v / tot for nm, v in cyc.items()}
            per[w]["_ms"] = d["ms"]
        print("%s (%s, %s reads, phase-profiling build): ms per phase group at w resident waves per SIMD" % (kn, cfg, n))
        proj = 0.0
        for g, members in GROUPS[which].items():
            ts = [sum(per[w][m] for m in members) for w in waves]
            a, b = fit(waves, ts)
            resid = max(abs(a + b / w - t) / t for w, t in zip(waves, ts))
            print("   %-5s  " % g + "  ".join("w=%d %.2f" % (w, t) for w, t in zip(waves, ts)) +
                  "   fit a=%.2f b=%.2f (worst residual %.1f %%)   -> w=8: %.2f" % (a, b, 100 * resid, a + b / 8))
            proj += (a + b / 6) if g == "rows" else (a + b / 8)
        whole = per[6]["_ms"]
        print("   whole kernel at w=6: %.2f ms;  rows at 6 waves + graph phases at 8 waves (a split kernel's floor, state passing free): %.2f ms = %.1f %%" % (
            whole, proj, 100.0 * (proj - whole) / whole))
    return 0


if __name__ == "__main__":
    sys.exit(main())
