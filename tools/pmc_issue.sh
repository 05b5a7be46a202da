#!/bin/bash
# VALU / SALU instruction issue rates per kernel (known-good SQ counters, one pass each): tools/pmc_issue.sh N
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
N=${1:-32768}; R=gpurun_out/pmci; rm -rf $R; mkdir -p $R
export C3_REPS=1
# (counter passes serialise kernel dispatch: k_window's consumer beside the first launch (round 6) would only wait out its bounded spin)
export C3_NO_WIN_CONSUMER=1
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/$c -o b -- python3 tools/phase_prof.py $N > $R/$c.log 2>&1
done
python3 - <<PY
import sqlite3, glob
from collections import defaultdict
val = defaultdict(dict); dur = {}
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU"):
    for f in glob.glob("$R/%s/*results.db" % c):
        db = sqlite3.connect(f)
        for kn, v, st, en in db.execute("select kernel_name, value, start, end from counters_collection where counter_name = ?", (c,)):
            n = kn.split("(")[0].replace("void ", "")
            val[n][c] = val[n].get(c, 0.0) + float(v); dur[n] = max(dur.get(n, 0), en - st)
for n, d in sorted(val.items(), key=lambda t: -dur.get(t[0], 0)):
    if not n.startswith("k_") or dur[n] < 1e6: continue
    simd_cycles = dur[n] * 1e-9 * 2.4e9 * 1024
    print("%-18s %.1f ms  VALU/SIMD-cycle %.3f (max 0.25)  SALU/SIMD-cycle %.3f  wave-cycles/SIMD-cycle %.2f  active_valu/SIMD-cycle %.3f" % (
        n, dur[n] / 1e6, d.get("SQ_INSTS_VALU", 0) / simd_cycles, d.get("SQ_INSTS_SALU", 0) / simd_cycles,
        d.get("SQ_WAVE_CYCLES", 0) / simd_cycles, d.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles))
PY
