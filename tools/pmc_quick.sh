#!/bin/bash
# quick HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE, separate passes) for N cfg2 reads: tools/pmc_quick.sh N OUTDIR
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
N=${1:-32768}; R=${2:-gpurun_out/pmcq}; rm -rf $R; mkdir -p $R
export C3_REPS=1
# (counter passes serialise kernel dispatch: k_window's consumer beside the first launch (round 6) would only wait out its bounded spin)
export C3_NO_WIN_CONSUMER=1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/$c -o b -- python3 tools/phase_prof.py $N > $R/$c.log 2>&1
done
python3 - <<PY
import sqlite3, glob
from collections import defaultdict
tot = defaultdict(lambda: [0.0, 0.0, 0])
for k, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for f in glob.glob("$R/%s/*results.db" % c):
        for kn, v in sqlite3.connect(f).execute("select kernel_name, value from counters_collection where counter_name = ?", (c,)):
            n = kn.split("(")[0].replace("void ", "")
            tot[n][k] += float(v); tot[n][2] += (k == 0)
for n, (f, w, calls) in sorted(tot.items(), key=lambda t: -(2 * t[1][0] + t[1][1])):
    if n.startswith("k_") and calls:
        print("%-22s calls=%d  fetch=%.1f GB  write=%.1f GB  per launch" % (n, calls, 2 * f * 1024 / calls / 1e9, w * 1024 / calls / 1e9))
PY
