#!/bin/bash
# Instruction-cache counters per kernel (SQC_ICACHE_*, one pass, kernel trace only): tools/pmc_icache.sh N cfg [lib ...]   (run through gpurun)
# Question (round 6): k_poa is ~70 KB of code on a 64 KB instruction cache shared by two CUs -- does adding code to its row loop cost through misses?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
N=${1:-16384}; CFG=${2:-cfg2}; shift 2
export C3_REPS=1 C3_NO_WIN_CONSUMER=1
for LIB in "${@:-c3poa_amd/lib/libc3poa_hip.so}"; do
  R=gpurun_out/pmcic_$(basename $LIB .so)_$CFG; rm -rf $R; mkdir -p $R
  export C3POA_LIB=$LIB
  timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_BUSY_CYCLES SQ_INSTS_VALU -d $R/p -o b -- python3 tools/phase_prof.py $N $CFG > $R/p.log 2>&1
  python3 - <<PY
import sqlite3, glob
from collections import defaultdict
val = defaultdict(dict); dur = {}
for f in glob.glob("$R/p/*results.db"):
    db = sqlite3.connect(f)
    for kn, cn, v, st, en in db.execute("select kernel_name, counter_name, value, start, end from counters_collection"):
        n = kn.split("(")[0].replace("void ", ""); val[n][cn] = val[n].get(cn, 0.0) + float(v); dur[n] = max(dur.get(n, 0), en - st)
print("# $LIB $CFG $N reads")
for n, d in sorted(val.items(), key=lambda t: -dur.get(t[0], 0)):
    if not n.startswith("k_") or dur[n] < 1e6: continue
    req = d.get("SQC_ICACHE_REQ", 0.0) or 1.0
    print("%-28s %7.2f ms  icache req %.3g  hits %.3g  misses %.3g (%.2f %% of requests)  duplicate misses %.3g  | valu insts %.3g" % (
        n, dur[n] / 1e6, req, d.get("SQC_ICACHE_HITS", 0), d.get("SQC_ICACHE_MISSES", 0), 100.0 * d.get("SQC_ICACHE_MISSES", 0) / req, d.get("SQC_ICACHE_MISSES_DUPLICATE", 0), d.get("SQ_INSTS_VALU", 0)))
PY
done
