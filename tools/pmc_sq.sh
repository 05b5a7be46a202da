#!/bin/bash
# SQ counters per kernel (two passes of <= 8 SQ counters, kernel trace only): tools/pmc_sq.sh N [cfg]
# What binds the DP kernels: VALU issue, SALU issue, waiting on memory / LDS, or instruction fetch?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
N=${1:-32768}; CFG=${2:-cfg2}; R=gpurun_out/pmcsq_$CFG; rm -rf $R; mkdir -p $R
export C3_REPS=1
# (counter passes serialise kernel dispatch: k_window's consumer beside the first launch (round 6) would only wait out its bounded spin)
export C3_NO_WIN_CONSUMER=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/p$i -o b -- python3 tools/phase_prof.py $N $CFG > $R/p$i.log 2>&1
done
python3 - <<PY | tee $R/summary.txt
import sqlite3, glob, ast
from collections import defaultdict
tm = {}
try:
    tm = ast.literal_eval([l for l in open("$R/p1.log").read().splitlines() if l.startswith("{'ms_")][0])
except Exception:
    pass
print("# tools/pmc_sq.sh $N $CFG: rocprofv3 --kernel-trace --pmc (two passes of 8 SQ counters) on python3 tools/phase_prof.py $N $CFG (shipped build)")
print("# counted DP cells of the batch: poa %s, polish full matrices %s, polish computed %s" % (tm.get("cells_poa"), tm.get("cells_polish"), tm.get("cells_polish_computed")))
import hashlib, os
hsh = hashlib.sha1()
for f_ in sorted(glob.glob("c3poa_amd/csrc/*.hip") + glob.glob("c3poa_amd/csrc/*.h")):
    hsh.update(os.path.basename(f_).encode() + b"\0" + open(f_, "rb").read())
# mean issue cost of a vector wave-instruction: the two classes of tools/ubench/valu_cost.hip (2.3 cycles: add / sub / logic / 16-bit min-max;
# 4.2 cycles: DPP, 32-bit min-max, compares, selects, lane reads) weighted by EACH KERNEL'S OWN instruction mix (tools/isa_cpi.py compiles the
# sources here and classifies the listing; round 6 -- one constant for every kernel had k_conk at 106 % of its issue cycles); 2.9 where the
# tool has no entry
CPI = 2.9
CPIK = {}
try:
    import subprocess, sys
    sys.path.insert(0, "tools")
    import isa_cpi
    CPIK = {k: v["cycles_per_inst"] for k, v in isa_cpi.main()["kernels"].items()}
except Exception as e_:
    print("# tools/isa_cpi.py failed (%s): every kernel at %.1f cycles per instruction" % (e_, CPI))
js = {"note": "tools/pmc_sq.sh $N $CFG: rocprofv3 --kernel-trace --pmc, SQ counters of the shipped build on tools/phase_prof.py", "cfg": "$CFG", "reads": $N,
      "kernel_src_sha": hsh.hexdigest()[:16], "cycles_per_inst_assumed": CPI, "cycles_per_inst_by_kernel": CPIK, "kernels": {}}
val = defaultdict(dict); dur = {}
for f in glob.glob("$R/p*/*results.db"):
    db = sqlite3.connect(f)
    for kn, cn, v, st, en in db.execute("select kernel_name, counter_name, value, start, end from counters_collection"):
        n = kn.split("(")[0].replace("void ", "")
        val[n][cn] = val[n].get(cn, 0.0) + float(v); dur[n] = max(dur.get(n, 0), en - st)
for n, d in sorted(val.items(), key=lambda t: -dur.get(t[0], 0)):
    if not n.startswith("k_") or dur[n] < 1e6: continue
    wc = d.get("SQ_WAVE_CYCLES", 1.0)
    print("%-10s %.1f ms" % (n, dur[n] / 1e6), " ".join("%s=%.3g" % (k.replace("SQ_", ""), v) for k, v in sorted(d.items())))
    print("           per wave-cycle: " + " ".join("%s=%.3f" % (k.replace("SQ_", ""), d[k] / wc) for k in sorted(d) if k != "SQ_WAVE_CYCLES"))
    cells = tm.get("cells_poa") if n.startswith("k_poa") else tm.get("cells_polish_computed") if n.startswith("k_window") else None
    if cells and "SQ_INSTS_VALU" in d:
        print("           per counted cell: %.3f vector wave-instructions (= %.0f lane operations), %.3f scalar instructions" % (
            d["SQ_INSTS_VALU"] / cells, 64 * d["SQ_INSTS_VALU"] / cells, d.get("SQ_INSTS_SALU", 0) / cells))
    # vector issue: SQ_BUSY_CYCLES is summed over the 32 shader engines -> / 32 = cycles of the kernel; 1 024 SIMDs
    cyc = d.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    kn = n.split("<")[0]
    if cyc and "SQ_INSTS_VALU" in d and kn not in js["kernels"]:
        ipc = d["SQ_INSTS_VALU"] / 1024.0 / cyc
        wide = "true>" in n and kn == "k_poa"
        cpi = CPIK.get("k_poa_wide" if wide else kn, CPI)
        js["kernels"][kn] = {"instance": n, "ms": dur[n] / 1e6, "insts_valu": d["SQ_INSTS_VALU"], "insts_salu": d.get("SQ_INSTS_SALU"), "kernel_cycles": cyc,
                             "cells": cells, "insts_per_cell": (d["SQ_INSTS_VALU"] / cells) if cells else None,
                             "insts_per_simd_cycle": ipc, "cycles_per_inst": cpi, "busy_frac": ipc * cpi}
        print("           vector issue: %.3f wave-instructions per SIMD cycle (%.2f GHz); x %.2f cycles per instruction (this kernel's listing) = %.0f %% of the issue cycles" % (ipc, cyc / dur[n], cpi, 100 * ipc * cpi))
import json
json.dump(js, open("$R/summary.json", "w"), indent=1)
PY
