#!/bin/bash
# Vector-memory-pipeline counters per kernel (TA / TCP / UTCL1 / TCC, one small group per pass, kernel trace only): tools/pmc_mem.sh N [cfg]
# Question: are the graph phases of k_poa / k_window bound by the LATENCY of dependent gathers (then more waves help) or by the
# THROUGHPUT of the CU's address / L1 pipeline (then only fewer, better coalesced vector memory instructions help)?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
N=${1:-32768}; CFG=${2:-cfg2}; R=gpurun_out/pmcmem_$CFG; rm -rf $R; mkdir -p $R
export C3_REPS=1
# (counter passes serialise kernel dispatch: k_window's consumer beside the first launch (round 6) would only wait out its bounded spin)
export C3_NO_WIN_CONSUMER=1
GROUPS_=(
 "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
 "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"
)
i=0
for P in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $R/p$i -o b -- python3 tools/phase_prof.py $N $CFG > $R/p$i.log 2>&1 || echo "pass $i failed: $P" >> $R/failed.txt
done
python3 - <<PY | tee $R/summary.txt
import sqlite3, glob
from collections import defaultdict
print("# tools/pmc_mem.sh $N $CFG: rocprofv3 --kernel-trace --pmc (one counter group per pass) on python3 tools/phase_prof.py $N $CFG (shipped build)")
val = defaultdict(dict); dur = {}
for f in glob.glob("$R/p*/*results.db"):
    db = sqlite3.connect(f)
    for kn, cn, v, st, en in db.execute("select kernel_name, counter_name, value, start, end from counters_collection"):
        n = kn.split("(")[0].replace("void ", "")
        val[n][cn] = val[n].get(cn, 0.0) + float(v); dur[n] = max(dur.get(n, 0), en - st)
for n, d in sorted(val.items(), key=lambda t: -dur.get(t[0], 0)):
    if not n.startswith("k_") or dur[n] < 1e6: continue
    print("%-28s %.1f ms" % (n, dur[n] / 1e6))
    for k in sorted(d): print("      %-44s %.4g" % (k, d[k]))
    cyc = d.get("SQ_BUSY_CYCLES", 0) / 32.0          # per shader engine -> kernel cycles
    if cyc:
        print("      kernel cycles (SQ_BUSY_CYCLES / 32 SEs): %.4g  = %.2f GHz" % (cyc, cyc / dur[n]))
        if "TA_TA_BUSY_sum" in d: print("      TA busy fraction (TA_TA_BUSY_sum / 256 CUs / cycles): %.3f" % (d["TA_TA_BUSY_sum"] / 256 / cyc))
        if "TCP_GATE_EN1_sum" in d: print("      TCP busy fraction (GATE_EN1 / 256 / cycles): %.3f   accesses per cycle per CU: %.3f" % (d["TCP_GATE_EN1_sum"] / 256 / cyc, d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 256 / cyc))
        if "TCP_PENDING_STALL_CYCLES_sum" in d: print("      TCP pending-stall fraction: %.3f   TA-data-stall: %.3f" % (d["TCP_PENDING_STALL_CYCLES_sum"] / 256 / cyc, d.get("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 0) / 256 / cyc))
    if d.get("TCP_UTCL1_REQUEST_sum"): print("      UTCL1 miss rate: %.4f" % (d.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / d["TCP_UTCL1_REQUEST_sum"]))
    if d.get("TCP_TCC_READ_REQ_sum"): print("      mean TCP->TCC read latency (cycles): %.0f" % (d.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / d["TCP_TCC_READ_REQ_sum"]))
    if d.get("TCC_REQ_sum"): print("      L2 hit rate: %.3f" % (d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1)))
    if d.get("SQ_WAVE_CYCLES"): print("      mean VMEM instructions in flight per wave: %.3f" % (d.get("SQ_INST_LEVEL_VMEM", 0) / d["SQ_WAVE_CYCLES"]))
PY
cat $R/failed.txt 2>/dev/null
