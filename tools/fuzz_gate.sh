#!/bin/bash
# Gate for every kernel commit (run on the GPU box through gpurun, ~4 GPU-minutes):
#   tools/fuzz_gate.sh [SEED0] [TAG]      e.g.  gpurun -- 'bash tools/fuzz_gate.sh 2001 r05'
# 2 500 reads through each of the three parity fuzzers (fuzz_parity: random concatemers incl. inserts up to 7 kb; fuzz_parity2: random
# splints, low-complexity inserts, chimeras, 100+ subreads; fuzz_parity3: random NON-DEFAULT configurations), all on device memory
# poisoned at allocation (C3_DEBUG_POISON=1), GPU path against the oracle.  The script -- not a hand -- appends ONE line per run to
# profiles/<TAG>_fuzz_parity.txt: date, kernel-source hash (the one bench.py / tools/collect_profiles.py compute), reads, mismatches.
# Exit code 1 when any read mismatches or a fuzzer dies.
cd "$(dirname "$0")/.."
SEED0=${1:-$(( ($(date +%s) / 60) % 100000 ))}; TAG=${2:-r05}
OUT=gpurun_out/fuzz_gate_$SEED0; mkdir -p $OUT profiles
export C3_DEBUG_POISON=1
SHA=$(python3 -c "import bench; print(bench.kernel_src_sha())")
fail=0
run() { # name, command...
  local name=$1; shift
  timeout 900 "$@" > $OUT/$name.log 2>&1 || { echo "fuzz_gate: $name exited with $?" >> $OUT/died.txt; fail=1; }
}
run p1a python3 tools/fuzz_parity.py 1500 $SEED0
run p1b python3 tools/fuzz_parity.py 1000 $((SEED0 + 1)) 7000
for k in 0 1 2 3 4 5 6 7 8 9; do run p2_$k python3 tools/fuzz_parity2.py 250 $((SEED0 + 10 + k)); done
for k in $(seq 0 24); do run p3_$k python3 tools/fuzz_parity3.py 100 $((SEED0 + 100 + k)); done
python3 - "$OUT" "$SHA" "$SEED0" "$TAG" <<'PY'
import glob, os, re, sys, time
out, sha, seed0, tag = sys.argv[1:5]
tot = {"p1": [0, 0], "p2": [0, 0], "p3": [0, 0]}
refused = 0
for f in sorted(glob.glob(os.path.join(out, "p*.log"))):
    key = os.path.basename(f)[:2]
    txt = open(f).read()
    refused += len(re.findall(r"refused by c3_create", txt))
    for m in re.finditer(r"reads (\d+)\s+(?:\([^)]*\)\s+)?mismatches (\d+)", txt):
        tot[key][0] += int(m.group(1)); tot[key][1] += int(m.group(2))
    if "MISMATCH" in txt:
        print(f + ":\n" + "\n".join(l for l in txt.splitlines() if "MISMATCH" in l)[:2000])
died = open(os.path.join(out, "died.txt")).read().strip().replace("\n", "; ") if os.path.exists(os.path.join(out, "died.txt")) else ""
reads = sum(v[0] for v in tot.values()); bad = sum(v[1] for v in tot.values())
line = "%s fuzz_gate kernels %s seeds %s..: fuzz_parity %d reads / %d mismatches, fuzz_parity2 %d / %d, fuzz_parity3 %d / %d (%d configurations refused loudly by c3_create / c3_set_splints); total %d reads, %d mismatches, C3_DEBUG_POISON=1%s" % (
    time.strftime("%Y-%m-%d %H:%M"), sha, seed0, tot["p1"][0], tot["p1"][1], tot["p2"][0], tot["p2"][1], tot["p3"][0], tot["p3"][1], refused, reads, bad,
    ("; DIED: " + died) if died else "")
print(line)
open(os.path.join("profiles", "%s_fuzz_parity.txt" % tag), "a").write(line + "\n")
open(os.path.join("gpurun_out", "fuzz_gate_last.txt"), "w").write(line + "\n")
sys.exit(1 if (bad or died or reads == 0) else 0)
PY
rc=$?
[ $fail -ne 0 ] && rc=1
exit $rc
