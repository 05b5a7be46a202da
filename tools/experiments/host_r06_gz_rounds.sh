# round 6: why rounds of more chunks than threads are slower standalone and faster in the command line -- phase times per round (run on the GPU box)
cd "$GRAFT_REPO_ROOT"
python - <<PY
import os, sys, subprocess
sys.path.insert(0, ".")
import bench
recs = bench.make_reads("cfg2", 100000, 0, 16)
with open("/tmp/in.fastq", "wb") as f:
    for i, r in enumerate(recs):
        f.write(("@r%08d\n%s\n+\n%s\n" % (i, r[0], r[1])).encode())
subprocess.check_call("gzip -6 -k -f /tmp/in.fastq", shell=True)
PY
g++ -O3 -std=c++17 -DC3_GZPAR_PROF tools/gzpar_prof.cpp -o /tmp/gzpar_prof -lz -lpthread
for cr in "2097152 16" "2097152 32" "1048576 32" "1048576 16"; do set -- $cr; echo "== chunk $1, $2 chunks per round, 16 threads"; /tmp/gzpar_prof /tmp/in.fastq.gz 16 $1 $2 2>&1 | awk 'NR<=4 || /total/'; done
