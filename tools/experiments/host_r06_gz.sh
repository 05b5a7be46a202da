# round 6: plain gzip input by several threads -- decoder throughput and the CLI on a .gz file (run on the GPU box: its 16-core quota)
cd "$GRAFT_REPO_ROOT"
N=${N:-200000}
python - <<PY
import gzip, os, sys, time, subprocess
sys.path.insert(0, ".")
from c3poa_amd import synth
import bench
t = time.time()
recs = bench.make_reads("cfg2", $N, 0, 16)
with open("/tmp/in.fastq", "wb") as f:
    for i, r in enumerate(recs):
        f.write(("@r%08d\n%s\n+\n%s\n" % (i, r[0], r[1])).encode())
print("fastq %.1f MB in %.1f s" % (os.path.getsize("/tmp/in.fastq") / 1e6, time.time() - t))
t = time.time(); subprocess.check_call("gzip -6 -k -f /tmp/in.fastq", shell=True); print("gzip -6: %.1f MB, %.1f s" % (os.path.getsize("/tmp/in.fastq.gz") / 1e6, time.time() - t))
PY
g++ -O3 -std=c++17 tools/gzpar_bench.cpp -o /tmp/gzpar_bench -lz -lpthread
/tmp/gzpar_bench /tmp/in.fastq.gz 1 2 4 8 12 16
for c in 262144 2097152 4194304 8388608; do CHUNK=$c /tmp/gzpar_bench /tmp/in.fastq.gz 8 16 | grep parallel; done
# the same reads through gzip -1 (what tools/cli_throughput.py --gz feeds the command line): 25-45 % of a chunk's bytes stay marked
zcat /tmp/in.fastq.gz | gzip -1 > /tmp/in1.fastq.gz
for c in 1048576 4194304; do CHUNK=$c /tmp/gzpar_bench /tmp/in1.fastq.gz 1 8 16 | grep -v zlib; done
# rounds of more chunks than threads (the threads take chunks as they get free)
for cr in "2097152 16" "1048576 32" "1048576 48" "2097152 32" "524288 64"; do set -- $cr; CHUNK=$1 ROUND=$2 /tmp/gzpar_bench /tmp/in.fastq.gz 16 | grep parallel; CHUNK=$1 ROUND=$2 /tmp/gzpar_bench /tmp/in1.fastq.gz 16 | grep parallel; done
