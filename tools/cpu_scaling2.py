import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import synth
from oracle import oracle_py as O
import bench
recs = bench.make_reads("cfg2", 8192, 0, 16)
for n in (256, 2048, 8192):
    t = time.perf_counter()
    O.process_batch(synth.SPLINT1, [(r[0], r[1]) for r in recs[:n]], [r[2] for r in recs[:n]], threads=16)
    dt = time.perf_counter() - t
    print(os.environ.get("C3O_MALLOPT"), "threads 16: %5d reads in %.2fs = %.1f reads/s" % (n, dt, n / dt), flush=True)
