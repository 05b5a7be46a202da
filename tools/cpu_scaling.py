#!/usr/bin/env python3
"""Diagnostic: oracle (CPU) throughput vs OpenMP thread count on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import synth
from oracle import oracle_py as O
recs = list(synth.generate("cfg2", n_reads=1024))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    if th > os.cpu_count():
        break
    n = min(len(recs), max(8, th * 8))
    t = time.perf_counter()
    O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs[:n]], [r[3] for r in recs[:n]], threads=th)
    dt = time.perf_counter() - t
    print("threads %3d: %4d reads in %.2fs = %.1f reads/s (%.2f per thread)" % (th, n, dt, n / dt, n / dt / th))
