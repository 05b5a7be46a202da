#!/usr/bin/env python3
"""Experiment: two handles (each with its own streams and slot scratch) running batches side by side from two host threads,
against one handle running the same batches one after the other.  Every kernel is a grid of persistent waves that pull
reads / windows from a queue, so a kernel's tail (waves that found the queue empty) and the host's work between launches
(work lists, counters) leave the device partly idle; a second handle's kernels can move into those slots.

    python tools/experiments/two_handles.py [reads_per_batch] [cfg] [steps]
"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from c3poa_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
recs = list(synth.generate(cfg, n_reads=min(n, 4096)))
recs = (recs * (n // len(recs) + 1))[:n]


def make():
    h = _lib.Handle(mdistcutoff=synth.CONFIGS[cfg]["mdist"])
    h.set_splints([synth.SPLINT1])
    h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
    h.run()                                                     # warm-up: allocations
    return h


def loop(h, k):
    for _ in range(k):
        h.run()


hs = [make(), make()]
for rep in range(2):
    t0 = time.perf_counter(); loop(hs[0], 2 * steps); t1 = time.perf_counter()
    one = 2 * steps * n / (t1 - t0)
    th = [threading.Thread(target=loop, args=(h, steps)) for h in hs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t1 = time.perf_counter()
    two = 2 * steps * n / (t1 - t0)
    print("%s %d reads per batch, %d batches: one handle %.1f k reads/s, two handles side by side %.1f k reads/s (%+.1f %%)" % (
        cfg, n, 2 * steps, one / 1e3, two / 1e3, 100.0 * (two - one) / one), flush=True)
r0, c0 = hs[0].results(with_consensus=True); r1, c1 = hs[1].results(with_consensus=True)
print("results of the two handles identical:", bool((r0 == r1).all()) and c0 == c1)
