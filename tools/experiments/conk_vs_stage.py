#!/usr/bin/env python3
"""Experiment: the conk phase of c3_batch_run with and without the next batch being staged beside it (H2D + 2-bit pack on the second stream).
    python tools/experiments/conk_vs_stage.py [reads] [cfg]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from c3poa_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
recs = list(synth.generate(cfg, n_reads=min(n, 4096)))
recs = (recs * (n // len(recs) + 1))[:n]
seqs = [r[1] for r in recs]; quals = [r[2] for r in recs]; strands = "".join(r[3] for r in recs)
lens = np.array([len(s) for s in seqs], dtype=np.int64)
off = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
host = _lib.PinnedBatch("".join(seqs).encode(), "".join(quals).encode(), off, strands)
h = _lib.Handle(mdistcutoff=synth.CONFIGS[cfg]["mdist"])
h.set_splints([synth.SPLINT1])
h.upload_pinned(host)
h.run()
keys = ("ms_conk", "ms_peaks", "ms_poa", "ms_prep", "ms_window")
fetcher = _lib.ResultFetcher(h, pinned=False)
for mode in ("run only", "stage beside run", "stage + result fetch", "run only", "stage + result fetch"):
    acc = {k: [] for k in keys}; wall = []
    for _ in range(3):
        t0 = time.perf_counter()
        if mode != "run only":
            h.stage_pinned(host)
        h.run()
        if mode == "stage + result fetch":
            fetcher.after_run()
        if mode != "run only":
            h.commit()
        wall.append((time.perf_counter() - t0) * 1e3)
        for k in keys: acc[k].append(h.last_timing[k])
    print("%-21s wall %.1f ms  " % (mode, min(wall)) + "  ".join("%s %.2f" % (k, min(v)) for k, v in acc.items()), flush=True)
fetcher.drain(); fetcher.close()
