"""First accepted band width per polish layer on noisy reads, CPU model of the band rules + certificate (tests/test_band_certificate_model.py): python tools/experiments/band_stats.py cfg2e15 6"""
import os, sys, pickle, collections
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'tests'))
import numpy as np
import test_band_certificate_model as M
from c3poa_amd import synth
from oracle import oracle_py as O
cfg, n = sys.argv[1], int(sys.argv[2])
hist = collections.Counter(); marg = collections.defaultdict(list)
for rec in synth.generate(cfg, n_reads=n):
    O.win_capture(True)
    try:
        O.process_batch(synth.SPLINT1, [(rec[1], rec[2])], [rec[3]], params=O.default_params(mdistcutoff=500), threads=1)
        als = O.win_captured()
    finally:
        O.win_capture(False)
    for a in als:
        rows, rowof, preds = M._rows(a)
        b0 = M._band(a, rows)
        if b0 is None:
            hist[("nb", 0)] += 1; continue
        start = b0[0]; acc = 0
        for cbm in range(start, 5):
            b = M._band(a, rows, cbm)
            if b is None: break
            cb, bw, lo = b
            H, D, succ = M._band_dp(a, rows, preds, lo, bw)
            bound = M._certificate(a, rows, preds, succ, H, lo, bw, cb)
            ops, bs = M._trace(a, rows, succ, H, D)
            if bound is not None: marg[cb].append(int(bs - bound))
            if bound is not None and bound < bs: acc = cb; break
        hist[(start, acc)] += 1
print(cfg, dict(hist))
for cb in sorted(marg): 
    m = np.array(marg[cb]); print("  cb", cb, "attempts", len(m), "accepted", int((m>0).sum()), "margin percentiles 5/25/50/75:", np.percentile(m,[5,25,50,75]).astype(int))
