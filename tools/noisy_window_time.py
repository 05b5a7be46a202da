#!/usr/bin/env python3
"""Kernel times on cfg2-shaped reads whose error rates are scaled (1.0 = the config's 4 / 2.5 / 3.5 % sub / ins / del): how the polish behaves when
more band certificates fail -- since round 5 a layer that needs wide UNBANDED rows sends its window to k_window's full-size launch.
    python tools/noisy_window_time.py [n_reads] [factor ...]        (C3POA_LIB selects the library)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import _lib, synth
from c3poa_amd.seqio import revcomp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
factors = [float(x) for x in sys.argv[2:]] or [1.0, 1.5, 2.0]
for f in factors:
    recs = []
    for i in range(min(n, 2048)):
        rng = np.random.default_rng([77, i])
        ins = synth._ACGT[rng.integers(0, 4, 1216)].tobytes().decode()
        clean = ins[-108:] + (synth.SPLINT1 + ins) * 3 + synth.SPLINT1 + ins[:108]
        strand = "+"
        if rng.random() < 0.5:
            clean, strand = revcomp(clean), "-"
        s, q = synth._mutate(rng, np.frombuffer(clean.encode(), dtype=np.uint8), sub=0.04 * f, ins=0.025 * f, dele=0.035 * f)
        recs.append((s.decode(), q.decode(), strand))
    recs = (recs * (n // len(recs) + 1))[:n]
    h = _lib.Handle(mdistcutoff=500)
    h.set_splints([synth.SPLINT1])
    h.upload([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs])
    best = None
    for _ in range(3):
        h.run(); t = h.timing()
        best = t if best is None else {k: (min(v, best[k]) if isinstance(v, float) else v) for k, v in t.items()}
    res, cons = h.results()
    print("%s errors x%.1f: reads %d ok %d  ms_poa %.2f ms_prep %.2f ms_window %.2f  windows %d second launch %d  band layers %d fallback %d  computed/full %.3f" % (
        os.path.basename(os.environ.get("C3POA_LIB", "libc3poa_hip.so")), f, n, int((res["status"] == 0).sum()), best["ms_poa"], best["ms_prep"], best["ms_window"],
        best["n_windows"], best["n_win_redo"], best["n_band_layers"], best["n_band_fallback"], best["cells_polish_computed"] / max(best["cells_polish"], 1)))
    h.close()
