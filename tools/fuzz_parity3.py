#!/usr/bin/env python3
"""Third randomised parity sweep: NON-DEFAULT configurations (every scoring / band / window parameter of c3_config is part of
the C-ABI; the reference hard-codes one value of each, the tests used only those):
   python tools/fuzz_parity3.py [n_reads] [seed]
A random configuration per run -- conk scoring and penalty, Savitzky-Golay window / order / passes, abPOA scoring (match,
mismatch, both gap pieces) and band (b, f), racon scoring, window length and quality threshold, dangling band, mdistcutoff,
zero-repeat switch -- on the reads of tools/fuzz_parity2.py's generator (shorter: the point is the parameters)."""
import importlib.util
import os
import sys
import numpy as np
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fuzz_parity2", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity2.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)


def random_config(rng):
    c = {}
    if os.environ.get("FUZZ_EXTREME"):                             # large scores: the 16-bit cells of every kernel must hand over or refuse
        c.update(conk_match=int(rng.choice([20, 60, 127])), conk_mismatch=-int(rng.choice([4, 50, 127])), conk_penalty=int(rng.choice([20, 200, 2000])))
        o1, e1 = int(rng.integers(1, 60)), int(rng.integers(1, 30))
        c.update(poa_match=int(rng.integers(5, 60)), poa_mismatch=int(rng.integers(1, 60)), poa_o1=o1, poa_e1=e1, poa_o2=int(rng.integers(o1, 200)), poa_e2=int(rng.integers(1, e1 + 1)))
        c.update(pol_match=int(rng.integers(1, 60)), pol_mismatch=-int(rng.integers(1, 60)), pol_gap=-int(rng.integers(1, 60)))
        c.update(poa_band_b=int(rng.integers(1, 120)), poa_band_f=float(rng.choice([0.0, 0.01, 0.1, 0.5])))
        c.update(sg_window=int(rng.choice([5, 41, 101, 127])), sg_iters=int(rng.integers(1, 7)), pol_window=int(rng.choice([50, 500, 1500])), mdistcutoff=int(rng.choice([20, 500])))
        return c
    if rng.random() < 0.7:
        m = int(rng.integers(1, 9)); c.update(conk_match=m, conk_mismatch=-int(rng.integers(1, 9)), conk_penalty=int(rng.integers(4, 41)))
    if rng.random() < 0.5:
        c.update(sg_window=int(rng.choice([11, 21, 31, 41, 51, 81])), sg_order=int(rng.choice([2, 3])), sg_iters=int(rng.integers(1, 5)))
    if rng.random() < 0.8:
        o1, e1 = int(rng.integers(1, 10)), int(rng.integers(1, 5))
        e2 = int(rng.integers(1, e1 + 1)); o2 = int(rng.integers(o1, 41))
        c.update(poa_match=int(rng.integers(1, 9)), poa_mismatch=int(rng.integers(1, 9)), poa_o1=o1, poa_e1=e1, poa_o2=o2, poa_e2=e2)
    if rng.random() < 0.5:
        c.update(poa_band_b=int(rng.integers(4, 40)), poa_band_f=float(rng.choice([0.0, 0.005, 0.01, 0.02, 0.05])))
    if rng.random() < 0.8:
        c.update(pol_match=int(rng.integers(1, 9)), pol_mismatch=-int(rng.integers(1, 9)), pol_gap=-int(rng.integers(1, 13)))
    if rng.random() < 0.5:
        c.update(pol_window=int(rng.choice([100, 200, 350, 500, 640, 800])), pol_q=int(rng.integers(0, 16)))
    if rng.random() < 0.3:
        c.update(dang_band=int(rng.choice([32, 64, 128])))
    c.update(mdistcutoff=int(rng.choice([100, 500, 500, 1000])))
    if rng.random() < 0.2:
        c.update(zero=0)
    return c


if __name__ == "__main__":
    from c3poa_amd import _lib
    from oracle import oracle_py as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(10_000 + seed)
    cfg = random_config(rng)
    print("seed %d cfg %s" % (seed, cfg), flush=True)
    splint, _md, reads, strands = fz.generate(n, 50_000 + seed)
    keep = [i for i, r in enumerate(reads) if len(r[0]) < 40_000]           # (the oracle's time: the long reads are fuzz_parity2's job)
    reads = [reads[i] for i in keep]; strands = [strands[i] for i in keep]
    try:
        h = _lib.Handle(**cfg)
        h.set_splints([splint])
    except _lib.C3Error as e:
        print("seed %d: refused by c3_create / c3_set_splints (%s)" % (seed, e))
        sys.exit(0)
    h.upload([r[0] for r in reads], [r[1] for r in reads], strands)
    h.run()
    res, cons = h.results()
    okeys = {k: v for k, v in cfg.items()}
    ores, ocons = O.process_batch(splint, reads, strands, params=O.default_params(**okeys), threads=16)
    bad = 0
    for i in range(len(reads)):
        o = ores[i]
        same = (int(res[i]["status"]) == o.status and cons[i] == ocons[i] and (o.status not in (0, 3) or
                (int(res[i]["n_sub"]) == o.n_sub and int(res[i]["n_peaks"]) == o.n_peaks)))
        if not same:
            bad += 1
            if bad <= 6:
                print("MISMATCH read %d len %d strand %s: gpu status %d n_sub %d n_peaks %d cons %d | oracle status %d n_sub %d n_peaks %d cons %d" % (
                    i, len(reads[i][0]), strands[i], res[i]["status"], res[i]["n_sub"], res[i]["n_peaks"], len(cons[i]), o.status, o.n_sub, o.n_peaks, len(ocons[i])))
    st = np.bincount(res["status"], minlength=6)
    t = h.timing()
    print("seed %d cfg %s: reads %d  mismatches %d  statuses %s  band layers %d fallback %d  POA second pass %d (beyond 16-bit: %d)" % (
        seed, cfg, len(reads), bad, st.tolist(), t["n_band_layers"], t["n_band_fallback"], t["n_poa_redo"], t["n_poa_redo16"]))
    sys.exit(1 if bad else 0)
