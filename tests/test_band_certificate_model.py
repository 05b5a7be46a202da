"""The exactness certificate of k_window's banded rows, checked WITHOUT a GPU and without any kernel code:
the oracle's capture hook hands over real window alignments (graph, layer, optimal path of the FULL matrix); a plain-Python
restatement of the band rules (band width by graph inflation, band start from the backbone position, cells outside = -inf, the
oracle's tie order) fills the band, evaluates the certificate bound exactly as specified and walks back.  Claim under test:
whenever the certificate accepts (bound < banded optimum), the banded traceback IS the full-matrix traceback."""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

MT, MM, G = 3, -5, -4
NEG = -10 ** 8
W_LEFT = {2: 70, 3: 105, 4: 140}


def _rows(a):
    rows = [None] + [int(v) for v in a["order"] if a["mask"][v]]
    rowof = {v: r for r, v in enumerate(rows) if v is not None}
    preds = [None] + [([rowof[u] for u in a["preds"][rows[r]] if u in rowof] or [0]) for r in range(1, len(rows))]
    return rows, rowof, preds


def _band(a, rows, cb_min=0):
    R, Q = len(rows) - 1, a["Q"]
    span = a["end"] - a["begin"] + 1
    need = (Q + 1 + 63) // 64
    cb = 2 if R * 4 < span * 5 else (3 if R * 2 < span * 3 else 4)
    cb = max(cb, cb_min)
    if cb > 4 or need > 10 or need <= cb or span < 1:
        return None
    bw = 64 * cb
    lo = np.zeros(R + 1, dtype=np.int64)
    cur = a["begin"] - 1
    for r in range(1, R + 1):
        v = rows[r]
        if v < a["blen"] and v > cur:
            cur = v
        lo[r] = min(max(((cur - a["begin"] + 1) * Q + span // 2) // span - W_LEFT[cb], 0), Q + 1 - bw)
    return cb, bw, lo


def _band_dp(a, rows, preds, lo, bw):
    R, Q, q = len(rows) - 1, a["Q"], a["query"]
    H = np.full((R + 1, Q + 1), NEG, dtype=np.int64)
    D = {}
    H[0][:min(bw, Q + 1)] = np.arange(min(bw, Q + 1)) * G
    succ = [[] for _ in range(R + 1)]
    for r in range(1, R + 1):
        for p in preds[r]:
            succ[p].append(r)
        base = a["base"][rows[r]]
        for j in range(lo[r], min(lo[r] + bw, Q + 1)):
            best, d = NEG, None
            if j > 0:
                s = MT if base == q[j - 1] else MM
                for p in preds[r]:
                    if H[p][j - 1] > NEG // 2 and H[p][j - 1] + s > best:
                        best, d = H[p][j - 1] + s, (0, p)
            for p in preds[r]:
                if H[p][j] > NEG // 2 and H[p][j] + G > best:
                    best, d = H[p][j] + G, (1, p)
            if j > lo[r] and H[r][j - 1] > NEG // 2 and H[r][j - 1] + G > best:
                best, d = H[r][j - 1] + G, (2, r)
            H[r][j] = best
            D[(r, j)] = d
    return H, D, succ


def _certificate(a, rows, preds, succ, H, lo, bw, cb):
    R, Q = len(rows) - 1, a["Q"]
    grp = [None] + [int(a["grp"][rows[r]]) for r in range(1, R + 1)]
    bidx, nb = np.zeros(R + 1, dtype=np.int64), 0
    for r in range(1, R + 1):
        nb += r == 1 or grp[r] != grp[r - 1]
        bidx[r] = nb
    bound = (bw - 1) * G + MT * (Q - bw + 1)
    for r in range(1, R + 1):
        hi, nbr = lo[r] + bw - 1, nb - bidx[r]
        if lo[r] - lo[r - 1] > 3:
            return None
        if any(p > 0 and lo[r] - lo[p] > 24 for p in preds[r]):
            return None
        if hi < Q:
            bound = max(bound, H[r][hi] + MT * (Q - hi))
        if preds[r] == [0] and lo[r] > 0:
            bound = max(bound, MT * min(Q, nbr + 1) + G * max(0, Q - nbr - 1))
        if succ[r]:
            dist = max(s - r for s in succ[r])
            ls = lo[r + dist] - lo[r]
            if dist > 64 or ls > 2 * cb:
                return None
            for b in range(ls):
                rem = Q - lo[r] - b
                bound = max(bound, H[r][lo[r] + b] + MT * min(rem, nbr) + G * max(0, rem - nbr))
    return bound


def _trace(a, rows, succ, H, D):
    R, Q = len(rows) - 1, a["Q"]
    bs, br = NEG, 0
    for r in range(1, R + 1):
        if not succ[r] and H[r][Q] > bs:
            bs, br = H[r][Q], r
    ops, r, j = [], br, Q
    while r > 0 or j > 0:
        if r == 0:
            ops.append((-1, j - 1)); j -= 1
            continue
        d = D.get((r, j))
        if d is None:
            return None, bs
        if d[0] == 2:
            ops.append((-1, j - 1)); j -= 1
        elif d[0] == 0:
            ops.append((rows[r], j - 1)); j -= 1; r = d[1]
        else:
            ops.append((rows[r], -1)); r = d[1]
    return ops[::-1], bs


def test_certificate_accepts_only_bands_that_hold_the_full_matrix_path():
    reads = [(r[1], r[2], r[3]) for r in synth.generate("cfg2", n_reads=2)]
    reads += [(r[1], r[2], r[3]) for r in synth.generate("cfg4", n_reads=1)][:1]
    rd, st = pickle.load(open(os.path.join(ROOT, "tests", "golden", "band_regress_read.pkl"), "rb"))     # 25 % error, ragged: many rejections
    reads.append((rd[0], rd[1], st))
    accepted = rejected = 0
    for k, (seq, qual, strand) in enumerate(reads):
        O.win_capture(True)
        try:
            O.process_batch(synth.SPLINT1, [(seq, qual)], [strand], params=O.default_params(mdistcutoff=1500 if k == 2 else 500), threads=1)
            als = O.win_captured()
        finally:
            O.win_capture(False)
        for a in als[:: (3 if k == 2 else 1)]:                                   # (cfg4: every third alignment, for time)
            rows, rowof, preds = _rows(a)
            cb_min = 0
            while True:
                bd = _band(a, rows, cb_min)
                if bd is None:
                    break
                cb, bw, lo = bd
                H, D, succ = _band_dp(a, rows, preds, lo, bw)
                ops, sb = _trace(a, rows, succ, H, D)
                bound = _certificate(a, rows, preds, succ, H, lo, bw, cb)
                assert sb <= a["score"]                                        # a band can only lose
                if bound is not None and sb > NEG // 2 and bound < sb:
                    accepted += 1
                    assert sb == a["score"] and ops == [(int(x), int(y)) for x, y in a["ops"]], (k, a["win"], a["layer"], cb)
                    break
                rejected += 1
                cb_min = cb + 1                                                # the kernel retries with the next wider band
    assert accepted >= 30 and rejected >= 5, (accepted, rejected)
