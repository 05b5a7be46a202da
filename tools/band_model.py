"""Model of k_window's banded row loop on real window alignments (CPU, numpy; design tool, not product).

The oracle's capture hook hands over every window alignment of a few synthetic reads (graph, mask, query, path).  For each one
the script computes the full DP, where the optimal path runs relative to the band centre that k_window derives from the
backbone position of a row, and whether the exactness certificate would accept the banded result for a
given band shape.  Usage: python tools/band_model.py cfg2 40 [wl wr]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

MT, MM, G = 3, -5, -4
NEG = -10 ** 8


def rows_of(a):
    """masked nodes in topological order -> (rows[1..R] node ids, rowof[node], preds per row (0 = virtual start row))"""
    rows = [None] + [int(v) for v in a["order"] if a["mask"][v]]
    rowof = {v: r for r, v in enumerate(rows) if v is not None}
    preds = [None]
    for r in range(1, len(rows)):
        p = [rowof[u] for u in a["preds"][rows[r]] if u in rowof]
        preds.append(p if p else [0])
    return rows, rowof, preds


def full_dp(a, lo=None, bw=None):
    """H[r][j]; with lo/bw: cells outside [lo[r], lo[r]+bw) are NEG"""
    rows, rowof, preds = rows_of(a)
    R, Q = len(rows) - 1, a["Q"]
    q = a["query"]
    H = np.full((R + 1, Q + 1), NEG, dtype=np.int64)
    j = np.arange(Q + 1)
    H[0] = j * G
    if lo is not None:
        H[0][(j < lo[0]) | (j >= lo[0] + bw)] = NEG
    for r in range(1, R + 1):
        sub = np.where(q == a["base"][rows[r]], MT, MM)
        key = np.full(Q + 1, NEG, dtype=np.int64)
        for p in preds[r]:
            key[1:] = np.maximum(key[1:], H[p][:-1] + sub)
            key = np.maximum(key, H[p] + G)
        if lo is not None:
            key[(j < lo[r]) | (j >= lo[r] + bw)] = NEG
        h = np.maximum.accumulate(key - G * j) + G * j
        if lo is not None:
            h[(j < lo[r]) | (j >= lo[r] + bw)] = NEG
        H[r] = np.maximum(h, NEG)
    succ = [[] for _ in range(R + 1)]
    for r in range(1, R + 1):
        for p in preds[r]:
            succ[p].append(r)
    ends = [r for r in range(1, R + 1) if not succ[r]]
    return H, rows, rowof, preds, succ, ends


def centre(a, rows):
    """expected column of every row: linear map of the backbone position reached so far onto the query"""
    R = len(rows) - 1
    blen, b, e, Q = a["blen"], a["begin"], a["end"], a["Q"]
    bb = np.zeros(R + 1, dtype=np.int64)
    cur = b - 1
    for r in range(1, R + 1):
        v = rows[r]
        if v < blen and v > cur:
            cur = v
        bb[r] = cur
    c = ((bb - b + 1) * Q + (e - b + 1) // 2) // (e - b + 1)
    c[0] = 0
    return c, bb


def analyse(a, wl, wr, stats):
    Hf, rows, rowof, preds, succ, ends = full_dp(a)
    R, Q = len(rows) - 1, a["Q"]
    assert max(Hf[r][Q] for r in ends) == a["score"], "full DP disagrees with the oracle score"
    c, bb = centre(a, rows)
    # path cells from the oracle's ops
    col = 0
    devs = []
    for node, qq in a["ops"]:
        if qq >= 0:
            col = qq + 1
        if node >= 0:
            devs.append(col - c[rowof[int(node)]])
    devs = np.array(devs)
    stats["dev_min"] = min(stats.get("dev_min", 0), devs.min()); stats["dev_max"] = max(stats.get("dev_max", 0), devs.max())
    stats.setdefault("devs", []).append(devs)
    bw = wl + wr + 1
    stats["cells_full"] = stats.get("cells_full", 0) + (R + 1) * (Q + 1)
    if Q + 1 <= bw:
        stats["cells_band"] = stats.get("cells_band", 0) + (R + 1) * (Q + 1)
        stats["n_small"] = stats.get("n_small", 0) + 1
        return
    lo = np.clip(c - wl, 0, Q + 1 - bw)
    lo = np.maximum.accumulate(lo)
    Hb = full_dp(a, lo, bw)[0]
    Sb = max(Hb[r][Q] for r in ends)
    inside = all(-wl <= d for d in devs)  # rough
    # certificate: shortest / longest row path to an end row
    sp = np.zeros(R + 1, dtype=np.int64); lp = np.zeros(R + 1, dtype=np.int64)
    for r in range(R, -1, -1):
        if succ[r]:
            sp[r] = 1 + min(sp[s] for s in succ[r]); lp[r] = 1 + max(lp[s] for s in succ[r])
    # cheap bounds: blocks remaining (lp upper bound); sp lower bound unknown -> 0
    best_exact = NEG; best_cheap = NEG
    # blocks after the block of row r (an upper bound of the rows any path can still visit: one node per aligned block)
    gr = [None] + [int(a["grp"][rows[r]]) for r in range(1, R + 1)]
    nb_after = np.zeros(R + 2, dtype=np.int64)
    for r in range(R - 1, -1, -1):
        nb_after[r] = nb_after[r + 1] + (1 if (r + 1 <= R and (r + 1 == R or gr[r + 1] != gr[r + 2] if r + 2 <= R else True)) else 0)
    # nb_after[r] = number of blocks that START after row r ... corrected below for rows inside a block
    blk_last = list(range(R + 1))
    for r in range(R - 1, 0, -1):
        if gr[r] == gr[r + 1]:
            blk_last[r] = blk_last[r + 1]
    nblk = np.zeros(R + 2, dtype=np.int64)          # nblk[r] = blocks among rows r..R (counted at their first row)
    for r in range(R, 0, -1):
        nblk[r] = nblk[r + 1] + (1 if (r == 1 or gr[r] != gr[r - 1]) else 0)
    hi = lo + bw - 1
    for r in range(0, R + 1):
        if not succ[r]:
            continue
        mlo = max(lo[s] for s in succ[r])
        # right exit from (r, hi[r]) if hi < Q
        if hi[r] < Q and Hb[r][hi[r]] > NEG // 2:
            rem = Q - hi[r]
            u = MT * rem + G * max(0, sp[r] - rem)
            best_exact = max(best_exact, Hb[r][hi[r]] + u)
            best_cheap = max(best_cheap, Hb[r][hi[r]] + MT * rem)
        # left exits: cells j in [lo[r], mlo) (vertical) -- diag exits need j + 1 < mlo
        for jj in range(lo[r], min(mlo, hi[r] + 1)):
            if Hb[r][jj] <= NEG // 2:
                continue
            rem = Q - jj
            u = MT * min(rem, lp[r]) + G * max(0, rem - lp[r])
            best_exact = max(best_exact, Hb[r][jj] + u)
            lpc = int(nblk[blk_last[r] + 1]) if r >= 1 else int(nblk[1])
            best_cheap = max(best_cheap, Hb[r][jj] + MT * min(rem, lpc) + G * max(0, rem - lpc))
    ok_truth = Sb == a["score"]
    stats["n"] = stats.get("n", 0) + 1
    stats["truth_ok"] = stats.get("truth_ok", 0) + ok_truth
    stats["cert_exact"] = stats.get("cert_exact", 0) + (best_exact < Sb)
    stats["cert_cheap"] = stats.get("cert_cheap", 0) + (best_cheap < Sb)
    stats.setdefault("margin", []).append(Sb - best_exact)
    stats.setdefault("bylayer", {}).setdefault(a["layer"], []).append((int(Sb - best_cheap), R, Q, a["end"] - a["begin"] + 1))
    if best_exact < Sb:
        assert ok_truth, "certificate accepted a wrong band result"
        stats["cells_band"] = stats.get("cells_band", 0) + (R + 1) * bw
    else:
        stats["cells_band"] = stats.get("cells_band", 0) + (R + 1) * bw + (R + 1) * (Q + 1)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    shapes = [(int(sys.argv[3]), int(sys.argv[4]))] if len(sys.argv) > 4 else [(63, 64), (95, 96), (80, 47), (110, 81)]
    recs = list(synth.generate(cfg, n_reads=n))
    O.win_capture(True)
    P = O.default_params(mdistcutoff=synth.CONFIGS[cfg]["mdist"])
    O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs], params=P, threads=1)
    als = O.win_captured()
    O.win_capture(False)
    print("%s: %d reads, %d window alignments" % (cfg, n, len(als)))
    for wl, wr in shapes:
        st = {}
        for a in als:
            analyse(a, wl, wr, st)
        devs = np.concatenate(st["devs"])
        print("band -%d..+%d (%d cols): path deviation from the centre min %d max %d, p0.1 %.0f p99.9 %.0f" % (
            wl, wr, wl + wr + 1, devs.min(), devs.max(), np.percentile(devs, 0.1), np.percentile(devs, 99.9)))
        n_ = max(st.get("n", 0), 1)
        print("   banded alignments %d (small, unbanded anyway: %d): band result == optimum %.4f, certificate(exact sp/lp) %.4f, "
              "certificate(cheap) %.4f, min/median margin %s/%s, cells computed / full = %.3f" % (
                  st.get("n", 0), st.get("n_small", 0), st.get("truth_ok", 0) / n_, st.get("cert_exact", 0) / n_, st.get("cert_cheap", 0) / n_,
                  min(st.get("margin", [0])), int(np.median(st.get("margin", [0]))), st["cells_band"] / st["cells_full"]))
        if os.environ.get("BYLAYER"):
            by_layer(st)


def by_layer(st):
    for k in sorted(st.get("bylayer", {})):
        v = st["bylayer"][k]
        print("      layer %2d: n %3d  cheap-cert pass %.2f  min margin %5d  mean R/span %.2f" % (
            k, len(v), np.mean([m > 0 for m, _, _, _ in v]), min(m for m, _, _, _ in v), np.mean([R / sp for _, R, _, sp in v])))


if __name__ == "__main__":
    main()
