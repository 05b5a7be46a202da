"""What a wrong guess of conk's match / mismatch values could cost (conk is external, its scoring unpinned; the call
site fixes only penalty = 20, C3POa.py:123).  The oracle's track -> call_peaks -> split (C3POa.py:123-155) under the frozen
scoring (5 / -4) against other plausible scorings, per config shape: reads whose split has the same structure (kept subreads,
dangling flags) and the largest shift of a subread boundary.   python tools/conk_scoring_sensitivity.py [reads per config]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import synth  # noqa: E402
from c3poa_amd.seqio import revcomp  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

FROZEN = (5, -4)
OTHERS = ((1, -1), (2, -3), (5, -5), (3, -4), (5, -3))


def split_of(seq, strand, match, mismatch, mdist):
    sp = synth.SPLINT1 if strand == "+" else revcomp(synth.SPLINT1)
    track = O.conk(sp, seq, 20, match, mismatch)
    return O.split(O.call_peaks(track, mdist), len(synth.SPLINT1), len(seq))


def compare(recs, mdist, scoring):
    """-> (reads, identical splits, same structure, boundary shifts of the same-structure reads)"""
    same = struct = 0
    shifts = []
    for r in recs:
        a, b = split_of(r[1], r[3], FROZEN[0], FROZEN[1], mdist), split_of(r[1], r[3], scoring[0], scoring[1], mdist)
        same += a == b
        if len(a["subs"]) == len(b["subs"]) and (a["has_front"], a["has_tail"]) == (b["has_front"], b["has_tail"]):
            struct += 1
            shifts.append(max([abs(x[0] - y[0]) for x, y in zip(a["subs"], b["subs"])] +
                              [abs(x[1] - y[1]) for x, y in zip(a["subs"], b["subs"])] + [0]))
    return len(recs), same, struct, np.array(shifts)


def noisy(recs, seed=11):
    """the same reads with a second round of errors (~19 % in all)"""
    out = []
    for i, r in enumerate(recs):
        s, q = synth._mutate(np.random.default_rng([seed, i]), np.frombuffer(r[1].encode(), dtype=np.uint8))
        out.append((r[0], s.decode(), q.decode(), r[3], r[4]))
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    print("# frozen scoring %d/%d, penalty 20; per config: reads, identical split, same structure, boundary shift (bases) max / mean / reads > 5" % FROZEN)
    for cfg in ("cfg1", "cfg3", "cfg4", "cfg1 x2 errors"):
        base = cfg.split()[0]
        recs = list(synth.generate(base, n_reads=n if base != "cfg4" else max(20, n // 5)))
        if cfg.endswith("errors"):
            recs = noisy(recs)
        for sc in OTHERS:
            k, same, struct, sh = compare(recs, synth.CONFIGS[base]["mdist"], sc)
            print("%-15s %2d/%-3d reads %4d identical %4d same structure %4d shift max %4d mean %.2f over5 %d" % (
                cfg, sc[0], sc[1], k, same, struct, sh.max() if len(sh) else -1, sh.mean() if len(sh) else -1, int((sh > 5).sum())))
