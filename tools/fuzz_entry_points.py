#!/usr/bin/env python3
"""Randomised sweep of the single-call entry points of the C ABI (the seams the reference's own functions are swapped at,
INTEGRATION.md) against the oracle, with inputs that do NOT come from a split read:
   python tools/fuzz_entry_points.py [n_cases] [seed]
* c3_determine_consensus: 1..40 "subreads" of unrelated or related sequences, lengths 1..6000 mixed in one call, optional dangling
  pieces, flat and random qualities (the dispatcher of bin/determine_consensus.py:10-47);
* c3_poa_msa: the same lists through the MSA entry (consensus + rows);
* c3_zero_repeats: arbitrary piece pairs, with and without an overlap, pieces of 1..9000 bases;
* c3_call_peaks: arbitrary score tracks (noise, plateaus, ramps, constant, length 1..30000)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from c3poa_amd import _lib, synth
from oracle import oracle_py as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
rnd = lambda L: acgt[rng.integers(0, 4, L)].tobytes().decode()   # noqa: E731


def noisy(s, err):
    if not s:
        return s, ""
    b, q = synth._mutate(rng, np.frombuffer(s.encode(), dtype=np.uint8), sub=err * 0.4, ins=err * 0.3, dele=err * 0.3)
    return b.decode(), q.decode()


h = _lib.Handle()
bad = {"determine_consensus": 0, "poa_msa": 0, "zero_repeats": 0, "call_peaks": 0}
done = dict.fromkeys(bad, 0)
for case in range(n_cases):
    kind = case % 4
    if kind == 0 or kind == 1:
        base = rnd(int(rng.choice([1, 5, 40, 300, 900, 1500, 3000, 6000])))
        n = int(rng.choice([1, 2, 2, 3, 4, 7, 12, 40]))
        subs, quals = [], []
        for k in range(n):
            r = rng.random()
            src = base if r < 0.7 else (rnd(int(rng.integers(1, len(base) + 50))) if r < 0.85 else base[int(rng.integers(0, len(base))):] or base)
            s, q = noisy(src, float(rng.choice([0.0, 0.05, 0.15])))
            if not s:
                s, q = "A", "I"
            if rng.random() < 0.15:
                q = "5" * len(s)
            subs.append(s); quals.append(q)
        if kind == 0:
            front = noisy(base[len(base) // 2:], 0.1) if rng.random() < 0.5 and len(base) > 10 else None
            tail = noisy(base[:len(base) // 3], 0.1) if rng.random() < 0.5 and len(base) > 10 else None
            front = front if front and front[0] else None
            tail = tail if tail and tail[0] else None
            got = tuple(h.determine_consensus(subs, quals, front, tail, return_draft=True))
            exp = tuple(O.determine_consensus(subs, quals, front, tail, return_draft=True)[:2])
            name = "determine_consensus"
        else:
            oc = len(subs) != 2                                      # (two sequences: the reference asks for the rows only and goes on to pairwise_consensus)
            om = len(subs) != 1                                      # (one sequence: the reference never aligns it -- determine_consensus.py:26-28 -- and the library returns no rows)
            got = tuple(list(x) for x in h.poa_msa(subs, out_cons=oc, out_msa=om))
            exp = tuple(list(x) for x in O.poa_msa(subs, out_cons=oc, out_msa=om)[:2])
            name = "poa_msa"
    elif kind == 2:
        ins = rnd(int(rng.choice([30, 400, 1300, 5000, 9000])))
        a, b = int(rng.integers(0, len(ins))), int(rng.integers(1, len(ins) + 1))
        d0, q0 = noisy(ins[:b] if rng.random() < 0.8 else rnd(b), 0.1)
        d1, q1 = noisy(ins[a:] if rng.random() < 0.8 else rnd(len(ins) - a), 0.1)
        if not d0 or not d1 or len(d0) * len(d1) > 14_000_000:
            continue
        ml = int(rng.choice([0, 100, 2000]))
        got = h.zero_repeats(d0, q0, d1, q1, ml)
        exp = O.zero_repeats(d0, q0, d1, q1, params=O.default_params(mdistcutoff=ml))
        exp = exp if len(exp) >= ml else ""                          # (the length cut-off is the caller's in the oracle -- c3o_pipeline.c -- and the entry point's in the library)
        name = "zero_repeats"
    else:
        L = int(rng.choice([1, 20, 41, 100, 3000, 30000]))
        t = int(rng.integers(0, 5))
        x = (rng.integers(0, 2000, L) if t == 0 else np.full(L, int(rng.integers(0, 500))) if t == 1 else np.arange(L) % int(rng.integers(1, 700)) if t == 2
             else (rng.integers(0, 40, L) + 3000 * (np.arange(L) % int(rng.integers(50, 3000)) < 5)) if t == 3 else np.repeat(rng.integers(0, 900, L // 37 + 1), 37)[:L])
        md = int(rng.choice([1, 20, 500, 5000]))
        try:
            got = [int(v) for v in h.call_peaks(x.astype(np.int32), md)]
        except _lib.C3Error as e:
            got = "error"
        try:
            exp = [int(v) for v in O.call_peaks(x.astype(np.int32), md)]
        except Exception:
            exp = "error"
        name = "call_peaks"
    done[name] += 1
    if name == "call_peaks" and exp == "error" and got in ([], "error"):
        continue                                                     # (a track shorter than half a smoothing window: the reference's padding raises, the library returns no peaks)
    if got != exp:
        bad[name] += 1
        if sum(bad.values()) <= 8:
            if name == "poa_msa":
                print("DIFFERENT poa_msa case %d: n %d lens %s | consensus equal %s (%d / %d) | rows equal %s (%d / %d rows of %s / %s)" % (
                    case, len(subs), [len(x) for x in subs][:12], got[0] == exp[0], len(got[0][0]) if got[0] else -1, len(exp[0][0]) if exp[0] else -1,
                    got[1] == exp[1], len(got[1]), len(exp[1]), len(got[1][0]) if got[1] else -1, len(exp[1][0]) if exp[1] else -1))
            elif name == "zero_repeats":
                print("DIFFERENT zero_repeats case %d: d0 %d d1 %d min_len %d: gpu %d | oracle %d bases" % (case, len(d0), len(d1), ml, len(got), len(exp)))
            else:
                print("DIFFERENT %s case %d: gpu %s | oracle %s" % (name, case, str(got)[:150], str(exp)[:150]))
print("seed %d: cases %s  differences %s" % (seed, done, bad))
sys.exit(1 if sum(bad.values()) else 0)
