#!/usr/bin/env python3
"""Second randomised parity sweep, GPU path against the oracle -- the dimensions tools/fuzz_parity.py leaves alone:
   python tools/fuzz_parity2.py [n_reads] [seed]
* a random splint per run (60..400 nt, sometimes low-complexity itself) and a random mdistcutoff;
* low-complexity inserts: two-letter alphabets, tandem repeats of period 1..12 with a few mutations, homopolymer blocks
  (every DP of the path is then full of ties: tie order is what is being compared);
* very many repeats of a short insert (up to 300 copies: more subreads than the 250 either side keeps, > 64 window layers);
* very long reads (up to ~150 kb);
* chimeras: two different inserts in one concatemer, a splint copy with a large deletion, a reverse-complemented block;
* inserts that contain a near copy of the splint's half.
One oracle call per run (one splint); prints the same summary line as fuzz_parity.py."""
import sys
import numpy as np
sys.path.insert(0, ".")
from c3poa_amd import synth
from c3poa_amd.seqio import revcomp

def generate(n, seed):
    """-> (splint, mdistcutoff, [(seq, qual)], [strand]); deterministic in (n, seed) -- tests replay single reads of a seed"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)


    def rnd(L, letters=4):
        if letters == 4:
            return acgt[rng.integers(0, 4, L)].tobytes().decode()
        sub = acgt[rng.permutation(4)[:letters]]
        return sub[rng.integers(0, letters, L)].tobytes().decode()


    def tandem(L):
        p = int(rng.integers(1, 13))
        unit = rnd(p)
        s = (unit * (L // p + 1))[:L]
        b = bytearray(s.encode())
        for _ in range(int(rng.integers(0, max(1, L // 40)))):         # a few point changes so that copies of the unit differ
            b[int(rng.integers(0, L))] = acgt[int(rng.integers(0, 4))]
        return b.decode()


    def low_complexity(L):
        k = int(rng.integers(0, 4))
        if k == 0:
            return rnd(L, 2)
        if k == 1:
            return tandem(L)
        if k == 2:                                                     # homopolymer blocks of 3..40
            out = []
            while sum(len(x) for x in out) < L:
                out.append("ACGT"[int(rng.integers(0, 4))] * int(rng.integers(3, 40)))
            return "".join(out)[:L]
        half = L // 2                                                  # half random, half tandem
        return rnd(half) + tandem(L - half)


    splint = rnd(int(rng.integers(60, 400))) if rng.random() < 0.8 else synth.SPLINT1
    if rng.random() < 0.15:
        splint = low_complexity(len(splint))
    mdist = int(rng.choice([100, 500, 500, 500, 1000, 2000]))
    reads, strands = [], []
    for i in range(n):
        kind = int(rng.integers(0, 10))
        if kind <= 2:                                                  # low-complexity insert, ordinary repeat count
            ins = low_complexity(int(rng.integers(80, 2500)))
            reps = int(rng.integers(0, 10))
        elif kind == 3:                                                # very many copies of a short insert
            ins = rnd(int(rng.integers(40, 260))) if rng.random() < 0.7 else low_complexity(int(rng.integers(40, 260)))
            reps = int(rng.integers(40, 300))
        elif kind == 4:                                                # very long read
            ins = rnd(int(rng.integers(1500, 9000)))
            reps = int(rng.integers(4, 16))
        elif kind == 5:                                                # the insert carries half a splint
            ins = rnd(int(rng.integers(300, 1500)))
            p = int(rng.integers(0, len(ins)))
            hs = splint[:len(splint) // 2] if rng.random() < 0.5 else splint[len(splint) // 2:]
            ins = ins[:p] + hs + ins[p:]
            reps = int(rng.integers(1, 8))
        else:
            ins = rnd(int(rng.integers(100, 2500)))
            reps = int(rng.integers(0, 12))
        k0, k1 = int(rng.integers(0, len(ins) + 1)), int(rng.integers(0, len(ins) + 1))
        units = [splint + ins for _ in range(reps)]
        if kind == 6 and reps >= 3:                                    # chimera: the second half repeats another insert
            other = rnd(int(rng.integers(100, 2500)))
            for r in range(reps // 2, reps):
                units[r] = splint + other
            if seed >= 100:                                            # (seeds >= 100: up to five unrelated inserts, copies interleaved)
                others = [ins, other] + [rnd(int(rng.integers(100, 2500))) for _ in range(int(rng.integers(1, 4)))]
                for r in range(reps):
                    units[r] = splint + others[int(rng.integers(0, len(others)))]
        if kind == 7 and reps >= 2:                                    # one splint copy lost most of itself
            r = int(rng.integers(0, reps))
            cut = int(rng.integers(len(splint) // 3, len(splint)))
            units[r] = splint[cut:] + ins
        if kind == 8 and reps >= 2:                                    # one unit reverse-complemented
            r = int(rng.integers(0, reps))
            units[r] = revcomp(units[r])
        clean = ins[len(ins) - k0:] + "".join(units) + splint + ins[:k1]
        err = float(rng.choice([0.0, 0.02, 0.08, 0.12, 0.2]))
        sb, qb = synth._mutate(rng, np.frombuffer(clean.encode(), dtype=np.uint8), sub=err * 0.4, ins=err * 0.25, dele=err * 0.35)
        s, q = sb.decode(), qb.decode()
        if rng.random() < 0.1:
            q = chr(33 + int(rng.integers(0, 60))) * len(s)            # flat qualities: every edge weight ties
        st = "+"
        if rng.random() < 0.5:
            s, q, st = revcomp(s), q[::-1], "-"
        if len(s):
            reads.append((s, q)); strands.append(st)


    return splint, mdist, reads, strands


if __name__ == "__main__":
    from c3poa_amd import _lib
    from oracle import oracle_py as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import os
    splint, mdist, reads, strands = generate(n, seed)
    splints, sids = [splint], [0] * len(reads)
    for k in range(1, int(os.environ.get("FUZZ_SPLINTS", "1"))):       # several splints of different lengths in ONE batch (reads interleaved)
        sp_k, _m, r_k, s_k = generate(max(8, n // 2), 1000 * k + seed)
        splints.append(sp_k); reads += r_k; strands += s_k; sids += [k] * len(r_k)
    if len(splints) > 1:
        perm = np.random.default_rng(seed).permutation(len(reads))
        reads = [reads[i] for i in perm]; strands = [strands[i] for i in perm]; sids = [sids[i] for i in perm]
    if os.environ.get("FUZZ_ONLY"):                                    # replay single reads of this seed (comma separated indices)
        keep = [int(x) for x in os.environ["FUZZ_ONLY"].split(",")]
        reads = [reads[i] for i in keep]; strands = [strands[i] for i in keep]; sids = [sids[i] for i in keep]
    h = _lib.Handle(mdistcutoff=mdist); h.set_splints(splints)
    sid_arr = np.array(sids, dtype=np.int16)
    nb = int(os.environ.get("FUZZ_BATCHES", "1"))
    if nb > 1:                                                         # ONE handle over several batches of very different sizes (buffers that grew, lists and flags of the batch before)
        cuts = sorted(set([0, len(reads)] + [int(x) for x in np.random.default_rng(seed).integers(0, len(reads) + 1, nb - 1)]))
        res_parts, cons = [], []
        for b0, b1 in zip(cuts, cuts[1:]):
            h.upload([r[0] for r in reads[b0:b1]], [r[1] for r in reads[b0:b1]], strands[b0:b1], sid_arr[b0:b1])
            h.run()
            r_, c_ = h.results()
            res_parts.append(r_.copy()); cons += list(c_)
        res = np.concatenate(res_parts)
    else:
        h.upload([r[0] for r in reads], [r[1] for r in reads], strands, sid_arr)
        h.run()
        res, cons = h.results()
    ores, ocons = [None] * len(reads), [None] * len(reads)
    for k, sp_k in enumerate(splints):                                 # (the oracle takes one splint per call)
        ix = [i for i in range(len(reads)) if sids[i] == k]
        o_, c_ = O.process_batch(sp_k, [reads[i] for i in ix], [strands[i] for i in ix], params=O.default_params(mdistcutoff=mdist), threads=16)
        for j, i in enumerate(ix):
            ores[i], ocons[i] = o_[j], c_[j]
    bad = 0
    for i in range(len(reads)):
        o = ores[i]
        same = (int(res[i]["status"]) == o.status and cons[i] == ocons[i] and (o.status not in (0, 3) or
                (int(res[i]["n_sub"]) == o.n_sub and int(res[i]["n_peaks"]) == o.n_peaks)))
        if not same:
            bad += 1
            if bad <= 10:
                print("MISMATCH read %d len %d strand %s: gpu status %d n_sub %d n_peaks %d cons %d | oracle status %d n_sub %d n_peaks %d cons %d" % (
                    i, len(reads[i][0]), strands[i], res[i]["status"], res[i]["n_sub"], res[i]["n_peaks"], len(cons[i]), o.status, o.n_sub, o.n_peaks, len(ocons[i])))
    st = np.bincount(res["status"], minlength=6)
    t = h.timing()
    print("seed %d splint %s nt mdist %d: reads %d (longest %d, most subreads %d)  mismatches %d  statuses OK/NA/NOPEAK/NOCONS/SHORT/LIMIT = %s  band layers %d fallback %d  POA second pass %d reads (beyond 16-bit cells / far arena: %d)" % (
        seed, "/".join(str(len(x)) for x in splints), mdist, len(reads), max(len(r[0]) for r in reads), int(res["n_sub"].max()), bad, st.tolist(), t["n_band_layers"], t["n_band_fallback"],
        t["n_poa_redo"], t["n_poa_redo16"]))
    sys.exit(1 if bad else 0)

