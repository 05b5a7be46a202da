"""CPU property tests of the oracle itself (SURVEY.md 4 "unit / property"): the banded/graph DP
stages against naive restatements on small random inputs, and invariants of the full path."""
import numpy as np

from c3poa_amd import synth
from c3poa_amd.seqio import revcomp
from oracle import oracle_py as O


def _naive_conk(splint, seq, match=5, mismatch=-4, pen=20):
    S, L = len(splint), len(seq)
    H = np.zeros((S + 1, L + 1), dtype=np.int64)
    tr = np.zeros(L, dtype=np.int64)
    for i in range(1, S + 1):
        for j in range(1, L + 1):
            s = match if splint[i - 1] == seq[j - 1] else mismatch
            H[i, j] = max(0, H[i - 1, j - 1] + s, H[i - 1, j] - pen, H[i, j - 1] - pen)
            d = (j - 1) - (i - 1)
            if d >= 0:
                tr[d] += H[i, j]
    return tr


def test_conk_vs_naive():
    rng = np.random.default_rng(0)
    for S, L in ((5, 30), (17, 64), (40, 41), (64, 200)):
        sp = "".join("ACGT"[i] for i in rng.integers(0, 4, S))
        sq = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
        sq = sq[:L // 3] + sp[:S // 2] + sq[L // 3 + S // 2:]
        assert O.conk(sp, sq).tolist() == _naive_conk(sp, sq[:L]).tolist()[:len(sq)]


def test_conk_peaks_at_splint_start():
    rng = np.random.default_rng(1)
    body = "".join("ACGT"[i] for i in rng.integers(0, 4, 3000))
    seq = body[:1000] + synth.SPLINT1 + body[1000:]
    tr = O.conk(synth.SPLINT1, seq)
    assert int(np.argmax(tr)) == 1000
    assert int(np.argmax(O.conk(revcomp(synth.SPLINT1), revcomp(seq)))) == len(seq) - 1000 - len(synth.SPLINT1)


def _global_linear_score(a, b, m=5, x=-4):
    # reference for the 2-sequence case: the POA of 2 sequences is a banded convex-gap global alignment
    def gap(k):
        return min(4 + 2 * k, 24 + k) if k else 0
    n, mm = len(a), len(b)
    NEG = -10 ** 9
    H = np.full((n + 1, mm + 1), NEG, dtype=np.int64)
    H[0, 0] = 0
    for i in range(n + 1):
        for j in range(mm + 1):
            if i == 0 and j == 0:
                continue
            best = NEG
            if i and j:
                best = max(best, H[i - 1, j - 1] + (m if a[i - 1] == b[j - 1] else x))
            for k in range(1, i + 1):
                best = max(best, H[i - k, j] - gap(k))
            for k in range(1, j + 1):
                best = max(best, H[i, j - k] - gap(k))
            H[i, j] = best
    return int(H[n, mm])


def _msa_score(rows, m=5, x=-4):
    def gap(k):
        return min(4 + 2 * k, 24 + k)
    a, b = rows
    s, i = 0, 0
    while i < len(a):
        if a[i] != "-" and b[i] != "-":
            s += m if a[i] == b[i] else x
            i += 1
            continue
        g = a if a[i] == "-" else b
        k = 0
        while i + k < len(a) and g[i + k] == "-" and (b if g is a else a)[i + k] != "-":
            k += 1
        s -= gap(k)
        i += k
    return s


def test_poa_two_sequences_is_optimal_global_alignment():
    rng = np.random.default_rng(3)
    for _ in range(12):
        L = int(rng.integers(20, 70))
        a = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
        b, _q = synth._mutate(rng, np.frombuffer(a.encode(), dtype=np.uint8), sub=0.05, ins=0.03, dele=0.03)
        b = b.decode()
        _c, rows, _cells = O.poa_msa([a, b], out_cons=False, out_msa=True)
        assert rows[0].replace("-", "") == a and rows[1].replace("-", "") == b
        assert all(not (x == "-" and y == "-") for x, y in zip(*rows))
        assert _msa_score(rows) == _global_linear_score(a, b)


def test_poa_rows_and_consensus_invariants():
    rng = np.random.default_rng(4)
    truth = "".join("ACGT"[i] for i in rng.integers(0, 4, 400))
    subs = [synth._mutate(rng, np.frombuffer(truth.encode(), dtype=np.uint8))[0].decode() for _ in range(7)]
    cons, rows, _ = O.poa_msa(subs)
    assert len(set(len(r) for r in rows)) == 1
    for r, s in zip(rows, subs):
        assert r.replace("-", "") == s
    assert synth.identity(cons[0], truth) > 0.97
    # first sequence alone: consensus is the sequence
    assert O.poa_msa([truth])[0] == [truth]
    assert O.poa_msa([]) == ([], [], 0)


def test_full_path_invariants_and_thread_independence():
    recs = list(synth.generate("cfg1", n_reads=24)) + list(synth.generate("cfg3", n_reads=8))
    reads = [(r[1], r[2]) for r in recs]
    st = [r[3] for r in recs]
    r1, c1 = O.process_batch(synth.SPLINT1, reads, st, threads=1)
    r8, c8 = O.process_batch(synth.SPLINT1, reads, st, threads=8)
    assert c1 == c8 and [x.status for x in r1] == [x.status for x in r8]
    for r, c, rec in zip(r1, c1, recs):
        assert r.status == 0
        assert all(r.sub_beg[k] < r.sub_end[k] for k in range(r.n_sub))
        assert all(r.peaks[k] < r.peaks[k + 1] for k in range(r.n_peaks - 1))
        assert abs(len(c) - len(rec[4])) < 0.05 * len(rec[4])
        assert synth.identity(c, rec[4]) > 0.9
    # a wrong strand assignment finds no splint peaks
    wrong = ["-" if s == "+" else "+" for s in st[:4]]
    rw, cw = O.process_batch(synth.SPLINT1, reads[:4], wrong, threads=2)
    assert all(x.status != 0 for x in rw) and all(c == "" for c in cw)
