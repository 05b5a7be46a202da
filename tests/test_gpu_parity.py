"""GPU parity: the HIP path (through the C ABI) must equal the CPU oracle bit for bit on the same
seeded inputs -- integer tracks, fp64 smoothed tracks, peaks, splits, drafts and consensi."""
import numpy as np
import pytest

from c3poa_amd import synth
from c3poa_amd.seqio import revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hnd():
    from c3poa_amd import _lib
    h = _lib.Handle()
    h.set_splints([synth.SPLINT1])
    yield h
    h.close()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle_py
    return oracle_py


def _rand_read(rng, L):
    seq = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
    qual = "".join(chr(33 + int(v)) for v in rng.integers(2, 41, L))
    return seq, qual


def test_conk_track_exact(hnd, O):
    rng = np.random.default_rng(1)
    reads, strands = [], []
    for rec in synth.generate("cfg1", n_reads=6):
        reads.append((rec[1], rec[2])); strands.append(rec[3])
    for L in (41, 63, 64, 65, 100, 283, 284, 285, 1000, 4999):
        reads.append(_rand_read(rng, L)); strands.append("+-"[L & 1])
    # a read that contains the splint verbatim, and one with non-ACGT bytes
    s, q = _rand_read(rng, 2000)
    reads.append((s[:700] + synth.SPLINT1 + s[700:], q[:700] + "I" * 284 + q[700:])); strands.append("+")
    s2 = list(s); s2[10] = "N"; s2[500] = "n"; s2[1999] = "R"
    reads.append(("".join(s2), q)); strands.append("-")
    hnd.upload([r[0] for r in reads], [r[1] for r in reads], strands)
    hnd.run(1)
    for i, (r, st) in enumerate(zip(reads, strands)):
        sp = synth.SPLINT1 if st == "+" else revcomp(synth.SPLINT1)
        ref = O.conk(sp, r[0])
        got = hnd.track(i)
        assert np.array_equal(ref, got), (i, len(r[0]), np.flatnonzero(ref != got)[:10])


def test_peaks_and_split_exact(hnd, O):
    recs = list(synth.generate("cfg1", n_reads=5)) + list(synth.generate("cfg3", n_reads=5)) + \
        list(synth.generate("cfg4", n_reads=1))
    rng = np.random.default_rng(2)
    extra = [_rand_read(rng, L) for L in (50, 300, 1500, 6000)]
    for k, (s, q) in enumerate(extra):
        recs.append(("x%d" % k, s, q, "+", ""))
    for md in (500, 1500):
        for i, rec in enumerate(recs):
            h2 = hnd
            h2.cfg.mdistcutoff = md
            # single-read batches so the smoothed track stays resident
            from c3poa_amd import _lib
            hh = _lib.Handle(mdistcutoff=md)
            hh.set_splints([synth.SPLINT1])
            hh.upload([rec[1]], [rec[2]], [rec[3]])
            hh.run(3)
            sp = synth.SPLINT1 if rec[3] == "+" else revcomp(synth.SPLINT1)
            tr = O.conk(sp, rec[1])
            pk, sm = O.call_peaks(tr, md, return_smoothed=True)
            got_sm = hh.smoothed(0)
            assert np.array_equal(sm.view(np.uint64), got_sm.view(np.uint64)), (i, md, np.abs(sm - got_sm).max())
            assert hh.raw_peaks(0).tolist() == pk.tolist(), (i, md)
            res, _ = hh.results(with_consensus=False)
            sp_o = O.split(pk, len(synth.SPLINT1), len(rec[1]))
            r = res[0]
            assert r["peaks"][:r["n_peaks"]].tolist() == sp_o["peaks"]
            assert [(int(r["sub_beg"][k]), int(r["sub_end"][k])) for k in range(r["n_sub"])] == sp_o["subs"]
            if sp_o["peaks"]:
                assert (r["has_front"], r["has_tail"]) == (sp_o["has_front"], sp_o["has_tail"])
                if r["has_front"]:
                    assert r["front_end"] == sp_o["front_end"]
                if r["has_tail"]:
                    assert r["tail_beg"] == sp_o["tail_beg"]
            hh.close()


def _subreads(rng, n, L, err=0.1):
    truth = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
    subs, quals = [], []
    for _ in range(n):
        s, q = synth._mutate(rng, np.frombuffer(truth.encode(), dtype=np.uint8), sub=err * 0.4, ins=err * 0.25, dele=err * 0.35)
        subs.append(s.decode()); quals.append(q.decode())
    return truth, subs, quals


@pytest.mark.parametrize("n,L", [(1, 300), (2, 200), (2, 1500), (3, 150), (3, 1500), (5, 800), (12, 1500), (3, 2600)])
def test_determine_consensus_exact(hnd, O, n, L):
    rng = np.random.default_rng(100 * n + L)
    truth, subs, quals = _subreads(rng, n, L)
    ft, fs, fq = _subreads(rng, 2, L)
    front = (fs[0][len(fs[0]) - L // 5:], fq[0][len(fs[0]) - L // 5:])
    tail = (fs[1][:L // 4], fq[1][:L // 4])
    for fr, tl in ((None, None), (front, tail), (front, None), (None, tail)):
        ref, ref_draft, _ = O.determine_consensus(subs, quals, fr, tl, return_draft=True)
        got, got_draft = hnd.determine_consensus(subs, quals, fr, tl, return_draft=True)
        assert got_draft == ref_draft, (n, L, fr is not None, tl is not None)
        assert got == ref, (n, L, fr is not None, tl is not None)


def test_poa_msa_rows_exact(hnd, O):
    rng = np.random.default_rng(5)
    for n, L in ((2, 120), (2, 900), (3, 400), (6, 300)):
        _, subs, _ = _subreads(rng, n, L, err=0.15)
        c_ref, m_ref, _ = O.poa_msa(subs, out_cons=(n != 2), out_msa=True)
        c_got, m_got = hnd.poa_msa(subs, out_cons=(n != 2), out_msa=True)
        assert m_got == m_ref
        assert c_got == c_ref
        for row, s in zip(m_got, subs):
            assert row.replace("-", "") == s


def test_divergent_and_ragged_subreads(hnd, O):
    rng = np.random.default_rng(9)
    # length-ragged subreads (0.8 .. 1.2x) stress the adaptive band
    truth = "".join("ACGT"[i] for i in rng.integers(0, 4, 1000))
    subs = [truth, truth[:820], truth[100:] + truth[:80], truth[:500] + truth[400:]]
    quals = ["".join(chr(33 + int(v)) for v in rng.integers(2, 41, len(s))) for s in subs]
    ref = O.determine_consensus(subs, quals)
    got = hnd.determine_consensus(subs, quals)
    assert got == ref
    # identical subreads, homopolymers
    subs = ["A" * 300, "A" * 300, "A" * 290]
    quals = ["5" * len(s) for s in subs]
    assert hnd.determine_consensus(subs, quals) == O.determine_consensus(subs, quals)


@pytest.mark.parametrize("cfg,n", [("cfg1", 48), ("cfg3", 24), ("cfg4", 4)])
def test_full_batch_exact(O, cfg, n):
    from c3poa_amd import _lib
    recs = list(synth.generate(cfg, n_reads=n))
    md = synth.CONFIGS[cfg]["mdist"]
    h = _lib.Handle(mdistcutoff=md)
    h.set_splints([synth.SPLINT1])
    strands = [r[3] for r in recs]
    strands[1] = "?"                      # one read without a splint assignment
    h.upload([r[1] for r in recs], [r[2] for r in recs], strands)
    h.run()
    res, cons = h.results()
    P = O.default_params(mdistcutoff=md)
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], strands, params=P, threads=8)
    for i in range(n):
        assert res[i]["status"] == ores[i].status, (i, res[i]["status"], ores[i].status)
        if ores[i].status in (0, 3):
            assert res[i]["n_sub"] == ores[i].n_sub
            assert res[i]["peaks"][:res[i]["n_peaks"]].tolist() == list(ores[i].peaks[:ores[i].n_peaks])
        assert cons[i] == ocons[i], i
    t = h.timing()
    ocells = sum(r.cells_poa for r in ores)
    assert t["cells_poa"] == ocells
    opol = sum(r.cells_polish for r in ores)
    assert t["cells_polish"] == opol
    # per-run figures: a second run of the same resident batch reports the same counts (no accumulation)
    h.run()
    t2 = h.timing()
    assert (t2["cells_conk"], t2["cells_poa"], t2["cells_polish"]) == (t["cells_conk"], t["cells_poa"], t["cells_polish"])
    h.close()
