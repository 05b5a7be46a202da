"""GPU tests that go beyond kernel-vs-oracle parity:
  * the HIP peak caller against the golden vectors produced by the REFERENCE'S OWN Python
  * the per-stage shims with the reference's call shapes
  * the drop-in CLI end to end: output tree, record text, -co, tail flush
  * size-independent properties at a larger batch (idempotence, shard-vs-whole equality, identity)"""
import gzip
import os
import types

import numpy as np
import pytest

from c3poa_amd import synth
from c3poa_amd.seqio import fastx_read, revcomp

pytestmark = pytest.mark.gpu


def test_call_peaks_against_reference_golden(golden):
    from c3poa_amd import _lib
    cases, sig = golden
    h = _lib.Handle()
    for c in cases["call_peaks"]:
        pk, sm = h.call_peaks(sig["track_" + c["name"]], c["min_dist"], return_smoothed=True)
        assert pk.tolist() == c["peaks"], (c["name"], c["min_dist"])
        if "sg3_" + c["name"] in sig.files:
            np.testing.assert_allclose(sm, sig["sg3_" + c["name"]], rtol=1e-9, atol=1e-7)
    h.close()


def test_shims_have_reference_call_shapes():
    from c3poa_amd import shims
    from oracle import oracle_py as O
    rec = next(iter(synth.generate("cfg1", n_reads=1)))
    sp = synth.SPLINT1 if rec[3] == "+" else revcomp(synth.SPLINT1)
    scores = shims.conk.conk(sp, rec[1], 20)                       # C3POa.py:123
    assert np.array_equal(scores, O.conk(sp, rec[1]))
    peaks = shims.call_peaks(scores, 500, 3, 41, 2)                # C3POa.py:124
    assert list(peaks) == O.call_peaks(scores, 500).tolist()
    assert shims.call_peaks(np.abs(np.random.default_rng(0).normal(1000, 300, 3000)).astype(int), 500, 3, 41, 2) == []
    sp_o = O.split(peaks, len(synth.SPLINT1), len(rec[1]))
    subs = [rec[1][b:e] for b, e in sp_o["subs"]]
    quals = [rec[2][b:e] for b, e in sp_o["subs"]]
    res = shims.msa_aligner(match=5).msa(subs, out_cons=True, out_msa=True)   # determine_consensus.py:43
    c_ref, m_ref, _ = O.poa_msa(subs)
    assert res.cons_seq == c_ref and res.msa_seq == m_ref
    res2 = shims.msa_aligner(match=5).msa(subs[:2], out_cons=False, out_msa=True)
    assert res2.msa_seq == O.poa_msa(subs[:2], out_cons=False)[1] and res2.cons_seq == []
    assert shims.msa_aligner(match=5).msa([], True, True).cons_seq == []
    dang = [rec[1][:sp_o["front_end"]], rec[1][sp_o["tail_beg"]:]]
    dq = [rec[2][:sp_o["front_end"]], rec[2][sp_o["tail_beg"]:]]
    args = types.SimpleNamespace(mdistcutoff=500, zero=True)
    cons, rep = shims.determine_consensus(args, rec[:3], subs, quals, dang, dq)
    assert rep == len(subs)
    assert cons == O.determine_consensus(subs, quals, (dang[0], dq[0]), (dang[1], dq[1]))


def _run_cli(tmp_path, recs, extra=()):
    import C3POa
    out = str(tmp_path / "out")
    os.makedirs(out + "/tmp")
    fq = str(tmp_path / "reads.fastq")
    with open(fq, "w") as fh:
        for r in recs:
            fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
        fh.write("@short\nACGT\n+\nIIII\n")
    fa = str(tmp_path / "splint.fasta")
    open(fa, "w").write(">Splint1\n%s\n" % synth.SPLINT1)
    synth.write_psl(out + "/tmp/splint_to_read_alignments.psl", recs[:-1])      # last read: no splint hit
    C3POa.main(C3POa.parse_args(["-r", fq, "-s", fa, "-o", out, "-g", "16"] + list(extra)))
    return out


def test_cli_output_tree_and_records(tmp_path):
    from oracle import oracle_py as O
    from c3poa_amd import records
    recs = list(synth.generate("cfg1", n_reads=40))          # 2 full groups of 16 + a tail of 8 (flushed)
    out = _run_cli(tmp_path, recs)
    assert sorted(os.listdir(out)) == ["Splint1", "c3poa.log", "tmp"]
    assert sorted(os.listdir(out + "/Splint1")) == ["R2C2_Consensus.fasta", "R2C2_Subreads.fastq"]
    log = open(out + "/c3poa.log").read().splitlines()
    assert log[0] == "C3POa version: v2.2.3" and log[1] == "Total reads: 41"
    assert log[2] == "No splint reads: 1 (2.44%)" and log[3] == "Under len cutoff: 1 (2.44%)"
    assert log[5] == "Reads after preprocessing: 39"
    got = {n: s for n, s, _q in fastx_read(out + "/Splint1/R2C2_Consensus.fasta")}
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs[:-1]], [r[3] for r in recs[:-1]], threads=8)
    exp = {}
    for r, res, cons in zip(recs[:-1], ores, ocons):
        if cons:
            exp[records.consensus_header(r[0], r[2], len(r[1]), res.n_sub, len(cons))[1:]] = cons
    assert got == exp and len(got) == 39
    subs = list(fastx_read(out + "/Splint1/R2C2_Subreads.fastq"))
    names = [s[0] for s in subs if s[0].startswith(recs[0][0] + "_")]
    assert names == [recs[0][0] + "_%d" % k for k in (1, 2, 3, 0, 4)]
    byname = {s[0]: s for s in subs}
    assert byname[recs[0][0] + "_1"][1] == recs[0][1][ores[0].sub_beg[0]:ores[0].sub_end[0]]


def test_cli_gzip_output(tmp_path):
    recs = list(synth.generate("cfg1", n_reads=5))
    out = _run_cli(tmp_path, recs, extra=("-co",))
    assert sorted(os.listdir(out + "/Splint1")) == ["R2C2_Consensus.fasta.gz", "R2C2_Subreads.fastq.gz"]
    txt = gzip.open(out + "/Splint1/R2C2_Consensus.fasta.gz", "rt").read()
    assert txt.count(">") == 4


def test_large_batch_properties():
    """cfg2-shaped batch: idempotence, shard == whole, identity against the synthetic truth, and exact
    equality with the oracle on a random sample."""
    from c3poa_amd import _lib
    from oracle import oracle_py as O
    n = 4096
    recs = list(synth.generate("cfg2", n_reads=n))
    h = _lib.Handle()
    h.set_splints([synth.SPLINT1])
    seqs, quals, st = [r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs]
    h.upload(seqs, quals, st)
    h.run()
    res1, cons1 = h.results()
    h.run()
    res2, cons2 = h.results()
    assert cons1 == cons2 and np.array_equal(res1["status"], res2["status"])           # idempotent
    half = n // 2
    h.upload(seqs[half:], quals[half:], st[half:])
    h.run()
    _r, cons_b = h.results()
    assert cons_b == cons1[half:]                                                        # a shard equals its slice
    t = h.timing()
    assert (res1["status"] == 0).sum() >= n - 2
    idx = np.random.default_rng(0).choice(n, 64, replace=False)
    ident = [synth.identity(cons1[i], recs[i][4]) for i in idx if cons1[i]]
    assert np.mean(ident) > 0.955 and np.min(ident) > 0.9
    ores, ocons = O.process_batch(synth.SPLINT1, [(recs[i][1], recs[i][2]) for i in idx], [recs[i][3] for i in idx], threads=16)
    assert [cons1[i] for i in idx] == ocons
    assert t["cells_conk"] > 0 and t["n_windows"] >= 3 * (half - 2)      # timing of the last (half) run
    h.close()


def test_pairwise_consensus_against_reference_golden(golden):
    """c3_pairwise_consensus (GPU, through the C ABI) == bin/consensus.py pairwise_consensus on the reference's golden cases"""
    from c3poa_amd import shims
    cases, _ = golden
    assert len(cases["pairwise"]) > 5
    for c in cases["pairwise"]:
        rows, quals = c["rows"], c["quals"]
        subs = [r.replace("-", "") for r in rows]
        assert shims.pairwise_consensus(rows, subs, quals) == c["cons"], c
    # identical subreads share the later quality (bin/consensus.py:77-79)
    assert shims.pairwise_consensus(["AC-GT", "ACG-T"], ["ACGT", "ACGT"], ["IIII", "5555"]) == \
        __import__("oracle.oracle_py", fromlist=["x"]).pairwise_consensus(["AC-GT", "ACG-T"], ["ACGT", "ACGT"], ["IIII", "5555"])


def test_full_size_cfg2_properties():
    """BASELINE cfg2 at its full size (100 000 reads in one batch): size-independent properties -- two half batches give
    the slices of the whole, a random sample equals the oracle exactly, every read but a handful yields a consensus"""
    import bench
    from c3poa_amd import _lib
    from oracle import oracle_py as O
    n = 100000
    recs = bench.make_reads("cfg2", n, 0, 16)                       # (seq, qual, strand, truth)
    seqs, quals, st = [r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs]
    h = _lib.Handle()
    h.set_splints([synth.SPLINT1])
    h.upload(seqs, quals, st)
    h.run()
    res, cons = h.results()
    assert (res["status"] == 0).sum() >= n - 20 and all(len(c) > 1000 for c, s in zip(cons, res["status"]) if s == 0)
    parts = []
    for lo, hi in ((0, n // 2), (n // 2, n)):
        h.upload(seqs[lo:hi], quals[lo:hi], st[lo:hi])
        h.run()
        parts += h.results()[1]
    assert parts == cons                                                             # sharding invariance
    h.close()
    idx = np.random.default_rng(5).choice(n, 48, replace=False)
    _ores, ocons = O.process_batch(synth.SPLINT1, [(seqs[i], quals[i]) for i in idx], [st[i] for i in idx], threads=16)
    assert [cons[i] for i in idx] == ocons
    ident = [synth.identity(cons[i], recs[i][3]) for i in idx[:24] if cons[i]]
    assert np.mean(ident) > 0.96


def _bumpy_track(rng, n, spacing, ties):
    """narrow bumps every `spacing` points on a low noisy floor: the median stays at the floor, every bump is a candidate"""
    x = np.arange(n)
    tr = np.abs(rng.normal(40, 12, n))
    centres = np.arange(spacing // 2, n - 20, spacing) + rng.integers(-spacing // 8, spacing // 8 + 1, size=len(np.arange(spacing // 2, n - 20, spacing)))
    amps = np.full(len(centres), 3000.0) if ties else rng.uniform(1500, 6000, len(centres))
    for c, a_ in zip(centres, amps):
        lo, hi = max(0, c - 40), min(n, c + 41)
        tr[lo:hi] += a_ * np.exp(-0.5 * ((x[lo:hi] - c) / 9.0) ** 2)
    tr = tr.astype(np.int32)
    return (tr // 32) * 32 if ties else tr


def test_call_peaks_with_few_and_with_many_candidates_equals_the_oracle():
    """k_peaks keeps up to 512 candidates in registers for the suppression loop and walks memory beyond that: tracks with tens, hundreds and
    more than a thousand local maxima above the height gate, equal values and plateaus (later candidate first), small and large distances:
    HIP == oracle"""
    from c3poa_amd import _lib
    from oracle import oracle_py as O
    rng = np.random.default_rng(2024)
    h = _lib.Handle()
    seen = {"registers": 0, "memory": 0}
    for n, spacing, ties in ((8000, 700, False), (60000, 150, False), (60000, 150, True), (200000, 150, False), (200000, 130, True), (120000, 260, False)):
        tr = _bumpy_track(rng, n, spacing, ties)
        n_cand = len(O.call_peaks(tr, 1))                 # (distance 1 keeps every candidate -- more than the 250 a read may have is the oracle's LIMIT too: only counted here)
        for md in (300, 1500, 4000):
            want = O.call_peaks(tr, md).tolist()
            if len(want) >= 250:
                continue
            got = h.call_peaks(tr, md).tolist()
            assert got == want, (n, spacing, ties, md, len(got), len(want))
            seen["memory" if n // spacing > 512 else "registers"] += len(want) > 0
    assert seen["registers"] >= 3 and seen["memory"] >= 3, seen
    h.close()
