"""BASELINE.json configs at real size on the GPU (cfg2 at its full size lives in test_gpu_golden_and_cli.py):

  cfg3  mixed 2-10 subreads / 1 kb insert     >= 50 000 reads in one batch
  cfg4  20 kb / 12 subreads / -d 1500          >= 2 000 reads (wide adaptive band, long graphs)
  cfg5  streamed FASTQ -> output files         the CLI pipeline (reader -> stage -> commit -> writer) over >= 3 GPU batches,
                                               with and without a PSL, several handles, and -n 2 where two GPUs exist

Size-independent properties at full size (shards == slices of the whole, every read yields a consensus, identity against the
synthetic truth) + bit-exact equality with the oracle on a random sample (reference: C3POa.py:110-173 per read,
C3POa.py:236-271 for the streamed loop).
"""
import os

import numpy as np
import pytest

from c3poa_amd import synth
from c3poa_amd.seqio import fastx_read

pytestmark = pytest.mark.gpu


def _whole_vs_shards_vs_oracle(cfg, n, n_sample, min_ok_frac, min_ident, shards=2):
    import bench
    from c3poa_amd import _lib
    from oracle import oracle_py as O
    md = synth.CONFIGS[cfg]["mdist"]
    recs = bench.make_reads(cfg, n, 0, 16)                          # (seq, qual, strand, truth)
    seqs, quals, st = [r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs]
    h = _lib.Handle(mdistcutoff=md)
    h.set_splints([synth.SPLINT1])
    h.upload(seqs, quals, st)
    h.run()
    res, cons = h.results()
    t = h.timing()
    assert (res["status"] == 0).sum() >= int(min_ok_frac * n)
    assert all(len(c) > 500 for c, s in zip(cons, res["status"]) if s == 0)
    # sharding invariance: equal slices processed alone give the slices of the whole
    parts, cells = [], 0
    for k in range(shards):
        lo, hi = n * k // shards, n * (k + 1) // shards
        h.upload(seqs[lo:hi], quals[lo:hi], st[lo:hi])
        h.run()
        parts += h.results()[1]
        cells += h.timing()["cells_poa"] + h.timing()["cells_polish"]
    assert parts == cons
    assert cells == t["cells_poa"] + t["cells_polish"]            # counted DP cells are a property of the reads, not of the batch
    h.close()
    idx = np.random.default_rng(11).choice(n, n_sample, replace=False)
    P = O.default_params(mdistcutoff=md)
    ores, ocons = O.process_batch(synth.SPLINT1, [(seqs[i], quals[i]) for i in idx], [st[i] for i in idx], params=P, threads=16)
    assert [cons[i] for i in idx] == ocons
    assert [int(res[i]["status"]) for i in idx] == [r.status for r in ores]
    assert [int(res[i]["n_sub"]) for i in idx] == [r.n_sub for r in ores]
    ident = [synth.identity(cons[i], recs[i][3]) for i in idx[:16] if cons[i]]
    assert np.mean(ident) > min_ident
    return res


def test_cfg3_50k_reads_mixed_repeats():
    res = _whole_vs_shards_vs_oracle("cfg3", 50000, 48, 0.999, 0.95)
    ns = res["n_sub"][res["status"] == 0]
    assert ns.min() <= 2 and ns.max() >= 10                        # the mix really covers 2..10 kept subreads


def test_cfg4_2k_long_reads_wide_band():
    res = _whole_vs_shards_vs_oracle("cfg4", 2048, 32, 0.999, 0.99)
    assert np.median(res["n_sub"][res["status"] == 0]) == 12


def test_cfgL_long_inserts_one_batch():
    """not a BASELINE config: 3 and 6 kb inserts, 3-5 repeats (reads of 10-32 kb) -- subreads beyond the LDS query copy and bands
    of 145-250 columns, i.e. the WIDE kernel instance of k_poa: sharding invariance, counted cells, oracle
    equality on a sample, and no read in the 32-bit pass"""
    res = _whole_vs_shards_vs_oracle("cfgL", 1536, 16, 0.995, 0.95)
    ns = res["n_sub"][res["status"] == 0]
    assert ns.min() >= 2 and ns.max() <= 6


@pytest.mark.parametrize("cfg,min_ok,min_ident", [("cfg2e15", 0.99, 0.91), ("cfg2e20", 0.97, 0.86)])
def test_noisy_reads_15_and_20_percent_errors(cfg, min_ok, min_ident):
    """round 6 (review item 2): the cfg2 shape with every error rate x 1.5 and x 2.0.  More band certificates of the polish fail there, more
    POA rows leave the one-chunk fast path, more windows take k_window's full-size launch -- the paths the 10 % configs barely touch.
    Whole batch against the oracle on a sample, sharding invariance, counted cells equal."""
    _whole_vs_shards_vs_oracle(cfg, 4096, 96, min_ok, min_ident)


# ---- cfg5: the streamed CLI -------------------------------------------------------------------------------------------

def _write_inputs(tmp_path, recs, with_psl):
    fq = str(tmp_path / "reads.fastq")
    with open(fq, "w") as fh:
        for i, r in enumerate(recs):
            fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
            if i % 97 == 5:
                fh.write("@short%d\nACGTACGT\n+\nIIIIIIII\n" % i)   # under the length cut-off, spread over the batches
    fa = str(tmp_path / "splint.fasta")
    open(fa, "w").write(">Splint1\n%s\n" % synth.SPLINT1)
    return fq, fa


def _cli(tmp_path, tag, fq, fa, recs, with_psl, env, extra=()):
    import C3POa
    out = str(tmp_path / tag)
    os.makedirs(out + "/tmp")
    if with_psl:
        synth.write_psl(out + "/tmp/splint_to_read_alignments.psl", recs)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        C3POa.main(C3POa.parse_args(["-r", fq, "-s", fa, "-o", out] + list(extra)))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


def _files(out):
    return (open(out + "/Splint1/R2C2_Consensus.fasta", "rb").read(), open(out + "/Splint1/R2C2_Subreads.fastq", "rb").read(),
            open(out + "/c3poa.log").read())


@pytest.mark.parametrize("with_psl", [True, False])
def test_cfg5_streamed_cli_multi_batch_equals_single_batch(tmp_path, with_psl):
    """reader -> c3_batch_stage -> c3_batch_commit -> writer across 4 GPU batches == the same input in ONE batch, byte for
    byte, with a PSL (table route) and without one (fused GPU splint finder route)"""
    from oracle import oracle_py as O
    from c3poa_amd import records
    n = 700
    recs = list(synth.generate("cfg5", n_reads=n))
    fq, fa = _write_inputs(tmp_path, recs, with_psl)
    one = _cli(tmp_path, "one", fq, fa, recs, with_psl, {"C3_GPU_BATCH_READS": "100000"})
    many = _cli(tmp_path, "many", fq, fa, recs, with_psl, {"C3_GPU_BATCH_READS": "200"})
    assert _files(one) == _files(many)
    # 4 GPU batches per range x 3 byte-range readers (parser threads) feeding the GPU: the same records, in the round-robin
    # order of the ranges
    ranged = _cli(tmp_path, "ranged", fq, fa, recs, with_psl, {"C3_GPU_BATCH_READS": "64", "C3_READERS_PER_GPU": "3", "C3_MIN_RANGE_BYTES": "1"})
    assert _sorted_records(ranged) == _sorted_records(one) and _files(ranged)[2] == _files(one)[2]
    got = {nm: s for nm, s, _q in fastx_read(many + "/Splint1/R2C2_Consensus.fasta")}
    assert len(got) >= n - 2
    log = open(many + "/c3poa.log").read().splitlines()
    assert log[1] == "Total reads: %d" % (n + len([i for i in range(n) if i % 97 == 5]))
    # bit-exact against the oracle on a sample of the records the pipeline wrote
    idx = list(range(0, n, 29))
    ores, ocons = O.process_batch(synth.SPLINT1, [(recs[i][1], recs[i][2]) for i in idx], [recs[i][3] for i in idx], threads=16)
    for i, r, c in zip(idx, ores, ocons):
        if c:
            assert got[records.consensus_header(recs[i][0], recs[i][2], len(recs[i][1]), r.n_sub, len(c))[1:]] == c
    if not with_psl:                                                 # the finder's PSL names every read once, in input order
        psl = open(many + "/tmp/splint_to_read_alignments.psl").read().splitlines()
        assert [l.split("\t")[9] for l in psl] == [r[0] for r in recs]
        assert open(one + "/tmp/splint_to_read_alignments.psl").read().splitlines() == psl


def _sorted_records(out):
    return (sorted(fastx_read(out + "/Splint1/R2C2_Consensus.fasta")), sorted(fastx_read(out + "/Splint1/R2C2_Subreads.fastq")))


def test_cfg5_two_handles_out_of_order_completion(tmp_path):
    """n_work > 1 (two worker threads, here on one GPU): buffer sets and result buffers are owned explicitly, so batches that
    finish out of order never see each other's names / subreads / qualities"""
    n = 900
    recs = list(synth.generate("cfg5", n_reads=n))
    fq, fa = _write_inputs(tmp_path, recs, True)
    one = _cli(tmp_path, "one", fq, fa, recs, True, {"C3_GPU_BATCH_READS": "100000"})
    two = _cli(tmp_path, "two", fq, fa, recs, True, {"C3_GPU_BATCH_READS": "64", "C3_HANDLES_PER_GPU": "2"})
    assert _sorted_records(one) == _sorted_records(two)
    assert open(one + "/c3poa.log").read() == open(two + "/c3poa.log").read()


def test_cfg5_two_gpus(tmp_path):
    """-n 2: one worker thread per GPU sharing the reader and the writer (C3POa.py:236 sized a process pool with -n)"""
    from c3poa_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    n = 900
    recs = list(synth.generate("cfg5", n_reads=n))
    fq, fa = _write_inputs(tmp_path, recs, True)
    one = _cli(tmp_path, "one", fq, fa, recs, True, {"C3_GPU_BATCH_READS": "100000"})
    two = _cli(tmp_path, "two", fq, fa, recs, True, {"C3_GPU_BATCH_READS": "100"}, extra=("-n", "2"))
    assert _sorted_records(one) == _sorted_records(two)
