"""GPU tests of the post-processing row: k_adapter == oracle bit for bit; CLI end to end."""
import os

import numpy as np
import pytest

from c3poa_amd import synth
from c3poa_amd.seqio import fastx_read, revcomp

pytestmark = pytest.mark.gpu


def _rand(rng, L):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, L))


def _noisy(rng, s, err):
    if not s:
        return s
    out, _q = synth._mutate(rng, np.frombuffer(s.encode(), dtype=np.uint8), sub=err * 0.4, ins=err * 0.25, dele=err * 0.35)
    return out.decode()


def test_adapter_finder_matches_oracle():
    from c3poa_amd import _lib
    from oracle import oracle_py as O
    rng = np.random.default_rng(31)
    adapters = [_rand(rng, 33), _rand(rng, 36), _rand(rng, 100), _rand(rng, 300)]          # 1, 1, 2 and 5 column chunks
    reads = []
    for i in range(40):
        a, b = adapters[i % 4], adapters[(i + 1) % 4]
        body = _rand(rng, int(rng.integers(200, 1500)))
        reads.append(_rand(rng, int(rng.integers(0, 50))) + _noisy(rng, a, 0.08) + body + _noisy(rng, revcomp(b), 0.08) + _rand(rng, int(rng.integers(0, 50))))
    reads += [_rand(rng, 2000), _rand(rng, 20), "ACGT", "A" * 150, adapters[0], revcomp(adapters[1])[:20], "NNNNNNNNNNNNNNNNNNNNNNNNN" + adapters[0] + "nnnn"]
    h = _lib.Handle()
    h.set_splints(adapters)
    h.upload(reads, ["!" * len(r) for r in reads], "?" * len(reads))
    tab = h.scan_adapters()
    exp = np.zeros_like(tab)
    for i, rd in enumerate(reads):
        for a, ad in enumerate(adapters):
            for rc in (0, 1):
                exp[i, a, rc] = O.adapter_align(rd, ad, bool(rc))
    assert np.array_equal(tab, exp)
    # the planted adapters are found where they were planted (strand, coordinates, nearly full length)
    for i in range(40):
        a, b = i % 4, (i + 1) % 4
        assert tab[i, a, 0, 5] >= 0.6 * len(adapters[a]) and tab[i, a, 0, 1] < 80
        assert tab[i, b, 1, 5] >= 0.6 * len(adapters[b]) and tab[i, b, 1, 2] > len(reads[i]) - 80
    assert tab[40, :, :, 0].max() < 40                                                      # random read: nothing convincing


def _write(path, recs):
    with open(path, "w") as fh:
        for n, s in recs:
            fh.write(">%s\n%s\n" % (n, s))


def test_post_cli_end_to_end(tmp_path):
    """consensus reads with 5'/3' adapters -> trimmed, re-oriented cDNA; PSL written and reused"""
    import C3POa_postprocessing as P
    rng = np.random.default_rng(8)
    a5, a3 = _rand(rng, 33), _rand(rng, 36)
    recs, truth = [], {}
    for i in range(60):
        cdna = _rand(rng, int(rng.integers(300, 1200)))
        pre, post = _rand(rng, int(rng.integers(5, 50))), _rand(rng, int(rng.integers(5, 50)))
        flip = i % 3 == 0
        seq = pre + (a3 if flip else a5) + cdna + revcomp(a5 if flip else a3) + post
        name = "c%03d_11.9_5000_3_%d" % (i, len(seq))
        recs.append((name, seq))
        truth[name + "_" + str(len(cdna))] = revcomp(cdna) if flip else cdna
    recs.append(("noadapter_10.0_3000_2_700", _rand(rng, 700)))
    fa, ad, out = str(tmp_path / "cons.fasta"), str(tmp_path / "adapters.fasta"), str(tmp_path / "out")
    _write(fa, recs)
    _write(ad, [("3Prime_adapter", a3), ("5Prime_adapter", a5)])
    n = P.main(P.parse_args(["-i", fa, "-a", ad, "-o", out, "-t"]))
    assert n == 60
    files = sorted(os.listdir(out))
    assert files == ["R2C2_full_length_consensus_reads.fasta", "R2C2_full_length_consensus_reads_left_splint.fasta",
                     "R2C2_full_length_consensus_reads_right_splint.fasta", "adapter_to_consensus_alignment.psl"]
    got = {r[0]: r[1] for r in fastx_read(out + "/R2C2_full_length_consensus_reads.fasta")}
    assert got == truth                                                    # exact adapters: trimmed at the exact junctions
    psl = open(out + "/adapter_to_consensus_alignment.psl").read().splitlines()
    assert all(len(l.split("\t")) == 21 for l in psl) and len(psl) >= 120
    mtime = os.stat(out + "/adapter_to_consensus_alignment.psl").st_mtime_ns
    P.main(P.parse_args(["-i", fa, "-a", ad, "-o", out]))                  # rerun without -t: PSL reused, untrimmed output
    assert os.stat(out + "/adapter_to_consensus_alignment.psl").st_mtime_ns == mtime
    untrimmed = {r[0]: r[1] for r in fastx_read(out + "/R2C2_full_length_consensus_reads.fasta")}
    assert set(untrimmed) == set(truth) and all(truth[k] in untrimmed[k] and len(untrimmed[k]) > len(truth[k]) for k in truth)


def test_post_cli_oligo_dt_demux(tmp_path):
    import C3POa_postprocessing as P
    rng = np.random.default_rng(9)
    adp = _rand(rng, 40)
    idx = [("dT_A", _rand(rng, 16)), ("dT_B", _rand(rng, 16)), ("dT_C", _rand(rng, 16))]
    recs, where = [], {}
    for i in range(30):
        cdna = _rand(rng, 500)
        k = i % 4
        tag = idx[k][1] if k < 3 else _rand(rng, 16)
        # undirectional library: the same adapter on both sides; the index sits right after the left adapter
        seq = _rand(rng, 20) + adp + tag + cdna + revcomp(adp) + _rand(rng, 20)
        name = "d%03d_12.0_4000_3_%d" % (i, len(seq))
        recs.append((name, seq))
        where[name] = idx[k][0] if k < 3 else "no_index_found"
    fa, ad, ix, out = (str(tmp_path / n) for n in ("cons.fasta", "adapter.fasta", "idx.fasta", "out"))
    _write(fa, recs); _write(ad, [("Adapter", adp)]); _write(ix, idx)
    P.main(P.parse_args(["-i", fa, "-a", ad, "-x", ix, "-o", out, "-u", "-t", "-n", "2", "-co"]))
    import gzip
    seen = {}
    for d in ("dT_A", "dT_B", "dT_C", "no_index_found"):
        assert os.path.exists(out + "/" + d + "/R2C2_full_length_consensus_reads.fasta.gz")
        for line in gzip.open(out + "/" + d + "/R2C2_full_length_consensus_reads.fasta.gz", "rt"):
            if line.startswith(">"):
                seen[line[1:].rsplit("_", 1)[0].strip()] = d
    assert seen == where
    tsv = open(out + "/R2C2_oligodT_multiplexing.tsv").read().splitlines()
    assert len(tsv) == 30 and all(len(l.split("\t")) == 3 for l in tsv)


def test_match_index_batch_matches_reference_golden():
    """GPU batch matcher == the reference's match_index on its golden cases, == the host statement on random pieces"""
    import json
    from c3poa_amd import _lib, postprocess
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post_cases.json")))
    names = [n for n, _s in g["indexes"]]
    seqs = [s for _n, s in g["indexes"]]
    h = _lib.Handle()
    got = h.match_index_batch([c["seq"] for c in g["match_index"]], seqs)
    assert [names[k] if k >= 0 else "-" for k in got] == [c["result"] for c in g["match_index"]]
    rng = np.random.default_rng(12)
    idx = [_rand(rng, 16), _rand(rng, 16), _rand(rng, 12), _rand(rng, 20), _rand(rng, 16)]       # mixed lengths: the `break` quirk
    pieces = []
    for i in range(3000):
        base = list(idx[i % 5])
        for _ in range(int(rng.integers(0, 4))):
            base[int(rng.integers(0, len(base)))] = "ACGT"[int(rng.integers(0, 4))]
        p = _rand(rng, int(rng.integers(0, 6))) + "".join(base) + _rand(rng, int(rng.integers(0, 6)))
        pieces.append(p[:int(rng.integers(0, 30))])
    pieces += ["", "A", "ACGT" * 16]
    assert np.array_equal(h.match_index_batch(pieces, idx), postprocess.match_batch_host(pieces, idx))
    assert np.array_equal(h.match_index_batch(pieces[:7], idx[:1]), np.full(7, -1))                # fewer than two indexes
    h.close()
