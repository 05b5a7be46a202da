"""CPU tests of the post-processing host logic (c3poa_amd/postprocess.py, c3_match_index) against golden outputs made
by running the reference's own parse_blat / match_index / write_fasta_file (tests/golden/make_golden_post.py)."""
import json
import os
import types

import pytest

from c3poa_amd import _lib, postprocess

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def post_golden():
    return json.load(open(os.path.join(HERE, "golden", "post_cases.json")))


def _tree(path):
    out = {}
    for root, _dirs, files in os.walk(path):
        for f in files:
            if not f.endswith(".psl"):
                out[os.path.relpath(os.path.join(root, f), path)] = open(os.path.join(root, f)).read()
    return out


def test_match_index_matches_reference(post_golden):
    names = [n for n, _s in post_golden["indexes"]]
    seqs = [s for _n, s in post_golden["indexes"]]
    got = []
    for c in post_golden["match_index"]:
        k = _lib.match_index(c["seq"], seqs)
        got.append(names[k] if k >= 0 else "-")
    assert got == [c["result"] for c in post_golden["match_index"]]
    assert {"-", "idx0", "idx1", "idx2", "idx3"} <= set(got)


def test_parse_and_write_match_reference(post_golden, tmp_path):
    for k, c in enumerate(post_golden["cases"]):
        path = str(tmp_path / ("case%d" % k)) + "/"
        os.makedirs(path)
        with open(path + postprocess.PSL_NAME, "w") as fh:
            fh.write("\n".join(c["psl"]) + "\n")
        idx_to_seq = {n: s for n, s in c["indexes"]}
        seq_to_idx = {s: n for n, s in c["indexes"]}
        f = c["flags"]
        args = types.SimpleNamespace(undirectional=f.get("u", False), barcoded=f.get("b", False), trim=f.get("t", False), threads=1)
        ad = postprocess.parse_blat(path + postprocess.PSL_NAME, c["reads"])
        postprocess.write_fasta_file(args, path, ad, c["reads"], seq_to_idx, idx_to_seq, match_batch=postprocess.match_batch_host)
        got = _tree(path)
        assert got == c["files"], "case %d (%s)" % (k, f)
        assert sum(len(v) for v in got.values()) > 1000


def test_psl_line_round_trips_through_parse_blat(tmp_path):
    # score, qS, qE, tS, tE, matches, mism, qBaseIns, tBaseIns, qNumIns, tNumIns, L
    e_plus = (60, 40, 70, 2, 33, 30, 1, 0, 0, 0, 0, 500)
    e_minus = (50, 430, 460, 0, 29, 28, 1, 1, 0, 1, 0, 500)
    psl = tmp_path / postprocess.PSL_NAME
    psl.write_text(postprocess.psl_line("r", 500, "5Prime_adapter", 33, "+", e_plus) + "\n" +
                   postprocess.psl_line("r", 500, "3Prime_adapter", 36, "-", e_minus) + "\n")
    assert all(len(l.split("\t")) == 21 for l in psl.read_text().splitlines())
    ad = postprocess.parse_blat(str(psl), {"r": "A" * 500})
    assert ad["r"]["+"][1] == ("5Prime_adapter", 30.0, 70 + (33 - 33))
    assert ad["r"]["-"][1] == ("3Prime_adapter", 28.0, 430 - (36 - 29))


def test_post_cli_flags():
    import C3POa_postprocessing as P
    a = P.parse_args(["-i", "c.fa", "-a", "ad.fa"])
    assert (a.output_path, a.undirectional, a.trim, a.barcoded, a.threads, a.groupSize, a.blatThreads, a.compress_output) == \
        (os.getcwd(), False, False, False, 1, 1000, False, False)
    a = P.parse_args(["-i", "c", "-a", "a", "-x", "idx", "-o", "o", "-u", "-t", "-b", "-n", "3", "-g", "9", "-bt", "-co", "-c", "cfg"])
    assert (a.index_file, a.output_path, a.undirectional, a.trim, a.barcoded, a.threads, a.groupSize, a.blatThreads,
            a.compress_output, a.config) == ("idx", "o", True, True, True, 3, 9, True, True, "cfg")
