"""CPU tests of the host-side mirror of the reference interface: record formatting and subread naming
(pinned by the golden `dispatch` / `header` cases captured from the reference's own Python), PSL
parsing, CLI flags."""
import os
import types

import pytest

from c3poa_amd import records, preprocess, synth


def test_consensus_header_matches_reference(golden):
    cases, _ = golden
    for c in cases["header"]:
        assert records.consensus_header("nm", c["qual"], c["L"], c["repeats"], c["cons_len"]) == c["header"]


def test_subread_naming_matches_reference(golden):
    cases, _ = golden
    for c in cases["dispatch"]:
        nsub, ndang = c["nsub"], c["ndang"]
        if nsub == 0:
            continue
        subs = ["ACGTACGTAC" * 3 + "ACGT"[i % 4] for i in range(nsub)]
        sq = ["I" * len(s) for s in subs]
        dang = ["TTTTGGGGCC" + "A" * j for j in range(ndang)]
        dq = ["5" * len(s) for s in dang]
        assert records.subread_records("rd", subs, sq, dang, dq) == c["subreads_fastq"]


def test_zero_repeat_records():
    txt = records.zero_repeat_records("rd", ["AAA", "CC"], ["III", "55"])
    assert txt == "@rd_0\nAAA\n+\nIII\n@rd_1\nCC\n+\n55\n"


def test_psl_parse_and_reuse(tmp_path):
    recs = list(synth.generate("cfg1", n_reads=6))
    tmp = str(tmp_path) + "/"
    synth.write_psl(tmp + "splint_to_read_alignments.psl", recs)
    # a weak hit that must be ignored (matches <= 50) and a second, better hit for read 0
    with open(tmp + "splint_to_read_alignments.psl", "a") as fh:
        fh.write("\t".join(["40", "0", "0", "0", "0", "0", "0", "0", "+", recs[1][0], "5000", "0", "284", "Other", "284", "0", "284", "1", "284,", "0,", "0,"]) + "\n")
        fh.write("\t".join(["283", "0", "0", "0", "0", "0", "0", "0", "-", recs[0][0], "5000", "0", "284", "Best", "284", "0", "284", "1", "284,", "0,", "0,"]) + "\n")
    d = {r[0]: [[None, 1, None]] for r in recs}
    d["nosplint"] = [[None, 1, None]]
    args = types.SimpleNamespace(reads=None, lencutoff=1000, splint_file=None)
    ad, aset, nos = preprocess.preprocess("blat-not-installed", args, tmp, d, len(d))
    assert nos == 1 and "nosplint" not in ad
    assert ad[recs[0][0]] == ["Best", "-"]
    assert ad[recs[1][0]] == ["Splint1", recs[1][3]]
    assert aset == {"Splint1", "Best"}


def test_missing_psl_without_blat_fails_loudly(tmp_path):
    args = types.SimpleNamespace(reads=None, lencutoff=1000, splint_file=None, splint_finder="blat")
    with pytest.raises(RuntimeError):
        preprocess.preprocess("definitely-not-a-binary", args, str(tmp_path) + "/", {}, 0)


def test_gpu_finder_psl_rows_pass_the_reference_filter(tmp_path):
    """rows written by the GPU splint finder are 21-column PSL and survive matches > 50 / qBaseInsert < 50"""
    assert preprocess.equivalent_matches(5 * 51 * 52 // 2) == 51 and preprocess.equivalent_matches(5 * 51 * 52 // 2 - 1) == 50
    assert preprocess.equivalent_matches(0) == 0
    row = preprocess.psl_row("r1", 5000, "Splint1", 284, "-", 120000, 430)
    cols = row.split("\t")
    assert len(cols) == 21 and cols[8] == "-" and cols[9] == "r1" and cols[13] == "Splint1"
    psl = tmp_path / "a.psl"
    psl.write_text(row + "\n" + preprocess.psl_row("r2", 900, "Splint1", 284, "+", 6000, 10) + "\n")
    ad, aset, none = preprocess.parse_psl(str(psl), {"r1": [[None, 1, None]], "r2": [[None, 1, None]]})
    assert ad == {"r1": ["Splint1", "-"]} and none == 1          # r2: equivalent matches 48 <= 50


def test_cli_flags_and_defaults():
    import C3POa
    a = C3POa.parse_args(["-r", "x.fq", "-s", "s.fa"])
    assert (a.lencutoff, a.mdistcutoff, a.numThreads, a.groupSize, a.zero, a.compress_output, a.blatThreads) == \
        (1000, 500, 1, 1000, True, False, False)
    a = C3POa.parse_args(["-r", "x", "-s", "s", "-z", "-co", "-b", "-l", "5", "-d", "1500", "-n", "8", "-g", "77", "-o", "out", "-c", "cfg"])
    assert (a.zero, a.compress_output, a.blatThreads, a.lencutoff, a.mdistcutoff, a.numThreads, a.groupSize, a.out_path, a.config) == \
        (False, True, True, 5, 1500, 8, 77, "out", "cfg")
    with pytest.raises(SystemExit):
        C3POa.parse_args(["-q", "5"])           # documented in the README but not a real flag


def test_config_reader(tmp_path, capsys):
    import C3POa
    p = tmp_path / "cfg"
    p.write_text("# comment\n\nracon\t/opt/racon\n")
    progs = C3POa.configReader(str(tmp_path), str(p))
    assert progs == {"racon": "/opt/racon", "blat": "blat"}
    assert "Using blat from your path" in capsys.readouterr().err


def test_fastx_reader_and_revcomp(tmp_path):
    from c3poa_amd.seqio import fastx_read, revcomp
    p = tmp_path / "r.fastq"
    p.write_text("@a comment\nACGT\n+\nIIII\n@b\nGG\n+\n55\n")
    assert list(fastx_read(str(p))) == [("a", "ACGT", "IIII"), ("b", "GG", "55")]
    f = tmp_path / "s.fasta"
    f.write_text(">s1 desc\nAC\nGT\n>s2\nTTT\n")
    assert list(fastx_read(str(f))) == [("s1", "ACGT", None), ("s2", "TTT", None)]
    assert revcomp("AACGTN") == "NACGTT"
