"""CPU tests of the native host I/O either side of the path (c3_reader_*, c3_write_group).
The reader is checked against the Python mirror of mm.fastx_read, the writer against the Python record formatter
whose formats are pinned by the golden `header` / `dispatch` cases captured from the reference."""
import gzip
import os
import types

import numpy as np
import pytest

from c3poa_amd import _lib, analyze, synth
from c3poa_amd.seqio import fastx_read


def _write_fastq(path, recs, opener=open, mode="w"):
    with opener(path, mode) as fh:
        for r in recs:
            fh.write("@%s extra words\n%s\n+\n%s\n" % (r[0], r[1], r[2]))


def _all(path, group, min_len=0, max_bases=0, sets=2):
    rd = _lib.Reader(path, n_sets=sets)
    out, short, sizes = [], 0, []
    while True:
        hb = rd.next(group, min_len, max_bases)
        short += hb.n_short
        if hb.n == 0:
            break
        sizes.append(hb.n)
        out += [hb.read(i) for i in range(hb.n)]
    rd.close()
    return out, short, sizes


def test_reader_matches_python_fastx(tmp_path):
    recs = [(r[0], r[1], r[2]) for r in synth.generate("cfg1", n_reads=23)]
    p = str(tmp_path / "a.fastq")
    _write_fastq(p, recs)
    got, short, sizes = _all(p, 10)
    assert got == list(fastx_read(p)) == recs and short == 0 and sizes == [10, 10, 3]
    pz = str(tmp_path / "a.fastq.gz")
    _write_fastq(pz, recs, gzip.open, "wt")
    assert _all(pz, 7)[0] == recs
    # length cut-off: skipped reads are counted, groups still hold `group` kept reads (C3POa.py:202-204,240-241)
    lens = sorted(len(r[1]) for r in recs)
    cut = lens[5]
    got, short, sizes = _all(p, 6, min_len=cut)
    assert short == 5 and got == [r for r in recs if len(r[1]) >= cut] and sizes == [6, 6, 6]
    # max_bases closes a group early
    got, _s, sizes = _all(p, 1000, max_bases=3 * lens[-1])
    assert got == recs and len(sizes) > 3


def test_reader_formats_and_edge_cases(tmp_path):
    p = tmp_path / "m.fa"
    p.write_text(">s1 desc\tmore\nAC\nGT\n\n>s2\nTTT\r\n>s3\n>s4\tx\nGG")            # multi-line, CRLF, empty record, no final newline
    got, _s, _z = _all(str(p), 10)
    assert got == [("s1", "ACGT", "!!!!"), ("s2", "TTT", "!!!"), ("s3", "", ""), ("s4", "GG", "!!")]
    assert [(n, s) for n, s, _q in got] == [(n, s) for n, s, _q in fastx_read(str(p))]
    q = tmp_path / "m.fq"
    q.write_text("@a\nACGT\nAC\n+a\n@III\nI@\n@b c\nGG\n+\n@@\n")                      # multi-line FASTQ, '@' inside qualities
    assert _all(str(q), 10)[0] == [("a", "ACGTAC", "@IIII@"), ("b", "GG", "@@")]
    e = tmp_path / "empty.fq"
    e.write_text("")
    assert _all(str(e), 10) == ([], 0, [])
    bad = tmp_path / "bad.fq"
    bad.write_text("@a\nACGT\n+\nII\n")
    with pytest.raises(ValueError):
        _all(str(bad), 10)
    bad.write_text("ACGT\n")
    with pytest.raises(ValueError):
        _all(str(bad), 10)
    with pytest.raises(OSError):
        _lib.Reader(str(tmp_path / "missing.fq"))


def test_reader_long_lines_cross_buffer_refills(tmp_path):
    rng = np.random.default_rng(3)
    recs = []
    for i in range(5):
        L = int(rng.integers(5_000_000, 9_000_000))                                   # > 16 MiB of lines in total, one > buffer/2
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].tobytes().decode()
        recs.append(("long%d" % i, s, "5" * L))
    p = str(tmp_path / "long.fastq")
    _write_fastq(p, recs)
    got, _s, _z = _all(p, 2, sets=1)
    assert [(n, len(s), len(q)) for n, s, q in got] == [(n, len(s), len(q)) for n, s, q in recs]
    assert all(g[1] == r[1] for g, r in zip(got, recs))


def _fake_results(rng, hb):
    """result records covering every branch of the writer"""
    res = np.zeros(hb.n, dtype=_lib.RESULT_DTYPE)
    cons = []
    for i in range(hb.n):
        L = int(hb.off[i + 1] - hb.off[i])
        kind = i % 9
        r = res[i]
        r["status"] = [_lib.ST_OK, _lib.ST_OK, _lib.ST_NO_PEAKS, _lib.ST_NOT_ASSIGNED, _lib.ST_NO_CONSENSUS,
                       _lib.ST_LIMIT, _lib.ST_OK, _lib.ST_NO_CONSENSUS, _lib.ST_TOO_SHORT][kind]
        ns = [3, 1, 0, 0, 2, 2, 0, 0, 0][kind]
        cuts = np.sort(rng.choice(np.arange(10, L - 10), size=ns + 1, replace=False))
        r["n_sub"] = ns
        for k in range(ns):
            r["sub_beg"][k], r["sub_end"][k] = cuts[k], cuts[k + 1]
        front, tail = [(1, 1), (0, 1), (0, 0), (1, 1), (1, 0), (1, 1), (1, 1), (1, 1), (0, 0)][kind]
        r["has_front"], r["has_tail"] = front, tail
        r["front_end"], r["tail_beg"] = cuts[0], cuts[-1]
        if kind in (6, 7) and ns == 0:
            r["front_end"], r["tail_beg"] = L // 3, 2 * L // 3
        c = "".join("ACGT"[j] for j in rng.integers(0, 4, int(rng.integers(50, 400)))) if r["status"] == _lib.ST_OK else ""
        cons.append(c)
    return res, cons


@pytest.mark.parametrize("zero", [True, False])
def test_writer_matches_python_records(tmp_path, zero):
    rng = np.random.default_rng(11)
    recs = [(r[0], r[1], r[2]) for r in synth.generate("cfg1", n_reads=45)]
    recs[7] = (recs[7][0], recs[7][1], "I" * len(recs[7][1]))                       # avg qual 40.0
    recs[8] = (recs[8][0], recs[8][1], "!" * len(recs[8][1]))                       # avg qual 0.0
    p = str(tmp_path / "a.fastq")
    _write_fastq(p, recs)
    hb = _lib.Reader(p).next(100)
    res, cons = _fake_results(rng, hb)
    splints = ["SplA", "SplB"]
    sid = np.array([i % 2 if i % 11 else -1 for i in range(hb.n)], dtype=np.int16)    # -1: not in adapter_dict
    res["status"][sid < 0] = _lib.ST_NOT_ASSIGNED
    # native writer
    raw = "".join(cons).encode()
    coff = np.zeros(hb.n + 1, dtype=np.int64)
    np.cumsum([len(c) for c in cons], out=coff[1:])
    cp = [str(tmp_path / ("c_%s.fa" % s)) for s in splints]
    sp = [str(tmp_path / ("s_%s.fq" % s)) for s in splints]
    _lib.write_group(hb, res, np.frombuffer(raw, dtype=np.uint8), coff, sid, cp, sp, zero)
    _lib.write_group(hb, res, np.frombuffer(raw, dtype=np.uint8), coff, sid, cp, sp, zero)      # appends
    # Python formatter (analyze.write_group -> records.py, pinned by the golden header/dispatch cases)
    out = str(tmp_path / "py") + "/"
    args = types.SimpleNamespace(out_path=out, zero=zero)
    ad = {recs[i][0]: [splints[sid[i]], "+"] for i in range(hb.n) if sid[i] >= 0}
    analyze.write_group(args, recs, res, cons, ad, 1)
    analyze.write_group(args, recs, res, cons, ad, 1)
    import os
    for s, c, q in zip(splints, cp, sp):
        exp_c = open(out + s + "/tmp1/R2C2_Consensus.fasta").read()
        exp_s = open(out + s + "/tmp1/subreads.fastq").read() if os.path.exists(out + s + "/tmp1/subreads.fastq") else ""
        assert open(c).read() == exp_c and len(exp_c) > 0
        assert open(q).read() == exp_s and len(exp_s) > 0


def test_avg_qual_text_matches_python_round(tmp_path):
    """header field avg_qual = str(round(sum/len, 2)) (C3POa.py:168): trailing zeros, .5 cases, long reads"""
    from c3poa_amd import records
    rng = np.random.default_rng(2)
    recs = []
    for i, L in enumerate([1, 3, 7, 8, 200, 201, 999, 1000, 4096, 12345]):
        for rep in range(4):
            q = "".join(chr(33 + int(v)) for v in rng.integers(0, 60, L))
            recs.append(("q%d_%d" % (i, rep), "A" * L, q))
    recs.append(("half", "AAAAAAAA", "".join(chr(33 + v) for v in (1, 0, 0, 0, 0, 0, 0, 0))))      # 0.125 -> 0.12
    recs.append(("half2", "AAAAAAAA", "".join(chr(33 + v) for v in (3, 0, 0, 0, 0, 0, 0, 0))))     # 0.375 -> 0.38
    p = str(tmp_path / "q.fastq")
    _write_fastq(p, recs)
    hb = _lib.Reader(p).next(1000)
    res = np.zeros(hb.n, dtype=_lib.RESULT_DTYPE)
    res["n_sub"] = 1
    res["sub_end"][:, 0] = 1
    coff = np.arange(hb.n + 1, dtype=np.int64)
    cp, sp = [str(tmp_path / "c.fa")], [str(tmp_path / "s.fq")]
    _lib.write_group(hb, res, np.frombuffer(b"A" * hb.n, dtype=np.uint8), coff, np.zeros(hb.n, dtype=np.int16), cp, sp)
    got = [l.strip() for l in open(cp[0]) if l.startswith(">")]
    assert got == [records.consensus_header(n, q, len(s), 1, 1) for n, s, q in recs]


def test_writer_threads_give_identical_files(tmp_path, monkeypatch):
    """groups of >= 4096 reads are formatted / written by several threads: same bytes, same record order"""
    rng = np.random.default_rng(4)
    recs = []
    for i in range(5000):
        L = int(rng.integers(60, 200))
        recs.append(("w%05d" % i, "".join("ACGT"[j] for j in rng.integers(0, 4, L)), "".join(chr(33 + int(v)) for v in rng.integers(2, 40, L))))
    p = str(tmp_path / "w.fastq")
    _write_fastq(p, recs)
    hb = _lib.Reader(p).next(10000)
    res, cons = _fake_results(rng, hb)
    raw = np.frombuffer("".join(cons).encode(), dtype=np.uint8)
    coff = np.zeros(hb.n + 1, dtype=np.int64)
    np.cumsum([len(c) for c in cons], out=coff[1:])
    sid = (np.arange(hb.n) % 3).astype(np.int16)
    outs = {}
    for T in ("1", "5"):
        monkeypatch.setenv("C3_WRITER_THREADS", T)
        cp = [str(tmp_path / ("c%s_%d.fa" % (T, s))) for s in range(3)]
        sp = [str(tmp_path / ("s%s_%d.fq" % (T, s))) for s in range(3)]
        for _rep in range(2):                                             # second call appends after the first
            _lib.write_group(hb, res, raw, coff, sid, cp, sp, True)
        outs[T] = [open(x).read() if os.path.exists(x) else "" for x in cp + sp]
    assert outs["1"] == outs["5"] and sum(1 for x in outs["1"] if x) >= 4
    args = types.SimpleNamespace(out_path=str(tmp_path / "py") + "/", zero=True)
    ad = {recs[i][0]: ["S%d" % sid[i], "+"] for i in range(hb.n)}
    analyze.write_group(args, recs, res, cons, ad, 1)
    analyze.write_group(args, recs, res, cons, ad, 1)
    assert outs["5"][3] == open(args.out_path + "S0/tmp1/subreads.fastq").read()
    assert outs["5"][1] == open(args.out_path + "S1/tmp1/R2C2_Consensus.fasta").read()


def test_native_assignment_matches_python_psl_parse(tmp_path):
    """c3_assign_* == the Python mirror of bin/preprocess.py:22-45 (filters, best-by-matches, first on ties, adapter_set)"""
    from c3poa_amd import preprocess
    rng = np.random.default_rng(6)
    names = ["rd%04d" % i for i in range(400)]
    recs = [(n, "ACGTACGTAC", "IIIIIIIIII") for n in names]
    fq = str(tmp_path / "r.fastq")
    _write_fastq(fq, recs)
    splints = ["SplB", "SplA", "SplC"]                               # CLI order = sorted(splint_dict)
    rows = []
    for i, n in enumerate(names):
        for _k in range(int(rng.integers(0, 4))):
            m = int(rng.choice([30, 50, 51, 120, 120, 200, 283]))
            g = int(rng.choice([0, 10, 49, 50, 80]))
            sp = str(rng.choice(["SplA", "SplB", "SplC", "NotInFile"]))
            st = "+-"[int(rng.integers(0, 2))]
            rows.append("\t".join([str(m), "0", "0", "0", "0", str(g), "0", "0", st, n, "5000", "0", "284", sp, "284", "0", "284", "1", "284,", "0,", "0,"]))
    psl = str(tmp_path / "a.psl")
    open(psl, "w").write("\n".join(rows) + "\n\n")
    # python mirror (rows naming a splint outside the file are dropped first: the native table ignores them)
    psl_py = str(tmp_path / "b.psl")
    open(psl_py, "w").write("\n".join(r for r in rows if "NotInFile" not in r) + "\n")
    ad, aset, no_splint = preprocess.parse_psl(psl_py, {n: [[None, 1, None]] for n in names})
    asg = _lib.Assigner(psl, sorted(splints))
    hb = _lib.Reader(fq).next(1000)
    sid, st, k = asg.batch(hb)
    order = sorted(splints)
    got = {names[i]: [order[sid[i]], chr(st[i])] for i in range(hb.n) if sid[i] >= 0}
    assert got == ad and k == len(ad) == len(names) - no_splint and 0 < k < len(names)
    seen, kept = asg.seen()
    assert seen == aset and kept == sum(1 for r in rows if "NotInFile" not in r and float(r.split("\t")[0]) > 50 and float(r.split("\t")[5]) < 50)
    asg.close()


def test_native_splint_psl_rows_match_python_statement(tmp_path):
    """c3_write_splint_psl writes, for every assigned read, exactly the row psl_row() states (perfect squares included)"""
    from c3poa_amd import preprocess
    rng = np.random.default_rng(13)
    n = 300
    recs = [("read_%d" % i, "ACGT" * int(rng.integers(3, 40)), None) for i in range(n)]
    recs = [(a, b, "I" * len(b)) for a, b, _ in recs]
    fq = str(tmp_path / "r.fastq")
    _write_fastq(fq, recs)
    hb = _lib.Reader(fq).next(1000)
    splints = [("SpA", 284), ("SpB", 97), ("SpC", 333)]
    tab = np.zeros((n, 3, 2, 4), dtype=np.int32)
    tab[..., 0] = rng.integers(0, 200000, (n, 3, 2))
    tab[..., 1] = rng.integers(-5, 200, (n, 3, 2))
    for k, m in enumerate((50, 51, 52, 100, 283, 284, 285, 400)):         # scores that are exactly match*m*(m+1)/2
        tab[k, :, :, 0] = 5 * m * (m + 1) // 2
    sid = rng.integers(-1, 3, n).astype(np.int16)
    st = bytes(rng.choice([43, 45], n).astype(np.uint8))
    psl = str(tmp_path / "f.psl")
    rows = _lib.write_splint_psl(hb, tab, sid, st, [s[0] for s in splints], [s[1] for s in splints], 5, psl)
    rows += _lib.write_splint_psl(hb, tab, sid, st, [s[0] for s in splints], [s[1] for s in splints], 5, psl)      # appends
    want = []
    for i in range(n):
        if sid[i] < 0:
            continue
        e = tab[i, sid[i], 1 if st[i] == 45 else 0]
        want.append(preprocess.psl_row(recs[i][0], len(recs[i][1]), splints[sid[i]][0], splints[sid[i]][1], chr(st[i]), e[0], e[1], 5))
    assert open(psl).read().splitlines() == want + want and rows == 2 * len(want) and len(want) > 100


def test_byte_range_readers_tile_the_file(tmp_path):
    """c3_reader_open_range: ranges that tile the file read every record exactly once, in order -- also when quality
    lines start with '@' (the FASTQ resync trap), with CRLF, and for FASTA (C3POa.py:236-256 shards reads over workers)"""
    rng = np.random.default_rng(3)
    recs = []
    for i in range(300):
        L = int(rng.integers(30, 400))
        s = "".join("ACGT"[k] for k in rng.integers(0, 4, L))
        q = "".join(chr(33 + int(v)) for v in rng.integers(0, 42, L))
        if i % 3 == 0:
            q = "@" + q[1:]                                  # quality line that looks like a header
        if i % 7 == 0:
            q = "@" + q[1:-1] + "+"
        recs.append(("r%d extra words" % i, s, q))
    for name, text in (("a.fastq", "".join("@%s\n%s\n+\n%s\n" % r for r in recs)),
                       ("b.fastq", "".join("@%s\r\n%s\r\n+\r\n%s\r\n" % r for r in recs)),
                       ("c.fasta", "".join(">%s\n%s\n" % (r[0], r[1]) for r in recs))):
        p = str(tmp_path / name)
        open(p, "w", newline="").write(text)
        size = os.path.getsize(p)
        whole = []
        rd = _lib.Reader(p, n_sets=1)
        while True:
            hb = rd.next(64)
            if hb.n == 0:
                break
            whole += [hb.read(i) for i in range(hb.n)]
        assert [w[0] for w in whole] == ["r%d" % i for i in range(300)]
        for cuts in ([0, size // 3, 2 * size // 3, size], [0, 1, 2, 50, size // 2, size // 2 + 1, size - 1, size],
                     sorted([0, size] + [int(x) for x in rng.integers(0, size, 9)])):
            got = []
            for b, e in zip(cuts, cuts[1:]):
                rr = _lib.Reader(p, n_sets=1, byte_range=(b, e))
                while True:
                    hb = rr.next(17)
                    if hb.n == 0:
                        break
                    got += [hb.read(i) for i in range(hb.n)]
                rr.close()
            assert got == whole, (name, cuts)


def test_tiny_input_does_not_pin_gigabytes(tmp_path):
    """the reader's (page-locked) buffers are sized by what the file can deliver, not by max_reads: a 100-read input asked
    for 131072-read batches with five buffer sets used to reserve ten buffers of that size"""
    recs = [(r[0], r[1], r[2]) for r in synth.generate("cfg1", n_reads=100)]
    p = str(tmp_path / "tiny.fastq")
    _write_fastq(p, recs)
    rd = _lib.Reader(p, n_sets=5)
    hb = rd.next(131072, 0, 1 << 30, set_index=0)
    assert hb.n == 100
    assert rd.reserved_bytes() < 8 * os.path.getsize(p) + (4 << 20), rd.reserved_bytes()
    assert [hb.read(i) for i in range(3)] == recs[:3]
    # a byte-range reader is bounded by its range, not by the file
    size = os.path.getsize(p)
    rr = _lib.Reader(p, n_sets=5, byte_range=(size // 2, size))
    hb2 = rr.next(131072, 0, 1 << 30, set_index=0)
    assert 0 < hb2.n < 100 and rr.reserved_bytes() < 6 * size + (4 << 20)
    rd.close(); rr.close()


def test_sets_taken_for_the_tail_of_an_input_are_sized_by_what_is_left(tmp_path):
    """round 5: a buffer set that is first used late in a file / byte range is sized by the bytes the reader can still deliver, not by
    the largest group so far -- in particular a set taken when NOTHING is left (the call that finds the end of the input) stays empty.
    Eight workers over a 20 GB input used to page-lock 50 GB this way: every range walked through its five sets at full batch size
    although two groups emptied it.  (Measured with the round-4 library on this very input: 3.06 x the file size reserved; now 1.5 x.)"""
    base = [(r[0], r[1], r[2]) for r in synth.generate("cfg1", n_reads=400)]
    recs = [("%s_%d" % (n, c), sq, q) for c in range(15) for n, sq, q in base]
    p = str(tmp_path / "m.fastq")
    _write_fastq(p, recs)
    size = os.path.getsize(p)
    rd = _lib.Reader(p, n_sets=5)
    got, per_set, ns = [], [], []
    for k in range(5):                                    # a first-in-first-out free list: every call takes a set nobody has used yet
        hb = rd.next(2500, 0, 1 << 30, set_index=k)
        got += [hb.read(i) for i in range(hb.n)]
        per_set.append(rd.reserved_bytes()); ns.append(hb.n)
    assert ns == [2500, 2500, 1000, 0, 0] and got == recs
    assert per_set[-1] < 1.8 * size, (per_set, size)
    assert per_set[4] - per_set[2] < (8 << 20), per_set                      # the two calls that found nothing reserved next to nothing
    rd.close()


def test_writer_appends_at_the_real_end_after_a_truncate(tmp_path):
    """reservations are keyed by the file and live only while a writer is in flight: truncating (or replacing) an output file
    between two c3_write_group calls -- without c3_writer_reset -- must not leave a NUL-filled hole"""
    rng = np.random.default_rng(4)
    recs = [(r[0], r[1], r[2]) for r in synth.generate("cfg1", n_reads=20)]
    p = str(tmp_path / "a.fastq")
    _write_fastq(p, recs)
    hb = _lib.Reader(p).next(100)
    res, cons = _fake_results(rng, hb)
    sid = np.zeros(hb.n, dtype=np.int16)
    raw = "".join(cons).encode()
    coff = np.zeros(hb.n + 1, dtype=np.int64)
    np.cumsum([len(c) for c in cons], out=coff[1:])
    cp, sp = [str(tmp_path / "c.fa")], [str(tmp_path / "s.fq")]
    link = str(tmp_path / "alias.fa")
    os.symlink(cp[0], link)                                               # a second spelling of the same file
    _lib.write_group(hb, res, np.frombuffer(raw, dtype=np.uint8), coff, sid, cp, sp, True)
    once_c, once_s = open(cp[0], "rb").read(), open(sp[0], "rb").read()
    assert once_c and once_s and b"\0" not in once_c
    open(cp[0], "w").close()                                              # truncate one file, replace the other
    os.remove(sp[0])
    _lib.write_group(hb, res, np.frombuffer(raw, dtype=np.uint8), coff, sid, [link], sp, True)
    assert open(cp[0], "rb").read() == once_c and open(sp[0], "rb").read() == once_s
    _lib.write_group(hb, res, np.frombuffer(raw, dtype=np.uint8), coff, sid, cp, sp, True)
    assert open(cp[0], "rb").read() == once_c * 2 and open(sp[0], "rb").read() == once_s * 2


def test_multi_line_fastq_ranges_are_reported_and_last_record_without_newline(tmp_path):
    """mm.fastx_read accepts multi-line FASTQ; a byte-range reader cannot resync inside one and must SAY so instead of returning
    an empty range (the records would be dropped silently).  The file's last record may lack its trailing newline."""
    rng = np.random.default_rng(8)
    recs = []
    for i in range(60):
        L = int(rng.integers(130, 300))
        recs.append(("m%d" % i, "".join("ACGT"[k] for k in rng.integers(0, 4, L)), "".join(chr(40 + int(v)) for v in rng.integers(0, 30, L))))
    wrap = lambda s: "\n".join(s[k:k + 60] for k in range(0, len(s), 60))      # noqa: E731
    p = str(tmp_path / "multi.fastq")
    open(p, "w").write("".join("@%s\n%s\n+\n%s\n" % (n, wrap(s), wrap(q)) for n, s, q in recs))
    size = os.path.getsize(p)
    whole = _lib.Reader(p, n_sets=1)
    hb = whole.next(1000)
    assert [hb.read(i) for i in range(hb.n)] == recs                         # the whole-file reader takes multi-line records
    mid = _lib.Reader(p, n_sets=1, byte_range=(size // 2, size))
    assert mid.range_lost()                                                  # ... a range reader reports that it cannot
    # 4-line FASTQ whose last record has no trailing newline: the range that starts at that record still finds it
    p2 = str(tmp_path / "nonl.fastq")
    text = "".join("@%s\n%s\n+\n%s\n" % r for r in recs)
    open(p2, "w").write(text[:-1])
    last_start = text.rindex("@m59")
    rr = _lib.Reader(p2, n_sets=1, byte_range=(last_start, -1))
    hb2 = rr.next(10)
    assert not rr.range_lost() and hb2.n == 1 and hb2.read(0) == recs[-1]
    rr2 = _lib.Reader(p2, n_sets=1, byte_range=(last_start - 5, -1))
    assert rr2.next(10).n == 1
    assert _lib.Reader(p2, n_sets=1).noqual() == 0
    pf = str(tmp_path / "x.fasta")
    open(pf, "w").write(">a\nACGT\n>b\nGGCC\n")
    rf = _lib.Reader(pf, n_sets=1)
    assert rf.next(10).n == 2 and rf.noqual() == 2


def _bgzf_write(path, data, block=0xff00, extra_first=False):
    """a BGZF file (bgzip / htslib layout): independent gzip members of <= 64 KiB, each with the 'BC' extra subfield holding the
    member's size - 1, and the empty end-of-file member"""
    import struct
    import zlib
    with open(path, "wb") as f:
        for i in range(0, len(data), block):
            chunk = data[i:i + block]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            xtra = (b"XY" + struct.pack("<H", 3) + b"abc") if extra_first else b""
            xlen = 6 + len(xtra)
            total = 12 + xlen + len(comp) + 8
            f.write(struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, xlen) + xtra + b"BC" + struct.pack("<HH", 2, total - 1)
                    + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def test_bgzf_input_is_inflated_in_parallel_and_equals_the_plain_file(tmp_path, monkeypatch):
    """bgzip-compressed FASTQ (mm.fastx_read reads any gzip, C3POa.py:201,239): the native reader locates the members by their
    headers and inflates them on several threads; records that straddle members, more members than one stretch holds (512),
    a foreign extra subfield in front of 'BC', one thread or five -- always the records of the plain file; a damaged member is an
    error, not a short file; ordinary gzip keeps the single-stream path"""
    import gzip as _gz
    recs = [(r[0], r[1], r[2]) for r in synth.generate("cfg1", n_reads=23)]
    plain = str(tmp_path / "a.fastq")
    _write_fastq(plain, recs)
    data = open(plain, "rb").read()
    for block, extra, threads in ((0xff00, False, "5"), (301, False, "5"), (301, False, "1"), (4096, True, "3")):
        pz = str(tmp_path / ("b%d_%d.fastq.gz" % (block, extra)))
        _bgzf_write(pz, data, block, extra)
        assert _gz.open(pz, "rb").read() == data                         # (a valid multi-member gzip file for everybody else)
        monkeypatch.setenv("C3_GZ_THREADS", threads)
        assert _all(pz, 7)[0] == recs, (block, extra, threads)
        assert len(data) // block + 1 > 512 or block != 301              # the small blocks really span several stretches
    monkeypatch.setenv("C3_NO_BGZF", "1")                                # the same file through zlib's stream reader
    assert _all(str(tmp_path / "b301_0.fastq.gz"), 7)[0] == recs
    monkeypatch.delenv("C3_NO_BGZF")
    # a flipped byte in the middle of a member: CRC / inflate error -> the reader fails loudly
    bad = bytearray(open(str(tmp_path / "b65280_0.fastq.gz"), "rb").read())
    bad[len(bad) // 2] ^= 0x55
    pb = str(tmp_path / "bad.fastq.gz")
    open(pb, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        _all(pb, 7)
    # a member whose ISIZE field was zeroed (its data would vanish without a trace if "size 0" meant "skip"): an error as well
    raw = open(str(tmp_path / "b65280_0.fastq.gz"), "rb").read()
    import struct
    first = struct.unpack("<H", raw[raw.index(b"BC\x02\x00") + 4:][:2])[0] + 1            # size of the first member
    pz0 = str(tmp_path / "isize0.fastq.gz")
    open(pz0, "wb").write(raw[:first - 4] + b"\0\0\0\0" + raw[first:])
    with pytest.raises(ValueError):
        _all(pz0, 7)
    # ... and so is a damaged ordinary gzip stream (zlib's reader reports it; it used to end the input as if the file were shorter)
    pg = str(tmp_path / "plain_bad.fastq.gz")
    zb = bytearray(_gz.compress(data))
    zb[len(zb) // 2] ^= 0x55
    open(pg, "wb").write(bytes(zb))
    with pytest.raises(ValueError):
        _all(pg, 7)
    # names-only pass and the length cut-off behave as on plain input
    got, short, _z = _all(str(tmp_path / "b301_0.fastq.gz"), 6, min_len=sorted(len(r[1]) for r in recs)[5])
    assert short == 5 and len(got) == 18


def test_compress_file_is_gzip_for_everybody_and_bgzf_for_this_reader(tmp_path, monkeypatch):
    """-co (C3POa.py:46-47,86-99: the final files are written through gzip.open): c3_compress_file deflates independent members on
    several threads.  The result must be what any gzip reader expects (Python's gzip gives the bytes back), carry the BGZF size
    subfield in every member (this package's reader then inflates it in parallel -- and yields the same records), end with the empty
    end-of-file member, work for an empty and for a multi-stretch input, with one thread or many, and leave no partial file behind on
    an error"""
    import gzip as _gz
    recs = [(r[0], r[1], r[2]) for r in synth.generate("cfg1", n_reads=300)]
    src = str(tmp_path / "sub.fastq")
    _write_fastq(src, recs)
    data = open(src, "rb").read()
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    for threads, level in ((1, 1), (3, 6), (0, 9)):
        dst = str(tmp_path / ("o%d.gz" % threads))
        assert _lib.compress_file(src, dst, level=level, threads=threads, remove=False) == dst
        z = open(dst, "rb").read()
        assert _gz.decompress(z) == data and z.endswith(eof)
        assert z[:4] == b"\x1f\x8b\x08\x04" and z[12:16] == b"BC\x02\x00"
        nm = 0; at = 0                                                      # walk the members by their size subfields
        while at < len(z):
            at += int.from_bytes(z[at + 16:at + 18], "little") + 1; nm += 1
        assert at == len(z) and nm == (len(data) + 0xff00 - 1) // 0xff00 + 1
        monkeypatch.setenv("C3_GZ_THREADS", "4")
        assert _all(dst, 64)[0] == recs                                     # the parallel BGZF path of the native reader
    assert len(z) < 0.6 * len(data)
    # empty input -> just the end-of-file member; default name + removal of the source (what -co does)
    e = str(tmp_path / "empty.fasta")
    open(e, "w").close()
    assert _lib.compress_file(e) == e + ".gz" and not os.path.exists(e) and open(e + ".gz", "rb").read() == eof and _gz.decompress(eof) == b""
    # an input of several stretches (threads x 64 members each)
    big = str(tmp_path / "big.txt")
    blob = (data * 4)[:9 * 1024 * 1024 + 12345]
    open(big, "wb").write(blob)
    _lib.compress_file(big, big + ".gz", threads=2, remove=False)
    assert _gz.decompress(open(big + ".gz", "rb").read()) == blob
    # errors: a source that does not exist, a destination that cannot be created
    with pytest.raises(_lib.C3Error):
        _lib.compress_file(str(tmp_path / "nope"), str(tmp_path / "nope.gz"))
    with pytest.raises(_lib.C3Error):
        _lib.compress_file(src, str(tmp_path / "no_such_dir" / "x.gz"), remove=False)
    assert os.path.exists(src)


def test_bgzf_input_in_byte_ranges_equals_one_reader(tmp_path):
    """a BGZF file cut into ranges of its INFLATED bytes (c3_bgzf_size + c3_reader_open_range): every record exactly once, in order,
    for cuts at every kind of place (inside records, on member borders, past the end); a plain gzip file refuses ranges"""
    import gzip
    import random
    rng = random.Random(21)
    recs = [("r%d" % i, "".join(rng.choice("ACGT") for _ in range(rng.randint(30, 9000))), None) for i in range(900)]
    recs = [(n, s, "".join(chr(rng.randint(35, 73)) for _ in s)) for n, s, _ in recs]
    fq = str(tmp_path / "a.fastq")
    with open(fq, "w") as fh:
        for r in recs:
            fh.write("@%s\n%s\n+\n%s\n" % r)
    bg = str(tmp_path / "a_bgzf.fastq.gz")
    _lib.compress_file(fq, bg, level=3, remove=False)
    size = _lib.bgzf_size(bg)
    assert size == os.path.getsize(fq)

    def read_range(path, beg, end):
        rd = _lib.Reader(path, n_sets=2, byte_range=(beg, end))
        out = []
        while True:
            hb = rd.next(300)
            if hb.n == 0:
                break
            out += [hb.read(i) for i in range(hb.n)]
        lost = rd.range_lost()
        rd.close()
        assert not lost
        return out

    for n_ranges in (2, 3, 7, 16):
        cuts = [size * k // n_ranges for k in range(n_ranges)] + [-1]
        got = []
        for k in range(n_ranges):
            got += read_range(bg, cuts[k], cuts[k + 1])
        assert got == recs, n_ranges
    # cuts on member borders (65 280 inflated bytes per member) and at random places
    for trial in range(6):
        cuts = sorted(set([0] + [rng.choice([65280 * rng.randrange(1, size // 65280), rng.randrange(1, size)]) for _ in range(5)])) + [-1]
        got = []
        for k in range(len(cuts) - 1):
            got += read_range(bg, cuts[k], cuts[k + 1])
        assert got == recs, cuts
    assert read_range(bg, size + 5, -1) == []
    # plain gzip: not BGZF, no ranges
    gz = str(tmp_path / "plain.fastq.gz")
    with gzip.open(gz, "wt") as fh:
        for r in recs[:50]:
            fh.write("@%s\n%s\n+\n%s\n" % r)
    assert _lib.bgzf_size(gz) == -1
    with pytest.raises(OSError):
        _lib.Reader(gz, byte_range=(100, -1))
