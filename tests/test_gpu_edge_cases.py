"""GPU edge cases against the oracle: ragged / degenerate reads, other splints (all conk templates),
several splints per batch, extreme qualities, capacity limits, per-stage runs."""
import os
import numpy as np
import pytest

from c3poa_amd import synth
from c3poa_amd.seqio import revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle_py
    return oracle_py


def _rand(rng, L):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, L))


def _qual(rng, L, lo=2, hi=41):
    return "".join(chr(33 + int(v)) for v in rng.integers(lo, hi, L))


def _concatemer(rng, splint, insert_len, n, k0, k1, err=0.1):
    ins = _rand(rng, insert_len)
    clean = ins[len(ins) - k0:] + (splint + ins) * n + splint + ins[:k1]
    s, q = synth._mutate(rng, np.frombuffer(clean.encode(), dtype=np.uint8), sub=err * 0.4, ins=err * 0.25, dele=err * 0.35)
    return s.decode(), q.decode()


def _compare(O, splints, reads, strands, sids, **cfg):
    """full pipeline on the GPU vs the oracle, read by read (the oracle takes one splint per call)"""
    from c3poa_amd import _lib
    h = _lib.Handle(**cfg)
    h.set_splints(splints)
    h.upload([r[0] for r in reads], [r[1] for r in reads], strands, np.array(sids, dtype=np.int16))
    h.run()
    res, cons = h.results()
    P = O.default_params(**{k: v for k, v in cfg.items() if k in ("mdistcutoff",)})
    for i, (r, st, sid) in enumerate(zip(reads, strands, sids)):
        ores, ocons = O.process_batch(splints[sid], [r], [st], params=P, threads=1)
        assert res[i]["status"] == ores[0].status, (i, int(res[i]["status"]), ores[0].status)
        assert cons[i] == ocons[0], i
        if ores[0].status in (0, 3):
            assert res[i]["n_sub"] == ores[0].n_sub and res[i]["n_peaks"] == ores[0].n_peaks
    h.close()
    return res, cons


def test_degenerate_reads(O):
    rng = np.random.default_rng(11)
    sp = synth.SPLINT1
    reads, strands = [], []
    reads.append((_rand(rng, 3000), _qual(rng, 3000))); strands.append("+")             # no splint at all
    s = _rand(rng, 2500)
    reads.append((s[:1200] + sp + s[1200:], _qual(rng, 2500 + len(sp)))); strands.append("+")   # one splint: 0 repeats
    reads.append((_rand(rng, 15), _qual(rng, 15))); strands.append("+")                   # too short
    reads.append((_rand(rng, 41), _qual(rng, 41))); strands.append("-")
    reads.append(_concatemer(rng, sp, 300, 2, 50, 60)); strands.append("+")              # 2 short subreads, dangling < 100
    reads.append(_concatemer(rng, sp, 700, 1, 150, 150)); strands.append("+")            # 1 subread + 2 dangling
    reads.append(_concatemer(rng, sp, 700, 1, 20, 20)); strands.append("+")              # 1 subread, no dangling -> nothing polished
    reads.append(_concatemer(rng, sp, 2500, 5, 400, 900)); strands.append("+")           # long inserts
    s, q = _concatemer(rng, sp, 900, 4, 120, 120)
    reads.append((revcomp(s), q[::-1])); strands.append("-")
    s2 = list(reads[4][0]); s2[5] = "N"; s2[100] = "n"; s2[-1] = "R"
    reads.append(("".join(s2).lower(), reads[4][1])); strands.append("+")                # lower case + non-ACGT
    reads.append(_concatemer(rng, sp, 1000, 3, 150, 150)); strands.append("?")           # not assigned
    res, cons = _compare(O, [sp], reads, strands, [0] * len(reads))
    assert [int(x) for x in res["status"][:4]] == [2, 3, 4, 2]
    assert res["status"][10] == 1 and cons[10] == ""


def test_many_repeats_and_long_read(O):
    rng = np.random.default_rng(12)
    sp = synth.SPLINT1
    reads = [_concatemer(rng, sp, 250, 30, 120, 120), _concatemer(rng, sp, 1200, 24, 500, 500, err=0.12)]
    res, cons = _compare(O, [sp], reads, ["+", "+"], [0, 0], mdistcutoff=400)
    assert res[0]["n_sub"] >= 25 and res[1]["n_sub"] >= 20 and all(cons)


@pytest.mark.parametrize("S", [40, 64, 65, 150, 284, 330, 400, 512])
def test_other_splint_lengths_cover_all_conk_templates(O, S):
    rng = np.random.default_rng(S)
    sp = _rand(rng, S)
    reads = [_concatemer(rng, sp, 900, 3, 150, 150), (_rand(rng, 2000), _qual(rng, 2000))]
    from c3poa_amd import _lib
    h = _lib.Handle()
    h.set_splints([sp])
    h.upload([r[0] for r in reads], [r[1] for r in reads], ["+", "-"])
    h.run(1)
    for i, st in enumerate("+-"):
        assert np.array_equal(h.track(i), O.conk(sp if st == "+" else revcomp(sp), reads[i][0]))
    h.close()
    _compare(O, [sp], reads[:1], ["+"], [0])


def test_several_splints_in_one_batch(O):
    rng = np.random.default_rng(13)
    sps = [synth.SPLINT1, _rand(rng, 120), _rand(rng, 500)]
    reads, strands, sids = [], [], []
    for k in range(9):
        sid = k % 3
        s, q = _concatemer(rng, sps[sid], 800 + 100 * k, 3, 130, 130)
        if k % 2:
            s, q, st = revcomp(s), q[::-1], "-"
        else:
            st = "+"
        reads.append((s, q)); strands.append(st); sids.append(sid)
    _compare(O, sps, reads, strands, sids)


def test_quality_extremes(O):
    rng = np.random.default_rng(14)
    s, _q = _concatemer(rng, synth.SPLINT1, 600, 4, 150, 150)
    for qual in ("!" * len(s), "~" * len(s), "".join("!~"[i & 1] for i in range(len(s))), "$" * len(s)):
        _compare(O, [synth.SPLINT1], [(s, qual)], ["+"], [0])


def test_mdistcutoff_variants(O):
    recs = list(synth.generate("cfg1", n_reads=3))
    for md in (100, 1000, 1600, 3000):
        _compare(O, [synth.SPLINT1], [(r[1], r[2]) for r in recs], [r[3] for r in recs], [0, 0, 0], mdistcutoff=md)


def test_too_many_subreads_is_a_status_not_a_crash():
    from c3poa_amd import _lib
    rng = np.random.default_rng(15)
    h = _lib.Handle()
    subs = [_rand(rng, 40) for _ in range(251)]
    with pytest.raises(_lib.C3Error):
        h.determine_consensus(subs, ["I" * 40] * 251)
    assert h.determine_consensus(subs[:3], ["I" * 40] * 3) != "" or True
    h.close()


def test_stage_masks_and_state_errors():
    from c3poa_amd import _lib
    rec = next(iter(synth.generate("cfg1", n_reads=1)))
    h = _lib.Handle()
    h.set_splints([synth.SPLINT1])
    h.upload([rec[1]], [rec[2]], [rec[3]])
    with pytest.raises(_lib.C3Error):
        h.track(0)                      # conk stage not run yet
    h.run(_lib.STAGE_CONK)
    h.track(0)
    with pytest.raises(_lib.C3Error):
        h.raw_peaks(0)
    h.run(_lib.STAGE_PEAKS)
    assert len(h.raw_peaks(0)) == 4
    h.run(_lib.STAGE_POA)
    d = h.draft(0)
    h.run(_lib.STAGE_POLISH)
    res, cons = h.results()
    assert res[0]["status"] == 0 and abs(len(cons[0]) - len(d)) < 60
    h.close()


def _zero_read(rng, a, b, err=True, n_ins=1300):
    ins = _rand(rng, n_ins)
    clean = ins[a:] + synth.SPLINT1 + ins[:b]
    if not err:
        return clean, "I" * len(clean), ins
    s, q = synth._mutate(rng, np.frombuffer(clean.encode(), dtype=np.uint8))
    return s.decode(), q.decode(), ins


def test_zero_repeat_rescue(O):
    """bin/determine_consensus.py:106-136: one splint, overlapping dangling pieces -> stitched consensus,
    repeats == 0, never polished; no overlap or -z -> NO_CONSENSUS."""
    from c3poa_amd import _lib
    rng = np.random.default_rng(21)
    reads, truths = [], []
    for a, b in ((686, 1040), (212, 1230), (376, 1006), (260, 424), (300, 983), (700, 600), (100, 1299)):
        s, q, ins = _zero_read(rng, a, b)
        reads.append((s, q)); truths.append(ins)
    s, q, ins = _zero_read(rng, 500, 900, err=False)
    reads.append((s, q)); truths.append(ins)
    reads.append((revcomp(reads[0][0]), reads[0][1][::-1])); truths.append(None)
    strands = ["+"] * 8 + ["-"]
    res, cons = _compare(O, [synth.SPLINT1], reads, strands, [0] * len(reads))
    assert [int(x) for x in res["status"]] == [0, 0, 0, 0, 0, 3, 0, 0, 0]
    assert all(int(r["n_sub"]) == 0 for r in res)
    h2 = len(synth.SPLINT1) // 2
    for i in (0, 1, 2, 4, 7):
        truth = synth.SPLINT1[h2:] + truths[i] + synth.SPLINT1[:h2]
        assert synth.identity(cons[i], truth) > (0.99 if i == 7 else 0.86)
    # -z : rescue disabled on both sides
    h = _lib.Handle(zero=0)
    h.set_splints([synth.SPLINT1])
    h.upload([r[0] for r in reads[:3]], [r[1] for r in reads[:3]], strands[:3])
    h.run()
    r2, c2 = h.results()
    assert [int(x) for x in r2["status"]] == [3, 3, 3] and c2 == ["", "", ""]
    P = O.default_params(zero=0)
    ores, ocons = O.process_batch(synth.SPLINT1, reads[:3], strands[:3], params=P, threads=1)
    assert [x.status for x in ores] == [3, 3, 3]
    h.close()


def test_zero_repeat_long_front_piece(O):
    """front pieces beyond the 4096 columns k_zero keeps in LDS (rows then live in global memory): the only size limit is the
    oracle's 16 M cells (oracle/c3o_zero.c, zr_max_cells), on either side of it the two agree"""
    rng = np.random.default_rng(24)
    reads = []
    for n_ins, a, b in ((6500, 1400, 2600), (6000, 1800, 3500), (7000, 1000, 2900), (4200, 50, 3000), (8200, 20, 1800)):
        s, q, _ = _zero_read(rng, a, b, n_ins=n_ins)
        reads.append((s, q))
    res, cons = _compare(O, [synth.SPLINT1], reads, ["+"] * len(reads), [0] * len(reads))
    assert [int(x) for x in res["status"]] == [0, 0, 3, 0, 0]                       # (6000 x 2900 cells: over the limit)
    assert all(int(r["n_sub"]) == 0 for r in res)
    assert len(cons[0]) > 6000 and len(cons[4]) > 8000


def test_window_beyond_the_first_launch_graph_and_scratch(O):
    """a read of 200+ copies of a short insert whose peaks are only partly called: subreads of several units each, window layers
    of 600+ bases that do not follow the draft -- one window graph grows past the first launch's node arrays and needs DP rows
    wider than its scratch; the full-size launch is sized from the batch (longest layer, sum of the layers) and must hold it
    (found by tools/fuzz_parity2.py seed 4, read 155: LIMIT on the GPU, a consensus in the oracle)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity2", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_parity2.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    splint, mdist, reads, strands = fz.generate(200, 4)
    res, cons = _compare(O, [splint], [reads[155]], [strands[155]], [0], mdistcutoff=mdist)
    assert int(res[0]["status"]) == 0 and len(cons[0]) > 2000


@pytest.mark.parametrize("seed,idx", [(55, [52, 83]), (93, [24, 44, 52, 70, 72]), (111, [2, 45, 48, 52, 53])])
def test_non_default_configurations_found_by_the_fuzzer(O, seed, idx, monkeypatch):
    """tools/fuzz_parity3.py, seeds 55 / 93 / 111 (a random non-default c3_config each): with abPOA match 1-2 against mismatch 8 an
    alignment prefers gaps and nearly every base becomes a node -- the last k_poa pass needs cells for `nodes x subread length`,
    not for the batch's typical band; with polishing windows of 100 bases a window consensus can be longer than 3 windows + 64 --
    the full-size k_window launch has output slots as long as its graph has nodes.  Both ended as LIMIT, the oracle finishes them."""
    import importlib.util
    from c3poa_amd import _lib
    spec = importlib.util.spec_from_file_location("fuzz_parity3", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_parity3.py"))
    fz3 = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz3)
    cfg = fz3.random_config(np.random.default_rng(10_000 + seed))
    splint, _md, reads, strands = fz3.fz.generate(100, 50_000 + seed)
    keep = [i for i, r in enumerate(reads) if len(r[0]) < 40_000]
    reads = [reads[keep[i]] for i in idx]; strands = [strands[keep[i]] for i in idx]
    # device memory poisoned at allocation: seed 93 (100-base windows, layers of 1 200+ bases) wrote past the first window launch's
    # query arrays -- invisible on the zeroed memory of a fresh process
    monkeypatch.setenv("C3_DEBUG_POISON", "1")
    h = _lib.Handle(**cfg)
    h.set_splints([splint])
    h.upload([r[0] for r in reads], [r[1] for r in reads], strands)
    h.run()
    res, cons = h.results()
    ores, ocons = O.process_batch(splint, reads, strands, params=O.default_params(**cfg), threads=8)
    assert [int(x) for x in res["status"]] == [o.status for o in ores] and list(cons) == list(ocons)
    assert sum(1 for o in ores if o.status == 0) >= len(idx) - 2
    h.close()


def test_zero_repeat_mixed_with_normal_reads(O):
    rng = np.random.default_rng(22)
    recs = list(synth.generate("cfg1", n_reads=6))
    reads = [(r[1], r[2]) for r in recs]
    strands = [r[3] for r in recs]
    for a, b in ((400, 800), (650, 700)):
        s, q, _ = _zero_read(rng, a, b)
        reads.insert(2, (s, q)); strands.insert(2, "+")
    res, cons = _compare(O, [synth.SPLINT1], reads, strands, [0] * len(reads))
    assert sum(int(r["n_sub"]) == 0 and int(r["status"]) == 0 for r in res) >= 1


def test_zero_repeats_entry_point_and_shim(O):
    from c3poa_amd import _lib, shims
    import types
    rng = np.random.default_rng(23)
    s, q, _ = _zero_read(rng, 450, 950)
    ores, _c = O.process_batch(synth.SPLINT1, [(s, q)], ["+"], threads=1)
    p = ores[0].front_end
    d0, q0, d1, q1 = s[:p], q[:p], s[p:], q[p:]
    ref = O.zero_repeats(d0, q0, d1, q1)
    h = _lib.Handle()
    assert h.zero_repeats(d0, q0, d1, q1, 500) == ref and len(ref) > 1400
    assert h.zero_repeats(d0, q0, d1, q1, 5000) == ""                    # shorter than the cutoff
    assert h.zero_repeats(_rand(rng, 700), "I" * 700, _rand(rng, 650), "I" * 650, 0) == ""   # no overlap
    h.close()
    args = types.SimpleNamespace(mdistcutoff=500, zero=True)
    assert shims.determine_consensus(args, ("rd", s, q), [], [], [d0, d1], [q0, q1]) == (ref, 0)
    args.zero = False
    assert shims.determine_consensus(args, ("rd", s, q), [], [], [d0, d1], [q0, q1]) == ("", 0)


def test_splint_finder_matches_oracle_and_truth(O):
    """c3_scan_splints (GPU replacement of the blat step) == oracle table bit for bit; assignment == ground truth"""
    from c3poa_amd import _lib
    rng = np.random.default_rng(77)
    splints = [synth.SPLINT1, _rand(rng, 200), _rand(rng, 97)]
    reads, truth = [], []
    for i in range(36):
        s = i % 3
        seq, _q = _concatemer(rng, splints[s], 700 + 40 * i, 2 + i % 3, 60, 60)
        st = "+-"[(i // 3) % 2]
        if st == "-":
            seq = revcomp(seq)
        reads.append(seq); truth.append((s, st))
    reads += [_rand(rng, 3000), _rand(rng, 500), "A" * 300, _rand(rng, 20)]      # nothing to find
    truth += [(-1, "?")] * 4
    h = _lib.Handle()
    h.set_splints(splints)
    h.upload(reads, ["I" * len(r) for r in reads], "?" * len(reads))
    tab, sid, st = h.scan_splints()
    otab, osid, ost = O.scan_splints(reads, splints)
    assert np.array_equal(tab, otab)
    assert np.array_equal(sid, osid) and st == ost
    assert [(int(a), chr(b)) for a, b in zip(sid, st)] == truth
    # a '-' strand read scores on the revcomp row: strand semantics of conk.conk(splint_dict[name][1], seq) (C3POa.py:119-122)
    assert tab[3, 0, 1, 0] > 4 * tab[3, 0, 0, 0]


def test_cli_gpu_splint_finder(tmp_path):
    """no PSL, no blat: the CLI assigns splints on the GPU, writes the PSL and reuses it on the next run"""
    import os
    import C3POa
    from c3poa_amd.seqio import fastx_read
    recs = list(synth.generate("cfg1", n_reads=12))
    rng = np.random.default_rng(5)
    out = str(tmp_path / "out")
    fq = str(tmp_path / "reads.fastq")
    with open(fq, "w") as fh:
        for r in recs:
            fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
        fh.write("@junk\n%s\n+\n%s\n" % (_rand(rng, 2000), "I" * 2000))
    other = _rand(rng, 240)
    others = []
    with open(fq, "a") as fh:                                                       # reads of a second splint, both strands
        for i in range(6):
            s, q = _concatemer(rng, other, 900, 3, 80, 80)
            if i % 2:
                s, q = revcomp(s), q[::-1]
            others.append(("o%02d" % i, s, q, "+-"[i % 2]))
            fh.write("@%s\n%s\n+\n%s\n" % (others[-1][0], s, q))
    fa = str(tmp_path / "splint.fasta")
    open(fa, "w").write(">Splint1\n%s\n>Other\n%s\n" % (synth.SPLINT1, other))
    C3POa.main(C3POa.parse_args(["-r", fq, "-s", fa, "-o", out]))
    psl = open(out + "/tmp/splint_to_read_alignments.psl").read().splitlines()
    assert len(psl) == 18 and all(len(l.split("\t")) == 21 for l in psl)
    assert [(l.split("\t")[9], l.split("\t")[8], l.split("\t")[13]) for l in psl] == \
        [(r[0], r[3], "Splint1") for r in recs] + [(r[0], r[3], "Other") for r in others]
    log = open(out + "/c3poa.log").read().splitlines()
    assert log[2].startswith("No splint reads: 1 ")
    first = sorted(fastx_read(out + "/Splint1/R2C2_Consensus.fasta"))
    assert len(first) == 12
    second = list(fastx_read(out + "/Other/R2C2_Consensus.fasta"))
    assert len(second) == 6 and {r[0].split("_")[0] for r in second} == {r[0] for r in others}
    assert all(int(r[0].split("_")[3]) == 3 for r in second)                        # repeats field of the header
    mtime = os.stat(out + "/tmp/splint_to_read_alignments.psl").st_mtime_ns
    C3POa.main(C3POa.parse_args(["-r", fq, "-s", fa, "-o", out]))                 # resume: PSL reused
    assert os.stat(out + "/tmp/splint_to_read_alignments.psl").st_mtime_ns == mtime
    assert sorted(fastx_read(out + "/Splint1/R2C2_Consensus.fasta")) == first


def test_window_consensus_global_scratch_path(O, monkeypatch):
    """graphs larger than the LDS sweep arrays take the global-scratch copy of the consensus sweep: same results"""
    monkeypatch.setenv("C3_DEBUG_WIN_LCAP", "64")                 # every window graph is "too large" for LDS
    recs = list(synth.generate("cfg1", n_reads=24))
    _compare(O, [synth.SPLINT1], [(r[1], r[2]) for r in recs], [r[3] for r in recs], [0] * len(recs))


def test_results_do_not_depend_on_slot_count_or_run():
    """same batch, different numbers of resident wave slots, repeated runs: bit-identical records and consensi
    (work is pulled from atomic queues, so scheduling differs; results must not)"""
    from c3poa_amd import _lib
    recs = list(synth.generate("cfg3", n_reads=96)) + list(synth.generate("cfg1", n_reads=32))
    ref = None
    for kw in ({}, {"slots_poa": 7, "slots_win": 5}, {"slots_poa": 64, "slots_win": 300}):
        h = _lib.Handle(**kw)
        h.set_splints([synth.SPLINT1])
        h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
        for _rep in range(2):
            h.run()
            res, cons = h.results()
            # the fixed-size arrays are only defined up to n_peaks / n_sub
            rec = tuple((int(r["status"]), int(r["n_peaks"]), tuple(r["peaks"][:r["n_peaks"]].tolist()), int(r["n_sub"]),
                         tuple(r["sub_beg"][:r["n_sub"]].tolist()), tuple(r["sub_end"][:r["n_sub"]].tolist()),
                         int(r["has_front"]), int(r["has_tail"]), int(r["front_end"]), int(r["tail_beg"]),
                         int(r["cons_len"]), int(r["draft_len"]), int(r["n_win"])) for r in res)
            key = (rec, tuple(cons))
            if ref is None:
                ref = key
            assert key[1] == ref[1], kw
            assert key[0] == ref[0], kw
            assert not res["peaks"][np.arange(256)[None, :] >= res["n_peaks"][:, None]].any()      # unused tails are zeroed
        h.close()
    assert sum(1 for c in ref[1] if c) >= 120


def test_handle_reuse_across_different_batches():
    """a handle that has processed other (larger, different) batches gives the same results as a fresh one:
    no state leaks between batches (work lists, counters, zero-repeat flags, window records, scratch)"""
    from c3poa_amd import _lib
    rng = np.random.default_rng(21)
    big = list(synth.generate("cfg3", n_reads=150))
    s = _rand(rng, 2600)
    zero = (s[:1300] + synth.SPLINT1 + s[700:], None)                      # one splint, overlapping ends: zero-repeat rescue
    small = list(synth.generate("cfg1", n_reads=9, start=500))
    small_reads = [(r[1], r[2], r[3]) for r in small] + [(zero[0], _qual(rng, len(zero[0])), "+"), (_rand(rng, 900), _qual(rng, 900), "+")]

    def run(h, reads):
        h.upload([r[0] for r in reads], [r[1] for r in reads], [r[2] for r in reads])
        h.run()
        res, cons = h.results()
        return [(int(r["status"]), int(r["n_sub"]), int(r["n_peaks"]), int(r["cons_len"])) for r in res], cons

    fresh = _lib.Handle(); fresh.set_splints([synth.SPLINT1])
    want = run(fresh, small_reads)
    fresh.close()
    h = _lib.Handle(); h.set_splints([synth.SPLINT1])
    run(h, [(r[1], r[2], r[3]) for r in big])
    run(h, small_reads[-2:] * 20)                                          # many zero-repeat / no-peak reads
    got = run(h, small_reads)
    h.close()
    assert got == want
    assert want[0][-2][0] in (0, 3) and any(c for c in want[1])


def test_abi_error_paths_return_codes_not_crashes():
    """misuse of the C ABI returns a negative c3_err (and c3_last_error text) -- nothing crosses the boundary as a crash"""
    import ctypes as C
    from c3poa_amd import _lib
    lib = _lib.load()
    h = _lib.Handle()
    hp = h.h
    seq = b"ACGT" * 300; q = b"I" * len(seq)
    off = np.array([0, len(seq)], dtype=np.int64)
    sid = np.zeros(1, dtype=np.int16); st = np.frombuffer(b"+", dtype=np.uint8)
    up = lambda o=off, s_=sid: lib.c3_batch_upload(hp, 1, seq, q, o.ctypes.data, s_.ctypes.data, st.ctypes.data)   # noqa: E731
    assert up() == -5 and b"c3_set_splints" in lib.c3_last_error(hp)            # C3_E_STATE: no splints yet
    assert lib.c3_batch_run(hp, 15) < 0                                           # nothing uploaded
    too_long = b"A" * 513
    o2 = np.array([0, 513], dtype=np.int64)
    assert lib.c3_set_splints(hp, 1, too_long, o2.ctypes.data) == -6              # C3_E_LIMIT
    h.set_splints([synth.SPLINT1])
    assert up(np.array([5, len(seq)], dtype=np.int64)) == -3                      # off[0] != 0
    assert up(off, np.array([3], dtype=np.int16)) == -3                           # splint_id out of range
    assert lib.c3_batch_upload(hp, 0, seq, q, off.ctypes.data, sid.ctypes.data, st.ctypes.data) == -3
    assert up() == 0
    out = np.zeros(10, dtype=np.int32)
    assert lib.c3_fetch_track(hp, 0, out.ctypes.data, 10) < 0                     # stage not run yet
    assert lib.c3_batch_run(hp, 15) == 0
    assert lib.c3_fetch_track(hp, 0, out.ctypes.data, 10) == -6                   # capacity too small
    assert lib.c3_fetch_track(hp, 7, out.ctypes.data, 10) == -3                   # read index out of range
    res = np.zeros(1, dtype=_lib.RESULT_DTYPE); coff = np.zeros(2, dtype=np.int64)
    rd = list(synth.generate("cfg1", n_reads=1))[0]
    h.upload([rd[1]], [rd[2]], [rd[3]]); h.run()
    tiny = np.zeros(8, dtype=np.uint8)
    assert lib.c3_batch_results(hp, res.ctypes.data, tiny.ctypes.data, 8, coff.ctypes.data) == -6 and coff[1] > 1000   # needed size reported
    assert lib.c3_scan_adapters(hp, None) < 0 and lib.c3_match_index_batch(hp, 0, None, None, 2, None, None, None) < 0
    assert lib.c3_pairwise_consensus(hp, b"AC-T", b"ACGT", 4, b"ACT", 2, b"III", b"ACGT", 4, b"IIII", C.create_string_buffer(8), 8, C.byref(C.c_int())) == -3
    h.close()


def test_staged_upload_pipeline_equals_plain_upload(tmp_path):
    """c3_batch_stage / c3_batch_commit (next batch copied on a second stream while the resident one runs) give the
    results of plain uploads, batch after batch; misuse returns C3_E_STATE"""
    from c3poa_amd import _lib
    recs = list(synth.generate("cfg1", n_reads=48))
    fq = str(tmp_path / "r.fastq")
    with open(fq, "w") as fh:
        for r in recs:
            fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
    strand_of = {r[0]: r[3] for r in recs}

    def plain(h, batch):
        h.upload([r[1] for r in batch], [r[2] for r in batch], [r[3] for r in batch])
        h.run()
        return h.results()[1]

    h = _lib.Handle(); h.set_splints([synth.SPLINT1])
    want = [plain(h, recs[i:i + 16]) for i in range(0, 48, 16)]
    assert h.lib.c3_batch_commit(h.h) == -5                                   # nothing staged
    rd = _lib.Reader(fq, n_sets=3)
    hbs = [rd.next(16) for _ in range(3)]
    st = [bytes(ord(strand_of[n]) for n in hb.names()) for hb in hbs]
    sid = np.zeros(16, dtype=np.int16)
    got = []
    h.upload_host(hbs[0], st[0], sid)
    for k in range(3):
        if k + 1 < 3:
            h.stage_host(hbs[k + 1], st[k + 1], sid)                           # overlaps the run below
            assert h.lib.c3_batch_stage(h.h, 16, hbs[k].c.seqs, hbs[k].c.quals, hbs[k].c.off, sid.ctypes.data, st[k]) == -5   # one at a time
        h.run()
        got.append(h.results()[1])
        if k + 1 < 3:
            h.commit()
    assert got == want
    h.close()


def test_results_snapshot_then_fetch_on_another_thread(tmp_path):
    """c3_batch_results_snapshot / c3_batch_results_fetch: the results of batch k are frozen on the device, the owner commits and
    runs batch k+1, and a second thread copies batch k out meanwhile -- every batch must equal its plain c3_batch_results;
    misuse returns C3_E_STATE (-5); a too small consensus buffer still delivers records + needed size"""
    import threading
    from c3poa_amd import _lib
    recs = list(synth.generate("cfg1", n_reads=96))
    fq = str(tmp_path / "r.fastq")
    with open(fq, "w") as fh:
        for r in recs:
            fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
    strand_of = {r[0]: r[3] for r in recs}
    h = _lib.Handle(); h.set_splints([synth.SPLINT1])
    want = []
    for i in range(0, 96, 32):
        b = recs[i:i + 32]
        h.upload([r[1] for r in b], [r[2] for r in b], [r[3] for r in b]); h.run()
        res, buf, coff = h.results_raw()
        want.append((res["status"].copy(), res["cons_len"].copy(), res["n_sub"].copy(), buf[:coff[-1]].tobytes(), coff.copy()))
    rd = _lib.Reader(fq, n_sets=3)
    hbs = [rd.next(32) for _ in range(3)]
    st = [bytes(ord(strand_of[n]) for n in hb.names()) for hb in hbs]
    sid = np.zeros(32, dtype=np.int16)
    rb0 = _lib.ResultBuffers()
    assert h.lib.c3_batch_results_fetch(h.h, rb0.fit(32, 64)[0].ctypes.data, None, 0, None) == -5       # no snapshot yet
    got, threads = [None] * 3, []
    h.upload_host(hbs[0], st[0], sid)
    for k in range(3):
        if k + 1 < 3:
            h.stage_host(hbs[k + 1], st[k + 1], sid)
        h.run()
        for t in threads:
            t.join()                                                            # the previous snapshot has been fetched
        shape = h.results_snapshot()
        assert h.lib.c3_batch_results_snapshot(h.h) == -5                       # one snapshot per handle
        rb = _lib.ResultBuffers(pinned=(k == 1))                                # pageable and page-locked destinations

        def fetch(k=k, rb=rb, shape=shape):
            res, buf, coff = h.results_fetch(rb, shape)
            got[k] = (res["status"].copy(), res["cons_len"].copy(), res["n_sub"].copy(), buf[:coff[-1]].tobytes(), coff.copy())
        threads = [threading.Thread(target=fetch)]
        threads[0].start()
        if k + 1 < 3:
            h.commit()                                                          # the next batch becomes resident while batch k is copied out
    for t in threads:
        t.join()
    for k in range(3):
        for a, b in zip(got[k], want[k]):
            assert (a == b) if isinstance(a, bytes) else np.array_equal(a, b), k
    # too small a consensus buffer: C3_E_LIMIT (-6) from the one-call form, records + offsets delivered
    res, buf, coff = _lib.ResultBuffers().fit(h.n, 8)
    rc = h.lib.c3_batch_results(h.h, res.ctypes.data, buf[:8].ctypes.data, 8, coff.ctypes.data)
    assert rc == -6 and coff[-1] == want[2][4][-1] and np.array_equal(res["cons_len"], want[2][1])
    h.close()


def test_scan_then_assign_equals_upload_with_known_strands():
    """the fused CLI route: upload unassigned -> c3_scan_splints -> c3_batch_assign -> run gives what an upload with the
    true splint / strand gives; c3_batch_assign rejects a splint row outside the table"""
    from c3poa_amd import _lib
    rng = np.random.default_rng(17)
    other = _rand(rng, 180)
    recs = list(synth.generate("cfg1", n_reads=20))
    reads = [(r[1], r[2], r[3], 0) for r in recs]
    for i in range(6):
        s, q = _concatemer(rng, other, 800, 3, 90, 90)
        if i % 2:
            s, q = revcomp(s), q[::-1]
        reads.append((s, q, "+-"[i % 2], 1))
    reads.append((_rand(rng, 2500), _qual(rng, 2500), "?", 0))
    h = _lib.Handle(); h.set_splints([synth.SPLINT1, other])
    h.upload([r[0] for r in reads], [r[1] for r in reads], [r[2] for r in reads], np.array([r[3] for r in reads], dtype=np.int16))
    h.run()
    want = h.results()[1]
    h.upload([r[0] for r in reads], [r[1] for r in reads], "?" * len(reads))
    _tab, sid, st = h.scan_splints()
    assert [chr(c) for c in st] == [r[2] for r in reads] and [int(x) for x in sid[:-1]] == [r[3] for r in reads[:-1]] and sid[-1] == -1
    bad = sid.copy(); bad[0] = 5
    assert h.lib.c3_batch_assign(h.h, bad.ctypes.data, st) == -3
    h.assign(sid, st)
    h.run()
    assert h.results()[1] == want and sum(1 for c in want if c) >= 25
    h.close()


def test_poa_second_pass_redoes_overflowing_reads(O, monkeypatch):
    """the first POA pass runs with scratch sized for the typical alignment; reads that overflow it are redone by a second
    launch with worst-case scratch -- same results, nobody ends in C3_ST_LIMIT"""
    from c3poa_amd import _lib
    recs = list(synth.generate("cfg3", n_reads=40))
    h = _lib.Handle()
    h.set_splints([synth.SPLINT1])
    h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
    h.run()
    res0, cons0 = h.results()
    assert h.timing()["n_poa_redo"] == 0
    monkeypatch.setenv("C3_DEBUG_POA_SMALL", "6000")          # alignments of > 96000 cells (or 6000 nodes) overflow the first pass
    h.run()
    res1, cons1 = h.results()
    t = h.timing()
    assert 0 < t["n_poa_redo"] < len(recs)
    assert cons1 == cons0 and np.array_equal(res1["status"], res0["status"])
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs], threads=8)
    assert cons1 == ocons
    assert t["cells_poa"] == sum(r.cells_poa for r in ores)
    h.close()
