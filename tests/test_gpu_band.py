"""k_window's banded rows: the band is an optimisation with a proof obligation, never a change of result.
The oracle always fills the full matrix; the HIP path must equal it
  * with the band and its certificate (default),
  * with the band switched off (the unbanded rows),
  * with every certificate declared failed (band attempted, thrown away, layer redone unbanded -- the fallback path),
and the timing counters must tell the three apart (cells computed vs cells of the full matrices, band layers, fallbacks)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import _lib, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(recs, mdist, mode):
    old = os.environ.get("C3_DEBUG_BAND")
    if mode:
        os.environ["C3_DEBUG_BAND"] = mode
    else:
        os.environ.pop("C3_DEBUG_BAND", None)
    try:
        h = _lib.Handle(mdistcutoff=mdist)
        h.set_splints([synth.SPLINT1])
        h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
        h.run()
        res, cons = h.results()
        t = h.timing()
        h.close()
    finally:
        if old is None:
            os.environ.pop("C3_DEBUG_BAND", None)
        else:
            os.environ["C3_DEBUG_BAND"] = old
    return res, cons, t


@pytest.mark.parametrize("cfg,n", [("cfg2", 96), ("cfg4", 24), ("cfg3", 64)])
def test_band_certificate_and_fallback_all_equal_the_full_matrix(cfg, n):
    recs = list(synth.generate(cfg, n_reads=n))
    md = synth.CONFIGS[cfg]["mdist"]
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs],
                                  params=O.default_params(mdistcutoff=md), threads=8)
    cells_full = sum(int(r.cells_polish) for r in ores)
    seen = {}
    for mode in (None, "off", "fail"):
        res, cons, t = _run(recs, md, mode)
        for i in range(n):
            assert res[i]["status"] == ores[i].status and cons[i] == ocons[i], (cfg, mode, i)
        assert t["cells_polish"] == cells_full                         # the algorithmic count never changes
        seen[mode] = t
    band, off, fail = seen[None], seen["off"], seen["fail"]
    assert off["n_band_layers"] == 0 and off["n_band_fallback"] == 0 and off["cells_polish_computed"] == cells_full
    assert band["n_band_layers"] > 0 and band["cells_polish_computed"] < 0.6 * cells_full
    assert band["n_band_fallback"] <= 0.05 * (band["n_band_layers"] + band["n_band_fallback"]) + 2
    # two launches of k_window: the first has DP scratch for the usual layer only; with the band on almost no window needs the
    # full-size second launch (the test below forces every window through it)
    assert band["n_win_redo"] <= 0.02 * band["n_windows"] + 1
    # "fail": every band-eligible layer makes ONE band attempt and is redone unbanded; normally a failed attempt is retried with
    # the next wider band first, so a layer can count several failed attempts before it is accepted (or goes unbanded)
    assert fail["n_band_layers"] == 0
    assert band["n_band_layers"] <= fail["n_band_fallback"] <= band["n_band_layers"] + band["n_band_fallback"]
    assert fail["cells_polish_computed"] > cells_full                  # band attempts + full matrices


def test_layers_longer_than_the_lds_query_arrays():
    """a subread with a 1.3 kb insertion inside one 500-base window: that layer has > 1024 bases, so its per-base row / node
    arrays live in global memory instead of LDS (W_QCAP) and its rows take the generic unbanded path"""
    rng = np.random.default_rng(31)
    recs = list(synth.generate("cfg2", n_reads=6))
    out = []
    for k, r in enumerate(recs):
        name, seq, qual, strand, truth = r
        if k % 2 == 0:
            at = 1750 + 300 + int(rng.integers(0, 200))               # inside the second subread
            ins = "".join("ACGT"[i] for i in rng.integers(0, 4, 1300))
            seq = seq[:at] + ins + seq[at:]
            qual = qual[:at] + "5" * len(ins) + qual[at:]
        out.append((name, seq, qual, strand, truth))
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in out], [r[3] for r in out], threads=6)
    res, cons, _t = _run(out, 500, None)
    for i in range(len(out)):
        assert res[i]["status"] == ores[i].status and cons[i] == ocons[i], i


def test_band_verify_on_device_and_the_fifth_in_edge_regression():
    """C3_DEBUG_BAND=verify aligns every accepted band layer a second time with the full matrix ON THE DEVICE and counts the layers
    whose tracebacks differ: must be 0.  The read is a fuzz find of round 3 (tools/fuzz_parity.py 2000 33, read 188): a window
    node with five in-edges of which the first two lie outside the layer's sub-graph -- the band descriptor build dropped the
    fifth edge's row from the first-four list, the DP took the virtual start row instead, one path differed by a tie."""
    import pickle
    read, strand = pickle.load(open(os.path.join(ROOT, "tests", "golden", "band_regress_read.pkl"), "rb"))
    recs = [("fuzz33_188", read[0], read[1], strand, "")] + list(synth.generate("cfg4", n_reads=6)) + list(synth.generate("cfg3", n_reads=24))
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs], threads=8)
    for mode in (None, "verify"):
        res, cons, t = _run(recs, 500, mode)
        for i in range(len(recs)):
            assert res[i]["status"] == ores[i].status and cons[i] == ocons[i], (mode, i)
        assert t["n_band_layers"] > 100 and t["n_band_mismatch"] == 0


def test_tiny_first_launch_scratch_sends_every_window_to_the_second_launch(monkeypatch):
    """C3_DEBUG_HCAP_DIV shrinks the first launch's DP scratch until no layer fits: every polished window is queued on the device
    and redone by the full-size launch; results, statuses and cell counts must not move"""
    recs = list(synth.generate("cfg2", n_reads=64)) + list(synth.generate("cfg3", n_reads=32, start=500))
    want, wcons, t0 = _run(recs, 500, None)
    monkeypatch.setenv("C3_DEBUG_HCAP_DIV", "100000")
    got, gcons, t1 = _run(recs, 500, None)
    assert gcons == wcons and np.array_equal(got["status"], want["status"]) and np.array_equal(got["cons_len"], want["cons_len"])
    assert t1["n_win_redo"] >= 0.9 * t1["n_windows"] > 0 and t0["n_win_redo"] <= 0.02 * t0["n_windows"] + 1
    for k in ("cells_polish", "cells_polish_computed", "n_band_layers", "n_band_fallback"):
        assert t1[k] == t0[k], k


@pytest.mark.parametrize("consumers", ["1", "7", "256", "off"])
def test_full_size_launch_as_a_consumer_beside_the_first_one(monkeypatch, consumers):
    """round 6: k_window's full-size launch runs BESIDE the first launch (a few resident waves on a stream of their own claim the windows
    the first launch hands over while it is still running; a third launch mops up).  With the tiny first-launch scratch every window takes
    that road: one consumer wave, a few, many, or none at all (C3_NO_WIN_CONSUMER: the serial order of round 5) -- the same bytes, the same
    counters, and a second batch on the same handle (the consumer count follows the previous batch's overflow)"""
    recs = list(synth.generate("cfg2e15", n_reads=64)) + list(synth.generate("cfg3", n_reads=32, start=700))
    want, wcons, t0 = _run(recs, 500, None)
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs], threads=8)
    assert wcons == ocons
    monkeypatch.setenv("C3_DEBUG_HCAP_DIV", "100000")
    if consumers == "off":
        monkeypatch.setenv("C3_NO_WIN_CONSUMER", "1")
    else:
        monkeypatch.setenv("C3_DEBUG_WIN_CONSUMERS", consumers)
    got, gcons, t1 = _run(recs, 500, None)
    assert gcons == wcons and np.array_equal(got["status"], want["status"]) and np.array_equal(got["cons_len"], want["cons_len"])
    assert t1["n_win_redo"] >= 0.9 * t1["n_windows"] > 0
    for k in ("cells_polish", "cells_polish_computed", "n_band_layers", "n_band_fallback", "n_windows"):
        assert t1[k] == t0[k], k
    # two batches through one handle
    monkeypatch.delenv("C3_DEBUG_WIN_CONSUMERS", raising=False)
    from c3poa_amd import _lib
    h = _lib.Handle(mdistcutoff=500)
    h.set_splints([synth.SPLINT1])
    for _ in range(2):
        h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
        h.run()
        assert h.results()[1] == wcons
    h.close()


def test_an_overflow_entry_that_appears_seconds_after_its_index_was_taken(monkeypatch):
    """round 6: the consumer beside the first launch claims an entry of the overflow list as soon as the COUNT says it exists; the entry
    itself is written a moment later -- and until the writer's store is visible device-wide.  The first version gave up on a claimed
    entry after about a second and lost the window (found by the fuzz gate on a batch whose first launch ran for seconds:
    profiles/r06_fuzz_parity.txt).  The test hook makes every producer wait 2.5 s between the two; no window may be lost."""
    recs = list(synth.generate("cfg2", n_reads=24))
    want, wcons, t0 = _run(recs, 500, None)
    monkeypatch.setenv("C3_DEBUG_HCAP_DIV", "100000")
    monkeypatch.setenv("C3_DEBUG_OVF_DELAY_MS", "2500")
    monkeypatch.setenv("C3_DEBUG_WIN_CONSUMERS", "64")
    got, gcons, t1 = _run(recs, 500, None)
    assert gcons == wcons and np.array_equal(got["status"], want["status"]) and np.array_equal(got["cons_len"], want["cons_len"])
    assert t1["n_win_redo"] >= 0.9 * t1["n_windows"] > 0 and t1["n_windows"] == t0["n_windows"]


def test_results_do_not_depend_on_what_fresh_device_memory_holds(monkeypatch):
    """every device buffer poisoned at allocation (C3_DEBUG_POISON): nothing may read a scratch cell it did not write -- with the
    usual scratch and with the tiny first-launch scratch that sends the windows through the abort + second-launch path"""
    recs = list(synth.generate("cfg2", n_reads=48)) + list(synth.generate("cfg3", n_reads=32, start=900)) + list(synth.generate("cfg4", n_reads=8, start=70))
    want, wcons, _ = _run(recs, 500, None)
    monkeypatch.setenv("C3_DEBUG_POISON", "1")
    for div in (None, "100000"):
        if div:
            monkeypatch.setenv("C3_DEBUG_HCAP_DIV", div)
        for mode in (None, "off"):
            got, gcons, _t = _run(recs, 500, mode)
            assert gcons == wcons and np.array_equal(got["status"], want["status"]), (div, mode)
