"""CPU oracle vs golden vectors captured from the reference's own Python
(tests/golden/make_golden.py).  Pins the oracle for: savitzky_golay, call_peaks, find_peaks,
rounding, subread split, pairwise_consensus, normalizeLen  (SURVEY.md 8(c))."""
import numpy as np
import pytest

from oracle import oracle_py as O


def test_savgol_matches_reference(golden):
    _, sig = golden
    names = [k[4:] for k in sig.files if k.startswith("sg1_")]
    assert names
    for nm in names:
        t = sig["track_" + nm].astype(np.float64)
        s1 = O.savgol(t)
        # tolerance: the reference sums via numpy dot (BLAS order) with pinv coefficients;
        # the oracle uses closed-form coefficients and a fixed fma order -> 1e-9 relative
        np.testing.assert_allclose(s1, sig["sg1_" + nm], rtol=1e-9, atol=1e-7)
        s3 = O.savgol(O.savgol(s1))
        np.testing.assert_allclose(s3, sig["sg3_" + nm], rtol=1e-9, atol=1e-7)


def test_savgol_coeffs_closed_form_equals_pinv():
    import ctypes as C
    c = (C.c_double * 41)()
    assert O.lib().c3o_savgol_coeffs(41, 2, c) == 0
    b = np.array([[k ** i for i in range(3)] for k in range(-20, 21)], dtype=float)
    m = np.linalg.pinv(b)[0]
    np.testing.assert_allclose(np.array(c[:]), m, atol=5e-15)


def test_call_peaks_exact(golden):
    cases, sig = golden
    for c in cases["call_peaks"]:
        pk = O.call_peaks(sig["track_" + c["name"]], c["min_dist"])
        assert pk.tolist() == c["peaks"], (c["name"], c["min_dist"])


def test_find_peaks_exact(golden):
    cases, _ = golden
    for c in cases["find_peaks"]:
        pk = O.find_peaks(c["x"], c["height"], c["distance"])
        assert pk.tolist() == c["peaks"], c


def test_rounding(golden):
    cases, _ = golden
    for c in cases["rounding"]:
        assert O.rounding(c["x"], c["base"]) == c["r"], c


def test_split(golden):
    cases, _ = golden
    for c in cases["split"]:
        r = O.split(c["raw_peaks"], c["S"], c["L"])
        if not c["called"]:
            assert r["peaks"] == []
            continue
        assert [e - b for b, e in r["subs"]] == c["sub_lens"], c
        d = []
        if r["has_front"]:
            d.append(r["front_end"])
        if r["has_tail"]:
            d.append(c["L"] - r["tail_beg"])
        assert d == c["dang_lens"], c


def test_pairwise_and_normalize(golden):
    cases, _ = golden
    for c in cases["pairwise"]:
        rows, quals = c["rows"], c["quals"]
        subs = [r.replace("-", "") for r in rows]
        assert O.normalize_len(rows[0], quals[0]) == c["norm"][0]
        assert O.normalize_len(rows[1], quals[1]) == c["norm"][1]
        assert O.pairwise_consensus(rows, subs, quals) == c["cons"], c


def test_pairwise_identical_subreads_share_later_quality():
    # bin/consensus.py:77-79: dict keyed by subread string
    rows = ["ACGT", "ACGT"]
    assert O.pairwise_consensus(rows, ["ACGT", "ACGT"], ["IIII", "5555"]) == "ACGT"
