"""CPU tests of the C-ABI shared library: it loads, exports every symbol declared in
include/c3poa.h, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "c3poa.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(c3_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from c3poa_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    assert set(_lib.EXPORTS) <= set(names)


def test_default_config_matches_reference_call_sites():
    from c3poa_amd import _lib
    c = _lib.default_config()
    assert (c.conk_penalty, c.sg_iters, c.sg_window, c.sg_order) == (20, 3, 41, 2)       # C3POa.py:111
    assert c.mdistcutoff == 500 and c.poa_match == 5                                         # C3POa.py:45, determine_consensus.py:30
    assert (c.poa_mismatch, c.poa_o1, c.poa_e1, c.poa_o2, c.poa_e2, c.poa_band_b) == (4, 4, 2, 24, 1, 10)
    assert abs(c.poa_band_f - 0.01) < 1e-15
    assert (c.pol_window, c.pol_q) == (500, 5)                                               # racon -q 5
    assert C.sizeof(_lib.ReadResult) == 4 * (10 + 3 * 256)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    from c3poa_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.C3Error) as e:
        _lib.Handle()
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under c3poa_amd/ or the CLI may reference it"""
    bad = []
    for base, _d, files in os.walk(os.path.join(ROOT, "c3poa_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|libc3oracle|c3o_[a-z]+\s*\(", txt):
                    bad.append(f)
    txt = open(os.path.join(ROOT, "C3POa.py")).read()
    if re.search(r"(from|import)\s+oracle|libc3oracle", txt):
        bad.append("C3POa.py")
    assert not bad, bad
