"""Oracle against golden vectors made by the REAL external libraries (tools/pin_external.py), for every stage whose file
exists under tests/golden/.  None can be generated in the build container (conk, pyabpoa, mappy, racon are not installed
and there is no network) -- these tests are the receiving end of the pin, and skip until a file is there."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _load(stage):
    p = os.path.join(GOLD, "external_%s.json" % stage)
    if not os.path.exists(p):
        pytest.skip("no %s: run tools/pin_external.py where %s is installed" % (os.path.basename(p), stage))
    return json.load(open(p))


def test_conk_track_against_the_real_conk():
    from oracle import oracle_py as O
    for c in _load("conk")["cases"]:
        assert O.conk(c["splint"], c["seq"], c["penalty"]).tolist() == c["track"]


def test_poa_against_the_real_pyabpoa():
    from oracle import oracle_py as O
    for c in _load("pyabpoa")["cases"]:
        cons, rows, _ = O.poa_msa(c["subs"], out_cons=c["cons"] is not None, out_msa=True)
        assert rows == c["msa"]
        if c["cons"] is not None:
            assert cons == [c["cons"]]


def test_polish_against_the_real_racon():
    from oracle import oracle_py as O
    for c in _load("racon")["cases"]:
        got = O.determine_consensus(c["subs"], c["quals"])
        assert got == c["polished"]


def test_pin_script_runs_without_the_libraries():
    """the script itself must work here: it reports what is missing instead of failing"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_external.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert all(s in out.stdout for s in ("conk", "pyabpoa", "mappy", "racon"))
