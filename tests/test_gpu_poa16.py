"""k_poa's 16-bit cells: the fast / near rows and the LDS ring keep every score relative to a per-row base
in 16 bits.  That is a representation, never a change of result: the oracle computes in 32 bits, and the HIP path must equal it
  * in the default set-up, WITHOUT handing a single read to the 32-bit second pass on the config shapes (a read is handed over
    when a stored score is neither plainly reachable nor plainly unreachable -- the guard -- or when a base difference is too
    large: a silent stream of such reads would only show up as time),
  * when the base follows the row maximum every few rows instead of every ~300 (C3_DEBUG_POA_RBSPAN: exercises the saturating
    re-basing, ring rows on different bases, the general row between them),
  * when every read goes through the 32-bit kernel instance (C3_DEBUG_POA32: what a handed-over read runs)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import _lib, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(recs, mdist, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        h = _lib.Handle(mdistcutoff=mdist)
        h.set_splints([synth.SPLINT1])
        h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
        h.run()
        res, cons = h.results()
        t = h.timing()
        h.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res, cons, t


def _ragged(recs, seed=5):
    """every third read loses or doubles a 25-90 base chunk in one repeat (wider, drifting bands; far predecessors)"""
    rng = np.random.default_rng(seed)
    out = []
    for k, r in enumerate(recs):
        name, seq, qual, strand, truth = r
        if k % 3 == 0 and len(seq) > 3000:
            at = int(rng.integers(1800, len(seq) - 400)); ln = int(rng.integers(25, 90))
            if k % 2:
                seq, qual = seq[:at] + seq[at + ln:], qual[:at] + qual[at + ln:]
            else:
                seq, qual = seq[:at] + seq[at:at + ln] + seq[at:], qual[:at] + qual[at:at + ln] + qual[at:]
        out.append((name, seq, qual, strand, truth))
    return out


@pytest.mark.parametrize("cfg,n", [("cfg2", 128), ("cfg3", 96), ("cfg4", 24)])
def test_sixteen_bit_cells_equal_the_oracle_and_hand_nothing_over(cfg, n):
    recs = _ragged(list(synth.generate(cfg, n_reads=n)))
    md = synth.CONFIGS[cfg]["mdist"]
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs],
                                  params=O.default_params(mdistcutoff=md), threads=8)
    cells = sum(int(r.cells_poa) for r in ores)
    # (C3_DEBUG_POA32 = 1: everything through the single-wave 32-bit instance, 2: through the eight-wave workgroup of the last pass)
    for env in ({}, {"C3_DEBUG_POA_RBSPAN": "3300"}, {"C3_DEBUG_POA_RBSPAN": "4000"}, {"C3_DEBUG_POA32": "1"}, {"C3_DEBUG_POA32": "2"}):
        res, cons, t = _run(recs, md, env)
        for i in range(n):
            assert res[i]["status"] == ores[i].status and cons[i] == ocons[i], (cfg, env, i)
        assert t["cells_poa"] == cells, (cfg, env)                     # counted cells: the same bands, row for row
        if "C3_DEBUG_POA32" not in env:
            assert t["n_poa_redo16"] == 0, (cfg, env, t["n_poa_redo16"])          # (n_poa_redo also counts reads whose scratch was too small)


def test_msa_rows_survive_a_moving_base():
    """stage probe: the MSA rows of c3_poa_msa with the base moving every other row equal the oracle's"""
    recs = list(synth.generate("cfg1", n_reads=4))
    old = os.environ.get("C3_DEBUG_POA_RBSPAN")
    os.environ["C3_DEBUG_POA_RBSPAN"] = "3300"
    try:
        h = _lib.Handle()
        for r in recs:
            subs = [r[1][250 + 1500 * k: 250 + 1500 * (k + 1)] for k in range(3)]
            gc, gm = h.poa_msa(subs)
            oc, om, _ = O.poa_msa(subs)
            assert gc == oc and gm == om
        h.close()
    finally:
        if old is None:
            os.environ.pop("C3_DEBUG_POA_RBSPAN", None)
        else:
            os.environ["C3_DEBUG_POA_RBSPAN"] = old


@pytest.mark.parametrize("cfg,n,envs", [
    ("cfgL", 10, ({}, {"C3_DEBUG_POA_WIDE": "0"}, {"C3_DEBUG_POA_RBSPAN": "3300"}, {"C3_DEBUG_POA32": "2"})),       # 3 kb / 6 kb inserts: WIDE ring by default; the last pass's workgroup kernel (rows of 3+ chunks by eight waves)
    ("cfg2", 64, ({"C3_DEBUG_POA_WIDE": "1"},)),                                            # the WIDE instance on ordinary reads
    ("cfg4", 12, ({"C3_DEBUG_POA_WIDE": "1"},)),
])
def test_long_subreads_and_the_wide_ring(cfg, n, envs):
    """subreads beyond the 1792-base LDS query copy / bands beyond 128 columns (abPOA has no length limit,
    bin/determine_consensus.py:30-47): the WIDE kernel instance (4 ring rows of 192 cells, three-chunk near rows, a query window
    that follows the band) equals the oracle, and so does the NARROW instance on the same reads (everything through its general
    rows); no read is handed to the 32-bit pass"""
    recs = _ragged(list(synth.generate(cfg, n_reads=n)), seed=9)
    md = synth.CONFIGS[cfg]["mdist"]
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs],
                                  params=O.default_params(mdistcutoff=md), threads=8)
    cells = sum(int(r.cells_poa) for r in ores)
    assert any(r.n_sub >= 3 for r in ores)
    for env in envs:
        res, cons, t = _run(recs, md, env)
        for i in range(n):
            assert res[i]["status"] == ores[i].status and cons[i] == ocons[i], (cfg, env, i)
        assert t["cells_poa"] == cells and (t["n_poa_redo16"] == 0 or "C3_DEBUG_POA32" in env), (cfg, env, t["n_poa_redo16"])
