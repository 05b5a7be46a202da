"""The last POA pass beside the polish of the other reads.

A read whose band has blown up (long inserts with a missed peak) or whose scores leave the 16-bit cells is redone by the LAST pass of
k_poa: a handful of reads, 100+ ms on a handful of CUs.  c3_batch_run now runs that pass on a stream of its own while k_prep / k_window /
k_stitch work on every other read, and polishes the stragglers in a small tail.  Nothing of that may show in the results: the same
bytes as the oracle, the same as the serial order (C3_NO_TAIL_OVERLAP=1), whatever reads are handed over -- here every fifth / every
second read (test hook C3_DEBUG_POA_PUNT_MOD: the host adds them to the last pass's list), on the cfg2 / cfg3 shapes, with zero-repeat reads in the batch (their rescue needs the
draft of the pass they are in: a straggler among them keeps the serial order) and when only the POA stage is run."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import _lib, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(recs, mdist, env, stages=None):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        h = _lib.Handle(mdistcutoff=mdist)
        h.set_splints([synth.SPLINT1])
        h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
        if stages is None:
            h.run()
        else:
            for st in stages:
                h.run(st)
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


def _oracle(recs, mdist):
    return O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs], params=O.default_params(mdistcutoff=mdist), threads=8)


@pytest.mark.parametrize("cfg,n,mod", [("cfg2", 160, 5), ("cfg3", 96, 2), ("cfg2", 64, 1)])
def test_last_pass_beside_the_polish_changes_no_byte(cfg, n, mod):
    recs = list(synth.generate(cfg, n_reads=n))
    md = synth.CONFIGS[cfg]["mdist"]
    ores, ocons = _oracle(recs, md)
    res_a, cons_a, ta = _run(recs, md, {"C3_DEBUG_POA_PUNT_MOD": str(mod)})
    res_b, cons_b, tb = _run(recs, md, {"C3_DEBUG_POA_PUNT_MOD": str(mod), "C3_NO_TAIL_OVERLAP": "1"})
    res_c, cons_c, tc = _run(recs, md, {})
    for i in range(n):
        assert int(res_a[i]["status"]) == ores[i].status and cons_a[i] == ocons[i], (cfg, i)
    assert cons_a == cons_b == cons_c and (res_a["status"] == res_b["status"]).all() and (res_a["status"] == res_c["status"]).all()
    handed = sum(1 for i in range(n) if i % mod == 0 and ores[i].n_sub >= 2)
    assert ta["n_poa_redo16"] == handed == tb["n_poa_redo16"] and tc["n_poa_redo16"] == 0
    if mod > 1:
        assert ta["ms_poa_tail"] > 0 and tb["ms_poa_tail"] == 0                # overlapped / in series
    else:
        assert ta["ms_poa_tail"] == 0                                          # every read is a straggler: nothing to run beside
    # the counters of the two polish passes add up to the one-pass figures
    for k in ("n_windows", "cells_polish", "cells_polish_computed", "n_band_layers"):
        assert ta[k] == tb[k] == tc[k], k
    assert ta["cells_poa"] == tb["cells_poa"] >= tc["cells_poa"]               # (the hook's reads are aligned twice: first passes + last pass)


def test_stragglers_among_zero_repeat_reads_and_stage_by_stage_runs():
    """zero-repeat reads (one splint, two dangling pieces) become 2-subread POA jobs whose rescue is finished by k_zero_finish right
    after the POA passes: when one of them is handed to the last pass the serial order is kept; and a POA-only call finishes the
    last pass before it returns (the polish of a later call finds every draft)"""
    import numpy as np
    rng = np.random.default_rng(3)
    recs = list(synth.generate("cfg2", n_reads=48))
    acgt = "ACGT"
    zr = []
    for k in range(12):
        ins = "".join(acgt[x] for x in rng.integers(0, 4, 900))
        seq = ins[300:] + synth.SPLINT1 + ins[:700]
        zr.append(("z%d" % k, seq, "I" * len(seq), "+", ""))
    allr = recs + zr
    md = 500
    ores, ocons = _oracle(allr, md)
    assert sum(1 for r in ores[len(recs):] if r.status == 0) >= 6             # rescued reads exist
    for env in ({"C3_DEBUG_POA_PUNT_MOD": "3"}, {"C3_DEBUG_POA_PUNT_MOD": "3", "C3_NO_TAIL_OVERLAP": "1"}):
        res, cons, t = _run(allr, md, env)
        for i in range(len(allr)):
            assert int(res[i]["status"]) == ores[i].status and cons[i] == ocons[i], (env, i)
    # stage by stage: conk + peaks, POA alone (the last pass must be complete when the call returns), polish
    res, cons, t = _run(recs, md, {"C3_DEBUG_POA_PUNT_MOD": "4"}, stages=[_lib.STAGE_CONK | _lib.STAGE_PEAKS, _lib.STAGE_POA, _lib.STAGE_POLISH])
    for i in range(len(recs)):
        assert int(res[i]["status"]) == ores[i].status and cons[i] == ocons[i], i
    assert t["ms_poa_tail"] == 0


def test_polish_again_on_the_resident_batch_after_an_overlapped_last_pass():
    """round 6 (advisor): after the overlapped last pass the handle used to keep the STRAGGLERS' draft statistics (longest draft, window
    count); a later c3_batch_run(C3_STAGE_POLISH) on the same resident batch sized k_prep's window tables from a handful of reads and
    most reads ended as LIMIT.  The handle now keeps the whole batch's figures: full run with stragglers, then the polish stage again."""
    recs = list(synth.generate("cfg2", n_reads=96))
    md = synth.CONFIGS["cfg2"]["mdist"]
    ores, ocons = _oracle(recs, md)
    res, cons, t = _run(recs, md, {"C3_DEBUG_POA_PUNT_MOD": "7"}, stages=[_lib.STAGES_ALL, _lib.STAGE_POLISH])
    for i in range(len(recs)):
        assert int(res[i]["status"]) == ores[i].status and cons[i] == ocons[i], i
    # ... and when every straggler fails to produce a draft worth a window the figures of the others must survive as well: POA + polish
    # in one call, then polish twice more
    res, cons, t = _run(recs, md, {"C3_DEBUG_POA_PUNT_MOD": "2"}, stages=[_lib.STAGES_ALL, _lib.STAGE_POLISH, _lib.STAGE_POLISH])
    for i in range(len(recs)):
        assert int(res[i]["status"]) == ores[i].status and cons[i] == ocons[i], i
