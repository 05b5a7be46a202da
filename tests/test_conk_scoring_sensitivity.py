"""conk's match / mismatch values are a guess (the library is external and unpinned, the call site C3POa.py:123 fixes
only penalty = 20).  This does not pin conk; it bounds what a wrong guess could cost: under three other plausible scorings the
split of a read (C3POa.py:127-155) keeps its structure -- the same number of kept subreads, the same dangling pieces -- and its
subread boundaries move by a handful of bases (they sit in the middle of the splint, which post-processing trims anyway).
Numbers over more reads and configs: tools/conk_scoring_sensitivity.py -> profiles/r04_conk_scoring_sensitivity.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

import pytest  # noqa: E402

from c3poa_amd import synth  # noqa: E402
import conk_scoring_sensitivity as CS  # noqa: E402


@pytest.mark.parametrize("cfg,n", [("cfg1", 60), ("cfg3", 60), ("cfg4", 12)])
@pytest.mark.parametrize("scoring", [(1, -1), (2, -3), (5, -5)])
def test_split_structure_survives_other_scorings(cfg, n, scoring):
    recs = list(synth.generate(cfg, n_reads=n))
    k, same, struct, shifts = CS.compare(recs, synth.CONFIGS[cfg]["mdist"], scoring)
    # the structure of the split is what decides which bases reach the consensus: at most one read in 12 may differ ...
    assert struct >= k - max(1, k // 12), (cfg, scoring, struct, k)
    # ... and where it is kept, the subread boundaries move by at most 5 bases (again one read in 12 may not: at cfg4 under 2/-3 one
    # read in 60 loses a peak and a boundary moves by a whole repeat -- the table in profiles/ lists it)
    assert (shifts <= 5).sum() >= len(shifts) - max(1, len(shifts) // 12), (cfg, scoring, shifts.max())


def test_the_frozen_scoring_is_compared_with_itself_exactly():
    recs = list(synth.generate("cfg1", n_reads=10))
    k, same, struct, shifts = CS.compare(recs, 500, CS.FROZEN)
    assert same == struct == k and shifts.max() == 0
