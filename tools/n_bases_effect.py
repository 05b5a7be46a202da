#!/usr/bin/env python3
"""How much does coding non-ACGT bytes as 'A'  cost?  abPOA gives N its own code (bin/determine_consensus.py:43);
the 2-bit packing here cannot.  Upper bound of the harm, measured with the CPU oracle (same coding as the kernels): consensus
identity against the synthetic truth for cfg2 reads as generated vs the same reads with a fraction of their bases replaced by N.
python tools/n_bases_effect.py [reads] [fraction ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import synth
from oracle import oracle_py as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fracs = [float(x) for x in sys.argv[2:]] or [0.0, 0.001, 0.005, 0.02]
recs = list(synth.generate("cfg2", n_reads=n))
rng = np.random.default_rng(123)
for f in fracs:
    reads = []
    for _name, seq, qual, _st, _truth in recs:
        s = np.frombuffer(seq.encode(), dtype=np.uint8).copy()
        s[rng.random(len(s)) < f] = ord("N")
        reads.append((s.tobytes().decode(), qual))
    res, cons = O.process_batch(synth.SPLINT1, reads, [r[3] for r in recs], threads=min(8, os.cpu_count() or 1))
    ids = [synth.identity(c, r[4]) for c, r in zip(cons, recs) if c]
    print("N fraction %.3f: %d / %d reads with a consensus, identity vs truth mean %.5f median %.5f" % (f, len(ids), n, float(np.mean(ids)), float(np.median(ids))))
