#!/usr/bin/env python3
"""Driver for `rocprofv3 --pc-sampling-*`: one resident batch, a few runs, nothing else (no child processes).
    rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 \
        -d gpurun_out/pcs -o pcs --output-format csv -- python3 tools/pc_sample_driver.py 8192 cfg2 3"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
recs = list(synth.generate(cfg, n_reads=min(n, 2048)))
recs = (recs * (n // len(recs) + 1))[:n]
h = _lib.Handle(mdistcutoff=synth.CONFIGS[cfg]["mdist"])
h.set_splints([synth.SPLINT1])
h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
for _ in range(reps):
    h.run()
t = h.timing()
print({k: round(v, 2) for k, v in t.items() if isinstance(v, float) and v})
