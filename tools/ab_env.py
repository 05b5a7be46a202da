"""Kernel times of one resident batch of DISTINCT reads with a run-time switch of the library off / on, alternating, same process and handle:
    python tools/ab_env.py cfg2 100000 C3_NO_WIN_CONSUMER [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from c3poa_amd import _lib, synth
cfg, n, var = sys.argv[1], int(sys.argv[2]), sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
recs = bench.make_reads(cfg, n, 0, bench.effective_cores())
h = _lib.Handle(mdistcutoff=synth.CONFIGS[cfg]["mdist"])
h.set_splints([synth.SPLINT1]); h.upload([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs])
h.run()
for rep in range(reps):
    for val in (None, "1"):
        if val is None:
            os.environ.pop(var, None)
        else:
            os.environ[var] = val
        h.run(); t = h.timing()
        print("%s %d distinct reads, %s=%s: ms_poa %.2f ms_prep %.2f ms_window %.2f (second launch %d) ms_total %.2f ms_wall %.2f" % (
            cfg, n, var, val, t["ms_poa"], t["ms_prep"], t["ms_window"], t["n_win_redo"], t["ms_total"], t["ms_wall"]), flush=True)
