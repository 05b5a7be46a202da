"""k_poa time of several builds of the library on the same DISTINCT reads of a config (round 6: 16 copies of 2 048 reads, what tools/ab_slots_poa.py
times, run in near lock-step and flattered one build by 4 %):    python tools/ab_poa_distinct.py cfg4 32768 lib1.so lib2.so ..."""
import os, subprocess, sys
cfg, n = sys.argv[1], sys.argv[2]
code = r'''
import sys, os; sys.path.insert(0, ".")
import bench
from c3poa_amd import _lib, synth
cfg, n = sys.argv[1], int(sys.argv[2])
recs = bench.make_reads(cfg, n, 0, bench.effective_cores())
h = _lib.Handle(mdistcutoff=synth.CONFIGS[cfg]["mdist"])
h.set_splints([synth.SPLINT1]); h.upload([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs])
best = None
for _ in range(3):
    h.run(); t = h.timing()
    best = t if best is None else {k: min(v, best[k]) if isinstance(v, float) else v for k, v in t.items()}
print("%s %d distinct reads: ms_poa=%.2f ms_prep=%.2f ms_window=%.2f ms_peaks=%.2f ms_conk=%.2f" % (cfg, n, best["ms_poa"], best["ms_prep"], best["ms_window"], best["ms_peaks"], best["ms_conk"]))
'''
for rep in range(int(os.environ.get("REPS", "2"))):
    for lib in sys.argv[3:]:
        r = subprocess.run([sys.executable, "-c", code, cfg, n], env=dict(os.environ, C3POA_LIB=lib), capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip() or r.stderr[-400:], flush=True)
