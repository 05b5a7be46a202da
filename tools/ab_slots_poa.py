"""k_poa time against the number of resident wave slots: python tools/ab_slots_poa.py N lib slots_poa [slots_poa ...]   (CFG=cfg4 ...)"""
import os, subprocess, sys
n, lib = sys.argv[1], sys.argv[2]
code = r'''
import sys; sys.path.insert(0, ".")
from c3poa_amd import _lib, synth
n = int(sys.argv[1]); cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
recs = list(synth.generate(cfg, n_reads=2048)) * (n // 2048)
h = _lib.Handle(slots_poa=int(sys.argv[2]), mdistcutoff=synth.CONFIGS[cfg]["mdist"])
h.set_splints([synth.SPLINT1]); h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
best = None
for _ in range(3):
    h.run(); t = h.timing()
    best = t if best is None else {k: min(v, best[k]) if isinstance(v, float) else v for k, v in t.items()}
print("slots_poa=%s ms_poa=%.2f ms_prep=%.2f ms_window=%.2f" % (sys.argv[2], best["ms_poa"], best["ms_prep"], best["ms_window"]))
'''
for s in sys.argv[3:]:
    r = subprocess.run([sys.executable, "-c", code, n, s] + ([os.environ["CFG"]] if "CFG" in os.environ else []), env=dict(os.environ, C3POA_LIB=lib), capture_output=True, text=True)
    print(os.path.basename(lib), r.stdout.strip() or r.stderr[-400:])
