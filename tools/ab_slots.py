"""k_window / k_poa time of one library build against the number of resident wave slots: python tools/ab_slots.py N lib slots_win [slots_win ...]"""
import os, subprocess, sys
n, lib = sys.argv[1], sys.argv[2]
code = r'''
import sys; sys.path.insert(0, ".")
from c3poa_amd import _lib, synth
n = int(sys.argv[1]); cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
recs = list(synth.generate(cfg, n_reads=2048)) * (n // 2048)
h = _lib.Handle(slots_win=int(sys.argv[2]), mdistcutoff=synth.CONFIGS[cfg]["mdist"])
h.set_splints([synth.SPLINT1]); h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
best = None
for _ in range(3):
    h.run(); t = h.timing()
    best = t if best is None else {k: min(v, best[k]) if isinstance(v, float) else v for k, v in t.items()}
print("slots_win=%s ms_window=%.2f ms_poa=%.2f fallback=%d/%d computed=%.3f" % (sys.argv[2], best["ms_window"], best["ms_poa"], best["n_band_fallback"], best["n_band_layers"] + best["n_band_fallback"], best["cells_polish_computed"] / max(best["cells_polish"], 1)))
'''
for s in sys.argv[3:]:
    env = dict(os.environ, C3POA_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code, n, s] + ([os.environ["CFG"]] if "CFG" in os.environ else []), env=env, capture_output=True, text=True)
    print(os.path.basename(lib), r.stdout.strip() or r.stderr[-400:])
