"""k_poa / k_window time against the number of resident wave slots (diagnostic): python tools/slots_sweep.py [n_reads]"""
import sys
sys.path.insert(0, ".")
from c3poa_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
recs = list(synth.generate("cfg2", n_reads=2048)) * (n // 2048)
for sp, sw in ((1024, 1024), (2048, 2048), (3072, 3072), (4096, 4096), (0, 0), (5632, 5120), (6144, 6144)):
    h = _lib.Handle(slots_poa=sp, slots_win=sw)
    h.set_splints([synth.SPLINT1])
    h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
    best = None
    for _ in range(3):
        h.run(); t = h.timing()
        best = t if best is None else {k: min(v, best[k]) if isinstance(v, float) else v for k, v in t.items()}
    print("slots poa=%d win=%d  ms_poa=%.1f ms_window=%.1f" % (sp, sw, best["ms_poa"], best["ms_window"]))
    h.close()
