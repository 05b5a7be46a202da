"""throughput of the GPU splint finder (c3_scan_splints) on cfg2-shaped reads"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from c3poa_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
recs = list(synth.generate("cfg2", n_reads=n))
rng = np.random.default_rng(1)
for k in (1, 4):
    splints = [synth.SPLINT1] + ["".join("ACGT"[i] for i in rng.integers(0, 4, 284)) for _ in range(k - 1)]
    h = _lib.Handle(); h.set_splints(splints)
    h.upload([r[1] for r in recs], [r[2] for r in recs], "?" * n)
    h.scan_splints()
    t = time.time(); tab, sid, st = h.scan_splints(); dt = time.time() - t
    ok = sum(1 for i, r in enumerate(recs) if sid[i] == 0 and chr(st[i]) == r[3])
    bases = sum(len(r[1]) for r in recs)
    print("splints=%d reads=%d correct=%d  %.1f ms  %.0f reads/s  %.2f Gcell/s" % (k, n, ok, dt * 1e3, n / dt, bases * 284 * 2 * k / dt / 1e9))
    h.close()
