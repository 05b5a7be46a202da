#!/usr/bin/env python3
"""Diagnostic: can conk + peaks of the NEXT batch hide under the POA / window kernels of the current one?
Two handles on one GPU, two host threads: A runs the whole path in a loop, B runs conk + peaks in a loop.
  python tools/overlap_probe.py [n_reads] [slots_poa] [slots_win]"""
import sys, threading, time
sys.path.insert(0, ".")
from c3poa_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
sp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
recs = list(synth.generate("cfg2", n_reads=2048))
recs = (recs * (n // len(recs) + 1))[:n]
args = ([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])


def mk(**kw):
    h = _lib.Handle(**kw); h.set_splints([synth.SPLINT1]); h.upload(*args); return h


a, b = mk(slots_poa=sp, slots_win=sw), mk()
a.run(); b.run(3)
t0 = time.perf_counter()
for _ in range(3):
    a.run()
ta = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
for _ in range(3):
    b.run(3)
tb = (time.perf_counter() - t0) / 3
print("alone: full path %.1f ms  %s" % (ta * 1e3, {k: round(v, 1) for k, v in a.last_timing.items() if k.startswith("ms_")}))
print("alone: conk+peaks %.1f ms" % (tb * 1e3))
stop = [False]; cnt = [0]


def spin():
    while not stop[0]:
        b.run(3); cnt[0] += 1


th = threading.Thread(target=spin); th.start()
time.sleep(0.2)
c0 = cnt[0]; t0 = time.perf_counter()
for _ in range(3):
    a.run()
tc = (time.perf_counter() - t0) / 3
c1 = cnt[0]
stop[0] = True; th.join()
print("together: full path %.1f ms per run, conk+peaks runs completed beside it: %.2f per full run  %s" % (
    tc * 1e3, (c1 - c0) / 3.0, {k: round(v, 1) for k, v in a.last_timing.items() if k.startswith("ms_")}))
print("serial cost of both = %.1f ms; overlapped = %.1f ms per (full + %.2f conk+peaks)" % ((ta + tb) * 1e3, tc * 1e3, (c1 - c0) / 3.0))
