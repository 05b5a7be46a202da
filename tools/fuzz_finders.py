"""Randomised check of the splint finder (4 splints, 700 reads) and the adapter finder (3 adapters, 350 reads) against their
oracles (diagnostic): python tools/fuzz_finders.py"""
import sys; sys.path.insert(0, ".")
import numpy as np
from c3poa_amd import _lib, synth
from c3poa_amd.seqio import revcomp
from oracle import oracle_py as O
rng = np.random.default_rng(9)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
rnd = lambda L: acgt[rng.integers(0, 4, L)].tobytes().decode()
splints = [synth.SPLINT1, rnd(150), rnd(64), rnd(333)]
reads = []
for i in range(600):
    sp = splints[i % 4]; ins = rnd(int(rng.integers(50, 1500)))
    clean = ins[:int(rng.integers(0, len(ins)))] + (sp + ins) * int(rng.integers(0, 4)) + sp[:int(rng.integers(0, len(sp) + 1))] + ins
    sb, _ = synth._mutate(rng, np.frombuffer(clean.encode(), dtype=np.uint8), sub=0.04, ins=0.025, dele=0.035)
    s = sb.decode()
    reads.append(revcomp(s) if i % 3 == 0 else s)
reads += [rnd(int(rng.integers(1, 3000))) for _ in range(100)]
h = _lib.Handle(); h.set_splints(splints)
h.upload(reads, ["!" * len(r) for r in reads], "?" * len(reads))
tab, sid, st = h.scan_splints()
otab, osid, ost = O.scan_splints(reads, splints)
print("scan_splints equal:", np.array_equal(tab, otab), np.array_equal(sid, osid), st == ost, "assigned", int((sid >= 0).sum()))
ads = [rnd(33), rnd(36), rnd(80)]
h.set_splints(ads)
rd2 = []
for i in range(300):
    body = rnd(int(rng.integers(100, 2500)))
    a, b = ads[i % 3], ads[(i + 1) % 3]
    sb, _ = synth._mutate(rng, np.frombuffer((rnd(int(rng.integers(0, 40))) + a + body + revcomp(b) + rnd(int(rng.integers(0, 40)))).encode(), dtype=np.uint8), sub=0.02, ins=0.01, dele=0.01)
    rd2.append(sb.decode())
rd2 += [rnd(int(rng.integers(1, 200))) for _ in range(50)]
h.upload(rd2, ["!" * len(r) for r in rd2], "?" * len(rd2))
t2 = h.scan_adapters()
ok = True
for i, r in enumerate(rd2):
    for a, ad in enumerate(ads):
        for rc in (0, 1):
            if not np.array_equal(t2[i, a, rc], O.adapter_align(r, ad, bool(rc))): ok = False
print("scan_adapters equal:", ok)
