#!/usr/bin/env python3
"""Randomised parity sweep, GPU path against the oracle (diagnostic, slower than the test suite):
   python tools/fuzz_parity.py [n_reads] [seed] [max insert, default 3500]
Concatemers with random insert length (60..3500), repeats (0..14), flank lengths, error rate (0..25 %), strand, quality
profile, occasional non-ACGT bytes / lower case, ragged repeats (a 25-90 base chunk missing or duplicated in some copies), plus
pure noise reads.  Prints the band counters of k_window (layers accepted by the certificate / redone unbanded)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from c3poa_amd import _lib, synth
from c3poa_amd.seqio import revcomp
from oracle import oracle_py as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 3500          # 7000: subreads beyond 6 kb (bands of 3-4 chunks in k_poa's WIDE instance)
rng = np.random.default_rng(seed)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
rnd = lambda L: acgt[rng.integers(0, 4, L)].tobytes().decode()   # noqa: E731
reads, strands = [], []
for i in range(n):
    kind = rng.integers(0, 20)
    if kind == 0:
        s = rnd(int(rng.integers(0, 4000))); q = "".join(chr(33 + int(v)) for v in rng.integers(0, 60, len(s)))
    else:
        ins = rnd(int(rng.integers(60, max_ins)))
        reps = int(rng.integers(0, 15)) if len(ins) < 1200 else int(rng.integers(0, 6))
        k0, k1 = int(rng.integers(0, len(ins))), int(rng.integers(0, len(ins)))
        if kind in (3, 4) and len(ins) > 300 and reps >= 2:
            # ragged repeats: some copies of the insert miss a 25-90 base chunk or carry it twice (layers that do not follow the
            # draft's diagonal: the banded window rows must fall back or certify, never differ)
            units = []
            for _r in range(reps):
                u = ins
                if rng.random() < 0.5:
                    c0 = int(rng.integers(30, len(ins) - 120)); ln = int(rng.integers(25, 90))
                    u = ins[:c0] + ins[c0 + ln:] if rng.random() < 0.5 else ins[:c0 + ln] + ins[c0:]
                units.append(synth.SPLINT1 + u)
            clean = ins[len(ins) - k0:] + "".join(units) + synth.SPLINT1 + ins[:k1]
        else:
            clean = ins[len(ins) - k0:] + (synth.SPLINT1 + ins) * reps + synth.SPLINT1 + ins[:k1]
        err = float(rng.choice([0.0, 0.03, 0.1, 0.1, 0.15, 0.25]))
        sb, qb = synth._mutate(rng, np.frombuffer(clean.encode(), dtype=np.uint8), sub=err * 0.4, ins=err * 0.25, dele=err * 0.35)
        s, q = sb.decode(), qb.decode()
        if kind == 1:
            q = "".join(chr(33 + int(v)) for v in rng.integers(0, 8, len(s)))            # very low qualities
        if kind == 2 and len(s) > 50:
            p = int(rng.integers(0, len(s) - 20)); s = s[:p] + "NNNNnnnnacgtRYKM" + s[p + 16:]
    st = "+"
    if rng.random() < 0.5:
        s, q, st = revcomp(s), q[::-1], "-"
    if rng.random() < 0.03:
        st = "?"
    reads.append((s, q)); strands.append(st)
keep = [i for i, r in enumerate(reads) if len(r[0]) > 0]
reads = [reads[i] for i in keep]; strands = [strands[i] for i in keep]
h = _lib.Handle(); h.set_splints([synth.SPLINT1])
h.upload([r[0] for r in reads], [r[1] for r in reads], strands)
h.run()
res, cons = h.results()
ores, ocons = O.process_batch(synth.SPLINT1, reads, strands, threads=16)
bad = 0
for i in range(len(reads)):
    o = ores[i]
    same = (int(res[i]["status"]) == o.status and cons[i] == ocons[i] and (o.status not in (0, 3) or
            (int(res[i]["n_sub"]) == o.n_sub and int(res[i]["n_peaks"]) == o.n_peaks)))
    if not same:
        bad += 1
        if bad <= 10:
            print("MISMATCH read %d len %d strand %s: gpu status %d n_sub %d cons %d | oracle status %d n_sub %d cons %d" % (
                i, len(reads[i][0]), strands[i], res[i]["status"], res[i]["n_sub"], len(cons[i]), o.status, o.n_sub, len(ocons[i])))
st = np.bincount(res["status"], minlength=6)
t = h.timing()
print("reads %d  mismatches %d  statuses OK/NA/NOPEAK/NOCONS/SHORT/LIMIT = %s  band layers %d fallback %d (verify mode: layers whose band and full tracebacks differ %d) computed/full cells %.3f  POA second pass %d reads (of them beyond 16-bit cells / far arena: %d)" % (
    len(reads), bad, st.tolist(), t["n_band_layers"], t["n_band_fallback"], t["n_band_mismatch"], t["cells_polish_computed"] / max(t["cells_polish"], 1),
    t["n_poa_redo"], t["n_poa_redo16"]))
if t["n_band_mismatch"]:
    sys.exit(2)
sys.exit(1 if bad else 0)
