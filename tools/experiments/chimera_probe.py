import sys, numpy as np
sys.path.insert(0, ".")
from c3poa_amd import _lib, synth
from oracle import oracle_py as O
rng = np.random.default_rng(5)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
rnd = lambda L: acgt[rng.integers(0, 4, L)].tobytes().decode()
sp = synth.SPLINT1
for nd, cp, L in ((4, 3, 1000), (6, 2, 1000), (8, 1, 900), (5, 2, 2500), (12, 1, 600)):
    inserts = [rnd(L + int(rng.integers(-50, 50))) for _ in range(nd)]
    units = [sp + inserts[k % nd] for k in range(nd * cp)]
    clean = rnd(150) + "".join(units) + sp + rnd(150)
    sb, qb = synth._mutate(rng, np.frombuffer(clean.encode(), dtype=np.uint8), sub=0.03, ins=0.02, dele=0.02)
    s, q = sb.decode(), qb.decode()
    h = _lib.Handle(); h.set_splints([sp]); h.upload([s], [q], ["+"]); h.run()
    res, cons = h.results(); t = h.timing()
    ores, ocons = O.process_batch(sp, [(s, q)], ["+"], threads=1)
    print(nd, cp, L, "gpu", int(res[0]["status"]), int(res[0]["n_sub"]), len(cons[0]), "| oracle", ores[0].status, ores[0].n_sub, len(ocons[0]), "same" if cons[0] == ocons[0] and int(res[0]["status"]) == ores[0].status else "DIFFERENT", "poa redo", t["n_poa_redo"], t["n_poa_redo16"])
    h.close()
