import sys; sys.path.insert(0, ".")
from c3poa_amd import _lib, synth
for k in (108, 30):
    cfg = dict(synth.CONFIGS["cfg2"]); cfg["k0"] = cfg["k1"] = k
    recs = list(synth.generate(cfg, n_reads=2048)) * 8
    h = _lib.Handle(); h.set_splints([synth.SPLINT1])
    h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs])
    best = None
    for _ in range(4):
        h.run(); t = h.timing(); best = t if best is None else {a: min(b, best[a]) if isinstance(b, float) else b for a, b in t.items()}
    res, _ = h.results(with_consensus=False)
    print(k, "front/tail:", int(res["has_front"].sum()), int(res["has_tail"].sum()), {a: round(b, 2) for a, b in best.items() if a.startswith("ms_")})
    h.close()
