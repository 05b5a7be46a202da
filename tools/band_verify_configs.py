import sys, os
sys.path.insert(0, ".")
os.environ["C3_DEBUG_BAND"] = "verify"
from c3poa_amd import _lib, synth
for cfg, n in (("cfg2", 20000), ("cfg3", 12000), ("cfg4", 4000), ("cfg2e15", 8000), ("cfg2e20", 8000)):
    recs = list(synth.generate(cfg, n_reads=min(n, 4000))) * (n // min(n, 4000))
    h = _lib.Handle(mdistcutoff=synth.CONFIGS[cfg]["mdist"]); h.set_splints([synth.SPLINT1])
    h.upload([r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs]); h.run(); t = h.timing(); h.close()
    print(cfg, "reads", len(recs), "band layers", t["n_band_layers"], "fallback", t["n_band_fallback"], "layers whose band and full tracebacks differ", t["n_band_mismatch"])
