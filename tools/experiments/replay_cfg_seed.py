"""replay reads of a tools/fuzz_parity3.py seed: python tools/experiments/replay_cfg_seed.py SEED i,j,k"""
import importlib.util, os, sys
import numpy as np
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("fz3", "tools/fuzz_parity3.py"); fz3 = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz3)
from c3poa_amd import _lib
from oracle import oracle_py as O
seed = int(sys.argv[1]); idx = [int(x) for x in sys.argv[2].split(",")]
cfg = fz3.random_config(np.random.default_rng(10_000 + seed))
splint, _md, reads, strands = fz3.fz.generate(100, 50_000 + seed)
keep = [i for i, r in enumerate(reads) if len(r[0]) < 40_000]
reads = [reads[keep[i]] for i in idx]; strands = [strands[keep[i]] for i in idx]
print(cfg, [len(r[0]) for r in reads], flush=True)
h = _lib.Handle(**cfg); h.set_splints([splint]); h.upload([r[0] for r in reads], [r[1] for r in reads], strands); h.run()
res, cons = h.results()
ores, ocons = O.process_batch(splint, reads, strands, params=O.default_params(**cfg), threads=8)
print([int(x) for x in res["status"]], [o.status for o in ores], list(cons) == list(ocons))
