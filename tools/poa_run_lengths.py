"""Run lengths of consecutive fast rows in the POA alignments of a config, counted by the ORACLE (diagnostic, CPU only; round 6,
review item 1c): how many rows lie in runs of k consecutive fast rows (one predecessor = the row above, band within 64 lanes),
and in runs whose band moved by exactly one column at both ends -- the prediction a skewed multi-row step (lane = row) needs.
    python tools/poa_run_lengths.py cfg2 200"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import synth
from oracle import oracle_py as O

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = O.lib()
S = (C.c_int64 * 64).in_dll(lib, "c3o_poa_rowstats")
R = (C.c_int64 * 192).in_dll(lib, "c3o_poa_runstats")
C.c_int.in_dll(lib, "c3o_poa_rowstats_on").value = 1
recs = list(synth.generate(cfg, n_reads=n))
p = O.default_params(mdistcutoff=synth.CONFIGS[cfg]["mdist"]) if cfg in synth.CONFIGS else O.default_params()
O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs], threads=1, params=p)
r = list(R); rows = S[0]


def summary(h, name):
    tot = sum(h)
    if not tot:
        print("  %s: none" % name); return
    acc, med = 0, None
    for k, x in enumerate(h):
        acc += x
        if med is None and acc * 2 >= tot:
            med = k + 1
    ge = lambda m: 100.0 * sum(h[m - 1:]) / tot
    print("  %s: %d rows (%.1f%% of all rows); row-weighted median run %s%d; rows in runs >=8 %.1f%%  >=16 %.1f%%  >=32 %.1f%%  >=64 %.1f%%"
          % (name, tot, 100.0 * tot / rows, ">=" if med == 64 else "", med, ge(8), ge(16), ge(32), ge(64)))
    print("    rows by run length 1..16: " + " ".join("%.1f" % (100.0 * x / tot) for x in h[:16]))


print("%s, %d reads, %d rows" % (cfg, n, rows))
summary(r[0:64], "runs of fast rows")
summary(r[64:128], "runs of fast rows whose band moved (+1, +1)")
fast = r[128] + r[129]
print("  fast rows with band step (+1,+1): %.1f%%; others by (dbeg: 1,0,2,other) x (dend: 1,0,2,other), %% of fast rows:" % (100.0 * r[128] / max(fast, 1)))
for a in range(4):
    print("    " + " ".join("%5.1f" % (100.0 * r[130 + 4 * a + b] / max(fast, 1)) for b in range(4)))
