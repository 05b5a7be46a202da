"""Row kinds of the POA alignments of a config, counted by the ORACLE (diagnostic, CPU only):
python tools/poa_row_stats.py cfg2 200"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import synth
from oracle import oracle_py as O

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = O.lib()
S = (C.c_int64 * 64).in_dll(lib, "c3o_poa_rowstats")
C.c_int.in_dll(lib, "c3o_poa_rowstats_on").value = 1
recs = list(synth.generate(cfg, n_reads=n))
p = O.default_params(mdistcutoff=synth.CONFIGS[cfg]["mdist"]) if cfg in synth.CONFIGS else O.default_params()
O.process_batch(synth.SPLINT1, [(r[1], r[2]) for r in recs], [r[3] for r in recs], threads=1, params=p)
s = list(S)
rows = s[0]
print("%s, %d reads: rows %d, cells/row %.1f" % (cfg, n, rows, s[1] / max(rows, 1)))
print("  fast %.1f%%  near %.1f%% (of which two chunks %.1f%%)  general %.1f%%" % (100 * s[2] / rows, 100 * s[3] / rows, 100 * s[5] / max(s[3], 1), 100 * s[4] / rows))
print("  fast rows that fail the no-wrap rule (wd + shift <= 64, lane 63 idle when the band stands): %.1f%%" % (100 * s[54] / max(s[2], 1)))
print("  fast rows by band shift <=0,1,2,3,>3: " + " ".join("%.1f%%" % (100 * x / max(s[2], 1)) for x in s[8:13]))
print("  near rows: 1 pred d=1(wide) %.1f%%, d=2 %.1f%%, d=3 %.1f%%; 2 preds (rows above) %.1f%%, other %.1f%%; 3 preds %.1f%%; 4 preds %.1f%%"
      % tuple(100 * x / max(s[3], 1) for x in s[16:23]))
print("  general rows: >4 preds %.1f%%, far pred %.1f%%, wide %.1f%%" % tuple(100 * x / max(s[4], 1) for x in s[24:27]))
print("  far predecessor distance 4-5 %.1f%% 6-7 %.1f%% 8-11 %.1f%% 12-15 %.1f%% 16+ %.1f%%" % tuple(100 * x / max(s[25], 1) for x in s[48:53]))
print("  cells of rows with a successor beyond the ring: %.2f%% with 6 ring rows, %.2f%% with 4" % (100 * s[56] / max(s[1], 1), 100 * s[57] / max(s[1], 1)))
print("  band width <=32 %.1f%% <=48 %.1f%% <=64 %.1f%% <=96 %.1f%% <=128 %.1f%% >128 %.1f%%" % tuple(100 * x / rows for x in s[28:34]))
print("  rows wider than 128: <=160 %.1f%% <=192 %.1f%% <=256 %.1f%% >256 %.1f%% (of all rows)" % tuple(100 * x / rows for x in s[58:62]))
print("  lowest real cell below its row maximum: %d; rows by that depth >-200 %.2f%% >-400 %.2f%% >-800 %.2f%% >-1600 %.3f%% below %.4f%%"
      % ((s[40],) + tuple(100 * x / rows for x in s[41:46])))
print("  largest step of the row maximum between consecutive rows: %d" % s[47])
