# k_window (shipped build): consensus pointer doubling with selects instead of nested branches, graph-loop loads unconditional at clamped
# indices (w_blocks, w_reorder, compaction, descriptor build, fusion) -- against the build before (_base)
L=c3poa_amd/lib
for cfg in cfg2 cfg3 cfg4; do
  n=32768; [ $cfg = cfg3 ] && n=16384; [ $cfg = cfg4 ] && n=8192
  for rep in 1 2; do for v in _base ""; do CFG=$cfg python tools/ab_slots.py $n $L/libc3poa_hip$v.so 6144; done; done
done
