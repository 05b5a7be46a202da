# k_window's band traceback: the walk's row / column made scalar at the top of every block and step (the bookkeeping moves to the scalar unit),
# against the same source without it (-DC3_EXP_TB_VEC)
L=c3poa_amd/lib
for cfg in cfg2 cfg3 cfg4; do
  n=8192; [ $cfg = cfg2 ] && n=32768; [ $cfg = cfg3 ] && n=16384
  for v in _tbvec _tbuni _tbvec _tbuni; do CFG=$cfg python tools/ab_slots.py $n $L/libc3poa_hip$v.so 6144; done
done
