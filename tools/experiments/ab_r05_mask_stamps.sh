# sub-graph fixpoint of k_window: stamps instead of a clearing sweep per turn (default) against the clearing sweep (-DC3_EXP_MASK_CLEAR)
L=c3poa_amd/lib
for cfg in cfg2 cfg3 cfg4; do
  n=8192; [ $cfg = cfg2 ] && n=32768; [ $cfg = cfg3 ] && n=16384
  for v in _maskclear "" _maskclear ""; do CFG=$cfg python tools/ab_slots.py $n $L/libc3poa_hip$v.so 6144; done
done
