# k_poa with poa_align as a real call (uniform copies of the slot pointers at entry; shipped build of this experiment) against the inlined one (_base)
L=c3poa_amd/lib
for rep in 1 2; do for v in _base ""; do python tools/ab_slots_poa.py 32768 $L/libc3poa_hip$v.so 6144; done; done
export CFG=cfg4
for rep in 1 2; do for v in _base ""; do python tools/ab_slots_poa.py 8192 $L/libc3poa_hip$v.so 6144; done; done
export CFG=cfgL
for rep in 1; do for v in _base ""; do python tools/ab_slots_poa.py 8192 $L/libc3poa_hip$v.so 6144; done; done
