# k_poa, poa_align as a real call: _base = WIDE instance, Ctx by reference (shipped before this run); "" = WIDE instance, Ctx by value;
# _calln = every non-32-bit instance, Ctx by value (-DC3_EXP_CALL_NARROW=1)
L=c3poa_amd/lib
for rep in 1 2; do for v in _base "" _calln; do python tools/ab_slots_poa.py 32768 $L/libc3poa_hip$v.so 6144; done; done
export CFG=cfg4
for rep in 1 2; do for v in _base _calln; do python tools/ab_slots_poa.py 8192 $L/libc3poa_hip$v.so 6144; done; done
export CFG=cfgL
for rep in 1 2; do for v in _base ""; do python tools/ab_slots_poa.py 8192 $L/libc3poa_hip$v.so 6144; done; done
