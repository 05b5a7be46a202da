# k_poa's near rows of up to 128 columns: two columns per lane (WIDE instance, shipped build) against two chunks of 64 (_nopair = the build before)
L=c3poa_amd/lib
export CFG=cfg4
for rep in 1 2 3; do for v in _nopair ""; do python tools/ab_slots_poa.py 8192 $L/libc3poa_hip$v.so 6144; done; done
export CFG=cfgL
for rep in 1 2; do for v in _nopair ""; do python tools/ab_slots_poa.py 8192 $L/libc3poa_hip$v.so 6144; done; done
export CFG=cfg2
for rep in 1; do for v in _nopair ""; do python tools/ab_slots_poa.py 32768 $L/libc3poa_hip$v.so 6144; done; done
