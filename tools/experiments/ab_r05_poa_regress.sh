L=c3poa_amd/lib
for v in "" _poa_r04 _poa_r04like _poa_nopunt _poa_oldcmp "" _poa_r04; do python tools/ab_slots_poa.py 32768 $L/libc3poa_hip$v.so 6144; done
export CFG=cfg4
for v in "" _poa_r04 _poa_r04like; do python tools/ab_slots_poa.py 8192 $L/libc3poa_hip$v.so 6144; done
