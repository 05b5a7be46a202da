# k_poa's consensus sweeps: pre-pass + register hand-over (consinl = inlined, consfn = as a real call) against the shipped sweeps (base)
L=c3poa_amd/lib
for rep in 1 2; do for v in base consinl consfn; do python tools/ab_slots_poa.py 32768 $L/libc3poa_hip_$v.so 6144; done; done
export CFG=cfg4
for rep in 1 2; do for v in base consinl consfn; do python tools/ab_slots_poa.py 8192 $L/libc3poa_hip_$v.so 6144; done; done
