# compiler-flag lottery for the two DP kernels (one flag set per variant library, built by /tmp/build_flag_variants.sh)
L=c3poa_amd/lib
python tools/ab_slots_poa.py 32768 $L/libc3poa_hip.so 6144
for v in ilp memclause iter_minreg bias100 bias0 nopostsched prealloc_sgpr trackers O2 aa; do python tools/ab_slots_poa.py 32768 $L/libc3poa_hip_k_poa_$v.so 6144; done
python tools/ab_slots_poa.py 32768 $L/libc3poa_hip.so 6144
python tools/ab_slots.py 32768 $L/libc3poa_hip.so 6144
for v in ilp memclause iter_minreg bias100 bias0 nopostsched prealloc_sgpr trackers O2 aa; do python tools/ab_slots.py 32768 $L/libc3poa_hip_k_polish_$v.so 6144; done
python tools/ab_slots.py 32768 $L/libc3poa_hip.so 6144
