# occupancy curves of the two DP kernels (shipped build): time against resident wave slots (256 CUs x 4 SIMDs x waves)
L=c3poa_amd/lib/libc3poa_hip.so
python tools/ab_slots_poa.py 32768 $L 3072 4096 5120 6144
python tools/ab_slots.py 32768 $L 3072 4096 5120
CFG=cfg4 python tools/ab_slots_poa.py 8192 $L 4096 5120 6144
