# k_window at 6 waves per SIMD with the LDS inside FIVE allocation granules of 1 280 bytes (6 400): 24 waves per CU really fit
L=c3poa_amd/lib
export C3_DEBUG_WIN_LDS=6400
python tools/ab_slots.py 32768 $L/libc3poa_hip_win5_l64.so 5120
python tools/ab_slots.py 32768 $L/libc3poa_hip_win6nw_l64.so 5120 5632 6144
unset C3_DEBUG_WIN_LDS
python tools/ab_slots.py 32768 $L/libc3poa_hip.so 5120
export C3_DEBUG_WIN_LDS=6400
export CFG=cfg4
python tools/ab_slots.py 8192 $L/libc3poa_hip_win5_l64.so 5120
python tools/ab_slots.py 8192 $L/libc3poa_hip_win6nw_l64.so 5120 6144
export CFG=cfg3
python tools/ab_slots.py 16384 $L/libc3poa_hip_win5_l64.so 5120
python tools/ab_slots.py 16384 $L/libc3poa_hip_win6nw_l64.so 6144
