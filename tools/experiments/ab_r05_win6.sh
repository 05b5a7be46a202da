L=c3poa_amd/lib
python tools/ab_slots.py 32768 $L/libc3poa_hip.so 5120
python tools/ab_slots.py 32768 $L/libc3poa_hip_win5nw.so 5120
python tools/ab_slots.py 32768 $L/libc3poa_hip_win6nw.so 6144 5120
python tools/ab_slots.py 32768 $L/libc3poa_hip_win6nw_wu2.so 6144
python tools/ab_slots.py 32768 $L/libc3poa_hip_win6.so 6144
python tools/ab_slots.py 32768 $L/libc3poa_hip.so 5120
export CFG=cfg4
python tools/ab_slots.py 8192 $L/libc3poa_hip.so 5120
python tools/ab_slots.py 8192 $L/libc3poa_hip_win6nw.so 5632 6144
python tools/ab_slots.py 8192 $L/libc3poa_hip_win6nw_wu2.so 5632
python tools/ab_slots.py 8192 $L/libc3poa_hip_win6.so 5632
