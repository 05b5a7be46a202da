# host pipeline ceiling (GPU stand-in) with the core-slot limiter (default) and without it (C3_HOST_THREADS=1000 = the round-4 behaviour)
python tools/host_ceiling.py 2000000 --workers 1,2,4,8 > gpurun_out/r05_host_ceiling_slots.txt 2> gpurun_out/r05_host_ceiling_slots.err
C3_HOST_THREADS=1000 python tools/host_ceiling.py 2000000 --workers 1,8 > gpurun_out/r05_host_ceiling_noslots.txt 2> gpurun_out/r05_host_ceiling_noslots.err
cat gpurun_out/r05_host_ceiling_slots.txt gpurun_out/r05_host_ceiling_noslots.txt
