# k_conk with the score lookups as LDS byte reads (five vector instructions per cell) against the build before (_conk6: six, v_bfe_i32 from a register table); k_peaks: four outputs per thread in the smoothing passes
L=c3poa_amd/lib
for rep in 1 2; do for v in _conk6 ""; do C3POA_LIB=$L/libc3poa_hip$v.so python tools/pc_sample_driver.py 32768 cfg2 4 | sed "s/^/libc3poa_hip$v cfg2 /"; done; done
for v in _conk6 ""; do C3POA_LIB=$L/libc3poa_hip$v.so python tools/pc_sample_driver.py 8192 cfg4 3 | sed "s/^/libc3poa_hip$v cfg4 /"; done
