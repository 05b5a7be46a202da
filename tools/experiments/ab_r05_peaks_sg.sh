# k_peaks variants: smoothing passes with one output per thread (old) / four per thread as real calls (new) / four per thread inlined (inl, inl5 = at most 96 registers)
L=c3poa_amd/lib
for rep in 1 2; do for v in old_1 inl5 inl5b; do C3POA_LIB=$L/libc3poa_hip_pk_$v.so python tools/pc_sample_driver.py 32768 cfg2 4 | sed "s/^/pk_$v cfg2 /" | cut -c1-100; done; done
