# k_poa steady rows (round 6): "" = steady rows on (shipped), _base = -DC3_STEADY=0 (the round-5 row loop), _stc = steady rows + poa_align as a real
# call in the NARROW instances, _st8 / _stc8 = the same two at eight waves per SIMD (64 VGPRs, ring of five rows: 5 088 bytes of LDS; 8 192 slots)
L=c3poa_amd/lib
V=${V:-"_base _stc"}
for cfgn in "cfg2 32768" "cfg3 16384" "cfg4 8192"; do set -- $cfgn; export CFG=$1
for rep in 1 2; do
  for v in $V ""; do python tools/ab_slots_poa.py $2 $L/libc3poa_hip$v.so 6144; done
  for v in ${V8:-}; do python tools/ab_slots_poa.py $2 $L/libc3poa_hip$v.so 8192; done
done; done
