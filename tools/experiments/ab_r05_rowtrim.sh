# k_window's banded fast rows, trimmed by hand: (rowtrim) 16-bit DPP scan, alignbit direction bits, counting edge address; (runloop) + runs of
# shift-1 fast rows as a loop of their own (no register copies at the merge of the row kinds), against the build before both (base)
L=c3poa_amd/lib
for cfg in cfg2 cfg3 cfg4; do
  n=8192; [ $cfg = cfg2 ] && n=32768; [ $cfg = cfg3 ] && n=16384
  for v in _base _rowtrim _rowtrim2 _base _rowtrim _rowtrim2; do CFG=$cfg python tools/ab_slots.py $n $L/libc3poa_hip$v.so 6144; done
done
