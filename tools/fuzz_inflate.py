#!/usr/bin/env python3
"""Differential fuzz of the reader's DEFLATE decoder (csrc/c3_inflate.hpp) against zlib: random payloads of seven kinds, levels 0-9, every
strategy, windows of 512 bytes to 32 KiB, memory levels 1-9, flushes in mid-stream, decoded in one piece and through chunks of 1 byte to 1 MiB;
then random bytes and bit-flipped streams, which must end in an error or some output -- never in a crash.  CPU only.
    python tools/fuzz_inflate.py SEED CASES"""
import ctypes as C, zlib, random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from c3poa_amd import _lib
lib=_lib.load(); lib.c3_debug_inflate.restype=C.c_long; lib.c3_debug_inflate.argtypes=[C.c_char_p,C.c_size_t,C.c_char_p,C.c_size_t,C.c_size_t]
rng=random.Random(int(sys.argv[1])); nprng=np.random.default_rng(int(sys.argv[1]))
def gen():
    kind=rng.randrange(7); n=rng.choice([0,1,2,3,7,100,1000,5000,33000,70000,200000])
    if kind==0: return bytes(nprng.integers(0,256,n,dtype=np.uint8))
    if kind==1: return bytes(nprng.integers(0,rng.choice([2,3,4,16]),n,dtype=np.uint8))
    if kind==2: return (b"".join(rng.choice([b"ACGT",b"AAAA",b"GATTACA",b"\n@read\n",b"IIII5555"]) for _ in range(n//4+1)))[:n]
    if kind==3:
        base=bytes(nprng.integers(65,70,max(1,n//5),dtype=np.uint8)); out=bytearray()
        while len(out)<n:
            out+=base
            if rng.random()<0.5 and len(out)>10: out[rng.randrange(len(out))]=rng.randrange(256)
        return bytes(out[:n])
    if kind==4: return bytes([rng.randrange(256)])*n
    if kind==5:
        p=rng.randrange(1,9); pat=bytes(nprng.integers(0,256,p,dtype=np.uint8)); return (pat*(n//p+1))[:n]
    blk=bytes(nprng.integers(0,256,32768,dtype=np.uint8)); return (blk*4)[:max(n,40000)]
cases=0
for it in range(int(sys.argv[2])):
    data=gen()
    level=rng.randrange(0,10); strat=rng.choice([zlib.Z_DEFAULT_STRATEGY,zlib.Z_FIXED,zlib.Z_HUFFMAN_ONLY,zlib.Z_RLE,zlib.Z_FILTERED]); wb=rng.randrange(9,16); ml=rng.randrange(1,10)
    co=zlib.compressobj(level,zlib.DEFLATED,-wb,ml,strat)
    if rng.random()<0.3 and len(data)>10:
        k=rng.randrange(1,len(data)); raw=co.compress(data[:k])+co.flush(rng.choice([zlib.Z_SYNC_FLUSH,zlib.Z_FULL_FLUSH]))+co.compress(data[k:])+co.flush()
    else: raw=co.compress(data)+co.flush()
    for chunk in (0, rng.choice([1,7,300,4096,40000,1<<16,1<<20])):
        out=C.create_string_buffer(len(data)+1)
        n=lib.c3_debug_inflate(raw,len(raw),out,len(data),chunk)
        assert n==len(data) and out.raw[:n]==data,(it,len(data),level,strat,wb,chunk,n)
        cases+=1
print('ok',cases)

# damaged input: random bytes, and valid streams with a flipped bit or cut short -- any result but a crash
data = gen()
bad = 0
for it in range(int(sys.argv[2]) * 5):
    if it % 3 == 0:
        raw = bytes(nprng.integers(0, 256, rng.randrange(1, 400), dtype=np.uint8))
    else:
        raw = bytearray(zlib.compress(gen() or b"x", rng.randrange(1, 10))[2:-4])
        if len(raw) > 1:
            raw[rng.randrange(len(raw))] ^= 1 << rng.randrange(8)
        raw = bytes(raw[:rng.randrange(1, len(raw) + 1)])
    out = C.create_string_buffer(300000)
    n = lib.c3_debug_inflate(raw, len(raw), out, 299000, rng.choice([0, 4096, 1 << 16]))
    bad += n < 0
print('damaged streams survived:', int(sys.argv[2]) * 5, 'errors reported:', bad)
