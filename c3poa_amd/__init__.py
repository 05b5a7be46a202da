"""c3poa_amd -- MI355X-native R2C2 consensus hot path (HIP kernels behind a C ABI).

Host-side mirror of the reference's call shapes lives in the sub-modules; the compute is in
c3poa_amd/lib/libc3poa_hip.so (sources: c3poa_amd/csrc, ABI: include/c3poa.h).
"""
__version__ = "0.1.0"
VERSION = "v2.2.3"   # C3POa version whose CLI / output tree is mirrored (C3POa.py:24)
