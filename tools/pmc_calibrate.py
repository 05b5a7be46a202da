#!/usr/bin/env python3
"""Runs the dword read/write calibration kernel (1 GiB each way) so rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
can be compared with a known byte count on this device (MI355X_MICROARCH.md, HBM section)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c3poa_amd import _lib
h = _lib.Handle()
h.lib.c3_debug_calibrate.argtypes = [C.c_void_p, C.c_longlong]
assert h.lib.c3_debug_calibrate(h.h, 1 << 30) == 0
print("calibration kernel done: 1073741824 bytes read, 1073741824 bytes written")
