#!/bin/bash
# The host half of the library (csrc/c3_io.cpp: FASTQ / BGZF reader, record formatter, writers, PSL table) under AddressSanitizer,
# CPU only (device code is not instrumented: GPU sanitizers are not available on this pool):  bash tools/host_asan.sh
set -e
cd "$(dirname "$0")/../c3poa_amd/csrc"
make -s -j8 >/dev/null
HIPCC=/opt/rocm/bin/hipcc
$HIPCC -O1 -g -fPIC -std=c++17 -fsanitize=address -fno-gpu-sanitize -shared-libsan -c c3_io.cpp -o build/c3_io_asan.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan -o ../lib/libc3poa_hip_hostasan.so \
  build/k_conk.o build/k_peaks.o build/k_poa.o build/k_poa_mw.o build/k_polish.o build/k_zero.o build/k_adapter.o build/c3_api.o build/c3_io_asan.o -lz
cd ../..
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 C3POA_LIB=c3poa_amd/lib/libc3poa_hip_hostasan.so \
  python -m pytest tests/test_host_io.py tests/test_stream_pipeline_cpu.py tests/test_postprocess_host.py -q -x
rm -f c3poa_amd/lib/libc3poa_hip_hostasan.so c3poa_amd/csrc/build/c3_io_asan.o
