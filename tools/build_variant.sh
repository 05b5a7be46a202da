#!/bin/bash
# A/B builds of the library with extra -D flags on one kernel file: tools/build_variant.sh NAME FILE.hip "-DFLAG ..."  ->  c3poa_amd/lib/libc3poa_hip_NAME.so
set -e
cd "$(dirname "$0")/../c3poa_amd/csrc"
NAME=$1; FILE=$2; FLAGS=$3
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-variable $FLAGS -c $FILE -o build/var_${NAME}_${FILE%.hip}.o
OBJ=$(ls build/k_conk.o build/k_peaks.o build/k_poa.o build/k_poa_mw.o build/k_polish.o build/k_zero.o build/k_adapter.o build/c3_api.o build/c3_io.o | grep -v "build/${FILE%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libc3poa_hip_${NAME}.so $OBJ build/var_${NAME}_${FILE%.hip}.o -lz
echo built ../lib/libc3poa_hip_${NAME}.so
