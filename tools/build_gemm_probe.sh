#!/bin/bash
# Builds tools/_probe/libhifihr_gemm_stamp.so: libhifihr.so with gemm.hip compiled -DHIFIHR_GEMM_STAMP (tools/gemm_stamp.py).
set -eu
cd "$(dirname "$0")/../hifihr_amd/csrc"
make -s
mkdir -p ../../tools/_probe
/opt/rocm/bin/hipcc -w -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DHIFIHR_GEMM_STAMP -c gemm.hip -o ../../tools/_probe/gemm_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^gemm.o$") ../../tools/_probe/gemm_stamp.o -ldl -o ../../tools/_probe/libhifihr_gemm_stamp.so
