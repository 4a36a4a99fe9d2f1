#!/bin/bash
# VERDICT r05 item 7 (measure, do not ship): one Winograd product on split-bf16 MFMA.
#   tools/_probe/split_bf16_probe              standalone: a plain LDS-tiled kernel, bf16x3 and bf16x6, errors vs float64
#   tools/_probe/libhifihr_split_bf16.so       libhifihr.so with gemm.hip compiled -DHIFIHR_PROBE_SPLIT_BF16: the PRODUCTION row-share kernel's
#                                              loaders and schedule with bf16x3 MFMAs (nt_rows_body<3>), driven by tools/split_bf16_probe.py
set -e
cd "$(dirname "$0")"
mkdir -p _probe
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 split_bf16_probe.hip -ldl -o _probe/split_bf16_probe
cd ../hifihr_amd/csrc
make -s
/opt/rocm/bin/hipcc -w -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DHIFIHR_PROBE_SPLIT_BF16 -c gemm.hip -o ../../tools/_probe/gemm_split_bf16.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^gemm.o$") ../../tools/_probe/gemm_split_bf16.o -ldl -o ../../tools/_probe/libhifihr_split_bf16.so
