#!/bin/bash
# Builds tools/_probe/libhifihr_halo_stamp.so: libhifihr.so with conv_halo.hip compiled -DHIFIHR_HALO_STAMP (tools/halo_stamp.py).
set -eu
cd "$(dirname "$0")/../hifihr_amd/csrc"
make -s
mkdir -p ../../tools/_probe
/opt/rocm/bin/hipcc -w -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DHIFIHR_HALO_STAMP -c conv_halo.hip -o ../../tools/_probe/halo_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^conv_halo.o$") ../../tools/_probe/halo_stamp.o -ldl -o ../../tools/_probe/libhifihr_halo_stamp.so
