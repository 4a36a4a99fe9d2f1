#!/bin/bash
# libhifihr.so with render.hip compiled -DHIFIHR_RENDER_STAMP2 (per-phase cycle sums of the v2 forward tile kernel) into tools/_probe/
set -eu
cd "$(dirname "$0")/../hifihr_amd/csrc"
make -s
mkdir -p ../../tools/_probe
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DHIFIHR_RENDER_STAMP2 -c render.hip -o ../../tools/_probe/render_stamp2.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^render.o$") ../../tools/_probe/render_stamp2.o -ldl -o ../../tools/_probe/libhifihr_render_stamp2.so
