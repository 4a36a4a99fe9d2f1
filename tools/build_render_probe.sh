#!/bin/bash
# libhifihr.so with render.hip compiled -DHIFIHR_RENDER_STAMP (per-tile cycle stamps) into tools/_probe/ for tools/render_stamp.py
set -eu
cd "$(dirname "$0")/../hifihr_amd/csrc"
make -s
mkdir -p ../../tools/_probe
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DHIFIHR_RENDER_STAMP -c render.hip -o ../../tools/_probe/render_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^render.o$") ../../tools/_probe/render_stamp.o -ldl -o ../../tools/_probe/libhifihr_render_stamp.so
