#!/bin/bash
# Builds tools/_probe/libhifihr_w2_ab<N>.so: libhifihr.so with conv_halo.hip compiled -DHIFIHR_W2_ABLATE=N (timing only, wrong results).
set -eu
cd "$(dirname "$0")/../hifihr_amd/csrc"
make -s
mkdir -p ../../tools/_probe
for n in "$@"; do
  /opt/rocm/bin/hipcc -w -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DHIFIHR_W2_ABLATE=$n -c conv_halo.hip -o ../../tools/_probe/w2_ab$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^conv_halo.o$") ../../tools/_probe/w2_ab$n.o -ldl -o ../../tools/_probe/libhifihr_w2_ab$n.so
done
