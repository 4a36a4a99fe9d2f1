#!/bin/bash
# Builds ablation variants of libhifihr.so (conv.hip with -DHIFIHR_CONV_PROBE=n) into tools/_probe/ for tools/conv_ablate.py.
set -eu
cd "$(dirname "$0")/../hifihr_amd/csrc"
make -s
mkdir -p ../../tools/_probe
for n in 1 2 3 4 -1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DHIFIHR_CONV_PROBE=$n -c conv.hip -o ../../tools/_probe/conv_p$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^conv.o$") ../../tools/_probe/conv_p$n.o -o ../../tools/_probe/libhifihr_p$n.so
done
ls -la ../../tools/_probe
