#!/bin/bash
# AddressSanitizer over the kernel sources on the CPU emulator: bash tools/asan_hostsim.sh [conv|render|wino ...]
# (GPU ASan / xnack+ builds are not available on the pool; ASan warns about swapcontext -- the emulator's fibers -- once, harmlessly.)
set -eu
cd "$(dirname "$0")/.."
make -s -C tests/hostsim -j8 asan
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python3 tools/asan_hostsim.py "$@"
