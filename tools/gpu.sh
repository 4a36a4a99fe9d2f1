#!/bin/bash
# Build libhifihr.so (and the hostsim emulator's library if asked) from the CURRENT sources, then hand the command to gpurun:
# a stale .so on the GPU box once cost a debugging round (the snapshot ships whatever is in the tree).
# usage: tools/gpu.sh <timeout seconds> '<command>'
set -eu
cd "$(dirname "$0")/.."
make -s -C hifihr_amd/csrc 2>&1 | grep -E "error|Error" && exit 1
make -s -C oracle >/dev/null 2>&1 || true
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
