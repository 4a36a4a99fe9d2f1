#!/bin/bash
# tools/_probe/libhifihr_prev.so = the library built from the csrc of a git revision (default HEAD), for tools/r06_ab_lib.sh
set -eu
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
rm -rf /tmp/_prev_src && mkdir -p /tmp/_prev_src/hifihr_amd /tmp/_prev_src/include
git archive $rev hifihr_amd/csrc include | tar -x -C /tmp/_prev_src
make -s -C /tmp/_prev_src/hifihr_amd/csrc 2>&1 | grep -E "error" && exit 1
mkdir -p tools/_probe
cp /tmp/_prev_src/hifihr_amd/libhifihr.so tools/_probe/libhifihr_prev.so
echo built tools/_probe/libhifihr_prev.so from $rev
