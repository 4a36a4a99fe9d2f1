#!/bin/bash
# same-box A/B of bench.py between two BUILDS of the library: bash tools/r06_ab_lib.sh <tag> <path of the other libhifihr.so> [runs]
# (on the GPU box, through gpurun; the other build is made beforehand, e.g. tools/build_prev_lib.sh = the csrc of git HEAD)
# -> gpurun_out/r06/<tag>_ab.txt: ms/step with the other build ("prev") and with the tree's ("this"), alternating
set -u
cd $GRAFT_REPO_ROOT
tag=$1; other=$2; runs=${3:-2}
O=gpurun_out/r06; mkdir -p $O
: > $O/${tag}_ab.txt
cp hifihr_amd/libhifihr.so /tmp/_this.so
for i in $(seq 1 $runs); do
  for v in prev this; do
    if [ $v = prev ]; then cp $other hifihr_amd/libhifihr.so; else cp /tmp/_this.so hifihr_amd/libhifihr.so; fi
    python3 bench.py --no-cpu-baseline --no-rooflines --steps 40 --warmup 10 > $O/_ab.log 2>&1
    ms=$(tail -1 $O/_ab.log | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo FAIL)
    echo "$v run $i: $ms ms/step" >> $O/${tag}_ab.txt
  done
done
cp /tmp/_this.so hifihr_amd/libhifihr.so
cat $O/${tag}_ab.txt
