#!/bin/bash
# same-box A/B of bench.py under an environment switch: bash tools/r06_ab.sh <tag> <ENVVAR> [runs]   (on the GPU box, through gpurun)
# -> gpurun_out/r06/<tag>_ab.txt: ms/step with ENVVAR=0 and ENVVAR=1, alternating, `runs` times each
set -u
cd $GRAFT_REPO_ROOT
tag=$1; var=$2; runs=${3:-2}
O=gpurun_out/r06; mkdir -p $O
: > $O/${tag}_ab.txt
for i in $(seq 1 $runs); do
  for v in 0 1; do
    env $var=$v python3 bench.py --no-cpu-baseline --no-rooflines --steps 40 --warmup 10 > $O/_ab.log 2>&1
    ms=$(tail -1 $O/_ab.log | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo FAIL)
    echo "$var=$v run $i: $ms ms/step" >> $O/${tag}_ab.txt
  done
done
cat $O/${tag}_ab.txt
