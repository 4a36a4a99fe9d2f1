cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-rooflines > gpurun_out/c3_$tag.json 2>/dev/null; python -c "import json; d=json.loads(open('gpurun_out/c3_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"; }
run default A=1
run nosplit HIFIHR_GEMM_TN_SPLIT=0
run nolight HIFIHR_LIGHT_BRANCH=0
run nomano HIFIHR_MANO_FUSED=0
run default2 A=1
