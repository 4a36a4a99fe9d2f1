# A/B of the layer-1 pair launch (conv_c64_bwd_pair_kernel) and its workgroup split, headline config, graph replay
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-rooflines > gpurun_out/c64_$tag.json 2>/dev/null; python -c "import json; d=json.loads(open('gpurun_out/c64_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"; }
run off HIFIHR_C64_PAIR=0
run p44 A=1
run p40 HIFIHR_C64_PAIR_DGRAD_PCT=40
run p48 HIFIHR_C64_PAIR_DGRAD_PCT=48
run off2 HIFIHR_C64_PAIR=0
run p44b A=1
run p36 HIFIHR_C64_PAIR_DGRAD_PCT=36
run p52 HIFIHR_C64_PAIR_DGRAD_PCT=52
