#!/bin/bash
# Everything profiles/r03_* is made of, in one GPU-box call (through gpurun): bash tools/collect_r03.sh [quick]
#   -> gpurun_out/r03/*  (copied into profiles/ by hand after a look)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
last() { tail -1 "$1" > "$2"; }
python3 bench.py > $O/bench_default.log 2>&1; last $O/bench_default.log $O/r03_bench_default.json
python3 bench.py --config 3 --steps 15 --warmup 3 > $O/b3.log 2>&1; last $O/b3.log $O/r03_bench_cfg3.json
python3 bench.py --config 5 --steps 15 --warmup 3 > $O/b5.log 2>&1; last $O/b5.log $O/r03_bench_cfg5.json
python3 bench.py --encoder res50 --steps 15 --warmup 3 --no-cpu-baseline > $O/b50.log 2>&1; last $O/b50.log $O/r03_bench_res50.json
python3 bench.py --config 3 --hand nimble-synthetic-uv --steps 15 --warmup 3 > $O/b3n.log 2>&1; last $O/b3n.log $O/r03_bench_cfg3_nimble_uv.json
python3 bench.py --config 5 --hand nimble-synthetic-uv --steps 15 --warmup 3 > $O/b5n.log 2>&1; last $O/b5n.log $O/r03_bench_cfg5_nimble_uv.json
python3 bench.py --config 5 --hand nimble-synthetic-uv --aa 1 --steps 15 --warmup 3 > $O/b5n1.log 2>&1; last $O/b5n1.log $O/r03_bench_cfg5_nimble_uv_aa1.json
for spec in "res18:" "cfg3:--config 3" "cfg5:--config 5" "res50:--encoder res50" "cfg3_nimble_uv:--config 3 --hand nimble-synthetic-uv"; do
  tag=${spec%%:*}; args=${spec#*:}
  bash tools/profile_bench.sh r03_$tag $args > /dev/null 2>&1
  cp gpurun_out/prof_r03_$tag/steady.md $O/r03_steady_state_$tag.md
  cp gpurun_out/prof_r03_$tag/kernel_stats.csv $O/r03_kernel_stats_$tag.csv
  [ "$tag" = res18 ] && cp gpurun_out/prof_r03_$tag/timeline.txt $O/r03_step_timeline_res18.txt
done
bash tools/kernel_traffic.sh > /dev/null 2>&1; cp gpurun_out/r03_kernel_traffic.json $O/ 2>/dev/null
bash tools/pmc_probe.sh render "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" render_only.py > $O/r03_pmc_render_sq_counters.txt 2>&1
python3 tools/time_render.py 2>&1 | grep -v amdgpu.ids > $O/r03_time_render.txt
python3 tools/time_wino_bn.py 2>&1 | grep "H =" > $O/r03_time_wino_bn.txt
python3 tools/time_conv_wino2.py 2>&1 | grep -v amdgpu.ids > $O/r03_time_conv_wino2.txt
if [ -f tools/_probe/libhifihr_halo_stamp.so ]; then python3 tools/wino2_stamp.py 2>&1 | grep -v amdgpu.ids > $O/r03_wino2_stamps.txt; fi
if [ -f tools/_probe/libhifihr_render_stamp2.so ]; then python3 tools/render_stamp2.py 2>&1 | grep -v amdgpu.ids > $O/r03_render_fwd_phase_stamps.txt; fi
if [ "${1:-}" != quick ]; then bash tools/ablation.sh > $O/r03_ablation.txt 2>&1; fi
ls -la $O | tail -40
