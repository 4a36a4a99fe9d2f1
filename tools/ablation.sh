#!/bin/bash
# What each dispatch decision buys on the FINAL tree: bench.py (BASELINE configs[1], hipGraph replay, 30 steps) with one switch off at a time.
# usage (GPU box, through gpurun): bash tools/ablation.sh > gpurun_out/r03_ablation.txt
cd $GRAFT_REPO_ROOT
run() {   # label, then VAR=value assignments / bench flags
  label=$1; shift
  envs=(); flags=()
  for a in "$@"; do if [[ $a == --* || $a =~ ^[0-9]+$ ]]; then flags+=("$a"); else envs+=("$a"); fi; done
  ms=$(env "${envs[@]}" python3 bench.py --no-cpu-baseline --no-rooflines --steps 30 --warmup 5 "${flags[@]}" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.0f img/s' % (d['ms_per_step'], d['value']))")
  printf "%-58s %s\n" "$label" "$ms"
}
run "default (final tree)"
run "batch-norm NOT fused into the Winograd transforms (HIFIHR_BN_WINO_FUSE=0)" HIFIHR_BN_WINO_FUSE=0
run "layer 1 on the direct halo kernel, not the one-launch Winograd (HIFIHR_CONV_WINO2=0)" HIFIHR_CONV_WINO2=0
run "batch-norm backward unfused, forward fused (HIFIHR_BN_WINO_BWD=0)" HIFIHR_BN_WINO_BWD=0
run "rasteriser forward on 16x16 tiles (HIFIHR_RENDER_TILE=16)" HIFIHR_RENDER_TILE=16
run "stem BN+ReLU+max-pool unfused (HIFIHR_BN_POOL=0)" HIFIHR_BN_POOL=0
run "1x1 convolutions on the implicit GEMM (HIFIHR_CONV1X1_GEMM=0)" HIFIHR_CONV1X1_GEMM=0
run "TN row-share kernel off (HIFIHR_GEMM_TN_ROWS=0)" HIFIHR_GEMM_TN_ROWS=0
run "NT row-share kernel off (HIFIHR_GEMM_ROWS=0)" HIFIHR_GEMM_ROWS=0
run "NT row-share kernel off + 1x1 on the implicit GEMM" HIFIHR_GEMM_ROWS=0 HIFIHR_CONV1X1_GEMM=0
run "layer-1 halo kernel off (HIFIHR_CONV_HALO=0)" HIFIHR_CONV_HALO=0
run "layer-1 halo weight gradient off (HIFIHR_CONV_HALO_WGRAD=0)" HIFIHR_CONV_HALO_WGRAD=0
run "Winograd F(2x2) instead of F(4x4) (HIFIHR_WINO_M=2)" HIFIHR_WINO_M=2
run "no Winograd at all (HIFIHR_WINOGRAD=0)" HIFIHR_WINOGRAD=0
run "stem kernel off (HIFIHR_CONV_STEM=0)" HIFIHR_CONV_STEM=0
run "plain F(4x4) tile geometry, no mosaic (HIFIHR_WINO_MOSAIC=0)" HIFIHR_WINO_MOSAIC=0
run "F(4x4) weight-gradient transform of small layers, narrow form (HIFIHR_WINO_DW_WIDE=0)" HIFIHR_WINO_DW_WIDE=0
run "strided forward convolutions on the implicit GEMM (HIFIHR_CONV_ROWS=0)" HIFIHR_CONV_ROWS=0
run "light estimator on the main stream, no side branch (HIFIHR_LIGHT_BRANCH=0)" HIFIHR_LIGHT_BRANCH=0
run "geometry loss terms on a side branch too (HIFIHR_GEOM_BRANCH=1)" HIFIHR_GEOM_BRANCH=1
run "MANO layer / joint regression as separate autograd nodes (HIFIHR_MANO_FUSED=0)" HIFIHR_MANO_FUSED=0
run "batched TN products with short reductions on the per-tile kernels (HIFIHR_GEMM_TN_SPLIT=0)" HIFIHR_GEMM_TN_SPLIT=0
run "F(4x4) backward-data and backward-weight products as two launches (HIFIHR_GEMM_PAIR=0)" HIFIHR_GEMM_PAIR=0
run "layer 1 data gradient and weight gradient as two launches (HIFIHR_C64_PAIR=0)" HIFIHR_C64_PAIR=0
run "eager launches, no hipGraph (--graph 0)" --graph 0
run "round 6: stem batch-norm reduction over every input pixel (HIFIHR_STEM_REDUCE_Y=0)" HIFIHR_STEM_REDUCE_Y=0
run "round 6: weight-gradient transforms / slab sums per layer, not deferred (HIFIHR_DEFER_DW=0)" HIFIHR_DEFER_DW=0
run "round 6: the deferred launch at the scope exit, not beside the stem backward (HIFIHR_DEFER_EARLY=0)" HIFIHR_DEFER_EARLY=0
run "round 6: TN products walk the zero rows behind the tile mosaic (HIFIHR_GEMM_TN_SKIP=0)" HIFIHR_GEMM_TN_SKIP=0
run "round 6: strided 3x3 + downsample 1x1 as two forward launches (HIFIHR_CONV_ROWS_PAIR=0)" HIFIHR_CONV_ROWS_PAIR=0
run "round 6: the downsample 1x1 data gradient as a launch + residual, not a tap of the strided 3x3 launch (HIFIHR_DGRAD_PLUS1X1=0)" HIFIHR_DGRAD_PLUS1X1=0
run "round 6: the downsample 1x1 weight gradient as a launch of its own (HIFIHR_WGRAD_PLUS1X1=0)" HIFIHR_WGRAD_PLUS1X1=0
run "round 6: TN products on contiguous shares, not XCD-coherent (HIFIHR_GEMM_TN_COHERENT=0)" HIFIHR_GEMM_TN_COHERENT=0
