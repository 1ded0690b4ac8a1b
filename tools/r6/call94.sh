#!/bin/bash
# SlowFast-R50 in its two-branch form (8 clips per branch): the GEMM routing thresholds that were tuned at 16 clips per plan
mkdir -p gpurun_out/r6
for rep in 1 2; do
  for knob in "gemm9=1" "gemm9h=0" "gemm9=0" "gemm9h_below=400" "gemm9_min_tiles=60" "gemm9h_min_tiles=48" "gemm9h_tr_min_tiles=100" "gemm8=0" "tapstream=0"; do
    timeout 300 python bench.py --workload slowfast_r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slowfast_r50 $knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_slowfast_routes_call94.txt
