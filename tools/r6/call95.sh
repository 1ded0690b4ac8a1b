#!/bin/bash
# the half-height eight-phase kernel's minimum tile count (default 96) in the two-branch forms of SlowFast-R50 and MViT-B, and in SlowFast's one-plan form
mkdir -p gpurun_out/r6
for rep in 1 2; do
  for knob in "gemm9h_min_tiles=96" "gemm9h_min_tiles=64" "gemm9h_min_tiles=48" "gemm9h_min_tiles=32" "gemm9h_min_tiles=16" "gemm9h_min_tiles=32,gemm9_min_tiles=60" "gemm9h_min_tiles=32,gemm9h_tr_min_tiles=100"; do
    timeout 300 python bench.py --workload slowfast_r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slowfast_r50 $knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
  for knob in "gemm9h_min_tiles=96" "gemm9h_min_tiles=32"; do
    timeout 300 python bench.py --workload slowfast_r50 --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slowfast_r50 streams 1 $knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
  for knob in "gemm9h_min_tiles=96" "gemm9h_min_tiles=48" "gemm9h_min_tiles=32" "gemm9h_min_tiles=32,gemm9h_tr_min_tiles=100" "gemm9_min_tiles=60"; do
    timeout 300 python bench.py --workload mvit_b_32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 $knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_gemm9h_min_tiles_call95.txt
