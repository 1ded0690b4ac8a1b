#!/bin/bash
# round 5, GPU call 13: the K = 320 / strided layers of SlowFast at the model's own grids under the candidate routings
OUT=gpurun_out/r5l; mkdir -p $OUT
for T in gemm8=1 gemm8=0 gemm8=0,gemm9=0 gemm9=2 gemm8=1 gemm8=0; do
  timeout 200 python tools/bench_gemm.py --tune=$T "real " "res3 slow" 2>&1 | grep -v "^$"
done | tee $OUT/bench_gemm_k320.txt
for T in gemm8=1 gemm8=0 gemm8=1 gemm8=0; do
    timeout 300 python bench.py --workload slowfast_r50 --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('slowfast_r50 $T', d['value'], d['ms_per_step'])"
done | tee $OUT/model_ab.txt
