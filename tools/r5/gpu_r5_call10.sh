#!/bin/bash
# round 5, GPU call 10: where the half-height tiles pay on MViT-B (its two sub-batch branches halve every tile count): per-layer
# on the branch's shapes under the four routings, then the routing threshold at model level (same box, interleaved)
OUT=gpurun_out/r5j; mkdir -p $OUT
for T in gemm9h=0 gemm9h=2 gemm9=2 gemm9=0,gemm9h=0 gemm9h=0 gemm9h=2; do
  timeout 200 python tools/bench_gemm.py --tune=$T "hb " "res4 slow" "res5 slow" 2>&1 | grep -v "^$"
done | tee $OUT/bench_gemm_half_mvit.txt
for W in mvit_b_32x3 slowfast_r50; do
  for T in gemm9h=0 gemm9h_below=200 gemm9h_below=300 gemm9h_below=600 gemm9h_below=100000 gemm9h=0 gemm9h_below=300; do
    timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$W $T', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/model_ab.txt
