#!/bin/bash
# round 5, GPU call 11: order of the six DMAs of a K tile in the half-height kernel (weights first / voxel rows first / 4 + 2)
OUT=gpurun_out/r5k; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k 'dma_order_variants' 2>&1 | tail -3 | tee $OUT/kernel_tests.txt; echo "variant_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
for T in gemm9h_var=0 gemm9h_var=1 gemm9h_var=2 gemm9h_var=0 gemm9h_var=1 gemm9h_var=2; do
  timeout 200 python tools/bench_gemm.py --tune=$T "res4 slow" "res5 slow" "hb mvit proj b4" "hb mvit qkv b15" 2>&1 | grep -v "^$"
done | tee $OUT/bench_gemm_half_var.txt
for W in slowfast_r50; do
  for T in gemm9h_var=0 gemm9h_var=1 gemm9h_var=2 gemm9h_var=0 gemm9h_var=1 gemm9h_var=2; do
    timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$W $T', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/model_ab.txt
cat $OUT/status.txt
