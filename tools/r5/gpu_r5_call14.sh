#!/bin/bash
# round 5, GPU call 14: the transposed half tile (256 voxels x 128 channels) for SlowFast res3's 128-channel 1x3x3 convs
OUT=gpurun_out/r5m; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k 'transposed_half or half_height' 2>&1 | tail -8 | tee $OUT/kernel_tests.txt; echo "kernel_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
for T in gemm9h_tr=0 gemm9h_tr=-1 gemm9h_tr=0 gemm9h_tr=-1; do
  timeout 200 python tools/bench_gemm.py --tune=$T "res3" "hb mvit proj b4" 2>&1 | grep -v "^$"
done | tee $OUT/bench_gemm_tr.txt
for T in gemm9h_tr=0 gemm9h_tr=-1 gemm9h_tr=0 gemm9h_tr=-1; do
    timeout 300 python bench.py --workload slowfast_r50 --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('slowfast_r50 $T', d['value'], d['ms_per_step'])"
done | tee $OUT/model_ab.txt
cat $OUT/status.txt
