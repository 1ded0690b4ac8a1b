#!/bin/bash
# round 5, GPU call 9: the eight-phase loop on 128 x 256 tiles (pv_gemm9h.hip): kernel tests, per-layer A/B on SlowFast's res4 /
# res5 shapes and the MViT shapes it can run, model-level A/B (same box, interleaved)
OUT=gpurun_out/r5i; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k 'quad_phase or large_tile or tap_rotation' 2>&1 | tail -15 | tee $OUT/kernel_tests.txt; echo "kernel_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
SEL="res4 res5 b15 b14 ksweep"
for T in gemm9h=0 gemm9h=1 gemm9h=2 gemm9h=0 gemm9h=1; do
  timeout 300 python tools/bench_gemm.py --tune=$T $SEL 2>&1 | grep -v "^$"
done | tee $OUT/bench_gemm_half.txt
for W in slowfast_r50 mvit_b_32x3; do
  for T in gemm9h=0 gemm9h=1 gemm9h=0 gemm9h=1 gemm9h_below=300; do
    timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$W $T', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/model_ab.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_full_geometry.py -q -x -k 'slowfast and not stress and not second and not calibrated' 2>&1 | tail -5 | tee $OUT/slowfast_tests.txt; echo "slowfast_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
cat $OUT/status.txt
