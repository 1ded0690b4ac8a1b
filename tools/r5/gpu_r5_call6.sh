#!/bin/bash
# round 5, GPU call 6: pv_gemm9 after the register clean-up (accumulators were half in scratch in call 5's build), split-K as a
# template variant with pairwise (deterministic) sums
OUT=gpurun_out/r5f; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k 'reduction_split or (quad_phase and not 70000 and not 40000)' 2>&1 | tail -4 | tee $OUT/quad_tests.txt; echo "quad_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
for T in gemm9=0 gemm9_var=0 gemm9_var=2 gemm9_splitk=0 gemm9=0 gemm9_var=2; do timeout 300 python tools/bench_gemm.py "sf conv_a res4" "sf conv_b res4" "sf conv_c res4" "sf conv_c res5" "sf conv_a res5" "sf conv_b res5" "mvit qkv b4" "mvit fc2 b4" "mvit qkv b14" "big" "ksweep M25k K" --tune=$T 2>&1 | grep -v amdgpu; done | tee $OUT/bench_gemm.txt
for W in slowfast_r50 mvit_b_32x3; do
  for T in gemm9=0 gemm9=1 gemm9_splitk=0; do
    timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$W $T', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/model_ab.txt
timeout 600 python -m pytest tests/test_gpu_full_geometry.py -q -x -k 'slowfast and bench_batch' 2>&1 | tail -30 | tee $OUT/model_tests.txt; echo "model_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
cat $OUT/status.txt
