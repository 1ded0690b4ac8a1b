#!/bin/bash
# round 4, GPU call 14: temporal fast path of the implicit GEMM's staging (gemm_tmode): tests of every dense-conv family + A/B
OUT=gpurun_out/r4n; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or lateral" > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/status.txt
tail -2 $OUT/kernels.log
timeout 400 python -m pytest tests/test_gpu_models.py tests/test_roi_head.py -q -x -m gpu -k "slowfast or resnet or r2plus1d or i3d or slow_r50 or c2d or csn or detection or roi" > $OUT/models.log 2>&1; echo "models rc=$?" >> $OUT/status.txt
tail -2 $OUT/models.log
for T in "gemm_tmode=1" "gemm_tmode=0" "gemm_tmode=1" "gemm_tmode=0"; do python tools/bench_gemm.py "sf conv_a res4" --tune=$T 2>&1 | grep -v amdgpu; done | tee $OUT/bench_gemm_tmode.txt
B="python bench.py --workload slowfast_r50 --no-secondary --no-cpu-baseline --no-sustained --no-roofline --steps 40 --warmup 10"
for rep in 1 2; do
  for T in "gemm_tmode=1" "gemm_tmode=0"; do
    $B --tune $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slowfast $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'])" | tee -a $OUT/ab_slowfast_tmode.txt
  done
done
timeout 300 python -m pytest tests/test_gpu_full_geometry.py -q -s -k "slowfast and north_star" > $OUT/full_geometry.log 2>&1; echo "full_geometry rc=$?" >> $OUT/status.txt
grep -v "^$" $OUT/full_geometry.log | tail -4 | cut -c1-220
cat $OUT/status.txt
