#!/bin/bash
# round 5, GPU call 7: pv_gemm9 as it ships (split-K and the dropped round-4 kernels removed): kernel tests, forced kernel tests,
# model suites of the families it serves, microbench, routing threshold A/B
OUT=gpurun_out/r5g; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k 'quad_phase or tap_rotation or large_tile or layernorm_fused or fused_mlp' 2>&1 | tail -4 | tee $OUT/kernel_tests.txt; echo "kernel_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tail -4 | tee $OUT/forced_kernel_tests.txt
import pytest
from pytorchvideo_amd import _lib as L
L.tune(gemm9=2)
raise SystemExit(pytest.main(["tests/test_gpu_kernels.py", "-q", "-k", "conv or lateral or linear", "--deselect", "tests/test_gpu_kernels.py::test_large_tile_gemm_kernel", "--deselect", "tests/test_gpu_kernels.py::test_temporal_conv_tap_rotation_and_uniform_tap_staging"]))
PY
echo "forced_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
for T in gemm9=0 gemm9=1 gemm9=0 gemm9=1; do timeout 300 python tools/bench_gemm.py "sf conv" "sf shortcut" "mvit" "big" "ksweep" --tune=$T 2>&1 | grep -v amdgpu; done | tee $OUT/bench_gemm.txt
for W in slowfast_r50 mvit_b_32x3; do
  for T in gemm9=0 gemm9=1 gemm9_min_tiles=120 gemm9_min_tiles=300 gemm9=0 gemm9=1; do
    timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$W $T', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/model_ab.txt
timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k 'slowfast or mvit or resnet or r2plus1d or csn' 2>&1 | tail -3 | tee $OUT/model_tests.txt; echo "model_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
cat $OUT/status.txt
