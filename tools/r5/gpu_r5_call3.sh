#!/bin/bash
# round 5, GPU call 3: pv_gemm9 v2 (epilogue tables through LDS, both halves' epilogues side by side, residual prefetch, fragment reads
# spread 4/4/8/8, one staging register per operand): kernel tests, forced kernel tests, microbench, model A/B; the new tap-rotation test
OUT=gpurun_out/r5c; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k 'quad_phase or tap_rotation' 2>&1 | tail -15 | tee $OUT/quad_tests.txt; echo "quad_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tail -8 | tee $OUT/forced_kernel_tests.txt
import pytest
from pytorchvideo_amd import _lib as L
L.tune(gemm9=2)
raise SystemExit(pytest.main(["tests/test_gpu_kernels.py", "-q", "-k", "conv or lateral or linear", "--deselect", "tests/test_gpu_kernels.py::test_large_tile_gemm_kernel", "--deselect", "tests/test_gpu_kernels.py::test_temporal_conv_tap_rotation_and_uniform_tap_staging"]))
PY
for T in gemm9=0 gemm9=2 gemm9=0 gemm9=2; do timeout 300 python tools/bench_gemm.py "sf conv" "sf shortcut" "mvit" "big" "ksweep" --tune=$T 2>&1 | grep -v amdgpu; done | tee $OUT/bench_gemm_gemm9.txt
for W in slowfast_r50 mvit_b_32x3; do
  for T in gemm9=0 gemm9_min_tiles=200 gemm9_min_tiles=100; do
    timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$W $T', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/model_ab.txt
cat $OUT/status.txt
