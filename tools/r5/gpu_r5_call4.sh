#!/bin/bash
# round 5, GPU call 4: pv_gemm9 v3 (multiply-free staging addresses; variants: bit 0 fragment read-ahead, bit 1 DMAs among the MFMAs):
# kernel tests per variant, forced kernel tests, microbench per variant, model A/B
OUT=gpurun_out/r5d; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k 'tap_rotation' 2>&1 | tail -3 | tee $OUT/rot_tests.txt; echo "rot_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
for V in 3 0 1 2; do
timeout 400 python - <<PY 2>&1 | grep -v amdgpu | tail -6 | tee $OUT/quad_tests_var$V.txt
import pytest
from pytorchvideo_amd import _lib as L
L.tune(gemm9_var=$V)
raise SystemExit(pytest.main(["tests/test_gpu_kernels.py", "-q", "-x", "-k", "quad_phase"]))
PY
echo "quad_tests var$V rc=${PIPESTATUS[0]}" >> $OUT/status.txt
done
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tail -4 | tee $OUT/forced_kernel_tests.txt
import pytest
from pytorchvideo_amd import _lib as L
L.tune(gemm9=2)
raise SystemExit(pytest.main(["tests/test_gpu_kernels.py", "-q", "-k", "conv or lateral or linear", "--deselect", "tests/test_gpu_kernels.py::test_large_tile_gemm_kernel", "--deselect", "tests/test_gpu_kernels.py::test_temporal_conv_tap_rotation_and_uniform_tap_staging"]))
PY
for T in gemm9=0 gemm9=2,gemm9_var=0 gemm9=2,gemm9_var=1 gemm9=2,gemm9_var=2 gemm9=2,gemm9_var=3 gemm9=0 gemm9=2,gemm9_var=0 gemm9=2,gemm9_var=3; do timeout 300 python tools/bench_gemm.py "sf conv_a res4" "sf conv_b res4" "sf conv_c res4" "sf conv_c res5" "sf conv_a res5" "mvit qkv b4" "mvit fc2 b4" "mvit qkv b14" "mvit proj" "big" "ksweep M25k K" --tune=$T 2>&1 | grep -v amdgpu; done | tee $OUT/bench_gemm_gemm9.txt
for W in slowfast_r50 mvit_b_32x3; do
  for T in gemm9=0 gemm9_min_tiles=200,gemm9_var=3 gemm9_min_tiles=200,gemm9_var=0; do
    timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$W $T', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/model_ab.txt
cat $OUT/status.txt
