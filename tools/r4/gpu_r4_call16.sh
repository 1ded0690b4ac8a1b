export PV_MI355X_LIB=pytorchvideo_amd/_lib/umode/libpv_mi355x.so
mkdir -p gpurun_out/r4p
python - <<'PY' 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r4p/umode_kernel_tests.txt
import pytest
from pytorchvideo_amd import _lib as L
L.tune(gemm_umode=1)
raise SystemExit(pytest.main(["tests/test_gpu_kernels.py", "-q", "-x", "-k", "conv or lateral"]))
PY
for T in "gemm_umode=1" "gemm_umode=0" "gemm_umode=1" "gemm_umode=0"; do python tools/bench_gemm.py "sf conv_b" --tune=$T 2>&1 | grep -v amdgpu; done | tee gpurun_out/r4p/bench_gemm_umode.txt
