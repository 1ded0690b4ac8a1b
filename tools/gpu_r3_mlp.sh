#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r3f; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_mlp or layernorm_fused" > $OUT/pytest_mlp.log 2>&1; tail -2 $OUT/pytest_mlp.log
python tools/bench_mlp.py --iters 30 > $OUT/bench_mlp.txt 2>&1; grep "abl=0\|ln_linear" $OUT/bench_mlp.txt
bash tools/gpu_r3_ab.sh mvit_b_32x3 "" "mlp_minw=1" 2>&1 | grep -v amdgpu
