#!/bin/bash
# attn_w64_kernel with the row-wise epilogue: tests, A/B, stamps
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -5
timeout 300 python tools/bench_attn.py 0 1 2>&1 | tee gpurun_out/r6/bench_attn_w64_call70.txt
export PV_MI355X_LIB=$PWD/pytorchvideo_amd/_lib/dev/libpv_mi355x.so
timeout 300 python tools/r6/attn_stamps.py 2>&1 | grep -v "first round" | tee gpurun_out/r6/attn_stamps_call70.txt
