#!/bin/bash
# attn_w64_kernel (pv_attn64.hip), first GPU visit: the attention kernel tests, then the A/B micro-benchmark on MViT-B's geometries
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -15
timeout 300 python tools/bench_attn.py 0 1 2>&1 | tee gpurun_out/r6/bench_attn_w64_call64.txt
