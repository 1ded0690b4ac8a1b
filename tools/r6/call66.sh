#!/bin/bash
# attn_w64_kernel, third form (two staging sets, pipelined exponentials, earlier pre-reads): tests, A/B, PMC
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -5
timeout 300 python tools/bench_attn.py 0 1 2>&1 | tee gpurun_out/r6/bench_attn_w64_call68.txt
bash tools/r6/pmc_attn.sh w64_call68 1 2>&1 | tee gpurun_out/r6/pmc_attn_w64_call68.txt
