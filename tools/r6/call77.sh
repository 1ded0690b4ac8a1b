#!/bin/bash
# attn_w64_kernel: v_max3 as plain (non-volatile) asm; micro-benchmark of the forms
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -2
timeout 300 python tools/bench_attn.py 1 3 4 2>&1 | tee gpurun_out/r6/bench_attn_w64_forms_call77.txt
