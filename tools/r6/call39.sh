#!/bin/bash
mkdir -p gpurun_out/r6
PV_PARITY_DUMP=gpurun_out/r6/parity_full_call39.jsonl timeout 900 python -m pytest tests/test_gpu_full_geometry.py -q -m gpu -x -k "reproducible_beside or (bench_batch and x3d) or (one_clip and x3d) or (teacher_forced_bf16 and x3d and not stress)" --durations=8 2>&1 | tail -20 | tee gpurun_out/r6/x3d_full_geometry_call39.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3d.py -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r6/kernel_tests_call39.log
