#!/bin/bash
# round 5, GPU call 18 (last of the budget): SlowFast evidence re-measured on the library with the tap rotation; every case of the
# three eight-phase kernel tests once more (8 workers: the CPU reference convolutions dominate)
bash tools/gpu_evidence.sh r5 slowfast_r50 2>&1 | tail -3 | cut -c1-300
OUT=gpurun_out/r5o; mkdir -p $OUT
timeout 140 python -m pytest tests/test_gpu_kernels.py -q -n 8 -k 'quad_phase or tap_rotation or large_tile' 2>&1 | tail -3 | tee $OUT/kernel_tests.txt
