#!/bin/bash
mkdir -p gpurun_out/r6
PV_PARITY_DUMP=gpurun_out/r6/parity_full_mvit_call50.jsonl timeout 1200 python -m pytest tests/test_gpu_full_geometry.py tests/test_gpu_models.py tests/test_gpu_checkpoint.py -q -m gpu -k "mvit or MViT or vision" --durations=5 2>&1 | tail -12
