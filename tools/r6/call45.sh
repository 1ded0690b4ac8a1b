#!/bin/bash
mkdir -p gpurun_out/r6
python tools/r6/dbg_ab.py 2>&1 | grep -v amdgpu.ids | tail -14 | tee gpurun_out/r6/dbg_ab_call45.txt
python tools/r6/dbg_x3ds2.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 > gpurun_out/r6/x3d_s_gate_order_call45.txt
timeout 600 python -m pytest tests/test_gpu_checkpoint.py -q -m gpu 2>&1 | tail -4
