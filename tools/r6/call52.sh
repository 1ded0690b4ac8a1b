#!/bin/bash
mkdir -p gpurun_out/r6
for t in "conv_route=0" "conv_route=1" "conv_route=3" "conv_route=0,gemm8=0" ; do
  python tools/bench_gemm.py "x3d " --tune=$t 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r6/bench_gemm_x3d_res5_routes_call52.txt
