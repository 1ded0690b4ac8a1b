#!/bin/bash
# every op of the four driver-run workloads beside every distinct neighbour kernel of the other sub-batch: bit-reproducible?
mkdir -p gpurun_out/r6
for w in x3d_m x3d_l mvit_b_32x3 slowfast_r50; do
  timeout 900 python tools/r6/neighbours.py $w 3 20 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tee gpurun_out/r6/neighbours_${w}_call41.txt | tail -6
done
