#!/bin/bash
# final HEAD: every op of SlowFast-R50 (with the conv_b -> conv_c launches) and MViT-B (new GEMM routes) beside every distinct kernel of the other sub-batch: bit-reproducible?
mkdir -p gpurun_out/r6
for w in slowfast_r50 mvit_b_32x3; do
  ( time timeout 420 python tools/r6/neighbours.py $w 3 20 ) 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tee gpurun_out/r6/neighbours_${w}_call101.txt | tail -6
done
