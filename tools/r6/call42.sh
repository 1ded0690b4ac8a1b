#!/bin/bash
mkdir -p gpurun_out/r6
for rep in 1 2 3; do
  for v in "fuse_ln_qkv_max_c=192" "fuse_ln_qkv_max_c=96"; do
    timeout 300 python bench.py --workload mvit_b_32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 $v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_ln_qkv_call42.txt
