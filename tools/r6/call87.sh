#!/bin/bash
# sub-batch branches per workload on the round-6 kernels: 2 (default) vs 3 vs 4
mkdir -p gpurun_out/r6
for rep in 1 2; do
  for w in mvit_b_32x3 x3d_m x3d_l; do
    for st in 2 3 4; do
      timeout 300 python bench.py --workload $w --streams $st --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w streams $st rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_streams_call87.txt
