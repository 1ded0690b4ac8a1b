#!/bin/bash
# SlowFast-R50: one plan (default) vs two sub-batch branches on the round-6 kernels
mkdir -p gpurun_out/r6
for rep in 1 2 3; do
  for st in 1 2; do
    timeout 300 python bench.py --workload slowfast_r50 --streams $st --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slowfast_r50 streams $st rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_slowfast_streams_call90.txt
