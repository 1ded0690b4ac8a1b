#!/bin/bash
# SlowFast-R50 branches 2 / 3 / 4, then the evidence pass of the final library + bench defaults for the four workloads and the default line
mkdir -p gpurun_out/r6
for rep in 1 2; do
  for st in 2 3 4; do
    timeout 300 python bench.py --workload slowfast_r50 --streams $st --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slowfast_r50 streams $st rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_slowfast_streams_call91.txt
bash tools/gpu_evidence.sh r6 x3d_m mvit_b_32x3 slowfast_r50 x3d_l 2>&1 | tail -40
python bench.py > gpurun_out/evidence_r6/bench_default_line.json 2> gpurun_out/evidence_r6/bench_default_line.err
python -c "
import json
d=json.load(open('gpurun_out/evidence_r6/bench_default_line.json'))
print('default line:', d['value'], d['ms_per_step'], {k:(v['value'], v['config']['streams']) for k,v in d['secondary'].items()}, 'cpu', d['cpu_baseline']['value'])
"
