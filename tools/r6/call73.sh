#!/bin/bash
# attn_w64_kernel in the model, single-plan form: A/B and the per-op times of attn.core
mkdir -p gpurun_out/r6
for rep in 1 2; do
  for knob in 0 1; do
    timeout 300 python bench.py --workload mvit_b_32x3 --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune attn_w64=$knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 streams 1 attn_w64=$knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_attn_w64_streams1_call73.txt
for knob in 0 1; do
  echo "attn_w64=$knob"
  PV_BENCH_VERBOSE=2 timeout 300 python bench.py --workload mvit_b_32x3 --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --tune attn_w64=$knob 2>&1 >/dev/null | grep "attn.core" | awk '{s+=$(NF-5); print} END {print "sum", s}'
done 2>&1 | tee -a gpurun_out/r6/model_ab_attn_w64_streams1_call73.txt
