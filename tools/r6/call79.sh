#!/bin/bash
# attn_w64_kernel: staging role wave-uniform by construction (8-wave form without waterfall loops), 5-deep fragment rings in the
# 256-register forms (no scratch in the tile loop): all attention tests incl. the per-form ones, micro-benchmark, model A/B
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attention" 2>&1 | tail -3
timeout 300 python tools/bench_attn.py 1 2 3 4 2>&1 | tee gpurun_out/r6/bench_attn_w64_forms_call79.txt
for rep in 1 2 3; do
  for knob in 0 2 3 4; do
    timeout 300 python bench.py --workload mvit_b_32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune attn_w64=$knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 attn_w64=$knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_attn_w64_forms_call79.txt
