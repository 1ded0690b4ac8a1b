#!/bin/bash
# attn_w64_kernel in the model: MViT tests, same-box A/B (attn_w64 = 0 / 1) on MViT-B
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_full_geometry.py -q -m gpu -x -k "attention or mvit or vit" 2>&1 | tail -4
for rep in 1 2 3; do
  for knob in 0 1; do
    timeout 300 python bench.py --workload mvit_b_32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune attn_w64=$knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 attn_w64=$knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_attn_w64_call72.txt
