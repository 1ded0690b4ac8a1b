#!/bin/bash
# round 6, call 14: fused bottleneck by stage and mode: kernel tests, A/B
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_bottleneck" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_models.py -q -x -k "x3d" 2>&1 | tail -3
for rep in 1 2; do
  for v in "28,block_stages_ab=16" "28,block_stages_ab=28" "24,block_stages_ab=16" "0,block_stages_ab=0"; do
    for w in x3d_m x3d_l; do
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune block_stages=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w block_stages=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_block_stages_call14.txt
