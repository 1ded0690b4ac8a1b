#!/bin/bash
# round 6, call 12: the fused bottleneck generalised to res2 / res3 (column tiles, row segments, one channel per lane): tests, micro-benchmark, A/B by stage
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_bottleneck or fused_mlp_rows" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_models.py -q -x -k "x3d" 2>&1 | tail -4
PV_MI355X_LIB=pytorchvideo_amd/_lib/dev/libpv_mi355x.so timeout 300 python tools/r6/bench_block.py 2>&1 | tee gpurun_out/r6/bench_block_call12.txt
for rep in 1 2; do
  for v in 28 16 24 0; do
    for w in x3d_m x3d_l; do
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune block_stages=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w block_stages=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_block_stages_call12.txt
