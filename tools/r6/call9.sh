#!/bin/bash
# round 6, call 9: fused bottleneck after the load/store wait fixes: kernel test, A/B, per-op time
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_bottleneck_block" 2>&1 | tail -3
for rep in 1 2; do
  for v in 1 0; do
    for w in x3d_m x3d_l; do
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune fuse_block=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w fuse_block=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_fuse_block_call9.txt
PV_BENCH_VERBOSE=1 timeout 300 python bench.py --workload x3d_m --steps 10 --warmup 3 --no-cpu-baseline --streams 1 2>&1 | grep -E "block.fused" | head -8 | tee gpurun_out/r6/x3d_m_res4_per_op_call9.txt
