#!/bin/bash
# round 6, call 6: the fused X3D bottleneck kernel (res4 blocks without squeeze-excitation): kernel test, model tests, A/B
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_bottleneck_block" > gpurun_out/r6/block_tests_call6.log 2>&1
tail -25 gpurun_out/r6/block_tests_call6.log
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_models.py -q -x -k "x3d" 2>&1 | tail -5
for rep in 1 2; do
  for v in 1 0; do
    for w in x3d_m x3d_l; do
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune fuse_block=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w fuse_block=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_fuse_block_call6.txt
PV_BENCH_VERBOSE=1 timeout 300 python bench.py --workload x3d_m --steps 10 --warmup 3 --no-cpu-baseline --streams 1 2>&1 | grep -E "block.fused|conv_c\|32x16x14x14|conv_b\|32x16x14x14|conv_a\|32x16x14x14" | head -20 | tee gpurun_out/r6/x3d_m_res4_per_op_call6.txt
export PV_PARITY_DUMP=$PWD/gpurun_out/r6/parity_full_call6.jsonl
rm -f $PV_PARITY_DUMP
timeout 900 python -m pytest tests/test_gpu_full_geometry.py -q -k "x3d" > gpurun_out/r6/full_geometry_call6.log 2>&1
tail -12 gpurun_out/r6/full_geometry_call6.log
