#!/bin/bash
# round 6, call 19: gated conv_c mode + routing defaults: kernel tests, X3D model + full-geometry tests, replay check, A/B
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_bottleneck" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_models.py -q -x -k "x3d" 2>&1 | tail -3
export PV_PARITY_DUMP=$PWD/gpurun_out/r6/parity_full_call19.jsonl; rm -f $PV_PARITY_DUMP
timeout 900 python -m pytest tests/test_gpu_full_geometry.py -q -k "x3d" 2>&1 | tail -8
unset PV_PARITY_DUMP
for rep in 1 2; do
  for v in "24,block_stages_gc=16" "24,block_stages_gc=0" "0,block_stages_ab=0,block_stages_gc=0"; do
    for w in x3d_m x3d_l; do
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune block_stages=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w block_stages=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_gated_conv_c_call19.txt
