#!/bin/bash
# the res2 instantiation with scalar v_fmac_f32 (no packed op with a cross-lane source in its own destination): kernel tests,
# determinism beside the two kernels that exposed it and beside every op, replay check, A/B
mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bottleneck" 2>&1 | tail -3 | tee gpurun_out/r6/block_tests_call37.log
PV_MI355X_LIB=$PWD/pytorchvideo_amd/_lib/dev/libpv_mi355x.so PV_BESIDE_ABL=0 timeout 900 python tools/r6/replay_locate.py 4 0 beside 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee gpurun_out/r6/replay_locate_beside_call37.txt
timeout 900 python tools/r6/replay_locate.py 28 28 besideall 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tee gpurun_out/r6/replay_locate_besideall_call37.txt
timeout 900 python tools/r6/replay_check.py x3d_m 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/replay_check_call37.txt
for rep in 1 2; do
  for v in "28,block_stages_ab=16" "24,block_stages_ab=16" "28,block_stages_ab=20"; do
    for w in x3d_m x3d_l; do
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune block_stages=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w block_stages=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_block_stages_call37.txt
