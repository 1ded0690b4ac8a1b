#!/bin/bash
# pw_stream_kernel's grid sized from the runtime's occupancy instead of the LDS footprint alone: tests, same-box A/B on all four workloads
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3d.py tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -3
OLD=$PWD/pytorchvideo_amd/_lib/old/libpv_mi355x.so
for rep in 1 2 3; do
  for w in x3d_m x3d_l slowfast_r50 mvit_b_32x3; do
    for lib in old new; do
      if [ $lib = old ]; then export PV_MI355X_LIB=$OLD; else unset PV_MI355X_LIB; fi
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w pw_stream grid $lib rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_pw_stream_occupancy_call55.txt
unset PV_MI355X_LIB
PV_BENCH_VERBOSE=2 python bench.py --workload x3d_m --streams 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep "conv_c|" | head -8
