#!/bin/bash
# stem_c4_dwt_kernel: two input frames in flight: kernel + X3D tests, same-box A/B against the previous library
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3d.py -q -m gpu -x -k "stem or x3d" 2>&1 | tail -3
OLD=$PWD/pytorchvideo_amd/_lib/old/libpv_mi355x.so
for rep in 1 2 3; do
  for w in x3d_m x3d_l; do
    for lib in old new; do
      if [ $lib = old ]; then export PV_MI355X_LIB=$OLD; else unset PV_MI355X_LIB; fi
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w stem two frames in flight $lib rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_stem_two_frames_call85.txt
unset PV_MI355X_LIB
for lib in old new; do
  if [ $lib = old ]; then export PV_MI355X_LIB=$OLD; else unset PV_MI355X_LIB; fi
  PV_BENCH_VERBOSE=2 timeout 300 python bench.py --workload x3d_m --streams 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep "op stem.conv01" | sed "s/^/$lib /"
done 2>&1 | tee -a gpurun_out/r6/model_ab_stem_two_frames_call85.txt
