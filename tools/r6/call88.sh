#!/bin/bash
# SlowFast-R50: the K = 320 / 640 / 1280 projection shortcuts and laterals on gemm8_kernel vs the 128 x 128 LDS-DMA kernel (gemm8 = 0)
mkdir -p gpurun_out/r6
for knob in "gemm8=1" "gemm8=0"; do
  echo "tune $knob"
  PV_BENCH_VERBOSE=2 timeout 300 python bench.py --workload slowfast_r50 --steps 10 --warmup 3 --no-cpu-baseline --tune $knob 2>&1 >/dev/null | grep "shortcut\|c320\|c640\|c1280" | awk '{print $2, $3, $4, $(NF-5)}'
done 2>&1 | tee gpurun_out/r6/slowfast_gemm8_routes_call88.txt
for rep in 1 2 3; do
  for knob in "gemm8=1" "gemm8=0"; do
    timeout 300 python bench.py --workload slowfast_r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slowfast_r50 $knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee -a gpurun_out/r6/slowfast_gemm8_routes_call88.txt
