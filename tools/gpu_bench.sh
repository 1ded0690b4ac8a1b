#!/bin/bash
# bench lines (+ per-op table) for a list of workloads.  Usage: tools/gpu_bench.sh <tag> wl1 wl2 ...
TAG=$1; shift
mkdir -p gpurun_out
for WL in "$@"; do
  PV_BENCH_VERBOSE=2 python bench.py --workload $WL --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_${WL}.json 2> gpurun_out/${TAG}_bench_${WL}.err
  cat gpurun_out/${TAG}_bench_${WL}.json; grep -v amdgpu.ids gpurun_out/${TAG}_bench_${WL}.err | grep -v "^  op" | tail -32
done
