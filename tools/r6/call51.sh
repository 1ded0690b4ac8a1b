#!/bin/bash
# per-op tables at the per-branch batch (what each of the two branches runs): which ops do not shrink with the batch?
mkdir -p gpurun_out/r6
for spec in "mvit_b_32x3 4" "mvit_b_32x3 8" "x3d_m 16" "x3d_m 32" "slowfast_r50 8" "slowfast_r50 16"; do
  set -- $spec
  PV_BENCH_VERBOSE=2 python bench.py --workload $1 --batch $2 --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --no-sustained 2> gpurun_out/r6/per_op_$1_b$2_call51.txt >/dev/null
  grep -c " op " gpurun_out/r6/per_op_$1_b$2_call51.txt
done
