#!/bin/bash
mkdir -p gpurun_out/r6
for rep in 1 2 3; do
  for v in 8388608 4194304 2097152; do
    timeout 300 python bench.py --workload mvit_b_32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune pool_stream_min_elems=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 pool_stream_min_elems=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_pool_stream_call49.txt
PV_BENCH_VERBOSE=2 python bench.py --workload mvit_b_32x3 --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --tune pool_stream_min_elems=2097152 2>&1 >/dev/null | grep "attn.pool\|op norm1" | tail -12
