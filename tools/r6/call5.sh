#!/bin/bash
# round 6, call 5: 16-row fused MLP with the single-pass LayerNorm prologue (no scratch): tests, H sweep, micro-benchmark, MViT-B A/B
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_mlp_rows" 2>&1 | tail -3
python tools/r6/mlp_h_sweep.py > gpurun_out/r6/mlp_h_sweep_call5.txt 2>&1; cat gpurun_out/r6/mlp_h_sweep_call5.txt
python tools/bench_mlp.py --iters 30 2>&1 | grep "rows/wave" > gpurun_out/r6/bench_mlp_call5.txt; cat gpurun_out/r6/bench_mlp_call5.txt
for rep in 1 2; do
  for v in 1 0; do
    python bench.py --workload mvit_b_32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune mlp_rows16=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b mlp_rows16=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_mlp16_call5.txt
