#!/bin/bash
# round 6, call 2: the 16-row fused MLP kernel -- kernel tests, micro-benchmark against the 32-row kernel, MViT-B A/B, parity
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_kernels.py -q -x -k "fused_mlp_rows" > gpurun_out/r6/mlp16_tests_call2.log 2>&1
tail -15 gpurun_out/r6/mlp16_tests_call2.log
python tools/bench_mlp.py --iters 30 > gpurun_out/r6/bench_mlp_call2.txt 2>&1
grep -v ln_linear gpurun_out/r6/bench_mlp_call2.txt | tail -30
for rep in 1 2; do
  for v in 1 0; do
    python bench.py --workload mvit_b_32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune mlp_rows16=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b mlp_rows16=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_mlp16_call2.txt
export PV_PARITY_DUMP=$PWD/gpurun_out/r6/parity_full_call2.jsonl
rm -f $PV_PARITY_DUMP
python -m pytest tests/test_gpu_full_geometry.py -q -k "mvit or x3d_l" > gpurun_out/r6/full_geometry_call2.log 2>&1
tail -15 gpurun_out/r6/full_geometry_call2.log
python -m pytest tests/test_gpu_models.py -q -k "mvit" 2>&1 | tail -3
