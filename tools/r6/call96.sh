#!/bin/bash
# on the new routing defaults (gemm9h_min_tiles 32, gemm9h_tr_min_tiles 100): neighbouring thresholds and the remaining routing knobs in the two-branch bench forms
mkdir -p gpurun_out/r6
run() { # workload knob rep
  timeout 300 python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune "$2" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 rep $3:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
}
for rep in 1 2; do
  for knob in "gemm9h_min_tiles=32" "gemm9h_min_tiles=96,gemm9h_tr_min_tiles=200" "gemm9_var=0" "gemm9_var=2" "conv_small_cin=64" "gemm9h_tr_min_tiles=50"; do run slowfast_r50 $knob $rep; done
  for knob in "gemm9h_min_tiles=32" "gemm9h_min_tiles=96,gemm9h_tr_min_tiles=200" "gemm9h_tr_min_tiles=50" "gemm9h_tr_min_tiles=150" "conv_small_cin=64" "conv_small_cin=192" "fuse_ln_qkv_max_c=96"; do run mvit_b_32x3 $knob $rep; done
done 2>&1 | tee gpurun_out/r6/model_ab_routes_call96.txt
