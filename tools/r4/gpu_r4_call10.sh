#!/bin/bash
# round 4, GPU call 10: two row sets per wave in the LayerNorm + qkv kernel (ln_linear_rs): test + same-box A/B
OUT=gpurun_out/r4j; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "layernorm_fused or linear_plus or fused_mlp" > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/status.txt
tail -3 $OUT/kernels.log
B="python bench.py --workload mvit_b_32x3 --no-secondary --no-cpu-baseline --no-sustained --steps 40 --warmup 10"
for rep in 1 2 3; do
  for T in "ln_linear_rs=2" "ln_linear_rs=1"; do
    $B --tune $T > $OUT/ab_mvit_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_mvit_${T//[=,]/_}_$rep.json')); print('mvit_b $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
for T in "ln_linear_rs=2" "ln_linear_rs=1"; do
  PV_BENCH_VERBOSE=2 $B --streams 1 --tune $T 2>&1 >/dev/null | grep "op attn.qkv" | head -3
done
cat $OUT/status.txt
