#!/bin/bash
# round 4, GPU call 5: four-waves-per-SIMD depthwise variant (dw_occ4): tests + same-box A/B
OUT=gpurun_out/r4e; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "depthwise or squeeze_excitation or linear_plus or token_pool or pool" > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/status.txt
tail -3 $OUT/kernels.log
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_models.py -q -k "x3d or csn" > $OUT/models.log 2>&1; echo "models rc=$?" >> $OUT/status.txt
tail -3 $OUT/models.log
B="python bench.py --no-secondary --no-cpu-baseline --no-sustained --steps 40 --warmup 10"
for rep in 1 2 3; do
  for T in "dw_occ4=1" "dw_occ4=0"; do
    $B --workload x3d_m --tune $T > $OUT/ab_x3d_m_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_x3d_m_${T//[=,]/_}_$rep.json')); print('x3d_m $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['launches_total'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
for T in "dw_occ4=1" "dw_occ4=0"; do
  $B --workload x3d_l --tune $T > $OUT/ab_x3d_l_${T//[=,]/_}.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/ab_x3d_l_${T//[=,]/_}.json')); print('x3d_l $T', d['value'], d['ms_per_step'], d['step_ms'])"
  PV_BENCH_VERBOSE=1 $B --workload x3d_m --streams 1 --tune $T 2>&1 >/dev/null | grep "conv_b\|conv_ab" | head -3
done
cat $OUT/status.txt
