#!/bin/bash
# round 4, GPU call 7: rocprofv3 evidence of the final library (kernel stats, op alignment, PMC) + the thin-tail GEMM tile A/B
OUT=gpurun_out/r4g; mkdir -p $OUT
B="python bench.py --no-secondary --no-cpu-baseline --no-sustained --steps 40 --warmup 10"
for rep in 1 2; do
  for T in "gemm_vt_tail=30" "gemm_vt_tail=0"; do
    $B --workload mvit_b_32x3 --tune $T > $OUT/ab_mvit_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_mvit_${T//[=,]/_}_$rep.json')); print('mvit_b $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'])"
    $B --workload slowfast_r50 --tune $T > $OUT/ab_sf_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_sf_${T//[=,]/_}_$rep.json')); print('slowfast $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'])"
  done
done
bash tools/gpu_evidence.sh r4 x3d_m mvit_b_32x3 slowfast_r50 2>&1 | tail -40
