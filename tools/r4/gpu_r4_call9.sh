#!/bin/bash
# round 4, GPU call 9: SlowFast's two pathways as two lanes of the launch plan (pv_plan_set_lane): tests + same-box A/B
OUT=gpurun_out/r4i; mkdir -p $OUT; rm -f $OUT/status.txt
export PV_PARITY_DUMP=$PWD/$OUT/parity_slowfast.jsonl; rm -f $PV_PARITY_DUMP
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_roi_head.py tests/test_transforms.py tests/test_head_comm.py tests/test_gpu_checkpoint.py -q -m gpu -k "slowfast or detection or roi or packer or split or gather or stream" > $OUT/models.log 2>&1; echo "models rc=$?" >> $OUT/status.txt
tail -3 $OUT/models.log
B="python bench.py --workload slowfast_r50 --no-secondary --no-cpu-baseline --no-sustained --steps 40 --warmup 10"
for rep in 1 2 3; do
  for T in "pathway_lanes=1" "pathway_lanes=0"; do
    $B --tune $T > $OUT/ab_sf_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_sf_${T//[=,]/_}_$rep.json')); print('slowfast $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['launches_total'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
for T in "pathway_lanes=1" "pathway_lanes=0"; do
  $B --streams 2 --tune $T > $OUT/ab_sf_streams2_${T//[=,]/_}.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/ab_sf_streams2_${T//[=,]/_}.json')); print('slowfast streams=2 $T', d['value'], d['ms_per_step'], d['step_ms'])"
done
timeout 1200 python -m pytest tests/test_gpu_full_geometry.py -q -s -k "slowfast and not stress" > $OUT/full_geometry.log 2>&1; echo "full_geometry rc=$?" >> $OUT/status.txt
grep -v "^$" $OUT/full_geometry.log | tail -7 | cut -c1-250
cat $OUT/status.txt
