#!/bin/bash
# round 4, GPU call 4: row-resident attention projection (MViT), temporal-tap rotation of the (3,1,1) implicit GEMM (SlowFast)
OUT=gpurun_out/r4d; mkdir -p $OUT; rm -f $OUT/status.txt
export PV_PARITY_DUMP=$PWD/$OUT/parity_full.jsonl; rm -f $PV_PARITY_DUMP
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "linear_plus or dense_conv or dilated or layernorm_fused or implicit or conv3d or lateral" > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/status.txt
tail -3 $OUT/kernels.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -k "slowfast or mvit or resnet or r2plus1d or c2d or i3d or slow_r50" > $OUT/models.log 2>&1; echo "models rc=$?" >> $OUT/status.txt
tail -4 $OUT/models.log
for T in "gemm_tap_rot=1" "gemm_tap_rot=0" "gemm_tap_rot=1" "gemm_tap_rot=0"; do python tools/bench_gemm.py "sf conv_a res4" --tune=$T 2>&1 | grep -v amdgpu; done | tee $OUT/bench_gemm_tap_rot.txt
B="python bench.py --no-secondary --no-cpu-baseline --no-sustained --steps 40 --warmup 10"
for rep in 1 2; do
  for T in "gemm_tap_rot=1" "gemm_tap_rot=0"; do
    $B --workload slowfast_r50 --tune $T > $OUT/ab_sf_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_sf_${T//[=,]/_}_$rep.json')); print('slowfast $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['launches_total'], d['roofline']['all_kernels_ms_per_step'])"
  done
  for T in "proj_rows=1" "proj_rows=0"; do
    $B --workload mvit_b_32x3 --tune $T > $OUT/ab_mvit_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_mvit_${T//[=,]/_}_$rep.json')); print('mvit_b $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['launches_total'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
PV_BENCH_VERBOSE=2 python bench.py --workload mvit_b_32x3 --streams 1 --no-secondary --no-cpu-baseline --no-sustained > $OUT/mvit_streams1.json 2> $OUT/mvit_per_op.txt
grep -v "^  op" $OUT/mvit_per_op.txt | grep "n=" | head -8
PV_BENCH_VERBOSE=2 python bench.py --workload slowfast_r50 --streams 1 --no-secondary --no-cpu-baseline --no-sustained > $OUT/sf_streams1.json 2> $OUT/sf_per_op.txt
grep -v "^  op" $OUT/sf_per_op.txt | grep "n=" | head -6
timeout 1500 python -m pytest tests/test_gpu_full_geometry.py -q -s -k "(slowfast or mvit) and not stress" > $OUT/full_geometry.log 2>&1; echo "full_geometry rc=$?" >> $OUT/status.txt
grep -v "^$" $OUT/full_geometry.log | tail -12
cd /tmp; export TMPDIR=/tmp; R=$OLDPWD; S=/tmp/calib2; rm -rf $S; mkdir -p $S
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $S/FETCH_SIZE -o c -- python $R/tools/calib_fetch.py run 6 > $R/$OUT/calib_run_rot.txt 2>&1
python $R/tools/calib_fetch.py fold $S 6 > $R/$OUT/calib_fetch_rot.md 2>&1
cd $R; grep CASE $OUT/calib_run_rot.txt; cat $OUT/calib_fetch_rot.md
cat $OUT/status.txt
