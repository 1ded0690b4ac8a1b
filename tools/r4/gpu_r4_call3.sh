#!/bin/bash
# round 4, GPU call 3: write-through SE gate, next-block norm1 in the fused MLP: tests, parity, same-box A/Bs, fetch calibration
OUT=gpurun_out/r4c; mkdir -p $OUT; rm -f $OUT/status.txt
export PV_PARITY_DUMP=$PWD/$OUT/parity_full.jsonl; rm -f $PV_PARITY_DUMP
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "squeeze_excitation or depthwise or se_gate or fused_mlp" > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/status.txt
tail -3 $OUT/kernels.log
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_models.py -q -k "x3d or csn or mvit" > $OUT/models.log 2>&1; echo "models rc=$?" >> $OUT/status.txt
tail -4 $OUT/models.log
B="python bench.py --no-secondary --no-cpu-baseline --no-sustained --steps 40 --warmup 10"
for rep in 1 2; do
  for T in "fuse_se_gate=1" "fuse_se_gate=0"; do
    $B --workload x3d_m --tune $T > $OUT/ab_x3d_m_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_x3d_m_${T//[=,]/_}_$rep.json')); print('x3d_m $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['launches_total'], d['roofline']['all_kernels_ms_per_step'])"
  done
  for T in "fuse_next_norm=1" "fuse_next_norm=0"; do
    $B --workload mvit_b_32x3 --tune $T > $OUT/ab_mvit_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_mvit_${T//[=,]/_}_$rep.json')); print('mvit_b $T rep$rep', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['launches_total'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
PV_BENCH_VERBOSE=2 python bench.py --workload x3d_m --streams 1 --no-secondary --no-cpu-baseline --no-sustained > $OUT/x3d_m_streams1.json 2> $OUT/x3d_m_per_op.txt
grep -v "^  op" $OUT/x3d_m_per_op.txt | grep "n=" | head -8
PV_BENCH_VERBOSE=2 python bench.py --workload mvit_b_32x3 --streams 1 --no-secondary --no-cpu-baseline --no-sustained > $OUT/mvit_streams1.json 2> $OUT/mvit_per_op.txt
grep -v "^  op" $OUT/mvit_per_op.txt | grep "n=" | head -12
timeout 1500 python -m pytest tests/test_gpu_full_geometry.py -q -s -k "x3d or mvit" > $OUT/full_geometry.log 2>&1; echo "full_geometry rc=$?" >> $OUT/status.txt
grep -v "^$" $OUT/full_geometry.log | tail -14
# SlowFast (3,1,1) fetch calibration (verdict r3 #6): counters in their own passes
cd /tmp; export TMPDIR=/tmp; R=$OLDPWD; S=/tmp/calib; rm -rf $S; mkdir -p $S
python $R/tools/calib_fetch.py run 6 > $R/$OUT/calib_run.txt 2>&1
for CTR in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $CTR | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $S/$tag -o c -- python $R/tools/calib_fetch.py run 6 > $S/$tag.log 2>&1 || echo "pass $tag failed: $(tail -2 $S/$tag.log)"
done
python $R/tools/calib_fetch.py fold $S 6 > $R/$OUT/calib_fetch.md 2>&1
cd $R; cat $OUT/calib_run.txt; cat $OUT/calib_fetch.md
cat $OUT/status.txt
