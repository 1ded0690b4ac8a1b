#!/bin/bash
# round 4, GPU call 2: in-launch SE gate + res5 conv_c routing: kernel tests, X3D parity at full geometry, same-box A/Bs
OUT=gpurun_out/r4b; mkdir -p $OUT
export PV_PARITY_DUMP=$PWD/$OUT/parity_x3d.jsonl; rm -f $PV_PARITY_DUMP
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "squeeze_excitation or depthwise or pointwise or se_gate" > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/status.txt
tail -3 $OUT/kernels.log
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_models.py -q -x -k "x3d or csn" > $OUT/models.log 2>&1; echo "models rc=$?" >> $OUT/status.txt
tail -3 $OUT/models.log
timeout 1200 python -m pytest tests/test_gpu_full_geometry.py -q -s -k "x3d" > $OUT/full_geometry_x3d.log 2>&1; echo "full_geometry rc=$?" >> $OUT/status.txt
grep -v "^$" $OUT/full_geometry_x3d.log | tail -12
B="python bench.py --workload x3d_m --no-secondary --no-cpu-baseline --no-sustained --steps 40 --warmup 10"
for rep in 1 2; do
  for T in "fuse_se_gate=1,pw_k14_nt4=1" "fuse_se_gate=0,pw_k14_nt4=1" "fuse_se_gate=1,pw_k14_nt4=0" "fuse_se_gate=0,pw_k14_nt4=0"; do
    $B --tune $T > $OUT/ab_${T//[=,]/_}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$OUT/ab_${T//[=,]/_}_$rep.json')); print('$T rep$rep', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['launches_total'], d['roofline']['all_kernels_ms_per_step'])"
  done
done
PV_BENCH_VERBOSE=2 python bench.py --workload x3d_m --streams 1 --no-secondary --no-cpu-baseline --no-sustained > $OUT/x3d_m_streams1.json 2> $OUT/x3d_m_per_op.txt
grep -v "^  op" $OUT/x3d_m_per_op.txt | grep "n=" | head -14
python bench.py --workload x3d_l --no-secondary --no-cpu-baseline --no-sustained --steps 30 > $OUT/x3d_l.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/x3d_l.json')); print('x3d_l', d['value'], d['ms_per_step'])"
cat $OUT/status.txt
