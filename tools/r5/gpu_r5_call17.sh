#!/bin/bash
# round 5, GPU call 17 (last minutes of the budget): temporal-tap rotation in the eight-phase kernels -- kernel cases, SlowFast
# parity at full geometry, L2 fill counter of res4 conv_a with / without, per-layer and model A/B
OUT=gpurun_out/r5n; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 150 python -m pytest tests/test_gpu_kernels.py -q -x -k 'temporal_tap_rotation' 2>&1 | tail -4 | tee $OUT/kernel_tests.txt; echo "kernel_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
timeout 200 python -m pytest tests/test_gpu_full_geometry.py -q -x -k 'slowfast and (one_clip or teacher_forced) and not stress' 2>&1 | tail -3 | tee $OUT/slowfast_parity.txt; echo "slowfast_parity rc=${PIPESTATUS[0]}" >> $OUT/status.txt
for T in gemm9_tap_rot=0 gemm9_tap_rot=1 gemm9_tap_rot=0 gemm9_tap_rot=1; do
    timeout 200 python bench.py --workload slowfast_r50 --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('slowfast_r50 $T', d['value'], d['ms_per_step'])"
done | tee $OUT/model_ab.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
for V in 0 1; do
  rm -rf /tmp/rot$V; timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/rot$V -o fetch -- python $R/tools/bench_gemm.py --tune=gemm9_tap_rot=$V "conv_a res4 slow" > /tmp/rot$V.log 2>&1
  python - <<PY | tee -a $R/$OUT/fetch_rot.txt
import csv, glob, collections
f = glob.glob('/tmp/rot$V/**/fetch_counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: [0.0, set()])
for row in csv.DictReader(open(f[0])):
    if row['Counter_Name'] == 'FETCH_SIZE' and 'gemm_quad' in row['Kernel_Name']:
        e = agg[row['Kernel_Name'][:60]]; e[0] += float(row['Counter_Value']); e[1].add(row['Dispatch_Id'])
for k, (v, ds) in agg.items():
    print('tap_rot=$V', k, 'dispatches', len(ds), 'L2 fill MB per launch (2 x FETCH_SIZE KB)', round(2 * v * 1024 / len(ds) / 1e6, 1))
PY
done
cd $R
for T in gemm9_tap_rot=0 gemm9_tap_rot=1 gemm9_tap_rot=0 gemm9_tap_rot=1; do
  timeout 100 python tools/bench_gemm.py --tune=$T "res4" "conv_a res5" 2>&1 | grep -v "^$"
done | tee $OUT/bench_gemm_rot.txt
cat $OUT/status.txt
