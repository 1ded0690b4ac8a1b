#!/bin/bash
# round 5, GPU call 8: every kernel test on the library as it ships; X3D suites (the in-launch SE gate left its kernels); the
# full-geometry parity file with the round's new gates (numbers dumped for profiles/r5/parity_full.jsonl); routing threshold A/B
OUT=gpurun_out/r5h; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x 2>&1 | tail -4 | tee $OUT/kernel_tests.txt; echo "kernel_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
timeout 600 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_models.py -q -x -k 'x3d or X3d or split_batch' 2>&1 | tail -3 | tee $OUT/x3d_tests.txt; echo "x3d_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
for W in slowfast_r50 mvit_b_32x3; do
  for T in gemm9_min_tiles=120 gemm9_min_tiles=64 gemm9_min_tiles=96 gemm9_min_tiles=120; do
    timeout 300 python bench.py --workload $W --no-cpu-baseline --no-secondary --no-sustained --no-roofline --tune $T 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$W $T', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/model_ab.txt
timeout 300 python bench.py --workload x3d_m --no-cpu-baseline --no-secondary --no-sustained --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('x3d_m', d['value'], d['ms_per_step'])" | tee -a $OUT/model_ab.txt
export PV_PARITY_DUMP=$PWD/$OUT/parity_full.jsonl; rm -f $PV_PARITY_DUMP
timeout 2400 python -m pytest tests/test_gpu_full_geometry.py -q -s --durations=8 2>&1 | grep -v "^$" | tail -60 | tee $OUT/full_geometry.txt; echo "full_geometry rc=${PIPESTATUS[0]}" >> $OUT/status.txt
cat $OUT/status.txt
