#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r3h; mkdir -p $OUT
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "handful_of_rows or linear_as_pointwise" > $OUT/pytest_head.log 2>&1; tail -4 $OUT/pytest_head.log
bash tools/gpu_r3_ab.sh x3d_m "" "head_rows=0" 2>&1 | grep -v amdgpu | grep "x3d_m\|head"
for WL in mvit_b_32x3 slowfast_r50; do PV_BENCH_VERBOSE=2 python bench.py --workload $WL --streams 1 --no-secondary --no-cpu-baseline --no-sustained > $OUT/$WL.json 2> $OUT/$WL.txt; grep "op head" $OUT/$WL.txt; done
python -m pytest tests/test_gpu_models.py tests/test_gpu_x3d.py -m gpu -x -q > $OUT/pytest_models.log 2>&1; tail -3 $OUT/pytest_models.log
