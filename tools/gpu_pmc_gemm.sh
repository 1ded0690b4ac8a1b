#!/bin/bash
# SQ counters for one GEMM shape of tools/bench_gemm.py under both pointwise routes.  Usage: tools/gpu_pmc_gemm.sh <tag> "<shape substring>"
TAG=$1; SHAPE=$2
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for route in 1 2; do
  python $R/tools/bench_gemm.py --tune=conv_route=$route "$SHAPE"
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_r$route -o sq -- python $R/tools/bench_gemm.py --tune=conv_route=$route "$SHAPE" > $R/gpurun_out/pmc_${TAG}_r$route.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_r$route -o fetch -- python $R/tools/bench_gemm.py --tune=conv_route=$route "$SHAPE" >> $R/gpurun_out/pmc_${TAG}_r$route.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_r$route -o write -- python $R/tools/bench_gemm.py --tune=conv_route=$route "$SHAPE" >> $R/gpurun_out/pmc_${TAG}_r$route.log 2>&1
done
find $R/gpurun_out/pmc_${TAG}_r1 $R/gpurun_out/pmc_${TAG}_r2 -name "*counter_collection.csv" | head
