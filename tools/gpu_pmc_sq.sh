#!/bin/bash
# SQ issue/stall counters per kernel for one workload.  Usage: tools/gpu_pmc_sq.sh <tag> <workload>
TAG=$1; WL=$2
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/sq_${TAG}_${WL} -o sq -- $CMD > $R/gpurun_out/${TAG}_sq_${WL}.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/sq_${TAG}_${WL} -o lds -- $CMD >> $R/gpurun_out/${TAG}_sq_${WL}.log 2>&1
ls $R/gpurun_out/sq_${TAG}_${WL}; tail -2 $R/gpurun_out/${TAG}_sq_${WL}.log
