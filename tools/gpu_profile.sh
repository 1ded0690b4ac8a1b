#!/bin/bash
# rocprofv3 evidence for one workload: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes.
# Usage: tools/gpu_profile.sh <tag> <workload>
TAG=$1; WL=$2
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${WL} -o trace -- $CMD > $R/gpurun_out/${TAG}_prof_${WL}.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_${WL} -o fetch -- $CMD >> $R/gpurun_out/${TAG}_prof_${WL}.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_${WL} -o write -- $CMD >> $R/gpurun_out/${TAG}_prof_${WL}.log 2>&1
ls -la $R/gpurun_out/prof_${TAG}_${WL}
tail -3 $R/gpurun_out/${TAG}_prof_${WL}.log
