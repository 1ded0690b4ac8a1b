#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Usage: tools/gpu_round.sh <tag> [workload]
TAG=${1:-r1}
WL=${2:-x3d_m}
R=$PWD
mkdir -p gpurun_out
PV_RUN_SLOW=1 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
PV_BENCH_VERBOSE=1 python bench.py --workload $WL --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_${WL}.json 2> gpurun_out/${TAG}_bench_${WL}.err
cat gpurun_out/${TAG}_bench_${WL}.json; tail -30 gpurun_out/${TAG}_bench_${WL}.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${WL} -o ${WL} -- python $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_rocprof_${WL}.log 2>&1
tail -3 $R/gpurun_out/${TAG}_rocprof_${WL}.log
find $R/gpurun_out/prof_${TAG}_${WL} -name '*stats*' | head
