#!/bin/bash
# The rocprofv3 evidence behind bench.py's roofline object, for a list of workloads, summarised ON the GPU box
# (the raw counter CSVs stay in scratch; gpurun_out/evidence_<tag>/ holds what profiles/ keeps):
#   1. bench line + per-op table of the single-plan form (--streams 1) and the default bench line
#   2. rocprofv3 --kernel-trace --stats of that command (+ the op-by-op alignment with the launch plan)
#   3. PMC passes (counters only): FETCH_SIZE | WRITE_SIZE | SQ issue + MFMA busy + GRBM_GUI_ACTIVE, replays only
# Usage: tools/gpu_evidence.sh <tag> wl1 [wl2 ...]
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/evidence_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for WL in "$@"; do
  S=/tmp/ev_${TAG}_${WL}; rm -rf $S; mkdir -p $S
  PV_BENCH_VERBOSE=2 python $R/bench.py --workload $WL --streams 1 --no-secondary --no-cpu-baseline > $OUT/${WL}_bench_streams1.json 2> $S/per_op.err
  grep -v amdgpu.ids $S/per_op.err > $OUT/${WL}_per_op.txt
  python $R/bench.py --workload $WL --no-secondary > $OUT/${WL}_bench_default.json 2>/dev/null
  CMD="python $R/bench.py --workload $WL --streams 1 --no-secondary --no-cpu-baseline --no-sustained --no-roofline --steps 10 --warmup 2"   # replays only: the in-situ per-op profiler would add partial replays to the trace
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $S -o trace -- $CMD > $S/trace.log 2>&1
  cp $(find $S -name 'trace_kernel_stats.csv' | head -1) $OUT/${WL}_kernel_stats.csv
  python $R/tools/align_trace.py $OUT/${WL}_per_op.txt $(find $S -name 'trace_kernel_trace.csv' | head -1) > $OUT/${WL}_rocprof_vs_events.md 2>$S/align.err || cat $S/align.err
  CMDP="python $R/bench.py --workload $WL --streams 1 --no-secondary --no-cpu-baseline --no-sustained --no-roofline --steps 5 --warmup 1"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $S -o fetch -- $CMDP > $S/fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $S -o write -- $CMDP > $S/write.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $S -o sq -- $CMDP > $S/sq.log 2>&1
  python $R/tools/summarize_evidence.py $WL $S $OUT/${WL}_per_op.txt > $OUT/${WL}_pmc.md 2> $S/sum.err || cat $S/sum.err
  tail -4 $OUT/${WL}_pmc.md
  python -c "import json; d=json.load(open('$OUT/${WL}_bench_default.json')); print('$WL default', d['value'], d['ms_per_step'], 'streams', d['config']['streams'])"
  python -c "import json; d=json.load(open('$OUT/${WL}_bench_streams1.json')); r=d['roofline']; print('$WL streams1', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['avg_launch_ms'])"
done
ls -la $OUT | head -40
