#!/bin/bash
# round 6, call 1: the default GPU suite with the thread cap / sharded oracle / slow marker (timing + new per-row numbers), then the default bench
mkdir -p gpurun_out/r6
export PV_PARITY_DUMP=$PWD/gpurun_out/r6/parity_full_call1.jsonl
rm -f $PV_PARITY_DUMP
( time python -m pytest tests -m gpu -q --durations=60 ) > gpurun_out/r6/suite_call1.log 2>&1
tail -80 gpurun_out/r6/suite_call1.log
unset PV_PARITY_DUMP
python bench.py > gpurun_out/r6/bench_call1.json 2> gpurun_out/r6/bench_call1.err
cat gpurun_out/r6/bench_call1.json | head -c 3000
