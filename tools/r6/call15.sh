#!/bin/bash
# round 6, call 15: the whole GPU suite (default + slow cases, durations), smoke
mkdir -p gpurun_out/r6
export PV_PARITY_DUMP=$PWD/gpurun_out/r6/parity_full_call15.jsonl
rm -f $PV_PARITY_DUMP
( time python -m pytest tests -m gpu -q --durations=25 ) > gpurun_out/r6/suite_default_call15.log 2>&1
tail -45 gpurun_out/r6/suite_default_call15.log
( time PV_RUN_SLOW=1 python -m pytest tests/test_gpu_full_geometry.py -m "gpu and slow" -q ) > gpurun_out/r6/suite_slow_call15.log 2>&1
tail -8 gpurun_out/r6/suite_slow_call15.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke_call15.log 2>&1; tail -6 gpurun_out/r6/smoke_call15.log
