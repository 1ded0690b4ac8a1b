#!/bin/bash
# the whole GPU suite as the driver runs it (default selection), then the slow cases, then smoke(), on the final kernels
mkdir -p gpurun_out/r6
( time PV_PARITY_DUMP=gpurun_out/r6/parity_full_call82.jsonl python -m pytest tests -m gpu -q --durations=25 ) > gpurun_out/r6/suite_default_call82.log 2>&1
tail -6 gpurun_out/r6/suite_default_call82.log
( time PV_RUN_SLOW=1 PV_PARITY_DUMP=gpurun_out/r6/parity_full_slow_call82.jsonl python -m pytest tests -m "gpu and slow" -q ) > gpurun_out/r6/suite_slow_call82.log 2>&1
tail -5 gpurun_out/r6/suite_slow_call82.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke_call82.log 2>&1; tail -3 gpurun_out/r6/smoke_call82.log
