#!/bin/bash
# final HEAD: the whole GPU suite (default selection, then the slow cases), smoke(), the default bench line
mkdir -p gpurun_out/r6 gpurun_out/evidence_r6
( time PV_PARITY_DUMP=gpurun_out/r6/parity_full_call89.jsonl python -m pytest tests -m gpu -q --durations=15 ) > gpurun_out/r6/suite_default_call89.log 2>&1
tail -6 gpurun_out/r6/suite_default_call89.log
( time PV_RUN_SLOW=1 PV_PARITY_DUMP=gpurun_out/r6/parity_full_slow_call89.jsonl python -m pytest tests -m "gpu and slow" -q ) > gpurun_out/r6/suite_slow_call89.log 2>&1
tail -5 gpurun_out/r6/suite_slow_call89.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke_call89.log 2>&1; tail -3 gpurun_out/r6/smoke_call89.log
( time python bench.py > gpurun_out/evidence_r6/bench_default_line.json 2> gpurun_out/evidence_r6/bench_default_line.err ) 2>&1 | tail -3
python -c "
import json
d=json.load(open('gpurun_out/evidence_r6/bench_default_line.json'))
print('default line:', d['value'], d['ms_per_step'], {k:(v['value'], v['steps']) for k,v in d['secondary'].items()}, 'cpu', d['cpu_baseline']['value'])
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], 'mvit', d['secondary']['mvit_b_32x3']['roofline']['kernel'], d['secondary']['mvit_b_32x3']['roofline']['frac'])
"
