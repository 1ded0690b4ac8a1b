#!/bin/bash
# final HEAD: the whole GPU suite (default selection, then the slow cases), smoke(), the evidence pass for the four workloads, the default bench line three times
mkdir -p gpurun_out/r6 gpurun_out/evidence_r6
( time PV_PARITY_DUMP=gpurun_out/r6/parity_full_call100.jsonl python -m pytest tests -m gpu -q --durations=15 ) > gpurun_out/r6/suite_default_call100.log 2>&1
tail -6 gpurun_out/r6/suite_default_call100.log
( time PV_RUN_SLOW=1 PV_PARITY_DUMP=gpurun_out/r6/parity_full_slow_call100.jsonl python -m pytest tests -m "gpu and slow" -q ) > gpurun_out/r6/suite_slow_call100.log 2>&1
tail -5 gpurun_out/r6/suite_slow_call100.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke_call100.log 2>&1; tail -3 gpurun_out/r6/smoke_call100.log
bash tools/gpu_evidence.sh r6 x3d_m mvit_b_32x3 slowfast_r50 x3d_l 2>&1 | tail -40
for i in 1 2 3; do
  python bench.py > gpurun_out/r6/bench_default_line_call100_$i.json 2> gpurun_out/r6/bench_default_line_call100_$i.err
  python -c "
import json
d=json.load(open('gpurun_out/r6/bench_default_line_call100_$i.json'))
print('run $i:', d['value'], {k:(v['value'], v['step_ms']['p50'], v['step_ms']['p90']) for k,v in d['secondary'].items()}, 'cpu', d['cpu_baseline']['value'])
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], 'mvit', d['secondary']['mvit_b_32x3']['roofline']['kernel'], d['secondary']['mvit_b_32x3']['roofline']['frac'])
"
done
cp gpurun_out/r6/bench_default_line_call100_3.json gpurun_out/evidence_r6/bench_default_line.json
cp gpurun_out/r6/bench_default_line_call100_3.err gpurun_out/evidence_r6/bench_default_line.err
