#!/bin/bash
# the default bench line (secondary legs with the pre-warm phase), three times in fresh processes
mkdir -p gpurun_out/r6 gpurun_out/evidence_r6
for i in 1 2 3; do
  python bench.py > gpurun_out/r6/bench_default_line_call92_$i.json 2> gpurun_out/r6/bench_default_line_call92_$i.err
  python -c "
import json
d=json.load(open('gpurun_out/r6/bench_default_line_call92_$i.json'))
print('run $i:', d['value'], {k:(v['value'], v['step_ms']['p50'], v['step_ms']['p90']) for k,v in d['secondary'].items()}, 'cpu', d['cpu_baseline']['value'])
"
done
cp gpurun_out/r6/bench_default_line_call92_3.json gpurun_out/evidence_r6/bench_default_line.json
cp gpurun_out/r6/bench_default_line_call92_3.err gpurun_out/evidence_r6/bench_default_line.err
