#!/bin/bash
# round 5, final GPU call: the whole GPU suite on the final library, smoke, the default bench line (with the adopted traffic.json)
OUT=gpurun_out/r5f; mkdir -p $OUT; rm -f $OUT/status.txt
export PV_PARITY_DUMP=$PWD/$OUT/parity_full.jsonl; rm -f $PV_PARITY_DUMP
timeout 2000 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "gpu_suite rc=$?" >> $OUT/status.txt
tail -4 $OUT/gpu_suite.log
timeout 400 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/status.txt
tail -5 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err; echo "bench rc=$?" >> $OUT/status.txt
python -c "
import json; d=json.load(open('$OUT/bench_default_line.json')); r=d['roofline']
print('x3d_m', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('model_hbm_frac'), r.get('model_hbm_frac_plan'), r.get('traffic'), r.get('traffic_population'))
for k,v in d.get('secondary',{}).items(): print(k, v['value'], v['ms_per_step'], v['roofline']['kernel'], v['roofline']['frac'], v['roofline'].get('traffic'))
print(d.get('cpu_baseline'))
"
cat $OUT/status.txt
