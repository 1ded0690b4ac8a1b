#!/bin/bash
# round 4, GPU call 1: the new parity gates, smoke, the default bench line
OUT=gpurun_out/r4a; mkdir -p $OUT
export PV_PARITY_DUMP=$PWD/$OUT/parity_full.jsonl
rm -f $PV_PARITY_DUMP
timeout 1500 python -m pytest tests/test_gpu_full_geometry.py -q -s > $OUT/full_geometry.log 2>&1; echo "full_geometry rc=$?" >> $OUT/status.txt
tail -5 $OUT/full_geometry.log
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/status.txt
tail -6 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err; echo "bench rc=$?" >> $OUT/status.txt
python -c "
import json; d=json.load(open('$OUT/bench_default_line.json')); r=d['roofline']
print('x3d_m', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('model_hbm_frac'), r.get('model_hbm_frac_plan'))
for k,v in d.get('secondary',{}).items(): print(k, v['value'], v['ms_per_step'], v['roofline']['kernel'], v['roofline']['frac'])
print(d.get('cpu_baseline'))
"
cat $OUT/status.txt
