#!/bin/bash
# Round-3 first visit: GPU parity suite, per-op tables of the two metric models (single-plan form), the default line.
R=$PWD
OUT=$R/gpurun_out/r3a
mkdir -p $OUT
( time python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
for WL in mvit_b_32x3 x3d_m; do
  PV_BENCH_VERBOSE=2 python bench.py --workload $WL --streams 1 --no-secondary --no-cpu-baseline > $OUT/${WL}_bench_streams1.json 2> $OUT/${WL}_per_op.err
  grep -v amdgpu.ids $OUT/${WL}_per_op.err > $OUT/${WL}_per_op.txt; rm -f $OUT/${WL}_per_op.err
  python -c "import json; d=json.load(open('$OUT/${WL}_bench_streams1.json')); r=d['roofline']; print('$WL streams1', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['kernels_ms_per_step'])"
done
( time python bench.py ) > $OUT/bench_default_line.json 2> $OUT/bench_default.err
tail -4 $OUT/bench_default.err
python -c "
import json; d=json.load(open('$OUT/bench_default_line.json'))
print('x3d_m', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
for k,v in d.get('secondary',{}).items(): print(k, v['value'], v['ms_per_step'], v['roofline']['kernel'], v['roofline']['frac'])
print(d.get('cpu_baseline'))"
