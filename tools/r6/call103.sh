#!/bin/bash
# the library rebuilt from the final sources (after the removed experiment): SlowFast / conv cases, smoke, the default bench line
mkdir -p gpurun_out/r6
( timeout 900 python -m pytest tests -m gpu -q -x -k "conv_b_and_pointwise or pointwise_conv_behind or slowfast or narrow_dense" 2>&1 | tail -3 ) | tee gpurun_out/r6/sanity_call103.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r6/sanity_call103.txt
python bench.py > gpurun_out/r6/bench_default_line_call103.json 2> /dev/null
python -c "
import json
d=json.load(open('gpurun_out/r6/bench_default_line_call103.json'))
print('default line:', d['value'], {k:v['value'] for k,v in d['secondary'].items()}, 'cpu', d['cpu_baseline']['value'])
" | tee -a gpurun_out/r6/sanity_call103.txt
