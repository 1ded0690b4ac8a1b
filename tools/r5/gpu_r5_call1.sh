#!/bin/bash
# round 5, GPU call 1: the generalised uniform-tap staging (gemm_umode, now on by default) through the dense-conv kernel tests and
# SlowFast's model / full-geometry suites; the 256-wide tile kernel FORCED on every workload GEMM shape (routing data); this box's baseline line
OUT=gpurun_out/r5a; mkdir -p $OUT; rm -f $OUT/status.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k 'conv or lateral' 2>&1 | tail -3 | tee $OUT/kernel_tests.txt; echo "kernel_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_full_geometry.py -q -x -k 'slowfast or resnet or r2plus1d or csn' 2>&1 | tail -3 | tee $OUT/model_tests.txt; echo "model_tests rc=${PIPESTATUS[0]}" >> $OUT/status.txt
for T in gemm8=1 gemm8=0 gemm8=2 gemm8=4; do timeout 300 python tools/bench_gemm.py "sf conv" "sf shortcut" "mvit" "big" --tune=$T 2>&1 | grep -v amdgpu; done | tee $OUT/bench_gemm_gemm8_modes.txt
timeout 900 python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err; echo "bench rc=$?" >> $OUT/status.txt
python -c "
import json; d=json.load(open('$OUT/bench_default_line.json')); r=d['roofline']
print('x3d_m', d['value'], d['ms_per_step'], r['kernel'], r['frac'])
for k,v in d.get('secondary',{}).items(): print(k, v['value'], v['ms_per_step'], v['roofline']['kernel'], v['roofline']['frac'])
print(d.get('cpu_baseline'))
"
cat $OUT/status.txt
