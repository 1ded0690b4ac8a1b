#!/bin/bash
# attn_w64_kernel in three forms (attn_w64 = 1: 64 rows / wave, one wave per SIMD; 2: 32 rows / wave, 8-wave workgroups; 3: 32 rows / wave,
# two 4-wave workgroups per CU): kernel tests per form, micro-benchmark, model A/B in both bench forms
mkdir -p gpurun_out/r6
for f in 1 2 3; do
  echo "tests, attn_w64=$f"
  timeout 600 python -c "
import sys, pytest
from pytorchvideo_amd import _lib as L
L.tune(attn_w64=$f)
sys.exit(pytest.main(['tests/test_gpu_kernels.py', '-q', '-m', 'gpu', '-x', '-k', 'attention']))" 2>&1 | tail -3
done
timeout 300 python tools/bench_attn.py 0 1 2 3 2>&1 | tee gpurun_out/r6/bench_attn_w64_forms_call74.txt
for rep in 1 2; do
  for knob in 0 1 2 3; do
    for st in 2 1; do
      timeout 300 python bench.py --workload mvit_b_32x3 --streams $st --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune attn_w64=$knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 streams $st attn_w64=$knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_attn_w64_forms_call74.txt
