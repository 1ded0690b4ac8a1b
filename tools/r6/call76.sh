#!/bin/bash
# attn_w64_kernel after the VALU trims (staging offsets, v_max3) + routing by size (attn_w64 = 4): tests per form, micro-benchmark, model A/B
mkdir -p gpurun_out/r6
for f in 1 3 4; do
  echo "tests, attn_w64=$f"
  timeout 600 python -c "
import sys, pytest
from pytorchvideo_amd import _lib as L
L.tune(attn_w64=$f)
sys.exit(pytest.main(['tests/test_gpu_kernels.py', '-q', '-m', 'gpu', '-x', '-k', 'attention']))" 2>&1 | tail -2
done
timeout 300 python tools/bench_attn.py 0 1 3 4 2>&1 | tee gpurun_out/r6/bench_attn_w64_forms_call76.txt
for rep in 1 2 3; do
  for knob in 0 3 4; do
    for st in 2 1; do
      timeout 300 python bench.py --workload mvit_b_32x3 --streams $st --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune attn_w64=$knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 streams $st attn_w64=$knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_attn_w64_forms_call76.txt
