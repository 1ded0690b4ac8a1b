#!/bin/bash
# MViT-B: which GEMM kernel for the 384 -> 384 projections of stage 3 (25096 rows: 297 tiles of 256 x 128 = 1.16 rounds of 256 CUs)?
mkdir -p gpurun_out/r6
for knob in "gemm9h=1" "gemm9h=0" "gemm9h=0,gemm9=0"; do
  echo "tune $knob"
  PV_BENCH_VERBOSE=2 timeout 300 python bench.py --workload mvit_b_32x3 --streams 1 --steps 10 --warmup 3 --no-cpu-baseline --tune $knob 2>&1 >/dev/null | grep "attn.proj\|attn.qkv\|mlp.fc" | awk '{print $2, $(NF-5)}' | sort | uniq -c | sort -k2 | head -30
done 2>&1 | tee gpurun_out/r6/mvit_gemm_routes_call86.txt
for rep in 1 2; do
  for knob in "gemm9h=1" "gemm9h=0"; do
    timeout 300 python bench.py --workload mvit_b_32x3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mvit_b_32x3 $knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee -a gpurun_out/r6/mvit_gemm_routes_call86.txt
