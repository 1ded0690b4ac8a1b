#!/bin/bash
# the 64-inner-channel conv_b -> conv_c variant with 3 / 4 voxel tiles per wave (one workgroup per CU: the whole register file is there): kernel cases, per-op, three interleaved rounds
mkdir -p gpurun_out/r6
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_b_and_pointwise" 2>&1 | tail -4 ) | tee gpurun_out/r6/kernel_cases_conv_bc_tm_call102.txt
for t in 2 3 4; do
  PV_BENCH_VERBOSE=2 timeout 300 python bench.py --workload slowfast_r50 --streams 1 --no-secondary --no-cpu-baseline --tune tap_pw2_tm=$t 2>&1 >/dev/null | grep -E "^  op conv_bc.*c64->64->256" | cut -c1-120 | sed "s/^/tm=$t /"
done | tee gpurun_out/r6/per_op_conv_bc_tm_call102.txt
run() { timeout 300 python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune "$2" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 rep $3:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do
  for knob in "tap_pw2_tm=2" "tap_pw2_tm=3" "tap_pw2_tm=4"; do run slowfast_r50 $knob $rep; done
done 2>&1 | tee gpurun_out/r6/model_ab_conv_bc_tm_call102.txt
