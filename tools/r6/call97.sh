#!/bin/bash
# three interleaved pairs: the pointwise streaming kernel up to 192 input channels (MViT-B stage 2; X3D res5 conv_a), gemm_quad_kernel's DMA placement before the barrier for every K (SlowFast-R50)
mkdir -p gpurun_out/r6
run() { # workload knob rep
  timeout 300 python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune "$2" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 rep $3:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
}
for rep in 1 2 3; do
  for knob in "conv_small_cin=128" "conv_small_cin=192"; do run mvit_b_32x3 $knob $rep; done
  for knob in "conv_small_cin=128" "conv_small_cin=192"; do run x3d_m $knob $rep; done
  for knob in "conv_small_cin=128" "conv_small_cin=192"; do run x3d_l $knob $rep; done
  for knob in "gemm9_var=-1" "gemm9_var=0"; do run slowfast_r50 $knob $rep; done
  for knob in "gemm9_var=-1" "gemm9_var=0"; do run mvit_b_32x3 $knob $rep; done
done 2>&1 | tee gpurun_out/r6/model_ab_small_cin_var_call97.txt
