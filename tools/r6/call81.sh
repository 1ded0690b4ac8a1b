#!/bin/bash
# attn_w64_kernel<96, 1, 4> (the default form): timing-only ablations (dev library)
mkdir -p gpurun_out/r6
export PV_MI355X_LIB=$PWD/pytorchvideo_amd/_lib/dev/libpv_mi355x.so
timeout 600 python tools/bench_attn.py 3 abl1f3 abl2f3 abl4f3 abl8f3 abl16f3 abl24f3 abl32f3 abl64f3 abl65f3 abl103f3 3 2>&1 | grep "attn_w64\|b1 \|b4 " | paste - - - | awk '{print $3, $9, $10, $11, "|", $17, $18, $19}' | tee gpurun_out/r6/bench_attn_w64_form3_ablations_call81.txt
