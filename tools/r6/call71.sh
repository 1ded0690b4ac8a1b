#!/bin/bash
# attn_w64_kernel (third form + row-wise epilogue): timing-only ablations (dev library)
mkdir -p gpurun_out/r6
export PV_MI355X_LIB=$PWD/pytorchvideo_amd/_lib/dev/libpv_mi355x.so
timeout 600 python tools/bench_attn.py 1 abl1 abl2 abl4 abl8 abl16 abl24 abl32 abl64 abl65 abl103 2>&1 | grep -v "b0 \|b2 \|b14\|b15\|b1 " | tee gpurun_out/r6/bench_attn_w64_ablations_call71.txt
