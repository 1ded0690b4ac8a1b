#!/bin/bash
# attn_w64_kernel: where a workgroup's time goes (clock stamps, dev library)
mkdir -p gpurun_out/r6
export PV_MI355X_LIB=$PWD/pytorchvideo_amd/_lib/dev/libpv_mi355x.so
timeout 300 python tools/r6/attn_stamps.py 2>&1 | tee gpurun_out/r6/attn_stamps_call69.txt
