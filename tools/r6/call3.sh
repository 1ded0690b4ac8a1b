#!/bin/bash
# round 6, call 3: ablations of the two fused-MLP kernels (development variant of the library: timing only)
mkdir -p gpurun_out/r6
PV_MI355X_LIB=pytorchvideo_amd/_lib/dev/libpv_mi355x.so python tools/bench_mlp.py --iters 30 2>&1 | grep -v ln_linear > gpurun_out/r6/bench_mlp_ablations_call3.txt
cat gpurun_out/r6/bench_mlp_ablations_call3.txt
