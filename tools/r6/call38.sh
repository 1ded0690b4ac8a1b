#!/bin/bash
# micro-test: the packed fp32 FMA whose destination pair is also its broadcast source, alone and beside an LDS-heavy kernel
mkdir -p gpurun_out/r6
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/pk_fma_inplace.hip -o /tmp/pk_fma_inplace 2>/dev/null
timeout 300 /tmp/pk_fma_inplace 2>&1 | tee gpurun_out/r6/pk_fma_inplace_call38.txt
