#!/bin/bash
# PMC of attn_w64_kernel on the micro-benchmark
bash tools/r6/pmc_attn.sh w64_call65 1 2>&1 | tee gpurun_out/r6/pmc_attn_w64_call65.txt
