#!/bin/bash
# PMC of attn_w64_kernel<96, 1, 4> (two 4-wave workgroups per CU) on the micro-benchmark
bash tools/r6/pmc_attn.sh w64f3_call75 3 2>&1 | grep -v "rocprofv3\|Opened" | tee gpurun_out/r6/pmc_attn_w64_form3_call75.txt
