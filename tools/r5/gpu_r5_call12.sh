#!/bin/bash
# round 5, GPU call 12: rocprofv3 evidence of the final library for the four driver-run workloads (kernel stats, op alignment,
# PMC passes per kernel symbol)
bash tools/gpu_evidence.sh r5 x3d_m mvit_b_32x3 slowfast_r50 x3d_l 2>&1 | tail -60
