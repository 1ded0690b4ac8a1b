#!/bin/bash
# round 6, call 83: rocprofv3 evidence of the library at this commit for the four driver-run workloads (kernel stats, op alignment,
# PMC passes per kernel symbol) in ONE pass, then the default bench line
bash tools/gpu_evidence.sh r6 x3d_m mvit_b_32x3 slowfast_r50 x3d_l 2>&1 | tail -70
python bench.py > gpurun_out/evidence_r6/bench_default_line.json 2> gpurun_out/evidence_r6/bench_default_line.err
head -c 1500 gpurun_out/evidence_r6/bench_default_line.json
