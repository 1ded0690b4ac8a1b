#!/bin/bash
# Round-3 micro visit: row-kernel ablations, attention geometries, SQ/LDS counters of the row kernels, the real-RCCL test.
R=$PWD
OUT=$R/gpurun_out/r3b
mkdir -p $OUT
python tools/bench_mlp.py --iters 30 > $OUT/bench_mlp.txt 2>&1; cat $OUT/bench_mlp.txt
python tools/bench_attn.py > $OUT/bench_attn.txt 2>&1; cat $OUT/bench_attn.txt
python -m pytest tests/test_head_comm.py -m gpu -x -q > $OUT/pytest_comm.log 2>&1; tail -5 $OUT/pytest_comm.log
python bench.py --workload x3d_m --head-comm require --no-secondary --no-cpu-baseline --no-roofline > $OUT/bench_headcomm.json 2> $OUT/bench_headcomm.err; cat $OUT/bench_headcomm.json; tail -3 $OUT/bench_headcomm.err
bash tools/gpu_pmc_mlp.sh gpurun_out/r3b/pmc_mlp; cat $OUT/pmc_mlp/pmc_summary.txt | head -80
rm -rf $OUT/pmc_mlp/pmc
