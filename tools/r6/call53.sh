#!/bin/bash
# the stem's compacted second output for res2.0's strided shortcut: kernel + x3d tests, then a same-box A/B
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3d.py tests/test_gpu_models.py tests/test_gpu_checkpoint.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_full_geometry.py -q -m gpu -x -k "x3d and not stress" 2>&1 | tail -3
for rep in 1 2 3; do
  for w in x3d_m x3d_l; do
    for v in 1 0; do
      timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune stem_compact_shortcut=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w stem_compact_shortcut=$v rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/r6/model_ab_stem_compact_shortcut_call53.txt
PV_BENCH_VERBOSE=2 python bench.py --workload x3d_m --streams 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep " op " | head -5
