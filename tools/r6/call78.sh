#!/bin/bash
# tap_stream_kernel: an XCD takes a contiguous run of chunks (tap_xcont = 1): kernel tests, SlowFast A/B, per-op times
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lateral or narrow or tap or stream or dense" 2>&1 | tail -3
for rep in 1 2 3; do
  for knob in 0 1; do
    timeout 300 python bench.py --workload slowfast_r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune tap_xcont=$knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slowfast_r50 tap_xcont=$knob rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_tap_xcont_call78.txt
for knob in 0 1; do
  echo "tap_xcont=$knob"
  PV_BENCH_VERBOSE=2 timeout 300 python bench.py --workload slowfast_r50 --steps 10 --warmup 3 --no-cpu-baseline --tune tap_xcont=$knob 2>&1 >/dev/null | grep "k3x1x1\|k1x3x3\|k7x1x1\|k5x1x1\|lateral\|fuse" | grep -v "c640\|c1024\|c1280\|c2048\|c256->\|c512->\|c128->128" | awk '{print $2, $(NF-5)}' | head -40
done 2>&1 | tee -a gpurun_out/r6/model_ab_tap_xcont_call78.txt | paste - - | head -0
tail -90 gpurun_out/r6/model_ab_tap_xcont_call78.txt | awk '/tap_xcont=/{k=$1} /^conv|^lat|^fuse/{t[k]+=$2} END{for (k in t) print k, "sum of the listed narrow layers:", t[k], "ms"}'
