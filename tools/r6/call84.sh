#!/bin/bash
# tap_stream_kernel: operand stream two K steps ahead (32- and 128-channel slabs): kernel tests, same-box A/B against the previous library
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -m gpu -x -k "lateral or narrow or tap or stream or dense or slowfast or csn or r2plus1d" 2>&1 | tail -3
OLD=$PWD/pytorchvideo_amd/_lib/old/libpv_mi355x.so
for rep in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export PV_MI355X_LIB=$OLD; else unset PV_MI355X_LIB; fi
    timeout 300 python bench.py --workload slowfast_r50 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slowfast_r50 tap prefetch depth $lib rep $rep:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"
  done
done 2>&1 | tee gpurun_out/r6/model_ab_tap_depth_call84.txt
unset PV_MI355X_LIB
for lib in old new; do
  if [ $lib = old ]; then export PV_MI355X_LIB=$OLD; else unset PV_MI355X_LIB; fi
  echo "lib=$lib"
  PV_BENCH_VERBOSE=2 timeout 300 python bench.py --workload slowfast_r50 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 >/dev/null | grep "k3x1x1\|k1x3x3\|lateral" | grep -v "c640\|c1024\|c1280\|c2048\|c256->\|c512->\|c128->128" | awk '{print $2, $(NF-5)}'
done 2>&1 | tee -a gpurun_out/r6/model_ab_tap_depth_call84.txt | tail -0
python - <<'PY'
rows={'old':[],'new':[]}; k=None
for l in open('gpurun_out/r6/model_ab_tap_depth_call84.txt'):
    l=l.strip()
    if l.startswith('lib='): k=l[4:]
    elif k and (l.startswith('conv') or l.startswith('lat')): rows[k].append(l.rsplit(' ',1))
tot=[0,0]
for (a,ta),(b,tb) in zip(rows['old'],rows['new']):
    print("%-44s %s -> %s" % (a,ta,tb)); tot[0]+=float(ta); tot[1]+=float(tb)
print("sum", tot)
PY
