#!/bin/bash
# A/B of emitter knobs on one box.  Usage: tools/gpu_r3_ab.sh <workload> "<tune1>" "<tune2>" ...
R=$PWD; OUT=$R/gpurun_out/r3ab; mkdir -p $OUT
WL=$1; shift
for T in "$@"; do
  TAG=$(echo "$T" | tr '=,' '__'); [ -z "$TAG" ] && TAG=base
  PV_BENCH_VERBOSE=1 python bench.py --workload $WL --streams 1 --no-secondary --no-cpu-baseline --no-sustained ${T:+--tune $T} > $OUT/${WL}_s1_$TAG.json 2> $OUT/${WL}_s1_$TAG.txt
  python -c "import json; d=json.load(open('$OUT/${WL}_s1_$TAG.json')); print('$WL streams1 [$T]', d['value'], d['ms_per_step'], d['roofline']['launches_total'])"
  grep "n=" $OUT/${WL}_s1_$TAG.txt | head -8
  python bench.py --workload $WL --no-secondary --no-cpu-baseline --no-roofline ${T:+--tune $T} > $OUT/${WL}_$TAG.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/${WL}_$TAG.json')); print('$WL default  [$T]', d['value'], d['ms_per_step'], d['sustained']['value'])"
done
