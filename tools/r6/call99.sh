#!/bin/bash
# conv_b -> conv_c in one launch (pv_conv3d pw2_*, tap_stream_kernel's PW2 mode): kernel cases, SlowFast-R50 parity with the option on, per-op table, three interleaved pairs
mkdir -p gpurun_out/r6
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_b_and_pointwise or pointwise_conv_behind or narrow_dense_conv or lateral" 2>&1 | tail -15 ) | tee gpurun_out/r6/kernel_cases_conv_bc_call99.txt
timeout 900 python tools/parity_full.py --workloads slowfast_r50 --fills trained_like --tune fuse_bc=1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    print({k: d.get(k) for k in ('workload','batch','streams','fp32_vs_oracle','bf16_vs_emulated_oracle','bf16_vs_fp32_oracle','storage_floor','top1_agree')})
" | tee gpurun_out/r6/parity_conv_bc_call99.txt
for t in 0 1; do
  PV_BENCH_VERBOSE=2 timeout 300 python bench.py --workload slowfast_r50 --streams 1 --no-secondary --no-cpu-baseline --tune fuse_bc=$t 2>&1 >/dev/null | grep -E "^  op (conv_b|conv_c|conv_bc)" > gpurun_out/r6/per_op_conv_bc_${t}_call99.txt
  echo "fuse_bc=$t: $(awk '{s+=$(NF-5)} END {print s, NR}' gpurun_out/r6/per_op_conv_bc_${t}_call99.txt) ms over conv_b / conv_c / conv_bc ops"
done
grep conv_bc gpurun_out/r6/per_op_conv_bc_1_call99.txt | cut -c1-150
run() { timeout 300 python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --tune "$2" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 rep $3:', d['value'], 'clips/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do
  for knob in "fuse_bc=0" "fuse_bc=1"; do run slowfast_r50 $knob $rep; done
done 2>&1 | tee gpurun_out/r6/model_ab_conv_bc_call99.txt
