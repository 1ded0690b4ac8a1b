#!/bin/bash
# SQ_INST_LEVEL_VMEM read through kernel durations only (no other counter's unit involved): level / (waves x duration) for the
# calibration kernels (1 / 4 loads in flight per wave by construction) and for the eight-phase GEMMs
R=$PWD
mkdir -p $R/gpurun_out/r6
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d /tmp/cal -o c -- $R/tools/micro/vmem_level.bin > /tmp/cal.log 2>&1 || tail -3 /tmp/cal.log
for SHAPE in "sf conv_a res4 slow" "sf conv_a res5 slow" "big gemm"; do
  TAG=$(echo $SHAPE | tr ' ' '_')
  timeout 200 rocprofv3 --pmc SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d /tmp/g_$TAG -o c -- python $R/tools/bench_gemm.py "$SHAPE" > /tmp/g_$TAG.log 2>&1 || tail -3 /tmp/g_$TAG.log
done
python - <<'PY' | tee $R/gpurun_out/r6/pmc_vmem_level_durations_call107.txt
import csv, glob, collections, re
def collect(d, pat):
    lvl, dur, waves = collections.defaultdict(float), collections.defaultdict(float), {}
    n = collections.Counter()
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            m = re.search(pat, row["Kernel_Name"])
            if m and row["Counter_Name"] == "SQ_INST_LEVEL_VMEM":
                k = m.group(1); lvl[k] += float(row["Counter_Value"]); n[k] += 1
                waves[k] = int(row["Grid_Size"]) // 64
    for f in glob.glob(d + "/*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            m = re.search(pat, row["Kernel_Name"])
            if m: dur[m.group(1)] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return {k: (lvl[k], dur[k], waves[k], n[k]) for k in lvl}
cal = collect("/tmp/cal", r"(chase_kernel<\d>)")
scale = None
for k, (l, ns, w, n) in sorted(cal.items()):
    r = l / (w * ns)      # level units per wave-nanosecond, summed over the same dispatches
    print("%s: %d dispatches, %d waves, %.3f ms in total, level %.4g -> %.5f units per wave-ns (%s loads in flight per wave by construction)" % (k, n, w, ns / 1e6, l, r, k[-2]))
    if k.endswith("<1>"): scale = r
import os
for d in sorted(glob.glob("/tmp/g_*")):
    if not os.path.isdir(d): continue
    for k, (l, ns, w, n) in collect(d, r"(\w*gemm\w*)").items():
        r = l / (w * ns)
        print("%s %s: %d dispatches, %d waves, %.1f us per dispatch (counter pass), level %.4g -> %.5f units per wave-ns = %.2f vector-memory instructions in flight per wave (calibration: %.5f per instruction)" % (
            os.path.basename(d), k, n, w, ns / n / 1e3, l, r, r / scale, scale))
PY
