#!/bin/bash
# vector-memory instructions in flight per wave, per kernel symbol, for the four workloads (single-plan form, replays only), calibrated in the same call
R=$PWD
mkdir -p $R/gpurun_out/r6
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d /tmp/cal -o c -- $R/tools/micro/vmem_level.bin > /tmp/cal.log 2>&1 || tail -3 /tmp/cal.log
for WL in x3d_m mvit_b_32x3 slowfast_r50 x3d_l; do
  timeout 400 rocprofv3 --pmc SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d /tmp/w_$WL -o c -- python $R/bench.py --workload $WL --streams 1 --no-secondary --no-cpu-baseline --no-sustained --no-roofline --steps 5 --warmup 1 > /tmp/w_$WL.log 2>&1 || tail -3 /tmp/w_$WL.log
done
python - <<'PY' | tee $R/gpurun_out/r6/vmem_in_flight_call108.txt
import csv, glob, collections, re, os
def collect(d, pat):
    lvl, dur, wns, n = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
    disp = {}
    for f in glob.glob(d + "/*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            m = re.search(pat, row["Kernel_Name"])
            if m: disp[row["Dispatch_Id"]] = (m.group(1), float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != "SQ_INST_LEVEL_VMEM" or row["Dispatch_Id"] not in disp: continue
            k, ns = disp[row["Dispatch_Id"]]
            lvl[k] += float(row["Counter_Value"]); dur[k] += ns; wns[k] += (int(row["Grid_Size"]) // 64) * ns; n[k] += 1
    return lvl, dur, wns, n
lvl, dur, wns, n = collect("/tmp/cal", r"(chase_kernel<\d>)")
scale = lvl["chase_kernel<1>"] / wns["chase_kernel<1>"]
print("calibration: chase_kernel<1> %.5f level units per wave-ns = 1 load in flight; chase_kernel<4> reads %.2f" % (scale, lvl["chase_kernel<4>"] / wns["chase_kernel<4>"] / scale))
for d in sorted(glob.glob("/tmp/w_*")):
    if not os.path.isdir(d): continue
    lvl, dur, wns, n = collect(d, r"(\w+_kernel)\b")
    tot = sum(dur.values())
    print("== %s (kernel time %.2f ms over the profiled replays)" % (os.path.basename(d)[2:], tot / 1e6))
    for k in sorted(dur, key=lambda k: -dur[k])[:12]:
        print("  %-28s %5.1f %% of kernel time, %4d dispatches, %7.1f us each, %6.2f vector-memory instructions in flight per wave (the whole grid counted as resident)" % (
            k, 100 * dur[k] / tot, n[k], dur[k] / n[k] / 1e3, lvl[k] / wns[k] / scale))
PY
