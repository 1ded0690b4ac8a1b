#!/bin/bash
# calibration of SQ_INST_LEVEL_VMEM: waves with exactly 1 / 4 dependent-load chains in flight (tools/micro/vmem_level.hip), then the GEMM numbers of call 105 read with it
R=$PWD
mkdir -p $R/gpurun_out/r6
cd /tmp; export TMPDIR=/tmp
$R/tools/micro/vmem_level.bin 2>&1 | tail -6 | tee $R/gpurun_out/r6/pmc_vmem_level_calibration_call106.txt
timeout 200 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/cal -o c -- $R/tools/micro/vmem_level.bin > /tmp/cal.log 2>&1 || tail -3 /tmp/cal.log
python - <<'PY' | tee -a $R/gpurun_out/r6/pmc_vmem_level_calibration_call106.txt
import csv, glob, collections, os, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("/tmp/cal/**/*counter_collection.csv", recursive=True) + glob.glob("/tmp/cal/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(chase_kernel<\d>)", row["Kernel_Name"])
        if not m: continue
        agg[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"]); n[m.group(1)].add(row["Dispatch_Id"])
for k, c in sorted(agg.items()):
    g = lambda x: c.get(x, 0.0)
    print("%s | dispatches %d | VMEM insts %.4g (expected %d per dispatch: 1024 waves x 2000 trips x chains) | level %.4g -> level / insts %.1f | wave cycles %.4g -> level / wave cycles %.3f (chains in flight all the time: %s)" % (
        k, len(n[k]), g("SQ_INSTS_VMEM"), 1024 * 2000 * int(k[-2]), g("SQ_INST_LEVEL_VMEM"), g("SQ_INST_LEVEL_VMEM") / max(g("SQ_INSTS_VMEM"), 1),
        g("SQ_WAVE_CYCLES"), g("SQ_INST_LEVEL_VMEM") / max(g("SQ_WAVE_CYCLES"), 1), k[-2]))
PY
