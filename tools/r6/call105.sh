#!/bin/bash
# the eight-phase GEMM kernels: how long is a DMA (global -> LDS) in flight?  SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM = average cycles a
# vector-memory instruction is outstanding; against the cycles a K tile's MFMAs take this says whether the loop's two-tile lookahead covers it
R=$PWD
mkdir -p $R/gpurun_out/r6
cd /tmp; export TMPDIR=/tmp
for SHAPE in "sf conv_a res4 slow" "sf conv_a res5 slow" "big gemm"; do
  TAG=$(echo $SHAPE | tr ' ' '_')
  for PASS in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    N=$(echo $PASS | cut -d' ' -f1)
    timeout 200 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d /tmp/lvl_$TAG -o $N -- python $R/tools/bench_gemm.py "$SHAPE" > /tmp/lvl_$TAG.$N.log 2>&1 || tail -3 /tmp/lvl_$TAG.$N.log
  done
  python - "$SHAPE" /tmp/lvl_$TAG <<'PY'
import csv, glob, sys, collections, os
shape, d = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        import re
        m = re.search(r"(\w*(gemm|conv|stream)\w*)", row["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
for k, c in agg.items():
    g = lambda x: c.get(x, 0.0)
    print("%s | %s | dispatches %d | VMEM insts %.3g level %.3g -> %.0f cycles in flight per VMEM inst | LDS insts %.3g level %.3g -> %.0f | wave cycles %.3g wait_any %.2f wait_inst_any %.2f wait_inst_lds %.2f active_inst %.2f | GUI_ACTIVE per dispatch %.0f" % (
        shape, k, len(n[k]), g("SQ_INSTS_VMEM"), g("SQ_INST_LEVEL_VMEM"), g("SQ_INST_LEVEL_VMEM") / max(g("SQ_INSTS_VMEM"), 1),
        g("SQ_INSTS_LDS"), g("SQ_INST_LEVEL_LDS"), g("SQ_INST_LEVEL_LDS") / max(g("SQ_INSTS_LDS"), 1), g("SQ_WAVE_CYCLES"),
        g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
        g("SQ_WAIT_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
        g("GRBM_GUI_ACTIVE") / max(len(n[k]), 1)))
PY
done 2>&1 | tee $R/gpurun_out/r6/pmc_vmem_level_call105.txt
