#!/bin/bash
# SQ counters of the kernels whose name contains <pattern> during one bench.py run.
# Usage: tools/gpu_pmc_kernel.sh <tag> <pattern> <bench args...>
TAG=$1; PAT=$2; shift 2
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmck_$TAG -o sq -- python $R/bench.py --no-cpu-baseline --no-secondary --no-sustained --steps 3 --warmup 1 "$@" > $R/gpurun_out/pmck_$TAG.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmck_$TAG -o lds -- python $R/bench.py --no-cpu-baseline --no-secondary --no-sustained --steps 3 --warmup 1 "$@" >> $R/gpurun_out/pmck_$TAG.log 2>&1
python - <<PY
import csv, collections, glob
for which in ("sq", "lds"):
    f = glob.glob("$R/gpurun_out/pmck_$TAG/**/%s_counter_collection.csv" % which, recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "$PAT" not in k: continue
        k = k.replace("void (anonymous namespace)::", "")[:44]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k, c in sorted(agg.items()):
        print("$TAG", which, k, "dispatches", len(n[k]))
        for name, v in sorted(c.items()):
            print("   %-28s %14.0f per dispatch" % (name, v / len(n[k])))
PY
