#!/bin/bash
# SQ / LDS counters of the attention kernels on the micro-benchmark (tools/bench_attn.py <knob>).  Usage: pmc_attn.sh <tag> <knob>
TAG=$1; KNOB=$2
R=$PWD
mkdir -p $R/gpurun_out/r6
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/bench_attn.py $KNOB"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_${TAG} -o sq -- $CMD > $R/gpurun_out/pmc_attn_${TAG}.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_${TAG} -o lds -- $CMD >> $R/gpurun_out/pmc_attn_${TAG}.log 2>&1
tail -3 $R/gpurun_out/pmc_attn_${TAG}.log
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$R/gpurun_out/pmc_attn_${TAG}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "attn" not in k: continue
        agg[k[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    wc = v["SQ_WAVE_CYCLES"] or 1.0
    print(k)
    for c in sorted(v): print("   %-34s %16.0f  (%.3f of wave cycles)" % (c, v[c], v[c] / wc))
PY
