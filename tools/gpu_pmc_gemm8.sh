#!/bin/bash
# SQ counters of one GEMM shape under the 128x128 kernel (gemm8=0) and the large-tile kernel (gemm8=2|4).
# Usage: tools/gpu_pmc_gemm8.sh "<shape substring>" [modes...]
SHAPE="$1"; shift
MODES=${@:-0 4}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for m in $MODES; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_g8_$m -o sq -- python $R/tools/bench_gemm.py --tune=gemm8=$m "$SHAPE" > $R/gpurun_out/pmc_g8_$m.log 2>&1
  tail -2 $R/gpurun_out/pmc_g8_$m.log
  python - <<PY
import csv, collections, glob
f = glob.glob("$R/gpurun_out/pmc_g8_$m/**/sq_counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:60]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    if row["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in agg.items():
    if "gemm" not in k: continue
    wc = c["SQ_WAVE_CYCLES"]
    print("mode $m", k, "launches", n[k])
    for name, v in sorted(c.items()):
        print("   %-28s %14.0f  %6.1f %% of wave cycles" % (name, v / n[k], 100 * v / wc if wc else 0))
PY
done
