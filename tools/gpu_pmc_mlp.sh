#!/bin/bash
# SQ / LDS counters of the fused row kernels (tools/bench_mlp.py incl. its ablation builds).  Usage: tools/gpu_pmc_mlp.sh <outdir>
R=$PWD; OUT=$R/$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc -o sq -- python $R/tools/bench_mlp.py --iters 2 > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -o lds -- python $R/tools/bench_mlp.py --iters 2 > $OUT/pmc_lds.log 2>&1
python - <<PY > $OUT/pmc_summary.txt
import csv, collections, glob, re
for which in ("sq", "lds"):
    f = glob.glob("$OUT/pmc/**/%s_counter_collection.csv" % which, recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "mlp_rows" not in k and "ln_linear" not in k: continue
        k = re.sub(r".*(mlp_rows_kernel|ln_linear_rows_kernel)", r"\1", k)[:60] + " grid=" + row.get("Grid_Size", "?")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k, c in sorted(agg.items()):
        print(which, k, "dispatches", len(n[k]))
        for name, v in sorted(c.items()):
            print("   %-28s %16.0f per dispatch" % (name, v / len(n[k])))
PY
