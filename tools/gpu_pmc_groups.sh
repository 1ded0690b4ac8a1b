#!/bin/bash
# SQ counters per (kernel, grid size) during one bench.py run: which launches are issue-bound, which are parked.
# Usage: tools/gpu_pmc_groups.sh <tag> <pattern> <bench args...>
TAG=$1; PAT=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/pmcg_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary --no-sustained --no-roofline --steps 3 --warmup 1"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/raw -o sq -- $B "$@" > $OUT/run.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/raw -o lds -- $B "$@" >> $OUT/run.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $OUT/raw -o mem -- $B "$@" >> $OUT/run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/raw -o mem2 -- $B "$@" >> $OUT/run.log 2>&1
python - <<PY > $OUT/summary.txt
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for which in ("sq", "lds", "mem", "mem2"):
    f = glob.glob("$OUT/raw/**/%s_counter_collection.csv" % which, recursive=True)
    if not f: continue
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "$PAT" not in k: continue
        k = k.replace("void (anonymous namespace)::", "")[:48] + " grid=" + row.get("Grid_Size", "?")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]].add(row["Dispatch_Id"])
for k, c in sorted(agg.items()):
    per = {name: v / max(len(n[k][name]), 1) for name, v in c.items()}
    wc = per.get("SQ_WAVE_CYCLES", 0) or 1
    print("%s  dispatches %d" % (k, len(n[k].get("SQ_WAVE_CYCLES", []))))
    print("   waves %d  wave_cycles/wave %.0f (x4 clk)  parked %.0f%%  issue-stalled %.0f%%  active %.0f%% (VALU %.0f%%)  VALU insts/wave %.0f  busy_cycles %.0f" % (
        per.get("SQ_WAVES", 0), wc / max(per.get("SQ_WAVES", 1), 1), 100 * per.get("SQ_WAIT_ANY", 0) / wc, 100 * per.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * per.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * per.get("SQ_ACTIVE_INST_VALU", 0) / wc, per.get("SQ_INSTS_VALU", 0) / max(per.get("SQ_WAVES", 1), 1), per.get("SQ_BUSY_CYCLES", 0)))
    w = max(per.get("SQ_WAVES", 1), 1)
    print("   per wave: LDS insts %.0f (active %.0f, issue-stall %.0f quad-cycles)  VMEM rd %.0f wr %.0f (active %.0f)  SALU %.0f (active %.0f)" % (
        per.get("SQ_INSTS_LDS", 0) / w, per.get("SQ_ACTIVE_INST_LDS", 0) / w, per.get("SQ_WAIT_INST_LDS", 0) / w, per.get("SQ_INSTS_VMEM_RD", 0) / w,
        per.get("SQ_INSTS_VMEM_WR", 0) / w, per.get("SQ_ACTIVE_INST_VMEM", 0) / w, per.get("SQ_INSTS_SALU", 0) / w, per.get("SQ_ACTIVE_INST_SCA", 0) / w))
    print("   GRBM_GUI_ACTIVE %.0f  FETCH_SIZE %.0f  WRITE_SIZE %.0f  TCC hit %.0f miss %.0f" % (per.get("GRBM_GUI_ACTIVE", 0), per.get("FETCH_SIZE", 0), per.get("WRITE_SIZE", 0), per.get("TCC_HIT_sum", 0), per.get("TCC_MISS_sum", 0)))
PY
rm -rf $OUT/raw
cat $OUT/summary.txt
