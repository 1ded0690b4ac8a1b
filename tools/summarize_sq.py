"""Fold the SQ / MFMA PMC passes (tools/gpu_pmc_sq.sh, tools/gpu_pmc_mfma.sh) into a markdown table.

    python tools/summarize_sq.py <workload> <sq_dir>

Per kernel family (top 8 by wave cycles): share of wave time spent issuing (any / VALU), parked on s_waitcnt or a
barrier (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY), VALU instructions per wave, LDS bank-conflict cycles per
active LDS cycle.
"""
import collections
import csv
import os
import re
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    if not os.path.exists(path):
        return agg
    for row in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
        k = re.sub(r"^void ", "", k)
        k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)
        k = re.sub(r"\(pv_.*", "", k)[:46]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    return agg


def main():
    wl, sq_dir = sys.argv[1:3]
    sq, lds = load(os.path.join(sq_dir, "sq_counter_collection.csv")), load(os.path.join(sq_dir, "lds_counter_collection.csv"))
    print("### %s\n" % wl)
    print("| kernel | share of wave cycles | issuing (VALU) | SQ_WAIT_ANY | SQ_WAIT_INST_ANY | VALU insts / wave | LDS conflict / active |")
    print("|---|---|---|---|---|---|---|")
    tot = sum(v["SQ_WAVE_CYCLES"] for v in sq.values()) or 1.0
    for k, v in sorted(sq.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:8]:
        wc = v["SQ_WAVE_CYCLES"] or 1.0
        l = lds.get(k, {})
        conflict = (l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"]) if l.get("SQ_LDS_IDX_ACTIVE") else 0.0
        print("| `%s` | %.0f %% | %.0f %% (%.0f %%) | %.0f %% | %.0f %%| %.0f | %.2f |" % (
            k, 100 * wc / tot, 100 * v["SQ_ACTIVE_INST_ANY"] / wc, 100 * v["SQ_ACTIVE_INST_VALU"] / wc,
            100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1.0),
            conflict))
    print()


if __name__ == "__main__":
    main()
