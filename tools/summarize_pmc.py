"""Fold rocprofv3 PMC passes (tools/gpu_profile.sh) into profiles/traffic.json.

HBM traffic per launch of a kernel family = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes:
FETCH_SIZE / WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE reports exactly half of the bytes of
wide coalesced reads (/opt/skills/guides/MI355X_MICROARCH.md, HBM section) -- WRITE_SIZE is used as is.

    python tools/summarize_pmc.py <prof_dir> <workload> <label>=<kernel regex> [...]
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def read_counter(path, name):
    per = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != name:
                continue
            k = row["Kernel_Name"]
            per[k][0] += 1
            per[k][1] += float(row["Counter_Value"])
    return per


def main():
    prof_dir, workload = sys.argv[1], sys.argv[2]
    fetch = read_counter(os.path.join(prof_dir, "fetch_counter_collection.csv"), "FETCH_SIZE")
    write = read_counter(os.path.join(prof_dir, "write_counter_collection.csv"), "WRITE_SIZE")
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    wl = data.setdefault(workload, {})
    for spec in sys.argv[3:]:
        label, rx = spec.split("=", 1)
        pat = re.compile(rx)
        n = sum(v[0] for k, v in fetch.items() if pat.search(k))
        fkb = sum(v[1] for k, v in fetch.items() if pat.search(k))
        nw = sum(v[0] for k, v in write.items() if pat.search(k))
        wkb = sum(v[1] for k, v in write.items() if pat.search(k))
        if n == 0 or nw == 0:
            print("no dispatches match", rx)
            continue
        fb, wb = 2.0 * fkb * 1024 / n, wkb * 1024 / nw
        wl[label] = {"kernel_regex": rx, "dispatches_profiled": n,
                     "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                     "hbm_bytes_per_launch": round(fb + wb),
                     "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, gfx950 FETCH_SIZE x2 correction"}
        print(label, wl[label])
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
