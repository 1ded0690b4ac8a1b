"""Copy one tools/gpu_evidence.sh result set from gpurun_out/ (scratch) into profiles/<round>/ (tracked) and merge
its PMC traffic figures into profiles/traffic.json, stamped with the commit the evidence was measured on.
    python tools/adopt_evidence.py r2 <commit>      # gpurun_out/evidence_r2/* -> profiles/r2/
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, commit = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "evidence_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    tj_path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tj_path))
    stamp = tj.setdefault("_measured_on", {})
    for f in sorted(glob.glob(os.path.join(src, "*"))):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
        if f.endswith("_pmc.md"):
            for line in open(f):
                if line.startswith("TRAFFIC "):
                    for wl, fams in json.loads(line[len("TRAFFIC "):]).items():
                        tj[wl] = fams          # replace the workload's families: they all come from one run
                        stamp[wl] = commit
                        print("traffic.json:", wl, sorted(fams), "@", commit)
    json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)
    open(tj_path, "a").write("\n")


if __name__ == "__main__":
    main()
