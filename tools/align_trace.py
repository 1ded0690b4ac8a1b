"""Match a rocprofv3 kernel trace with the launch plan: every op of the plan is one kernel launch, so the i-th
dispatch after the first kernel of a replay is op i.  Prints, per op label family, the kernel symbols behind it and
the rocprofv3 average duration (End - Start timestamps) next to the HIP-event time bench.py measured in-process.

    python tools/align_trace.py <per_op.txt from PV_BENCH_VERBOSE=2> <trace_kernel_trace.csv> [first-kernel substring]
"""
import collections
import csv
import re
import sys


def main(per_op, trace, first="stem"):
    ops = []
    for line in open(per_op):
        m = re.match(r"\s+op (.+?)\s+([0-9.]+) ms", line)
        if m:
            ops.append((m.group(1).split("|")[0], float(m.group(2))))
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    # MViT's pooling ops on large grids launch a second, tiny kernel for the cls row: not an op of its own
    rows = [r for r in rows if "_prefix_kernel" not in r["Kernel_Name"]]
    starts = []
    for i, r in enumerate(rows):   # a replay begins at the first-kernel symbol, at least one plan length after the previous one
        if first in r["Kernel_Name"] and (not starts or i - starts[-1] >= len(ops)):
            starts.append(i)
    fam = lambda l: l.split(".")[0] if l.startswith(("conv_b", "conv_ab")) else l
    ev, rp, sym, n_replays = collections.OrderedDict(), collections.defaultdict(float), collections.defaultdict(set), 0
    for label, ms in ops:
        ev[fam(label)] = ev.get(fam(label), 0.0) + ms
    for s in starts:
        seg = rows[s:s + len(ops)]
        if len(seg) < len(ops) or any(first in r["Kernel_Name"] for r in seg[len(ops) // 2:]):
            continue    # truncated, or not a whole replay (another one starts inside it)
        n_replays += 1
        for (label, _), r in zip(ops, seg):
            rp[fam(label)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            name = re.sub(r"^void \(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
            name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
            sym[fam(label)].add(name[:48])
    print("| op family | launches | rocprofv3 ms / replay (avg of %d replays) | HIP events ms (bench.py) | kernels |" % n_replays)
    print("|---|---|---|---|---|")
    counts = collections.Counter(fam(l) for l, _ in ops)
    for k in sorted(ev, key=lambda k: -ev[k]):
        print("| %s | %d | %.4f | %.4f | %s |" % (k, counts[k], rp[k] / n_replays, ev[k], ", ".join(sorted(sym[k]))))
    print("| **total** | %d | %.4f | %.4f | |" % (len(ops), sum(rp.values()) / n_replays, sum(ev.values())))


if __name__ == "__main__":
    main(*sys.argv[1:])
