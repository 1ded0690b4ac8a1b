"""Fold the PMC passes of tools/gpu_evidence.sh into one markdown table per workload (runs on the GPU box).

    python tools/summarize_evidence.py <workload> <scratch dir with fetch/write/sq passes> <per_op.txt>

Per kernel symbol (top by time): dispatches per replay, average duration (kernel trace of the SQ pass), HBM bytes
per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (separate passes; gfx950 reports half of the bytes of wide
coalesced reads, /opt/skills/guides/MI355X_MICROARCH.md, HBM section), and from the SQ pass:
  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): busy cycles are summed over the
  1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (calibrated on a 32768 x 4096 x 4096 GEMM: 33.5 M MFMAs x 32 cycles
  against its measured 961 TFLOP/s = 38 % of 2.5 PFLOP/s),
  VALU issue share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES,
  issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES  (quad-cycle counters of the same unit: ratios are exact).
Also prints a JSON line `TRAFFIC {...}` with the per-family HBM bytes for profiles/traffic.json.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    k = re.sub(r"\(anonymous namespace\)::", "", name)
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)
    return re.sub(r"\(pv_.*", "", k)[:52]


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[k].add(row["Dispatch_Id"])
    return agg, {k: len(v) for k, v in n.items()}


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


FAMILIES = {   # op family of bench.py's roofline object -> kernel symbols behind it
    "x3d_m": {"conv_c": r"pw_stream_kernel<\d, \d, (true|false), \d+, false, (true|false)>|conv_igemm", "conv_ab": r"pwdw_plane_kernel",
              "conv_b": r"dw3_plane_kernel", "stem.conv01": r"stem_c4_dwt_kernel"},
    "mvit_b_32x3": {"attn.core": r"attn_(pipe_)?kernel", "gemm": r"gemm_glds_kernel", "layernorm": r"layernorm", "stream": r"pw_stream_kernel"},
    "slowfast_r50": {"conv_a": r"gemm_glds_kernel<false|pw_stream_kernel|conv_igemm", "gemm": r"gemm_glds_kernel", "stem.conv": r"stem7_kernel|stem_c4_kernel",
                     "narrow+lateral": r"tap_stream_kernel"},
}


def by_plan_order(d, per_op):
    """HBM bytes per op family, attributing dispatches to plan ops BY ORDER (every op is one launch; the PMC passes run
    whole replays only): exact where one kernel symbol serves several families.  {} if the passes do not line up."""
    ops = []
    for line in open(per_op):
        m = re.match(r"\s+op (.+?)\s+([0-9.]+) ms", line)
        if m:
            lab = m.group(1).split("|")[0]
            ops.append(lab.split(".")[0] if lab.startswith(("conv_b", "conv_ab")) else lab)
    out = {}
    for key, pat, ctr in (("fetch", "fetch_counter_collection.csv", "FETCH_SIZE"), ("write", "write_counter_collection.csv", "WRITE_SIZE")):
        disp = {}
        for row in csv.DictReader(open(find(d, pat))):
            if row["Counter_Name"] == ctr:
                e = disp.setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], 0.0])
                e[1] += float(row["Counter_Value"])
        seq = [disp[i] for i in sorted(disp) if "_prefix_kernel" not in disp[i][0]]   # (an op's second, cls-row launch)
        starts = [i for i, (name, _) in enumerate(seq) if "stem" in name]
        reps, last = [], -len(ops)
        for i in starts:
            if i - last >= len(ops) and i + len(ops) <= len(seq):
                reps.append(i)
                last = i
        if not reps:
            return {}
        acc, cnt = collections.defaultdict(float), collections.Counter(ops)
        for i in reps:
            for lab, (_, v) in zip(ops, seq[i:i + len(ops)]):
                acc[lab] += v * 1024
        for lab in acc:
            out.setdefault(lab, {})[key] = acc[lab] / (len(reps) * cnt[lab])
            out[lab]["n"] = len(reps) * cnt[lab]
    res = {}
    for lab, v in out.items():
        if "fetch" in v and "write" in v:
            res[lab] = {"attribution": "plan order (dispatch i of a replay = op i)", "dispatches_profiled": v["n"],
                        "fetch_bytes_per_launch": round(2 * v["fetch"]), "write_bytes_per_launch": round(v["write"]),
                        "hbm_bytes_per_launch": round(2 * v["fetch"] + v["write"]),
                        "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, gfx950 FETCH_SIZE x2 correction"}
    return res


def per_op_traffic(d, per_op):
    """Markdown rows, one per plan op: algorithmic bytes (bench.py's per-op table: GB/s x ms) beside the PMC bytes of the same
    dispatch (plan order), sorted by the bytes moved above the algorithmic count.  [] if the passes do not line up."""
    ops = []
    for line in open(per_op):
        m = re.match(r"\s+op (.+?)\s+([0-9.]+) ms\s+([0-9.]+) GB/s", line)
        if m:
            ops.append((m.group(1), float(m.group(2)), float(m.group(2)) * float(m.group(3)) * 1e6))
    per = [dict(label=l, ms=ms, alg=a, fetch=0.0, write=0.0) for l, ms, a in ops]
    for key, pat, ctr in (("fetch", "fetch_counter_collection.csv", "FETCH_SIZE"), ("write", "write_counter_collection.csv", "WRITE_SIZE")):
        disp = {}
        for row in csv.DictReader(open(find(d, pat))):
            if row["Counter_Name"] == ctr:
                e = disp.setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], 0.0])
                e[1] += float(row["Counter_Value"])
        seq = [disp[i] for i in sorted(disp) if "_prefix_kernel" not in disp[i][0]]
        starts = [i for i, (name, _) in enumerate(seq) if "stem" in name]
        reps, last = [], -len(ops)
        for i in starts:
            if i - last >= len(ops) and i + len(ops) <= len(seq):
                reps.append(i)
                last = i
        if not reps:
            return []
        for i in reps:
            for o, (_, v) in zip(per, seq[i:i + len(ops)]):
                o[key] += v * 1024 / len(reps)
    rows = []
    for o in per:
        hbm = 2 * o["fetch"] + o["write"]
        rows.append((hbm - o["alg"], "| `%s` | %.4f | %.1f | %.1f | %.1f | %.1f | %.2f |" % (
            o["label"][:64], o["ms"], o["alg"] / 1e6, 2 * o["fetch"] / 1e6, o["write"] / 1e6, hbm / 1e6, hbm / max(o["alg"], 1.0))))
    return [r for _, r in sorted(rows, key=lambda t: -t[0])]


def main():
    wl, d = sys.argv[1], sys.argv[2]
    fetch, nf = counters(find(d, "fetch_counter_collection.csv"))
    write, nw = counters(find(d, "write_counter_collection.csv"))
    sq, ns = counters(find(d, "sq_counter_collection.csv"))
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(find(d, "sq_kernel_trace.csv"))):
        dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    tot = {k: sum(v) for k, v in dur.items()}
    print("### %s  (bench.py --streams 1, replays only; %d kernel symbols)\n" % (wl, len(tot)))
    print("| kernel | dispatches | avg us (PMC pass) | HBM MB / launch (2F+W) | MFMA util | VALU issue | parked (WAIT_ANY) | issue-stalled |")
    print("|---|---|---|---|---|---|---|---|")
    for k in sorted(tot, key=lambda k: -tot[k])[:14]:
        c = sq.get(k, {})
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui else 0.0
        hb = (2.0 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) / max(nf.get(k, 1), 1)
              + write.get(k, {}).get("WRITE_SIZE", 0.0) / max(nw.get(k, 1), 1)) * 1024 / 1e6
        print("| `%s` | %d | %.1f | %.1f | %.1f %% | %.0f %% | %.0f %% | %.0f %% |" % (
            k, len(dur[k]), tot[k] / len(dur[k]), hb, 100 * mf, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc))
    traffic = by_plan_order(d, sys.argv[3]) if len(sys.argv) > 3 else {}
    rows = per_op_traffic(d, sys.argv[3]) if len(sys.argv) > 3 else []
    if rows:
        print("\nPer op (plan order), sorted by the bytes moved above the algorithmic count -- top 24 of %d:\n" % len(rows))
        print("| op | ms (HIP events) | algorithmic MB | fetch MB (2 x FETCH_SIZE) | write MB | HBM MB | HBM / algorithmic |")
        print("|---|---|---|---|---|---|---|")
        print("\n".join(rows[:24]))
        print()
    for label, rx in FAMILIES.get(wl, {}).items():
        if traffic:
            break     # exact attribution available: dispatch i of a replay is op i of the launch plan
        pat = re.compile(rx)
        ks = [k for k in nf if pat.search(k)]
        n1, n2 = sum(nf[k] for k in ks), sum(nw.get(k, 0) for k in ks)
        if not n1 or not n2:
            continue
        fb = 2.0 * sum(fetch[k]["FETCH_SIZE"] for k in ks) * 1024 / n1
        wb = sum(write[k]["WRITE_SIZE"] for k in ks if k in write) * 1024 / n2
        traffic[label] = {"kernel_regex": rx, "dispatches_profiled": n1, "fetch_bytes_per_launch": round(fb),
                          "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb),
                          "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, gfx950 FETCH_SIZE x2 correction"}
    # per kernel SYMBOL (template arguments folded): every dispatch of the symbol in the passes -- the population
    # bench.py's roofline object averages `alg_bytes_per_launch` over
    by_sym = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for k in nf:
        e = by_sym[k.split("<")[0]]
        e[0] += nf[k]
        e[1] += fetch[k].get("FETCH_SIZE", 0.0)
        e[2] += nw.get(k, 0)
        e[3] += write.get(k, {}).get("WRITE_SIZE", 0.0)
    traffic["_by_kernel"] = {
        sym: {"dispatches_profiled": e[0], "fetch_bytes_per_launch": round(2.0 * e[1] * 1024 / e[0]), "write_bytes_per_launch": round(e[3] * 1024 / e[2]),
              "hbm_bytes_per_launch": round(2.0 * e[1] * 1024 / e[0] + e[3] * 1024 / e[2]),
              "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, gfx950 FETCH_SIZE x2 correction"}
        for sym, e in by_sym.items() if e[0] and e[2]}
    print("\nTRAFFIC " + json.dumps({wl: traffic}))


if __name__ == "__main__":
    main()
