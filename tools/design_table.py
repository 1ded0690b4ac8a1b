"""Markdown rows of DESIGN.md's "Measured" table from one evidence set (profiles/<tag>/<workload>_bench_*.json).
    python tools/design_table.py r5
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = [("x3d_m", "X3D-M 16×224², b=32 (`configs[1]`, bench default; 2 branches)"),
         ("mvit_b_32x3", "MViT-B 32×3 224², b=8 (`configs[3]`; 2 branches)"),
         ("slowfast_r50", "SlowFast-R50 8×8 256², b=16 (`configs[2]`; 2 branches)"),
         ("x3d_l", "X3D-L 16×224², b=32 per GPU (`configs[4]`, one GPU of the eight)")]


def main():
    tag = sys.argv[1]
    d = os.path.join(ROOT, "profiles", tag)
    print("| Workload | clips/s (default form) | ms/step (p10 / p50 / p90) | single plan | dominant kernel symbol: launches, ms/step, per launch | "
          "algorithmic rate → `roofline.frac` | PMC traffic / algorithmic bytes per launch | whole model: SURVEY byte model · plan's own bytes · MFMA | "
          "kernels ≥ 5 % of the step (ms) | CPU baseline (threads: clips/s) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for wl, label in NAMES:
        a = json.load(open(os.path.join(d, wl + "_bench_default.json")))
        b = json.load(open(os.path.join(d, wl + "_bench_streams1.json")))
        r, r1 = a["roofline"], b["roofline"]
        unit = r["unit"]
        rate = "%.2f TB/s" % (r["achieved"] / 1e3) if unit == "GB/s" else "%.0f TFLOP/s" % r["achieved"]
        second = ""
        if r.get("second_roof"):
            s2 = r["second_roof"]
            second = "; %s roof: %.1f of %.1f %s = %.2f" % (s2.get("bound"), s2.get("achieved", 0), s2.get("peak", 0), s2.get("unit", ""), s2.get("frac", 0))
        others = {k: v for k, v in (r.get("second_roofs") or {}).items() if k != r["kernel"]}
        if others:      # every fp32-stencil symbol of the step against the VALU roof, dominant or not (round 6)
            second += "; other stencil symbols on the valu roof: " + ", ".join(
                "`%s` %.2f (%.2f ms)" % (k, v.get("frac", 0), v.get("kernel_ms_per_step", 0)) for k, v in sorted(others.items()))
        # PMC traffic of the SAME evidence pass (profiles/traffic.json, per kernel symbol); the JSON line itself was printed before
        # that pass was adopted and still carries the previous round's figure
        ent = (tj.get(wl, {}).get("_by_kernel") or {}).get(r["kernel"])
        tr = ent["hbm_bytes_per_launch"] if ent else r.get("traffic")
        traffic = "%.1f / %.1f MB = %.2f" % (tr / 1e6, r["alg_bytes_per_launch"] / 1e6, tr / r["alg_bytes_per_launch"]) if tr else "—"
        whole = "%.2f · %.2f · %.3f" % (r.get("model_hbm_frac") or 0, r.get("model_hbm_frac_plan") or 0, r.get("model_mfma_frac") or 0)
        ks = ", ".join("`%s` %.2f" % (k, v) for k, v in r["kernels_ms_per_step"].items())
        cb = a.get("cpu_baseline") or {}
        cpu = "; ".join("%s: %s" % kv for kv in sorted(cb.get("threads_sweep", {}).items(), key=lambda t: int(t[0])))
        sm = a["step_ms"]
        print("| %s | **%.0f** (sustained %.0f) | %.2f (%.2f / %.2f / %.2f) | %.0f / %.2f ms | `%s`: %d, %.2f ms, %.1f µs | %s = **%.2f**%s | %s | %s | %s | %s |" % (
            label, a["value"], a.get("sustained", {}).get("value", 0), a["ms_per_step"], sm["p10"], sm["p50"], sm["p90"],
            b["value"], b["ms_per_step"], r["kernel"], r["launches_per_step"], r["kernel_ms_per_step"], r["avg_launch_ms"] * 1e3,
            rate, r["frac"], second, traffic, whole, ks, cpu))


if __name__ == "__main__":
    main()
