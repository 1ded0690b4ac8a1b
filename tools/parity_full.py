"""Parity of the deploy form at the FULL BASELINE.json geometries against the CPU oracle (run on the GPU box).

For every workload: fp32 deploy vs the oracle; bf16 deploy vs (a) the oracle on the kernels' quantisation of the
weights and the input (kernel isolation) and (b) the UNQUANTISED fp32 oracle -- the north-star comparator.
Metric: max|d| / max|oracle| over the logits.  Prints one line per case; `--json` appends them to a file.

    python tools/parity_full.py [--workloads x3d_m,x3d_l,slowfast_r50,mvit_b_32x3] [--fills reference_style,deterministic]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def case(workload, fill, batch=1):
    from bench import make_model, oracle_forward, synth_input
    from oracle.weights import deterministic_fill, quantize_like_kernels, reference_style_fill
    from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
    out = {"workload": workload, "fill": fill, "batch": batch}
    torch.manual_seed(0)
    m, shape = make_model(workload)
    (reference_style_fill if fill == "reference_style" else deterministic_fill)(m, 0).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth_input(shape, batch, 99)
    fn = oracle_forward(workload)
    t0 = time.time()
    with torch.no_grad():
        want = fn(sd, x)
        sd_q = quantize_like_kernels(sd)
        xq = [t.bfloat16().float() for t in x] if isinstance(x, list) else x.bfloat16().float()
        want_q = fn(sd_q, xq)
    out["oracle_s"] = round(time.time() - t0, 2)
    out["logit_absmax"] = round(want.abs().max().item(), 4)
    out["logit_std"] = round(want.std().item(), 4)
    rel = lambda a, b: (a.float().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-9)
    transmute_model(m, "mi355x")
    for dtype, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        xd = [t.cuda().to(dtype) for t in x] if isinstance(x, list) else x.cuda().to(dtype)
        dm = convert_to_deployable_form(m, xd, dtype=dtype)
        got = dm(list(xd) if isinstance(xd, list) else xd).float().cpu()
        if tag == "fp32":
            out["fp32_vs_oracle"] = rel(got, want)
        else:
            out["bf16_vs_quantised_oracle"] = rel(got, want_q)
            out["bf16_vs_fp32_oracle"] = rel(got, want)
            out["quantised_oracle_vs_fp32_oracle"] = rel(want_q, want)
            out["top1_agree"] = bool((got.argmax(1) == want.argmax(1)).all())
        del dm
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="x3d_m,x3d_l,slowfast_r50,mvit_b_32x3")
    ap.add_argument("--fills", default="reference_style,deterministic")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    rows = []
    for w in a.workloads.split(","):
        for f in a.fills.split(","):
            r = case(w, f)
            rows.append(r)
            print(json.dumps(r), flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)
