"""What 16-bit storage alone costs against the fp32 reference -- CPU only, no kernel of this repo involved.

For every BASELINE workload (one clip, full geometry, calibrated weights = trained-like BatchNorm statistics, logits
O(1-10)) the fp32 oracle is compared with the SAME oracle evaluated with (a) the dense weights and the input rounded
to the format and (b) additionally every tensor the 16-bit deploy form stores rounded where it is stored
(oracle/functional.py::storage_emulation).  Exact fp32 arithmetic everywhere else.  Metric: max|d| / max|logits|.
This is the floor under `bf16_vs_fp32_oracle` of tools/parity_full.py: no arithmetic that holds bf16 weights and
activations can be closer to the fp32 reference on these instances.  `*_self_sensitivity_1ulp` is the floor under
`bf16_vs_emulated_oracle`: the 16-bit evaluation against ITSELF after a one-fp32-ulp nudge of every stored value.

    python tools/storage_floor.py [--json profiles/r3/storage_floor.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402


def floors(workload, fill="calibrated", formats=("bf16", "fp16"), sensitivity=True):
    from bench import oracle_forward, synth_input
    from oracle import functional as OF
    from oracle.weights import quantize_like_kernels
    from parity_full import filled_model, rel
    m, shape = filled_model(workload, fill)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth_input(shape, 1, 99)
    fn = oracle_forward(workload)
    out = {"workload": workload, "fill": fill}
    with torch.no_grad():
        want = fn(sd, x)
        out["logit_absmax"] = round(want.abs().max().item(), 4)
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            if name not in formats:
                continue
            sd_b = quantize_like_kernels(sd)
            sd_q = {k: (sd[k].to(dt).float() if sd_b[k] is not sd[k] else sd[k]) for k in sd}
            xq = [t.to(dt).float() for t in x] if isinstance(x, list) else x.to(dt).float()
            out[name + "_weights_only"] = rel(fn(sd_q, xq), want)
            with OF.storage_emulation(dt):
                base = fn(sd_q, xq)
            out[name + "_storage"] = rel(base, want)
            # how reproducible is that evaluation itself?  The same oracle with every stored value nudged by about one
            # fp32 ulp (x (1 + 1e-7 N(0,1))) BEFORE it is rounded to the format: a few roundings flip, and the random-weight
            # network amplifies them.  An implementation whose fp32 arithmetic differs from the oracle's in the last bit
            # (accumulation order, FMA contraction, exp / sigmoid approximations) cannot agree with it better than this.
            if not sensitivity:
                continue
            g = torch.Generator().manual_seed(1)
            OF._STORE = lambda t: (t * (1 + 1e-7 * torch.randn(t.shape, generator=g))).to(dt).float()
            try:
                out[name + "_self_sensitivity_1ulp"] = rel(fn(sd_q, xq), base)
            finally:
                OF._STORE = None
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="x3d_m,x3d_l,slowfast_r50,mvit_b_32x3")
    ap.add_argument("--fills", default="trained_like,calibrated")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    rows = []
    for w in a.workloads.split(","):
        for f in a.fills.split(","):
            rows.append(floors(w, f))
            print(json.dumps(rows[-1]), flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)
