"""Where does the bf16 deviation of a deep X3D come from?  (run on the GPU box)

For every block of the deploy form: (a) the block ALONE on the quantised oracle's input of that block
(per-block kernel error) and (b) the deviation of the chained forward after that block (accumulated).
A smooth geometric growth of (b) with flat (a) = amplification by the network instance; a jump = a kernel bug.

    python tools/x3d_depth_probe.py [--workload x3d_l] [--fill reference_style|deterministic]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="x3d_l")
    ap.add_argument("--fill", default="reference_style")
    a = ap.parse_args()
    from bench import make_model, synth_input
    from oracle import functional as OF
    from oracle.weights import deterministic_fill, quantize_like_kernels, reference_style_fill
    from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
    torch.manual_seed(0)
    m, shape = make_model(a.workload)
    (reference_style_fill if a.fill == "reference_style" else deterministic_fill)(m, 0).eval()
    x = synth_input(shape, 1, 99)
    sd_q, x_q = quantize_like_kernels({k: v.clone() for k, v in m.state_dict().items()}, x)
    with torch.no_grad():
        want, outs = OF.x3d_forward(sd_q, x_q, shape[1], shape[2], return_blocks=True)
        _, outs32 = OF.x3d_forward(m.state_dict(), x, shape[1], shape[2], return_blocks=True)
    transmute_model(m, "mi355x")
    dm = convert_to_deployable_form(m, x.cuda().bfloat16(), dtype=torch.bfloat16)
    rel = lambda g, w: (g.float().cpu() - w).abs().max().item() / max(w.abs().max().item(), 1e-9)
    rms = lambda g, w: ((g.float().cpu() - w).pow(2).mean().sqrt() / w.pow(2).mean().sqrt().clamp_min(1e-12)).item()
    ins = [x_q] + outs[:-1]
    cur = x.cuda().bfloat16()
    print("blk  alone(max/absmax)  alone(rms)   chained(max/absmax)  chained(rms)  weights-only(rms, oracle q vs fp32)  absmax")
    for i, (blk, xin, w, w32) in enumerate(zip(dm.blocks, ins, outs, outs32)):
        alone = blk(xin.cuda().bfloat16()).clone()
        cur = blk(cur).clone()
        print("%3d  %.3e          %.3e    %.3e            %.3e     %.3e                          %.3g" % (
            i, rel(alone, w), rms(alone, w), rel(cur, w), rms(cur, w), rms(w, w32), w.abs().max().item()), flush=True)


if __name__ == "__main__":
    main()
