"""slowfast_r50_detection on one MI355X: clips/s of the whole deploy form (backbone + RoI head, one graph replay) and
the per-op device time of the head.  Synthetic clips (BASELINE's SlowFast shape) and boxes.

    python tools/bench_detection.py [--batch 16] [--boxes-per-clip 8] [--steps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model  # noqa: E402
from pytorchvideo_amd.models import hub  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--boxes-per-clip", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    torch.manual_seed(0)
    B, R = a.batch, a.batch * a.boxes_per_clip
    model = hub.slowfast_r50_detection().eval()
    transmute_model(model, "mi355x")
    g = torch.Generator().manual_seed(0)
    fast = torch.randn((B, 3, 32, 256, 256), generator=g).to(torch.bfloat16).cuda()
    slow = fast[:, :, torch.linspace(0, 31, 8).long()].contiguous()
    xy = torch.rand((R, 2), generator=g) * 128.0
    wh = torch.rand((R, 2), generator=g) * 120.0 + 8.0
    boxes = torch.cat([torch.arange(R).remainder(B).float()[:, None], xy, xy + wh], 1).cuda()
    dm = convert_to_deployable_form(model, ([slow, fast], boxes), dtype=torch.bfloat16)
    for _ in range(a.warmup):
        dm([slow, fast], boxes)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.steps):
        scores = dm([slow, fast], boxes)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    assert tuple(scores.shape) == (R, 80) and bool(torch.isfinite(scores).all())
    print("slowfast_r50_detection bf16 batch %d, %d boxes: %.3f ms/step, %.1f clips/s, %.0f boxes/s"
          % (B, R, ms, B / ms * 1e3, R / ms * 1e3))
    ops = dm._pv_session.profile(iters=5)
    total = sum(o[2] for o in ops)
    print("per-op total %.3f ms over %d launches; head ops:" % (total, len(ops)))
    for label, _, t, alg, _ in ops:
        if label.startswith(("det.", "head.pool")):
            print("  %-40s %8.4f ms  alg %8.2f MB  -> %6.1f GB/s" % (label.split("|")[0], t, alg / 1e6, alg / 1e6 / max(t, 1e-6)))


if __name__ == "__main__":
    main()
