"""Teacher-forced per-block bf16 parity at the FULL BASELINE.json geometries (run on the GPU box).

For every residual block / MultiScaleBlock of a workload (X3D-M 26, X3D-L 55, SlowFast-R50 16 + 16, MViT-B 16) plus its
stem and head:

    oracle block input (what the fp32 oracle feeds that block, rounded to bf16 -- the deploy form's storage type)
        -> the block ALONE in its bf16 deploy form (its own session, its own ingest: conv_a / conv_b / SE / conv_c /
           shortcut fusions exactly as inside the full plan)
        -> compared with the fp32 oracle's output of that block on the same input (fp32 weights, fp32 arithmetic)

Metric: max|d| / max|oracle block output|, bound 1e-2 (the north star's bf16 bar) per block.  No error of an earlier
block reaches a later one, so a kernel whose arithmetic moves shows up in exactly the blocks it serves -- the isolation
the chaotic end-to-end comparison cannot give (round-3 verdict, weak #3).  The token stream of MViT stays fp32 in the
deploy form, so its block inputs are handed over unrounded.

    python tools/parity_blocks.py [--workloads x3d_m,x3d_l,slowfast_r50,mvit_b_32x3] [--fill trained_like] [--json out.jsonl]
"""
import argparse
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402


def _rel(got, want):
    return (got.float().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-12)


def _q(t):
    return t.bfloat16().float()


def _deploy_single(module, xin, dtype=torch.bfloat16):
    """One stem / res block / head alone: transmute, convert for xin's size, run, return a host fp32 tensor."""
    from pytorchvideo_amd.accelerator.mi355x import blocks as B
    blk = B.transmute_single_io(copy.deepcopy(module))
    assert blk is not None, "block declined by the transmuter: %s" % type(module).__name__
    blk.convert(tuple(xin.shape), dtype=dtype)
    got = blk(xin.cuda().to(dtype)).float().cpu()
    del blk
    return got


def _x3d_blocks(m, sd, x, clip_len, crop):
    import math
    from oracle import functional as OF
    rows = []
    xin = _q(x)
    want = OF.x3d_stem(sd, xin)
    rows.append(("blocks.0 (stem)", _rel(_deploy_single(m.blocks[0], xin), want)))
    h = want
    for s in range(1, 5):
        for i, rb in enumerate(m.blocks[s].res_blocks):
            p = "blocks.%d.res_blocks.%d" % (s, i)
            xin = _q(h)
            want = OF.x3d_res_block(sd, xin, p, (1, 2, 2) if i == 0 else (1, 1, 1))
            rows.append((p, _rel(_deploy_single(rb, xin), want)))
            h = want
    side = int(math.ceil(crop / 32))
    xin = _q(h)
    want = OF.x3d_head(sd, xin, "blocks.5", (clip_len, side, side))
    rows.append(("blocks.5 (head)", _rel(_deploy_single(m.blocks[5], xin), want)))
    return rows


def _slowfast_blocks(m, sd, x):
    from oracle import functional as OF
    from pytorchvideo_amd.accelerator.mi355x import blocks as B
    rows = []

    def fuse(s, f, p):
        w = sd[p + ".conv_fast_to_slow.weight"]
        z = torch.nn.functional.conv3d(f, w, stride=(4, 1, 1), padding=(w.shape[2] // 2, 0, 0))
        return torch.cat([s, torch.relu(OF._bn(z, sd, p + ".norm"))], 1)

    # stems + first lateral fusion: the MultiPathWayWithFuse container as one block
    xs, xf = _q(x[0]), _q(x[1])
    s = OF.res_basic_stem(sd, xs, "blocks.0.multipathway_blocks.0")
    f = OF.res_basic_stem(sd, xf, "blocks.0.multipathway_blocks.1")
    want_s = fuse(s, f, "blocks.0.multipathway_fusion")
    blk = B.transmute_multipath(copy.deepcopy(m.blocks[0]))
    blk.convert([tuple(xs.shape), tuple(xf.shape)], dtype=torch.bfloat16)
    got = blk([xs.cuda().bfloat16(), xf.cuda().bfloat16()])
    rows.append(("blocks.0 (stems + fusion) slow", _rel(got[0], want_s)))
    rows.append(("blocks.0 (stems + fusion) fast", _rel(got[1], f)))
    del blk, got
    s = want_s
    spatial = (1, 2, 2, 2)
    for st in range(4):
        for pw, name in ((0, "slow"), (1, "fast")):
            h = s if pw == 0 else f
            stage = m.blocks[st + 1].multipathway_blocks[pw]
            for i, rb in enumerate(stage.res_blocks):
                p = "blocks.%d.multipathway_blocks.%d.res_blocks.%d" % (st + 1, pw, i)
                xin = _q(h)
                sb = (1, spatial[st], spatial[st]) if i == 0 else (1, 1, 1)
                want = OF.bottleneck_res_block(sd, xin, p, (1, 1, 1), sb)
                rows.append((p, _rel(_deploy_single(rb, xin), want)))
                h = want
            if pw == 0:
                s = h
            else:
                f = h
        if st < 3:
            # lateral fusion site on the teacher-forced trajectory (its own kernel is checked in tests/test_gpu_kernels.py)
            s = fuse(s, f, "blocks.%d.multipathway_fusion" % (st + 1))
    # PoolConcatPathway + head
    xs, xf = _q(s), _q(f)
    pooled = torch.cat([torch.nn.functional.avg_pool3d(xs, (8, 7, 7), stride=1),
                        torch.nn.functional.avg_pool3d(xf, (32, 7, 7), stride=1)], 1)
    want = OF.res_basic_head(sd, pooled, "blocks.6")
    pc = B.transmute_multipath(copy.deepcopy(m.blocks[5]))
    pc.convert([tuple(xs.shape), tuple(xf.shape)], dtype=torch.bfloat16)
    mid = pc([xs.cuda().bfloat16(), xf.cuda().bfloat16()])
    mid = mid[0] if isinstance(mid, list) else mid
    rows.append(("blocks.5 (pool + concat)", _rel(mid, pooled)))
    rows.append(("blocks.6 (head)", _rel(_deploy_single(m.blocks[6], _q(pooled)), OF.res_basic_head(sd, _q(pooled), "blocks.6"))))
    del pc, mid
    return rows, want


def _mvit_blocks(m, sd, x, cfg):
    import torch.nn.functional as F
    from oracle import functional as OF
    from pytorchvideo_amd.accelerator.mi355x import blocks as B
    rows = []
    # the prologue on the host (its kernels -- patch embedding with the position tables -- are in the full-plan test)
    xq = _q(x)
    y = F.conv3d(xq, sd["patch_embed.patch_model.weight"], sd.get("patch_embed.patch_model.bias"),
                 stride=cfg.get("conv_patch_embed_stride", (2, 4, 4)), padding=cfg.get("conv_patch_embed_padding", (1, 3, 3)))
    thw = [y.shape[2], y.shape[3], y.shape[4]]
    y = y.flatten(2).transpose(1, 2)
    c = "cls_positional_encoding"
    y = torch.cat((sd[c + ".cls_token"].expand(y.shape[0], -1, -1), y), dim=1)
    pos = sd[c + ".pos_embed_spatial"].repeat(1, thw[0], 1) + torch.repeat_interleave(sd[c + ".pos_embed_temporal"], thw[1] * thw[2], dim=1)
    y = y + torch.cat([sd[c + ".pos_embed_class"], pos], 1)
    for i, (heads, kq, sq, kkv, skv) in enumerate(OF.mvit_schedule(cfg)):
        want, thw2 = OF.multiscale_block(sd, y, thw, "blocks.%d" % i, heads, kq, sq, kkv, skv, True,
                                         cfg.get("residual_pool", False), cfg.get("dim_mul_in_att", False))
        blk = B.Mi355xMViTBlock(copy.deepcopy(m.blocks[i]))
        blk.convert(tuple(y.shape), dtype=torch.bfloat16, thw=tuple(thw))
        got, got_thw = blk(y.cuda(), list(thw))
        assert list(got_thw) == list(thw2)
        rows.append(("blocks.%d" % i, _rel(got, want)))
        del blk, got
        y, thw = want, thw2
    return rows


def blocks_case(workload, fill="trained_like"):
    """[(block name, max|d| / max|oracle block output|)] for every block of `workload` at its BASELINE geometry."""
    from bench import synth_input
    from parity_full import filled_model
    m, shape = filled_model(workload, fill)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth_input(shape, 1, 99)
    with torch.no_grad():
        if workload in ("x3d_m", "x3d_l"):
            rows = _x3d_blocks(m, sd, x, 16, 224)
        elif workload == "slowfast_r50":
            rows, _ = _slowfast_blocks(m, sd, x)
        else:
            from pytorchvideo_amd.models.hub import mvit_video_base_32x3_config as cfg
            rows = _mvit_blocks(m, sd, x, cfg)
    torch.cuda.empty_cache()
    return rows


def defect_case(workload, block, scale=1.05, fills=("calibrated", "trained_like")):
    """Defect injection: `block` = "blocks.S.res_blocks.I" of an X3D model, teacher-forced; the deploy form is
    built twice -- as is, and with conv_c's filter bank scaled by `scale` (a 5 % arithmetic defect in one kernel's operand) --
    and both are compared with the fp32 oracle's output of the UNTOUCHED block.  {fill: {"clean": e, "defect": e}}."""
    assert workload in ("x3d_m", "x3d_l")
    from bench import synth_input
    from oracle import functional as OF
    from parity_full import filled_model
    S, I = int(block.split(".")[1]), int(block.split(".")[3])
    out = {"workload": workload, "block": block, "scale": scale}
    for fill in fills:
        m, shape = filled_model(workload, fill)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        x = synth_input(shape, 1, 99)
        with torch.no_grad():
            h = OF.x3d_stem(sd, _q(x))
            for s in range(1, S + 1):
                for i in range(len(m.blocks[s].res_blocks)):
                    if (s, i) == (S, I):
                        break
                    h = OF.x3d_res_block(sd, _q(h), "blocks.%d.res_blocks.%d" % (s, i), (1, 2, 2) if i == 0 else (1, 1, 1))
            xin = _q(h)
            want = OF.x3d_res_block(sd, xin, block, (1, 2, 2) if I == 0 else (1, 1, 1))
            rb = m.blocks[S].res_blocks[I]
            clean = _rel(_deploy_single(rb, xin), want)
            bad = copy.deepcopy(rb)
            bad.branch2.conv_c.weight.mul_(scale)
            defect = _rel(_deploy_single(bad, xin), want)
        out[fill] = {"clean": clean, "defect": defect}
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="x3d_m,x3d_l,slowfast_r50,mvit_b_32x3")
    ap.add_argument("--fill", default="trained_like")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    for w in a.workloads.split(","):
        rows = blocks_case(w, a.fill)
        worst = max(rows, key=lambda r: r[1])
        rec = {"workload": w, "fill": a.fill, "blocks": len(rows), "worst": worst[1], "worst_block": worst[0],
               "per_block": {k: float("%.3e" % v) for k, v in rows}}
        print(json.dumps(rec), flush=True)
        if a.json:
            with open(a.json, "a") as f:
                f.write(json.dumps(rec) + "\n")
