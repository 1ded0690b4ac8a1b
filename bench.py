"""Forward-throughput benchmark of the MI355X path (clips/s), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload x3d_m|x3d_l|slowfast_r50|mvit_b_32x3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward of the hot path over one batch of synthetic clips that is already
resident in HBM in the reference's own layout ([B,3,T,H,W], bf16); every rank processes a
fixed per-GPU batch (weak scaling) and the step ends with the single head collective
(all_gather of the [B_local, 400] fp32 logits).  Rank 0 prints ONE JSON line.

N=1 default workload = BASELINE.json configs[1]: X3D-M, bf16, [32,3,16,224,224].
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFS = 2500.0    # dense bf16

# algorithmic FLOPs / bytes per clip (probe of the reference op graph)
WORKLOADS = {
    "x3d_m": dict(batch=32, gflop=9.465, mb=365.0, bound="hbm",
                  desc="create_x3d(input_clip_length=16,input_crop_size=224) [B,3,16,224,224]"),
    "x3d_l": dict(batch=32, gflop=18.325, mb=604.0, bound="hbm",
                  desc="create_x3d(16,224,depth_factor=5.0) [B,3,16,224,224]"),
    "slowfast_r50": dict(batch=16, gflop=131.42, mb=740.0, bound="mfma",
                         desc="create_slowfast(model_depth=50) slow [B,3,8,256,256] + fast [B,3,32,256,256]"),
    "mvit_b_32x3": dict(batch=8, gflop=339.92, mb=1465.0, bound="mfma",
                        desc="create_multiscale_vision_transformers(**mvit_video_base_32x3_config) [B,3,32,224,224]"),
}


def make_model(name):
    """(original-form model, input shape(s), oracle forward) of a workload."""
    from oracle import functional as OF  # only used by the cpu_baseline leg
    if name in ("x3d_m", "x3d_l"):
        from pytorchvideo_amd.models import create_x3d
        kw = dict(input_clip_length=16, input_crop_size=224)
        if name == "x3d_l":
            kw["depth_factor"] = 5.0
        return create_x3d(**kw), (3, 16, 224, 224), lambda sd, x: OF.x3d_forward(sd, x, 16, 224)
    if name == "slowfast_r50":
        from pytorchvideo_amd.models import create_slowfast
        return (create_slowfast(model_depth=50), [(3, 8, 256, 256), (3, 32, 256, 256)],
                lambda sd, x: OF.slowfast_forward(sd, x[0], x[1]))
    if name == "mvit_b_32x3":
        from pytorchvideo_amd.models import create_multiscale_vision_transformers
        from pytorchvideo_amd.models.hub import mvit_video_base_32x3_config as cfg
        return (create_multiscale_vision_transformers(**cfg), (3, 32, 224, 224),
                lambda sd, x: OF.mvit_forward(sd, x, cfg))
    raise SystemExit("unknown workload %s" % name)


def synth_input(shape, batch, seed):
    """Synthetic clips; SlowFast's slow pathway is the temporally subsampled fast pathway
    (PackPathway: uniform_temporal_subsample, transforms/functional.py:134-160)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if isinstance(shape, list):
        fast = torch.randn((batch,) + tuple(shape[1]), generator=g)
        idx = torch.linspace(0, shape[1][1] - 1, shape[0][1]).long()
        return [fast[:, :, idx].contiguous(), fast]
    return torch.randn((batch,) + tuple(shape), generator=g)


def build_model(name, batch, device, dtype):
    from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
    from pytorchvideo_amd.utils import randomize_norm_stats
    torch.manual_seed(0)
    model, shape, _ = make_model(name)
    randomize_norm_stats(model, 0)
    model.eval()
    x = synth_input(shape, batch, 1234 + int(os.environ.get("RANK", "0")))
    x = [t.to(dtype).to(device) for t in x] if isinstance(x, list) else x.to(dtype).to(device)
    transmute_model(model, "mi355x")
    deployed = convert_to_deployable_form(model, x, dtype=dtype)
    return model, deployed, x


def cpu_baseline(name):
    """The oracle (a CPU port of the reference forward) timed on this box's host cores, on a
    bounded sample of the same workload.  Baseline, not target."""
    from oracle.weights import reference_style_fill
    # big hosts (256 hw threads) run torch-CPU slower when oversubscribed: cap the thread count
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    m, shape, oracle_fn = make_model(name)
    reference_style_fill(m, 0).eval()
    sd = m.state_dict()
    b = {"x3d_m": 2, "x3d_l": 2, "slowfast_r50": 1, "mvit_b_32x3": 1}[name]
    x = synth_input(shape, b, 7)
    with torch.no_grad():
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            oracle_fn(sd, x)
            best = min(best, time.perf_counter() - t0)
    return {"value": round(b / best, 3), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "%d clips, fp32, torch-CPU oracle (oracle/functional.py), %d threads, best of 2" % (b, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="x3d_m")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: workload's)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-h2d", action="store_true",
                    help="also time steps that upload the (pinned) host input first; reported as pcie_inclusive, never as value")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    from pytorchvideo_amd.parallel import gather_logits

    wl = WORKLOADS[args.workload]
    batch = args.batch or wl["batch"]
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    _, model, x = build_model(args.workload, batch, device, dtype)
    if args.no_graph:
        model.__dict__["_pv_use_graph"] = False

    def step():
        return gather_logits(model(list(x) if isinstance(x, list) else x), global_batch=batch * world)

    for _ in range(args.warmup):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # one event per step on the launch stream (a few microseconds each): the spread of the step time, SURVEY 8d
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pct = lambda q: round(step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))], 4)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert out.shape == (batch * world, 400) and torch.isfinite(out).all()

    pcie = None
    if args.with_h2d:   # the same steps with the host -> device upload of the batch inside the timed region
        xs = x if isinstance(x, list) else [x]
        hs = [t.cpu().pin_memory() for t in xs]
        for _ in range(2):
            for t, h in zip(xs, hs):
                t.copy_(h, non_blocking=True)
            out = step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            for t, h in zip(xs, hs):
                t.copy_(h, non_blocking=True)
            out = step()
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        pcie = {"value": round(batch * args.steps / e1, 2), "unit": "clips/s (this rank)", "ms_per_step": round(e1 / args.steps * 1e3, 4),
                "upload_bytes_per_step": int(sum(h.numel() * h.element_size() for h in hs))}
        # ... and with the pre-path transforms fused into the ingest (pytorchvideo_amd.transforms.DevicePacker):
        # one uint8 clip at the fast frame rate is uploaded, every pathway is subsampled / scaled / normalised
        # on the device
        from pytorchvideo_amd.transforms import DevicePacker
        ratios = (xs[-1].shape[2] // xs[0].shape[2], 1) if len(xs) == 2 else None
        packer = DevicePacker(model, (0.45,) * 3, (0.225,) * 3, div255=True, frame_ratios=ratios)
        u8 = torch.randint(0, 256, tuple(xs[-1].shape), dtype=torch.uint8).pin_memory()
        d8 = torch.empty_like(u8, device=device)
        for _ in range(2):
            d8.copy_(u8, non_blocking=True)
            packer(d8)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            d8.copy_(u8, non_blocking=True)
            packer(d8)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t2
        pcie["u8_packed"] = {"value": round(batch * args.steps / e2, 2), "ms_per_step": round(e2 / args.steps * 1e3, 4),
                             "upload_bytes_per_step": int(u8.numel())}

    # per-kernel device time (HIP events on the launch stream, inside this process)
    sess = model._pv_session
    prof = sess.profile(iters=3)
    agg = {}
    for label, kind, ms, alg_bytes, flops in prof:
        label = label.split("|")[0]
        a = agg.setdefault(label.split(".")[0] if label.startswith(("conv_b", "conv_ab")) else label, [0, 0.0, 0, 0])
        a[0] += 1
        a[1] += ms
        a[2] += alg_bytes
        a[3] += flops
    total_kernel_ms = sum(v[1] for v in agg.values())
    dom_label, dom = max(agg.items(), key=lambda kv: kv[1][1])
    # the dominant kernel's roofline: HBM when its arithmetic intensity is below the ridge
    # (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B), MFMA otherwise
    dom_gbs = dom[2] / (dom[1] * 1e-3) / 1e9 if dom[1] > 0 else 0.0
    dom_tfs = dom[3] / (dom[1] * 1e-3) / 1e12 if dom[1] > 0 else 0.0
    mfma_bound = dom[2] > 0 and dom[3] / dom[2] > MFMA_PEAK_TFS * 1e12 / (HBM_PEAK_GBS * 1e9)

    # HBM traffic of the dominant kernel family from the committed rocprofv3 PMC passes
    # (tools/gpu_profile.sh + tools/summarize_pmc.py -> profiles/traffic.json), bytes per launch
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic = tj.get(args.workload, {}).get(dom_label, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        clips_s = batch * world * args.steps / elapsed
        line = {
            "metric": "clips/sec forward", "value": round(clips_s, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "step_ms": {"p10": pct(0.1), "p50": pct(0.5), "p90": pct(0.9)},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "per_gpu_batch": batch,
                       "global_batch": batch * world, "parallelism": "dp%d (batch-sharded, logits all_gather)" % world,
                       "weights": "random-init, randomised BN stats", "hip_graph": not args.no_graph},
            "roofline": {
                "bound": "mfma" if mfma_bound else "hbm", "kernel": dom_label, "launches_per_step": dom[0],
                "achieved": round(dom_tfs if mfma_bound else dom_gbs, 1),
                "peak": MFMA_PEAK_TFS if mfma_bound else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": round(dom_tfs / MFMA_PEAK_TFS if mfma_bound else dom_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                "alg_bytes_per_launch": int(dom[2] / max(dom[0], 1)), "avg_launch_ms": round(dom[1] / max(dom[0], 1), 5),
                "kernel_ms_per_step": round(dom[1], 4), "all_kernels_ms_per_step": round(total_kernel_ms, 4),
                "model_hbm_frac": round(clips_s / world * wl["mb"] * 1e6 / (HBM_PEAK_GBS * 1e9), 4),
                "model_mfma_frac": round(clips_s / world * wl["gflop"] * 1e9 / (MFMA_PEAK_TFS * 1e12), 4),
            },
        }
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload)
        if os.environ.get("PV_BENCH_VERBOSE") == "2":
            for label, kind, ms, alg_bytes, flops in prof:
                print("  op %-60s %8.4f ms %8.1f GB/s %8.2f TF/s" % (label, ms, alg_bytes / max(ms, 1e-9) / 1e6,
                                                                   flops / max(ms, 1e-9) / 1e9), file=sys.stderr)
        if os.environ.get("PV_BENCH_VERBOSE"):
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print("  %-16s n=%3d %8.3f ms  %8.1f GB/s  %7.2f TF/s" % (
                    k, v[0], v[1], v[2] / max(v[1], 1e-9) / 1e6, v[3] / max(v[1], 1e-9) / 1e9), file=sys.stderr)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
