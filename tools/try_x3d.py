"""Scratch GPU check: X3D deploy form vs the original form (CPU fp32), per block and whole net."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.weights import deterministic_fill, seeded_input
from pytorchvideo_amd.models import create_x3d
from pytorchvideo_amd.accelerator import transmute_model, convert_to_deployable_form


def run(dtype, B=2, T=4, S=160):
    m = create_x3d(input_clip_length=T, input_crop_size=S)
    deterministic_fill(m, 0)
    m.eval()
    x = seeded_input((B, 3, T, S, S), 0)
    refs = [x]
    with torch.no_grad():
        for b in m.blocks:
            refs.append(b(refs[-1]))
    ref = refs[-1]
    transmute_model(m, "mi355x")
    xd = x.cuda().to(dtype)
    dm = convert_to_deployable_form(m, xd, dtype=dtype, use_graph=False)
    out = dm(xd).float().cpu()
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(dtype, "NET max abs err %.3e  ref absmax %.3e  rel %.3e" % (err, scale, err / scale), flush=True)
    for i, blk in enumerate(dm.blocks):
        o = blk(refs[i].cuda().to(dtype)).float().cpu()
        e = (o - refs[i + 1]).abs().max().item()
        s = refs[i + 1].abs().max().item()
        print("  block %d %-28s err %.3e absmax %.3e rel %.3e" % (i, blk._get_name(), e, s, e / s), flush=True)
    return dm


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run(torch.float32)
    run(torch.bfloat16)
    # quick timing of X3D-M bf16 batch 8
    m = create_x3d(input_clip_length=16, input_crop_size=224)
    deterministic_fill(m, 0)
    transmute_model(m, "mi355x")
    x = torch.randn(8, 3, 16, 224, 224, device="cuda", dtype=torch.bfloat16)
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, use_graph=False)
    for _ in range(3):
        dm(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        dm(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    print("X3D-M B=8 bf16: %.3f ms/iter, %.1f clips/s" % (dt * 1e3, 8 / dt))
    prof = dm._pv_session.profile(3)
    agg = {}
    for label, kind, ms, ab, fl in prof:
        a = agg.setdefault(label, [0, 0.0, 0, 0])
        a[0] += 1; a[1] += ms; a[2] += ab; a[3] += fl
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-16s n=%3d %8.3f ms  %7.1f GB/s  %7.2f TF/s" % (k, v[0], v[1], v[2] / max(v[1], 1e-9) / 1e6, v[3] / max(v[1], 1e-9) / 1e9))
