import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import functional as OF
from oracle.weights import quantize_like_kernels, reference_style_fill, seeded_input
from gpu_util import rel_err
from pytorchvideo_amd import _lib as L
from pytorchvideo_amd.models import hub
from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
src = hub.x3d_s(); reference_style_fill(src, 3); src.eval()
sd = {k: v.clone() for k, v in src.state_dict().items()}
x = seeded_input((2, 3, 13, 160, 160), 3)
want = OF.x3d_forward(*quantize_like_kernels(sd, x), 13, 160)
transmute_model(src, "mi355x")
xd = x.cuda().bfloat16()
from pytorchvideo_amd.accelerator.mi355x import tuning
print("fp32:", rel_err(convert_to_deployable_form(src, x.cuda(), dtype=torch.float32)(x.cuda()), OF.x3d_forward(sd, x, 13, 160)), flush=True)
want_fp32 = OF.x3d_forward(sd, x, 13, 160)
print("quantised oracle vs fp32 oracle: %.3e" % rel_err(want, want_fp32), flush=True)
for stages, ab_, opts in ((0x1c, 0x10, {}), (0, 0, {}), (0, 0, {"fuse_ab": False}), (0, 0, {"fuse_ab": False, "fuse_stem": False, "fuse_shortcut": False})):
    L.tune(block_stages=stages, block_stages_ab=ab_)
    for k_, v_ in opts.items():
        tuning.OPTIONS[k_] = v_
    dm = convert_to_deployable_form(src, xd, dtype=torch.bfloat16)
    labels = [o[3].split("|")[0] for o in dm._pv_session.ops]
    out = dm(xd)
    print("block_stages 0x%02x ab 0x%02x %s: block.fused %d, rel err vs quantised oracle %.3e, vs fp32 oracle %.3e" % (stages, ab_, opts, labels.count("block.fused"), rel_err(out, want), rel_err(out, want_fp32)), flush=True)
