"""Run in a fresh interpreter by tests/test_reference_plugin.py (only where the reference tree exists): the MI355X
target registered in the REFERENCE's registry, driven by the REFERENCE's own transmute_model / convert driver on the
REFERENCE's own model classes.  Prints one 'ok <what>' line per check."""
import sys

import torch

from oracle import ref_shim

ref_shim.install()                       # stand-ins for fvcore / torchvision, reference tree on sys.path

import pytorchvideo_amd.accelerator as A                                                     # noqa: E402
from pytorchvideo.accelerator.deployment.common.model_transmuter import (                   # noqa: E402
    EFFICIENT_BLOCK_TRANSMUTER_REGISTRY, transmute_model)
from pytorchvideo.accelerator.deployment.mobile_cpu.utils.model_conversion import convert_to_deployable_form  # noqa: E402
from pytorchvideo.accelerator.efficient_blocks.efficient_block_base import EfficientBlockBase  # noqa: E402
from pytorchvideo_amd.accelerator.mi355x import conversion as CV                              # noqa: E402
from pytorchvideo_amd.accelerator.mi355x.session import Session                               # noqa: E402

assert A.efficient_blocks.INSIDE_PYTORCHVIDEO and A.EfficientBlockBase is EfficientBlockBase
assert A.EFFICIENT_BLOCK_TRANSMUTER_REGISTRY is EFFICIENT_BLOCK_TRANSMUTER_REGISTRY and "mi355x" in EFFICIENT_BLOCK_TRANSMUTER_REGISTRY
print("ok registered in the reference's registry")


def labels(sess):
    return [o[3] for o in sess.ops]


def chain(model, x):
    sess = Session(dtype=torch.bfloat16)
    assert CV._chain_net_blocks(model, CV._batch_of(x), sess, torch.bfloat16, x) is not None
    return sess


def check_net(name, ref_factory, mirror_factory, cfg, x, boxes=None):
    torch.manual_seed(0)
    ref = ref_factory(**cfg).eval()
    torch.manual_seed(0)
    mir = mirror_factory(**cfg).eval()
    call = (lambda m: m(list(x) if isinstance(x, list) else x, boxes)) if boxes is not None else \
           (lambda m: m(list(x) if isinstance(x, list) else x))
    with torch.no_grad():
        want = call(ref)
    keys = list(ref.state_dict().keys())
    transmute_model(ref, target_device="mi355x")             # the reference's walker, the reference's modules
    A.transmute_model(mir, "mi355x")
    backbone = ref.model if boxes is not None else ref
    assert all(isinstance(b, EfficientBlockBase) for b in backbone.blocks), [type(b).__name__ for b in backbone.blocks]
    assert list(ref.state_dict().keys()) == keys
    with torch.no_grad():
        assert torch.equal(call(ref), want)                  # original form of the adopted reference modules
    # the launch plan emitted from the reference's module tree is the one emitted from the host mirror
    xb = [t.bfloat16() for t in x] if isinstance(x, list) else x.bfloat16()
    a, b = chain(backbone, xb), chain(mir.model if boxes is not None else mir, xb)
    assert labels(a) == labels(b) and len(a.ops) > 10
    if boxes is not None:
        assert type(ref.detection_head).__name__ == "Mi355xRoIHeadBlock"
        ref.detection_head.convert(None, session=a, input_ref=backbone._pv_output, num_boxes=boxes.shape[0])
        assert [l.split("|")[0] for l in labels(a)[-2:]] == ["det.roi_align", "det.proj"]
    print("ok", name, len(a.ops), "launches")


import types                                                                                  # noqa: E402

import pytorchvideo_amd.models as MM                                                          # noqa: E402
from pytorchvideo.models.csn import create_csn                                                # noqa: E402
from pytorchvideo.models.r2plus1d import create_r2plus1d                                      # noqa: E402
from pytorchvideo.models.resnet import create_resnet, create_resnet_with_roi_head as ref_det  # noqa: E402
from pytorchvideo.models.slowfast import create_slowfast, create_slowfast_with_roi_head as ref_sf_det  # noqa: E402
from pytorchvideo.models.vision_transformers import create_multiscale_vision_transformers     # noqa: E402
from pytorchvideo.models.x3d import create_x3d                                                # noqa: E402

RM = types.SimpleNamespace(create_x3d=create_x3d, create_csn=create_csn, create_r2plus1d=create_r2plus1d,
                           create_resnet=create_resnet, create_slowfast=create_slowfast,
                           create_multiscale_vision_transformers=create_multiscale_vision_transformers)

x = torch.randn(2, 3, 4, 64, 64)
check_net("x3d", RM.create_x3d, MM.create_x3d, dict(input_clip_length=4, input_crop_size=64, model_num_class=10), x)
check_net("csn", RM.create_csn, MM.create_csn, dict(model_num_class=10, head_pool_kernel_size=(1, 2, 2)), x)
check_net("r2plus1d", RM.create_r2plus1d, MM.create_r2plus1d, dict(model_num_class=10, head_pool_kernel_size=(1, 2, 2)), x)
check_net("resnet", RM.create_resnet, MM.create_resnet, dict(model_num_class=10, head_pool_kernel_size=(4, 2, 2)), x)
fast = torch.randn(1, 3, 8, 64, 64)
sf = [fast[:, :, ::4].contiguous(), fast]
check_net("slowfast", RM.create_slowfast, MM.create_slowfast,
          dict(model_depth=18, model_num_class=10, head_pool_kernel_sizes=((2, 2, 2), (8, 2, 2))), sf)
boxes = torch.tensor([[0, 4.0, 6.0, 40.0, 50.0], [1, 0.0, 0.0, 63.0, 63.0], [1, 20.5, 10.25, 30.0, 61.0]])
check_net("resnet detection", ref_det, MM.create_resnet_with_roi_head, dict(model_num_class=10), x, boxes)
check_net("slowfast detection", ref_sf_det, MM.create_slowfast_with_roi_head,
          dict(model_num_class=10, head_pool_kernel_sizes=((2, 1, 1), (8, 1, 1))), sf, boxes[:1])

# MViT: the reference's MultiScaleBlocks become MI355X blocks; whole-model plan from the reference's tree
cfg = dict(spatial_size=32, temporal_size=4, depth=2, patch_embed_dim=32, num_heads=1, head_num_classes=5,
           pool_q_stride_size=[[1, 1, 2, 2]], pool_kv_stride_adaptive=[1, 2, 2], pool_kvq_kernel=[3, 3, 3],
           embed_dim_mul=[[1, 2.0]], atten_head_mul=[[1, 2.0]])
torch.manual_seed(0)
ref = RM.create_multiscale_vision_transformers(**cfg).eval()
torch.manual_seed(0)
mir = MM.create_multiscale_vision_transformers(**cfg).eval()
xm = torch.randn(2, 3, 4, 32, 32)
with torch.no_grad():
    want = ref(xm)
transmute_model(ref, target_device="mi355x")
A.transmute_model(mir, "mi355x")
assert all(type(b).__name__ == "Mi355xMViTBlock" and isinstance(b, EfficientBlockBase) for b in ref.blocks)
with torch.no_grad():
    assert torch.equal(ref(xm), want)
plans = []
for m in (ref, mir):
    s = Session(dtype=torch.bfloat16)
    assert CV._is_fusable_mvit(m, xm.bfloat16()) and CV._try_fuse_mvit(m, s, torch.bfloat16, xm.bfloat16())
    plans.append(labels(s))
assert plans[0] == plans[1]
print("ok mvit", len(plans[0]), "launches")

# The reference's convert driver reaches our convert(): without a GPU that must fail loudly, never fall back to torch.
torch.manual_seed(0)
m = RM.create_x3d(input_clip_length=4, input_crop_size=64, model_num_class=10).eval()
transmute_model(m, target_device="mi355x")
if not torch.cuda.is_available():
    try:
        convert_to_deployable_form(m, x)
    except RuntimeError as e:
        assert "no CPU fallback" in str(e), str(e)
        print("ok the reference's convert driver calls Mi355xBlock.convert (loud failure without a GPU)")
    else:
        sys.exit("convert_to_deployable_form succeeded without a GPU")
else:
    dm = convert_to_deployable_form(m, x)
    with torch.no_grad():
        got = dm(x.cuda()).float().cpu()
    assert (got - m(x)).abs().max().item() <= 2e-2 * m(x).abs().max().item()
    print("ok the reference's convert driver produced a working deploy form")
print("ALL OK")
