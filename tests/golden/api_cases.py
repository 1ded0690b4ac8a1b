"""The public symbols of the path whose call surface must stay verbatim, and the seeded
constructions used to pin the RNG consumption order of the factories.  Shared by make_api_golden.py (REAL reference)
and tests/test_api_surface.py (this package)."""
import hashlib
import importlib
import inspect

import torch

SYMBOLS = [
    ("models.x3d", "create_x3d"), ("models.x3d", "create_x3d_stem"), ("models.x3d", "create_x3d_bottleneck_block"),
    ("models.x3d", "create_x3d_res_block"), ("models.x3d", "create_x3d_res_stage"), ("models.x3d", "create_x3d_head"),
    ("models.x3d", "ProjectedPool"),
    ("models.resnet", "create_resnet"), ("models.resnet", "create_bottleneck_block"), ("models.resnet", "create_res_block"),
    ("models.resnet", "create_res_stage"), ("models.resnet", "create_resnet_with_roi_head"),
    ("models.resnet", "BottleneckBlock"), ("models.resnet", "ResBlock"), ("models.resnet", "ResStage"),
    ("models.slowfast", "create_slowfast"), ("models.slowfast", "create_slowfast_with_roi_head"),
    ("models.slowfast", "FuseFastToSlow"), ("models.slowfast", "PoolConcatPathway"), ("models.slowfast", "FastToSlowFusionBuilder"),
    ("models.csn", "create_csn"), ("models.r2plus1d", "create_r2plus1d"), ("models.r2plus1d", "create_2plus1d_bottleneck_block"),
    ("models.vision_transformers", "create_multiscale_vision_transformers"),
    ("models.stem", "create_res_basic_stem"), ("models.stem", "create_conv_patch_embed"), ("models.stem", "ResNetBasicStem"),
    ("models.stem", "PatchEmbed"),
    ("models.head", "create_res_basic_head"), ("models.head", "create_vit_basic_head"), ("models.head", "create_res_roi_pooling_head"),
    ("models.head", "ResNetBasicHead"), ("models.head", "ResNetRoIHead"), ("models.head", "VisionTransformerBasicHead"),
    ("models.head", "SequencePool"),
    ("models.net", "Net"), ("models.net", "MultiPathWayWithFuse"), ("models.net", "DetectionBBoxNetwork"),
    ("models.weight_init", "init_net_weights"),
    ("layers.convolutions", "create_conv_2plus1d"), ("layers.convolutions", "ConvReduce3D"), ("layers.convolutions", "Conv2plus1d"),
    ("layers.attention", "Mlp"), ("layers.attention", "MultiScaleAttention"), ("layers.attention", "MultiScaleBlock"),
    ("layers.positional_encoding", "SpatioTemporalClsPositionalEncoding"),
    ("layers.swish", "Swish"), ("layers.drop_path", "DropPath"), ("layers.utils", "round_width"), ("layers.utils", "round_repeats"),
]

_MV = dict(spatial_size=32, temporal_size=4, depth=2, patch_embed_dim=16, num_heads=1, head_num_classes=5)
SEEDED = [
    ("x3d_xs", "models.x3d", "create_x3d", dict(input_clip_length=4, input_crop_size=160)),
    ("slow_r50", "models.resnet", "create_resnet", dict(model_num_class=10)),
    ("slow_r50_detection", "models.resnet", "create_resnet_with_roi_head", dict(model_num_class=10)),
    ("slowfast_r18", "models.slowfast", "create_slowfast", dict(model_depth=18, model_num_class=10)),
    ("slowfast_r50_detection", "models.slowfast", "create_slowfast_with_roi_head", dict(model_num_class=10)),
    ("csn_r50", "models.csn", "create_csn", dict(model_num_class=10)),
    ("r2plus1d_r50", "models.r2plus1d", "create_r2plus1d", dict(model_num_class=10)),
    ("mvit_pooled", "models.vision_transformers", "create_multiscale_vision_transformers",
     dict(_MV, pool_q_stride_size=[[1, 1, 2, 2]], pool_kv_stride_adaptive=[1, 2, 2], pool_kvq_kernel=[3, 3, 3],
          embed_dim_mul=[[1, 2.0]], atten_head_mul=[[1, 2.0]])),
    ("mvit_no_cls_joint_pos", "models.vision_transformers", "create_multiscale_vision_transformers",
     dict(_MV, cls_embed_on=False, sep_pos_embed=False, dropout_rate_block=0.1, droppath_rate_block=0.2)),
]


# the hub entry points of the reference's hubconf.py that lie on the path (models/hub/*.py), built with their defaults
HUB = ["x3d_xs", "x3d_s", "x3d_m", "x3d_l", "slow_r50", "c2d_r50", "i3d_r50", "slow_r50_detection", "slowfast_r50",
       "slowfast_r101", "slowfast_16x8_r101_50_50", "slowfast_r50_detection", "csn_r101", "r2plus1d_r50", "mvit_base_16",
       "mvit_base_16x4", "mvit_base_32x3"]


def rounding_table(root):
    """round_width / round_repeats (layers/utils.py:19-49) over the ranges the factories use and beyond."""
    u = importlib.import_module(root + ".layers.utils")
    widths = [[w, m, mn, dv, c, u.round_width(w, m, mn, dv, c)]
              for w in (1, 3, 8, 12, 24, 54, 96, 100, 192, 432, 2048) for m in (0.0, 0.0625, 0.25, 0.9, 1.0, 1.5, 2.0, 2.25, 5.0)
              for mn, dv in ((8, 8), (1, 1), (None, 8), (16, 4)) for c in (False, True)]
    repeats = [[r, m, u.round_repeats(r, m)] for r in (1, 2, 3, 5, 11, 25) for m in (0.0, 0.5, 1.0, 2.2, 5.0)]
    return {"round_width": widths, "round_repeats": repeats}


def _norm(v):
    """A default value without object identity: functions / classes by name (the mirror's callables are its own)."""
    if isinstance(v, (tuple, list)):
        return "(" + ", ".join(_norm(e) for e in v) + ")"
    if inspect.isclass(v):
        return "cls:" + v.__name__
    if callable(v):
        return "fn:" + getattr(v, "__name__", type(v).__name__)
    return repr(v)


def signature_of(obj):
    return [[n, p.kind.name, None if p.default is inspect.Parameter.empty else _norm(p.default)]
            for n, p in inspect.signature(obj).parameters.items()]


def resolve(root, module, name):
    return getattr(importlib.import_module(root + "." + module), name)


def seeded_fingerprint(root, module, name, cfg, seed=123):
    """Construct under torch.manual_seed(seed): sha256 over (key, shape, bytes) of the state_dict, and the next
    random number -- equal only if the factory drew the same numbers in the same order."""
    torch.manual_seed(seed)
    model = resolve(root, module, name)(**cfg)
    nxt = torch.rand(1).item()
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(str(tuple(v.shape)).encode())
        h.update(v.detach().cpu().contiguous().reshape(-1).numpy().tobytes())
    return {"sha256": h.hexdigest(), "next_rand": nxt, "n_tensors": len(model.state_dict())}
