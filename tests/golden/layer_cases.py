"""Layer-level parity cases shared by make_layer_golden.py (runs the REAL reference) and
tests/test_reference_gates.py (runs this package's host mirrors).  The option sweeps follow the reference's
own layer tests (tests/test_layers_convolutions.py, test_layers_attention.py:16-140,
test_layers_positional_encoding.py, test_models_vision_transformers.py:20-120), which are shape-only there;
here the VALUES of the reference are recorded.

A case = (name, module path below the package root, class or factory name, kwargs, input spec).
Input spec: ("tensor", shape) | ("tokens", shape, thw) for modules called as m(x, thw) | ("list", [shapes]) for the
SlowFast containers that take a list of pathway tensors."""
import torch.nn as nn

_ATT = dict(dim=10, num_heads=2)
_TOK = ("tokens", (3, 81, 10), (4, 4, 5))   # cls + 4x4x5 grid (torch's avg_pool3d rejects a window larger than the grid)
_TOK_NOCLS = ("tokens", (3, 80, 10), (4, 4, 5))
_POOL = dict(kernel_q=(3, 3, 3), kernel_kv=(3, 3, 3), stride_q=(2, 2, 1), stride_kv=(1, 1, 5))

LAYER_CASES = [
    ("conv_reduce3d", "layers.convolutions", "ConvReduce3D",
     dict(in_channels=4, out_channels=8, kernel_size=((1, 1, 1), (3, 3, 3), (1, 3, 3)),
          stride=((1, 1, 1), (1, 1, 1), None), padding=((0, 0, 0), (1, 1, 1), (0, 1, 1)),
          dilation=((2, 2, 2), (1, 1, 1), None), groups=(1, 2, None), bias=(True, False, None)),
     ("tensor", (2, 4, 3, 7, 14))),
    ("conv2plus1d_t_then_xy", "layers.convolutions", "create_conv_2plus1d",
     dict(in_channels=4, out_channels=8, inner_channels=6, conv_xy_first=False, kernel_size=(3, 3, 3),
          stride=(2, 2, 2), padding=(1, 1, 1), norm=nn.BatchNorm3d, activation=nn.ReLU),
     ("tensor", (2, 4, 5, 7, 14))),
    ("conv2plus1d_xy_then_t", "layers.convolutions", "create_conv_2plus1d",
     dict(in_channels=4, out_channels=8, conv_xy_first=True, kernel_size=(3, 3, 3), stride=(1, 2, 2),
          padding=(1, 1, 1), bias=True, norm=None, activation=None),
     ("tensor", (1, 4, 3, 14, 7))),
    ("mlp_gelu", "layers.attention", "Mlp", dict(in_features=10, hidden_features=20, out_features=30, act_layer=nn.GELU),
     ("tensor", (8, 10))),
    ("mlp_relu_nobias", "layers.attention", "Mlp",
     dict(in_features=20, hidden_features=10, out_features=20, act_layer=nn.ReLU, bias_on=False), ("tensor", (4, 7, 20))),
    ("attn_plain", "layers.attention", "MultiScaleAttention", dict(_ATT), _TOK),
    ("attn_dim_out", "layers.attention", "MultiScaleAttention", dict(_ATT, dim_out=20, qkv_bias=True), _TOK),
    ("attn_pool_conv", "layers.attention", "MultiScaleAttention", dict(_ATT, **_POOL), _TOK),
    ("attn_pool_conv_dim_out", "layers.attention", "MultiScaleAttention", dict(_ATT, dim_out=20, **_POOL), _TOK),
    ("attn_pool_no_cls", "layers.attention", "MultiScaleAttention", dict(_ATT, has_cls_embed=False, **_POOL), _TOK_NOCLS),
    ("attn_pool_avg", "layers.attention", "MultiScaleAttention", dict(_ATT, pool_mode="avg", **_POOL), _TOK),
    ("attn_pool_max_first", "layers.attention", "MultiScaleAttention",
     dict(_ATT, pool_mode="max", pool_first=True, **_POOL), _TOK),
    ("attn_fused_qkv_no_residual", "layers.attention", "MultiScaleAttention",
     dict(_ATT, separate_qkv=False, residual_pool=False, bias_on=False, **_POOL), _TOK),
    ("attn_dense_pool_conv", "layers.attention", "MultiScaleAttention", dict(_ATT, depthwise_conv=False, **_POOL), _TOK),
    ("block_widen", "layers.attention", "MultiScaleBlock", dict(dim=10, dim_out=20, num_heads=2), _TOK),
    ("block_widen_in_att", "layers.attention", "MultiScaleBlock", dict(dim=10, dim_out=20, num_heads=2, dim_mul_in_att=True), _TOK),
    ("block_pool_q", "layers.attention", "MultiScaleBlock",
     dict(dim=10, dim_out=20, num_heads=2, kernel_q=(3, 3, 3), stride_q=(2, 2, 1)), _TOK),
    ("block_pool_kv_no_cls", "layers.attention", "MultiScaleBlock",
     dict(dim=10, dim_out=10, num_heads=2, kernel_kv=(3, 3, 3), stride_kv=(1, 2, 5), has_cls_embed=False, qkv_bias=True,
          mlp_ratio=2.0), _TOK_NOCLS),
    ("posenc_sep_cls", "layers.positional_encoding", "SpatioTemporalClsPositionalEncoding",
     dict(embed_dim=12, patch_embed_shape=(2, 3, 4), sep_pos_embed=True, has_cls=True), ("tensor", (2, 24, 12))),
    ("posenc_joint_no_cls", "layers.positional_encoding", "SpatioTemporalClsPositionalEncoding",
     dict(embed_dim=12, patch_embed_shape=(2, 3, 4), sep_pos_embed=False, has_cls=False), ("tensor", (2, 24, 12))),
]

_V = ("tensor", (2, 8, 4, 14, 14))
LAYER_CASES += [
    # ---- building blocks of the conv families (rows a3-a11), the options the factories expose
    ("x3d_stem", "models.x3d", "create_x3d_stem", dict(in_channels=3, out_channels=8), ("tensor", (2, 3, 4, 16, 16))),
    ("x3d_bottleneck_se", "models.x3d", "create_x3d_bottleneck_block", dict(dim_in=8, dim_inner=16, dim_out=8), _V),
    ("x3d_bottleneck_no_se_stride1", "models.x3d", "create_x3d_bottleneck_block",
     dict(dim_in=8, dim_inner=24, dim_out=16, conv_stride=(1, 1, 1), se_ratio=0.0, inner_act=nn.ReLU), _V),
    ("x3d_res_block_projection", "models.x3d", "create_x3d_res_block", dict(dim_in=8, dim_inner=16, dim_out=16), _V),
    ("x3d_res_block_identity", "models.x3d", "create_x3d_res_block",
     dict(dim_in=8, dim_inner=16, dim_out=8, conv_stride=(1, 1, 1)), _V),
    ("x3d_res_stage", "models.x3d", "create_x3d_res_stage", dict(depth=3, dim_in=8, dim_inner=16, dim_out=16), _V),
    ("x3d_head", "models.x3d", "create_x3d_head",
     dict(dim_in=8, dim_inner=16, dim_out=32, num_classes=5, pool_kernel_size=(4, 7, 7), activation=None), _V),
    ("x3d_head_softmax_bn_lin5", "models.x3d", "create_x3d_head",
     dict(dim_in=8, dim_inner=16, dim_out=32, num_classes=5, pool_kernel_size=(2, 7, 7), bn_lin5_on=True), _V),
    ("res_stem_no_pool_elu", "models.stem", "create_res_basic_stem",
     dict(in_channels=3, out_channels=8, conv_kernel_size=(1, 5, 5), conv_padding=(0, 2, 2), pool=None, activation=nn.ELU),
     ("tensor", (2, 3, 4, 16, 16))),
    ("res_stem_avg_pool", "models.stem", "create_res_basic_stem", dict(in_channels=3, out_channels=8, pool=nn.AvgPool3d),
     ("tensor", (1, 3, 4, 16, 16))),
    ("bottleneck_dilated_grouped", "models.resnet", "create_bottleneck_block",
     dict(dim_in=8, dim_inner=8, dim_out=16, conv_a_kernel_size=(3, 1, 1), conv_a_padding=(1, 0, 0), conv_b_kernel_size=(1, 3, 3),
          conv_b_stride=(1, 1, 1), conv_b_padding=(0, 2, 2), conv_b_dilation=(1, 2, 2), conv_b_num_groups=4), _V),
    ("res_block_strided_shortcut", "models.resnet", "create_res_block",
     dict(dim_in=8, dim_inner=4, dim_out=16, bottleneck=None, conv_a_stride=(2, 1, 1), conv_b_stride=(1, 2, 2)), _V),
    ("res_stage_per_block_kernels", "models.resnet", "create_res_stage",
     dict(depth=3, dim_in=8, dim_inner=4, dim_out=16, bottleneck=None, conv_a_kernel_size=[(3, 1, 1), (1, 1, 1)],
          conv_a_padding=[(1, 0, 0), (0, 0, 0)]), _V),
    ("res_head_max_pool_no_average", "models.head", "create_res_basic_head",
     dict(in_features=8, out_features=5, pool=nn.MaxPool3d, pool_kernel_size=(2, 7, 7), output_with_global_average=False), _V),
    ("res_head_adaptive_softmax", "models.head", "create_res_basic_head",
     dict(in_features=8, out_features=5, pool=nn.AdaptiveAvgPool3d, output_size=(1, 2, 2), activation=nn.Softmax), _V),
    ("fuse_fast_to_slow", "models.slowfast", "_fuse_case", dict(), ("list", [(2, 16, 2, 7, 7), (2, 4, 8, 7, 7)])),
    ("pool_concat_pathway", "models.slowfast", "_pool_concat_case", dict(), ("list", [(2, 16, 2, 7, 7), (2, 4, 8, 7, 7)])),
]


def _fuse_case(root):
    """FastToSlowFusionBuilder(...).create_module (models/slowfast.py:623-694) -> FuseFastToSlow."""
    import importlib
    sf = importlib.import_module(root + ".models.slowfast")
    return sf.FastToSlowFusionBuilder(slowfast_channel_reduction_ratio=4, conv_fusion_channel_ratio=2, conv_kernel_size=(5, 1, 1),
                                      conv_stride=(4, 1, 1), max_stage_idx=3).create_module(fusion_dim_in=16, stage_idx=1)


def _pool_concat_case(root):
    import importlib
    sf = importlib.import_module(root + ".models.slowfast")
    return sf.PoolConcatPathway(retain_list=True, pool=nn.ModuleList([nn.AvgPool3d((2, 1, 1)), nn.AvgPool3d((8, 1, 1))]), dim=1)


SPECIAL_BUILDERS = {"_fuse_case": _fuse_case, "_pool_concat_case": _pool_concat_case}


def build_case(root, module, attr, kwargs):
    """Instantiate a case from the package `root` ('pytorchvideo' = the reference, 'pytorchvideo_amd' = this repo).
    `bottleneck=None` in kwargs stands for that package's own create_bottleneck_block."""
    import importlib
    if attr in SPECIAL_BUILDERS:
        return SPECIAL_BUILDERS[attr](root)
    mod = importlib.import_module(root + "." + module)
    kwargs = dict(kwargs)
    if "bottleneck" in kwargs and kwargs["bottleneck"] is None:
        kwargs["bottleneck"] = importlib.import_module(root + ".models.resnet").create_bottleneck_block
    return getattr(mod, attr)(**kwargs)


_MV = dict(depth=2, patch_embed_dim=16, num_heads=1, head_num_classes=5, pool_q_stride_size=[[1, 1, 2, 2]],
           pool_kv_stride_adaptive=[1, 2, 2], pool_kvq_kernel=[3, 3, 3], embed_dim_mul=[[1, 2.0]], atten_head_mul=[[1, 2.0]])

# create_multiscale_vision_transformers input forms (SURVEY 8b "Module/API surface to preserve")
MVIT_CASES = [
    ("mvit_video", dict(_MV, spatial_size=32, temporal_size=4), (2, 3, 4, 32, 32)),
    ("mvit_image_2d_patch", dict(_MV, spatial_size=32, temporal_size=1, use_2d_patch=True, conv_patch_embed_kernel=(7, 7),
                                 conv_patch_embed_stride=(4, 4), conv_patch_embed_padding=(3, 3)), (2, 3, 32, 32)),
    ("mvit_pre_embedded_tokens", dict(_MV, spatial_size=8, temporal_size=2, enable_patch_embed=False, input_channels=16,
                                      cls_embed_on=False), (2, 128, 16)),
    ("mvit_rect_mean_pool_head", dict(_MV, spatial_size=(32, 48), temporal_size=4, cls_embed_on=False, sep_pos_embed=False,
                                      head_activation=nn.Softmax), (1, 3, 4, 32, 48)),
]
