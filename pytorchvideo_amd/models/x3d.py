"""X3D (reference: pytorchvideo/models/x3d.py).  Same keyword-only factory signatures,
module tree and state_dict keys; compute in this *original form* is plain torch ops (CPU
plumbing / training), the MI355X deploy form is produced by
`pytorchvideo_amd.accelerator.transmute_model(model, "mi355x")` + `convert_to_deployable_form`.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ..layers.convolutions import Conv2plus1d
from ..layers.squeeze_excitation import SqueezeExcitation
from ..layers.swish import Swish
from ..layers.utils import round_repeats, round_width, set_attributes
from .head import ResNetBasicHead
from .net import Net
from .resnet import BottleneckBlock, ResBlock, ResStage
from .stem import ResNetBasicStem, _act, _norm


def create_x3d_stem(*, in_channels, out_channels, conv_kernel_size=(5, 3, 3), conv_stride=(1, 2, 2),
                    conv_padding=(2, 1, 1), norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1,
                    activation=nn.ReLU):
    """Spatial 1xkxk conv then depthwise temporal kx1x1 conv, BN, ReLU (reference:
    x3d.py:19-102).  NB the reference stores the *spatial* conv in the `conv_t` slot and the
    *temporal* one in `conv_xy` (x3d.py:83-88); the state_dict keys depend on that swap."""
    kt, kh, kw = conv_kernel_size
    spatial = nn.Conv3d(in_channels, out_channels, kernel_size=(1, kh, kw),
                        stride=(1, conv_stride[1], conv_stride[2]),
                        padding=(0, conv_padding[1], conv_padding[2]), bias=False)
    temporal = nn.Conv3d(out_channels, out_channels, kernel_size=(kt, 1, 1), stride=(conv_stride[0], 1, 1),
                         padding=(conv_padding[0], 0, 0), bias=False, groups=out_channels)
    return ResNetBasicStem(
        conv=Conv2plus1d(conv_t=spatial, norm=None, activation=None, conv_xy=temporal),
        norm=_norm(norm, out_channels, norm_eps, norm_momentum),
        activation=_act(activation),
        pool=None,
    )


def create_x3d_bottleneck_block(*, dim_in, dim_inner, dim_out, conv_kernel_size=(3, 3, 3),
                                conv_stride=(1, 2, 2), norm=nn.BatchNorm3d, norm_eps=1e-5,
                                norm_momentum=0.1, se_ratio=0.0625, activation=nn.ReLU, inner_act=Swish):
    """1x1x1 -> depthwise 3x3x3 [-> SE] -> Swish -> 1x1x1 (reference: x3d.py:105-228).
    norm_b is `Sequential(BN, SE-or-Identity)`, hence the `norm_b.1.block.*` SE keys."""
    # construction order follows the reference so that a given RNG seed yields the same weights
    conv_a = nn.Conv3d(dim_in, dim_inner, kernel_size=(1, 1, 1), bias=False)
    conv_b = nn.Conv3d(dim_inner, dim_inner, kernel_size=conv_kernel_size, stride=conv_stride,
                       padding=[k // 2 for k in conv_kernel_size], bias=False, groups=dim_inner,
                       dilation=(1, 1, 1))
    se = (SqueezeExcitation(num_channels=dim_inner, num_channels_reduced=round_width(dim_inner, se_ratio),
                            is_3d=True)
          if se_ratio > 0.0 else nn.Identity())
    bn_b = nn.Identity() if norm is None else norm(num_features=dim_inner, eps=norm_eps, momentum=norm_momentum)
    return BottleneckBlock(
        conv_a=conv_a,
        norm_a=_norm(norm, dim_inner, norm_eps, norm_momentum),
        act_a=_act(activation),
        conv_b=conv_b,
        norm_b=nn.Sequential(bn_b, se),
        act_b=_act(inner_act),
        conv_c=nn.Conv3d(dim_inner, dim_out, kernel_size=(1, 1, 1), bias=False),
        norm_c=_norm(norm, dim_out, norm_eps, norm_momentum),
    )


def create_x3d_res_block(*, dim_in, dim_inner, dim_out, bottleneck=create_x3d_bottleneck_block,
                         use_shortcut=True, conv_kernel_size=(3, 3, 3), conv_stride=(1, 2, 2),
                         norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, se_ratio=0.0625,
                         activation=nn.ReLU, inner_act=Swish):
    """(reference: x3d.py:231-324).  The shortcut conv exists when width or stride change;
    its BN only when the width changes (so res2.0 has a bare strided conv)."""
    widens = dim_in != dim_out
    skip_conv = None
    if (widens or np.prod(conv_stride) > 1) and use_shortcut:
        skip_conv = nn.Conv3d(dim_in, dim_out, kernel_size=(1, 1, 1), stride=conv_stride, bias=False)
    skip_norm = norm(num_features=dim_out) if (norm is not None and widens and use_shortcut) else None
    return ResBlock(
        branch1_conv=skip_conv,
        branch1_norm=skip_norm,
        branch2=bottleneck(dim_in=dim_in, dim_inner=dim_inner, dim_out=dim_out,
                           conv_kernel_size=conv_kernel_size, conv_stride=conv_stride, norm=norm,
                           norm_eps=norm_eps, norm_momentum=norm_momentum, se_ratio=se_ratio,
                           activation=activation, inner_act=inner_act),
        activation=_act(activation),
        branch_fusion=lambda x, y: x + y,
    )


def create_x3d_res_stage(*, depth, dim_in, dim_inner, dim_out, bottleneck=create_x3d_bottleneck_block,
                         conv_kernel_size=(3, 3, 3), conv_stride=(1, 2, 2), norm=nn.BatchNorm3d,
                         norm_eps=1e-5, norm_momentum=0.1, se_ratio=0.0625, activation=nn.ReLU,
                         inner_act=Swish):
    """(reference: x3d.py:327-408).  SE sits in every other block, starting with the first."""
    blocks = [
        create_x3d_res_block(
            dim_in=dim_in if i == 0 else dim_out, dim_inner=dim_inner, dim_out=dim_out,
            bottleneck=bottleneck, conv_kernel_size=conv_kernel_size,
            conv_stride=conv_stride if i == 0 else (1, 1, 1), norm=norm, norm_eps=norm_eps,
            norm_momentum=norm_momentum, se_ratio=(se_ratio if (i + 1) % 2 else 0.0),
            activation=activation, inner_act=inner_act)
        for i in range(depth)
    ]
    return ResStage(res_blocks=nn.ModuleList(blocks))


class ProjectedPool(nn.Module):
    """pre_conv -> BN -> act -> pool -> post_conv -> [BN] -> act (reference: x3d.py:742-806)."""

    def __init__(self, *, pre_conv=None, pre_norm=None, pre_act=None, pool=None, post_conv=None,
                 post_norm=None, post_act=None) -> None:
        super().__init__()
        set_attributes(self, locals())
        assert self.pre_conv is not None
        assert self.pool is not None
        assert self.post_conv is not None

    def forward(self, x):
        for name in ("pre_conv", "pre_norm", "pre_act", "pool", "post_conv", "post_norm", "post_act"):
            op = getattr(self, name)
            if op is not None:
                x = op(x)
        return x


def create_x3d_head(*, dim_in, dim_inner, dim_out, num_classes, pool_act=nn.ReLU,
                    pool_kernel_size=(13, 5, 5), norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1,
                    bn_lin5_on=False, dropout_rate=0.5, activation=nn.Softmax,
                    output_with_global_average=True):
    """(reference: x3d.py:411-536)"""
    pool = nn.AdaptiveAvgPool3d((1, 1, 1)) if pool_kernel_size is None else nn.AvgPool3d(pool_kernel_size, stride=1)
    projected_pool = ProjectedPool(
        pre_conv=nn.Conv3d(dim_in, dim_inner, kernel_size=(1, 1, 1), bias=False),
        pre_norm=norm(num_features=dim_inner, eps=norm_eps, momentum=norm_momentum),
        pre_act=_act(pool_act),
        pool=pool,
        post_conv=nn.Conv3d(dim_inner, dim_out, kernel_size=(1, 1, 1), bias=False),
        post_norm=norm(num_features=dim_out, eps=norm_eps, momentum=norm_momentum) if bn_lin5_on else None,
        post_act=_act(pool_act),
    )
    if activation is None:
        act_module = None
    elif activation == nn.Softmax:
        act_module = activation(dim=1)
    elif activation == nn.Sigmoid:
        act_module = activation()
    else:
        raise NotImplementedError("{} is not supported as an activationfunction.".format(activation))
    return ResNetBasicHead(
        proj=nn.Linear(dim_out, num_classes, bias=True),
        activation=act_module,
        pool=projected_pool,
        dropout=nn.Dropout(dropout_rate) if dropout_rate > 0 else None,
        output_pool=nn.AdaptiveAvgPool3d(1) if output_with_global_average else None,
    )


def create_x3d(*, input_channel=3, input_clip_length=13, input_crop_size=160, model_num_class=400,
               dropout_rate=0.5, width_factor=2.0, depth_factor=2.2, norm=nn.BatchNorm3d, norm_eps=1e-5,
               norm_momentum=0.1, activation=nn.ReLU, stem_dim_in=12, stem_conv_kernel_size=(5, 3, 3),
               stem_conv_stride=(1, 2, 2),
               stage_conv_kernel_size=((3, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)),
               stage_spatial_stride=(2, 2, 2, 2), stage_temporal_stride=(1, 1, 1, 1),
               bottleneck=create_x3d_bottleneck_block, bottleneck_factor=2.25, se_ratio=0.0625,
               inner_act=Swish, head_dim_out=2048, head_pool_act=nn.ReLU, head_bn_lin5_on=False,
               head_activation=None, head_output_with_global_average=True):
    """X3D model builder (reference: x3d.py:539-739).  X3D-M = (clip 16, crop 224)."""
    torch._C._log_api_usage_once("PYTORCHVIDEO.model.create_x3d")
    stem_dim_out = round_width(stem_dim_in, width_factor)
    blocks = [create_x3d_stem(
        in_channels=input_channel, out_channels=stem_dim_out, conv_kernel_size=stem_conv_kernel_size,
        conv_stride=stem_conv_stride, conv_padding=[k // 2 for k in stem_conv_kernel_size], norm=norm,
        norm_eps=norm_eps, norm_momentum=norm_momentum, activation=activation)]

    # base widths double per stage (rounded to 8) before the global width factor is applied
    base_widths = [stem_dim_in]
    for _ in range(3):
        base_widths.append(round_width(base_widths[-1], 2.0, divisor=8))
    base_depths = [1, 2, 5, 3]

    dim_in = stem_dim_out
    for i in range(len(base_depths)):
        dim_out = round_width(base_widths[i], width_factor)
        dim_inner = int(bottleneck_factor * dim_out)
        blocks.append(create_x3d_res_stage(
            depth=round_repeats(base_depths[i], depth_factor), dim_in=dim_in, dim_inner=dim_inner,
            dim_out=dim_out, bottleneck=bottleneck, conv_kernel_size=stage_conv_kernel_size[i],
            conv_stride=(stage_temporal_stride[i], stage_spatial_stride[i], stage_spatial_stride[i]),
            norm=norm, norm_eps=norm_eps, norm_momentum=norm_momentum, se_ratio=se_ratio,
            activation=activation, inner_act=inner_act))
        dim_in = dim_out

    spatial_stride = stem_conv_stride[1] * np.prod(stage_spatial_stride)
    temporal_stride = stem_conv_stride[0] * np.prod(stage_temporal_stride)
    assert input_clip_length >= temporal_stride, "Clip length doesn't match temporal stride!"
    assert input_crop_size >= spatial_stride, "Crop size doesn't match spatial stride!"
    side = int(math.ceil(input_crop_size / spatial_stride))
    blocks.append(create_x3d_head(
        dim_in=dim_out, dim_inner=dim_inner, dim_out=head_dim_out, num_classes=model_num_class,
        pool_act=head_pool_act, pool_kernel_size=(input_clip_length // temporal_stride, side, side),
        norm=norm, norm_eps=norm_eps, norm_momentum=norm_momentum, bn_lin5_on=head_bn_lin5_on,
        dropout_rate=dropout_rate, activation=head_activation,
        output_with_global_average=head_output_with_global_average))
    return Net(blocks=nn.ModuleList(blocks))
