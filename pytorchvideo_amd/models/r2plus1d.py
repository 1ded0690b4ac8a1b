"""R(2+1)D (reference: pytorchvideo/models/r2plus1d.py): conv_b is a Conv2plus1d
(3x1x1 -> BN -> ReLU -> 1x3x3, layers/convolutions.py:159-188)."""
from functools import partial

import torch
import torch.nn as nn

from ..layers.convolutions import create_conv_2plus1d
from .head import create_res_basic_head
from .net import Net
from .resnet import _MODEL_STAGE_DEPTH, _half, create_bottleneck_block, create_res_stage
from .stem import create_res_basic_stem


def create_2plus1d_bottleneck_block(*, dim_in, dim_inner, dim_out, conv_a_kernel_size=(1, 1, 1),
                                    conv_a_stride=(1, 1, 1), conv_a_padding=(0, 0, 0), conv_a=nn.Conv3d,
                                    conv_b_kernel_size=(3, 3, 3), conv_b_stride=(2, 2, 2),
                                    conv_b_padding=(1, 1, 1), conv_b_num_groups=1, conv_b_dilation=(1, 1, 1),
                                    conv_b=create_conv_2plus1d, conv_c=nn.Conv3d, norm=nn.BatchNorm3d,
                                    norm_eps=1e-5, norm_momentum=0.1, activation=nn.ReLU):
    """(reference: r2plus1d.py:14-120).  The `conv_b` argument is ignored like in the
    reference: a (2+1)D conv carrying the block's norm/activation is always used."""
    return create_bottleneck_block(
        dim_in=dim_in, dim_inner=dim_inner, dim_out=dim_out, conv_a_kernel_size=conv_a_kernel_size,
        conv_a_stride=conv_a_stride, conv_a_padding=conv_a_padding, conv_a=conv_a,
        conv_b_kernel_size=conv_b_kernel_size, conv_b_stride=conv_b_stride, conv_b_padding=conv_b_padding,
        conv_b_num_groups=conv_b_num_groups, conv_b_dilation=conv_b_dilation,
        conv_b=partial(create_conv_2plus1d, norm=norm, norm_eps=norm_eps, norm_momentum=norm_momentum,
                       activation=activation),
        conv_c=conv_c, norm=norm, norm_eps=norm_eps, norm_momentum=norm_momentum, activation=activation)


def create_r2plus1d(*, input_channel=3, model_depth=50, model_num_class=400, dropout_rate=0.0,
                    norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, activation=nn.ReLU,
                    stem_dim_out=64, stem_conv_kernel_size=(1, 7, 7), stem_conv_stride=(1, 2, 2),
                    stage_conv_a_kernel_size=((1, 1, 1),) * 4, stage_conv_b_kernel_size=((3, 3, 3),) * 4,
                    stage_conv_b_num_groups=(1, 1, 1, 1), stage_conv_b_dilation=((1, 1, 1),) * 4,
                    stage_spatial_stride=(2, 2, 2, 2), stage_temporal_stride=(1, 1, 2, 2),
                    stage_bottleneck=(create_2plus1d_bottleneck_block,) * 4, head_pool=nn.AvgPool3d,
                    head_pool_kernel_size=(4, 7, 7), head_output_size=(1, 1, 1), head_activation=nn.Softmax,
                    head_output_with_global_average=True):
    """(reference: r2plus1d.py:123-313)"""
    torch._C._log_api_usage_once("PYTORCHVIDEO.model.create_r2plus1d")
    assert model_depth in _MODEL_STAGE_DEPTH.keys(), f"{model_depth} is not in {_MODEL_STAGE_DEPTH.keys()}"
    depths = _MODEL_STAGE_DEPTH[model_depth]
    blocks = [create_res_basic_stem(
        in_channels=input_channel, out_channels=stem_dim_out, conv_kernel_size=stem_conv_kernel_size,
        conv_stride=stem_conv_stride, conv_padding=_half(stem_conv_kernel_size), pool=None, norm=norm,
        activation=activation)]
    dim_in, dim_out = stem_dim_out, stem_dim_out * 4
    for i in range(len(depths)):
        blocks.append(create_res_stage(
            depth=depths[i], dim_in=dim_in, dim_inner=dim_out // 4, dim_out=dim_out,
            bottleneck=stage_bottleneck[i], conv_a_kernel_size=stage_conv_a_kernel_size[i],
            conv_a_stride=[1, 1, 1], conv_a_padding=_half(stage_conv_a_kernel_size[i]),
            conv_b_kernel_size=stage_conv_b_kernel_size[i],
            conv_b_stride=(stage_temporal_stride[i], stage_spatial_stride[i], stage_spatial_stride[i]),
            conv_b_padding=_half(stage_conv_b_kernel_size[i]), conv_b_num_groups=stage_conv_b_num_groups[i],
            conv_b_dilation=stage_conv_b_dilation[i], norm=norm, activation=activation))
        dim_in, dim_out = dim_out, dim_out * 2
    blocks.append(create_res_basic_head(
        in_features=dim_in, out_features=model_num_class, pool=head_pool, output_size=head_output_size,
        pool_kernel_size=head_pool_kernel_size, dropout_rate=dropout_rate, activation=head_activation,
        output_with_global_average=head_output_with_global_average))
    return Net(blocks=nn.ModuleList(blocks))
