"""3-D ResNet building blocks and builder (reference: pytorchvideo/models/resnet.py).

Every conv model of the path (X3D, SlowFast, CSN, R(2+1)D, plain ResNet) is assembled from
`BottleneckBlock` inside `ResBlock` inside `ResStage`; the attribute names below are the
state_dict keys of the model zoo (`blocks.N.res_blocks.M.branch2.conv_a.weight`, ...).
The acoustic variants (resnet.py:151,1022) are out of scope.
"""
from typing import Callable

import numpy as np
import torch
import torch.nn as nn

from ..layers.utils import set_attributes
from .head import create_res_basic_head, create_res_roi_pooling_head
from .net import DetectionBBoxNetwork, Net
from .stem import _act, _norm, create_res_basic_stem


class BottleneckBlock(nn.Module):
    """conv_a -> norm_a -> act_a -> conv_b -> norm_b -> act_b -> conv_c -> norm_c
    (reference: resnet.py:1288-1365).  norm_c is flagged so that init zeroes its gamma."""

    def __init__(self, *, conv_a=None, norm_a=None, act_a=None, conv_b=None, norm_b=None, act_b=None,
                 conv_c=None, norm_c=None) -> None:
        super().__init__()
        set_attributes(self, locals())
        assert all(op is not None for op in (self.conv_a, self.conv_b, self.conv_c))
        if self.norm_c is not None:
            self.norm_c.block_final_bn = True

    def forward(self, x):        # spelled out op by op so that the block stays TorchScript-able like the reference's
        x = self.conv_a(x)
        if self.norm_a is not None:
            x = self.norm_a(x)
        if self.act_a is not None:
            x = self.act_a(x)
        x = self.conv_b(x)
        if self.norm_b is not None:
            x = self.norm_b(x)
        if self.act_b is not None:
            x = self.act_b(x)
        x = self.conv_c(x)
        if self.norm_c is not None:
            x = self.norm_c(x)
        return x


class ResBlock(nn.Module):
    """act(fusion(shortcut(x), branch2(x))) (reference: resnet.py:1137-1189)."""

    def __init__(self, branch1_conv=None, branch1_norm=None, branch2=None, activation=None,
                 branch_fusion: Callable = None) -> None:
        super().__init__()
        set_attributes(self, locals())
        assert self.branch2 is not None

    def forward(self, x):
        if self.branch1_conv is None:
            shortcut = x
        else:
            shortcut = self.branch1_conv(x)
            if self.branch1_norm is not None:
                shortcut = self.branch1_norm(shortcut)
        x = self.branch_fusion(shortcut, self.branch2(x))
        if self.activation is not None:
            x = self.activation(x)
        return x


class ResStage(nn.Module):
    """A run of ResBlocks (reference: resnet.py:1368-1400)."""

    def __init__(self, res_blocks: nn.ModuleList) -> None:
        super().__init__()
        self.res_blocks = res_blocks

    def forward(self, x):
        for block in self.res_blocks:
            x = block(x)
        return x


def _trivial_sum(x, y):
    return x + y


def create_bottleneck_block(*, dim_in, dim_inner, dim_out, conv_a_kernel_size=(3, 1, 1),
                            conv_a_stride=(2, 1, 1), conv_a_padding=(1, 0, 0), conv_a=nn.Conv3d,
                            conv_b_kernel_size=(1, 3, 3), conv_b_stride=(1, 2, 2), conv_b_padding=(0, 1, 1),
                            conv_b_num_groups=1, conv_b_dilation=(1, 1, 1), conv_b=nn.Conv3d,
                            conv_c=nn.Conv3d, norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1,
                            activation=nn.ReLU):
    """T x1x1 conv, 1x3x3 (or grouped / (2+1)D) conv, 1x1x1 conv, each with BN
    (reference: resnet.py:17-148)."""
    return BottleneckBlock(
        conv_a=conv_a(in_channels=dim_in, out_channels=dim_inner, kernel_size=conv_a_kernel_size,
                      stride=conv_a_stride, padding=conv_a_padding, bias=False),
        norm_a=_norm(norm, dim_inner, norm_eps, norm_momentum),
        act_a=_act(activation),
        conv_b=conv_b(in_channels=dim_inner, out_channels=dim_inner, kernel_size=conv_b_kernel_size,
                      stride=conv_b_stride, padding=conv_b_padding, bias=False,
                      groups=conv_b_num_groups, dilation=conv_b_dilation),
        norm_b=_norm(norm, dim_inner, norm_eps, norm_momentum),
        act_b=_act(activation),
        conv_c=conv_c(in_channels=dim_inner, out_channels=dim_out, kernel_size=(1, 1, 1), bias=False),
        norm_c=_norm(norm, dim_out, norm_eps, norm_momentum),
    )


def create_res_block(*, dim_in, dim_inner, dim_out, bottleneck, use_shortcut=False,
                     branch_fusion=_trivial_sum, conv_a_kernel_size=(3, 1, 1), conv_a_stride=(2, 1, 1),
                     conv_a_padding=(1, 0, 0), conv_a=nn.Conv3d, conv_b_kernel_size=(1, 3, 3),
                     conv_b_stride=(1, 2, 2), conv_b_padding=(0, 1, 1), conv_b_num_groups=1,
                     conv_b_dilation=(1, 1, 1), conv_b=nn.Conv3d, conv_c=nn.Conv3d, conv_skip=nn.Conv3d,
                     norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1,
                     activation_bottleneck=nn.ReLU, activation_block=nn.ReLU):
    """Residual block; a projection shortcut (1x1x1 conv with the combined stride, + BN)
    appears when the shape changes or when forced (reference: resnet.py:326-462)."""
    skip_stride = tuple(a * b for a, b in zip(conv_a_stride, conv_b_stride))
    reshapes = dim_in != dim_out or np.prod(skip_stride) != 1
    skip_norm = None
    if use_shortcut or (norm is not None and reshapes):
        skip_norm = norm(num_features=dim_out, eps=norm_eps, momentum=norm_momentum)
    skip_conv = None
    if reshapes or use_shortcut:
        skip_conv = conv_skip(dim_in, dim_out, kernel_size=(1, 1, 1), stride=skip_stride, bias=False)
    return ResBlock(
        branch1_conv=skip_conv,
        branch1_norm=skip_norm,
        branch2=bottleneck(
            dim_in=dim_in, dim_inner=dim_inner, dim_out=dim_out,
            conv_a_kernel_size=conv_a_kernel_size, conv_a_stride=conv_a_stride,
            conv_a_padding=conv_a_padding, conv_a=conv_a,
            conv_b_kernel_size=conv_b_kernel_size, conv_b_stride=conv_b_stride,
            conv_b_padding=conv_b_padding, conv_b_num_groups=conv_b_num_groups,
            conv_b_dilation=conv_b_dilation, conv_b=conv_b, conv_c=conv_c,
            norm=norm, norm_eps=norm_eps, norm_momentum=norm_momentum,
            activation=activation_bottleneck,
        ),
        activation=_act(activation_block),
        branch_fusion=branch_fusion,
    )


def create_res_stage(*, depth, dim_in, dim_inner, dim_out, bottleneck, conv_a_kernel_size=(3, 1, 1),
                     conv_a_stride=(2, 1, 1), conv_a_padding=(1, 0, 0), conv_a=nn.Conv3d,
                     conv_b_kernel_size=(1, 3, 3), conv_b_stride=(1, 2, 2), conv_b_padding=(0, 1, 1),
                     conv_b_num_groups=1, conv_b_dilation=(1, 1, 1), conv_b=nn.Conv3d, conv_c=nn.Conv3d,
                     norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, activation=nn.ReLU):
    """`depth` residual blocks; only the first one strides / changes width.  conv_a kernel
    and padding may be given per block and are cycled (reference: resnet.py:465-598)."""
    if isinstance(conv_a_kernel_size[0], int):
        conv_a_kernel_size = [conv_a_kernel_size]
    if isinstance(conv_a_padding[0], int):
        conv_a_padding = [conv_a_padding]
    kernels = (list(conv_a_kernel_size) * depth)[:depth]
    paddings = (list(conv_a_padding) * depth)[:depth]
    blocks = []
    for i in range(depth):
        first = i == 0
        blocks.append(create_res_block(
            dim_in=dim_in if first else dim_out, dim_inner=dim_inner, dim_out=dim_out,
            bottleneck=bottleneck,
            conv_a_kernel_size=kernels[i], conv_a_stride=conv_a_stride if first else (1, 1, 1),
            conv_a_padding=paddings[i], conv_a=conv_a,
            conv_b_kernel_size=conv_b_kernel_size, conv_b_stride=conv_b_stride if first else (1, 1, 1),
            conv_b_padding=conv_b_padding, conv_b_num_groups=conv_b_num_groups,
            conv_b_dilation=conv_b_dilation, conv_b=conv_b, conv_c=conv_c,
            norm=norm, norm_eps=norm_eps, norm_momentum=norm_momentum,
            activation_bottleneck=activation, activation_block=activation,
        ))
    return ResStage(res_blocks=nn.ModuleList(blocks))


_MODEL_STAGE_DEPTH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _half(ks):
    return [k // 2 for k in ks]


def _conv_b_padding(kernel, dilation):
    """Spatial padding follows the dilation when dilated, else kernel//2 (resnet.py:778-791)."""
    return (kernel[0] // 2,
            dilation[1] if dilation[1] > 1 else kernel[1] // 2,
            dilation[2] if dilation[2] > 1 else kernel[2] // 2)


def create_resnet(*, input_channel=3, model_depth=50, model_num_class=400, dropout_rate=0.5,
                  norm=nn.BatchNorm3d, activation=nn.ReLU, stem_dim_out=64,
                  stem_conv_kernel_size=(3, 7, 7), stem_conv_stride=(1, 2, 2), stem_pool=nn.MaxPool3d,
                  stem_pool_kernel_size=(1, 3, 3), stem_pool_stride=(1, 2, 2), stem=create_res_basic_stem,
                  stage1_pool=None, stage1_pool_kernel_size=(2, 1, 1),
                  stage_conv_a_kernel_size=((1, 1, 1), (1, 1, 1), (3, 1, 1), (3, 1, 1)),
                  stage_conv_b_kernel_size=((1, 3, 3), (1, 3, 3), (1, 3, 3), (1, 3, 3)),
                  stage_conv_b_num_groups=(1, 1, 1, 1),
                  stage_conv_b_dilation=((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
                  stage_spatial_h_stride=(1, 2, 2, 2), stage_spatial_w_stride=(1, 2, 2, 2),
                  stage_temporal_stride=(1, 1, 1, 1), bottleneck=create_bottleneck_block,
                  head=create_res_basic_head, head_pool=nn.AvgPool3d, head_pool_kernel_size=(4, 7, 7),
                  head_output_size=(1, 1, 1), head_activation=None, head_output_with_global_average=True):
    """Slow / C2D / I3D-style 3-D ResNet (reference: resnet.py:601-841)."""
    torch._C._log_api_usage_once("PYTORCHVIDEO.model.create_resnet")
    assert model_depth in _MODEL_STAGE_DEPTH.keys(), f"{model_depth} is not in {_MODEL_STAGE_DEPTH.keys()}"
    depths = _MODEL_STAGE_DEPTH[model_depth]
    n = len(depths)
    if isinstance(stage_conv_a_kernel_size[0], int):
        stage_conv_a_kernel_size = (stage_conv_a_kernel_size,) * n
    if isinstance(stage_conv_b_kernel_size[0], int):
        stage_conv_b_kernel_size = (stage_conv_b_kernel_size,) * n
    if isinstance(stage_conv_b_dilation[0], int):
        stage_conv_b_dilation = (stage_conv_b_dilation,) * n
    if isinstance(bottleneck, Callable):
        bottleneck = [bottleneck] * n

    blocks = [stem(
        in_channels=input_channel, out_channels=stem_dim_out, conv_kernel_size=stem_conv_kernel_size,
        conv_stride=stem_conv_stride, conv_padding=_half(stem_conv_kernel_size), pool=stem_pool,
        pool_kernel_size=stem_pool_kernel_size, pool_stride=stem_pool_stride,
        pool_padding=_half(stem_pool_kernel_size), norm=norm, activation=activation)]
    dim_in, dim_out = stem_dim_out, stem_dim_out * 4
    for i in range(n):
        ka = stage_conv_a_kernel_size[i]
        pad_a = _half(ka) if isinstance(ka[0], int) else [_half(k) for k in ka]
        blocks.append(create_res_stage(
            depth=depths[i], dim_in=dim_in, dim_inner=dim_out // 4, dim_out=dim_out,
            bottleneck=bottleneck[i], conv_a_kernel_size=ka,
            conv_a_stride=(stage_temporal_stride[i], 1, 1), conv_a_padding=pad_a,
            conv_b_kernel_size=stage_conv_b_kernel_size[i],
            conv_b_stride=(1, stage_spatial_h_stride[i], stage_spatial_w_stride[i]),
            conv_b_padding=_conv_b_padding(stage_conv_b_kernel_size[i], stage_conv_b_dilation[i]),
            conv_b_num_groups=stage_conv_b_num_groups[i], conv_b_dilation=stage_conv_b_dilation[i],
            norm=norm, activation=activation))
        dim_in, dim_out = dim_out, dim_out * 2
        if i == 0 and stage1_pool is not None:
            blocks.append(stage1_pool(kernel_size=stage1_pool_kernel_size,
                                      stride=stage1_pool_kernel_size, padding=(0, 0, 0)))
    if head is not None:
        blocks.append(head(
            in_features=dim_in, out_features=model_num_class, pool=head_pool,
            output_size=head_output_size, pool_kernel_size=head_pool_kernel_size,
            dropout_rate=dropout_rate, activation=head_activation,
            output_with_global_average=head_output_with_global_average))
    return Net(blocks=nn.ModuleList(blocks))


def create_resnet_with_roi_head(*, input_channel=3, model_depth=50, model_num_class=80, dropout_rate=0.5,
                                norm=nn.BatchNorm3d, activation=nn.ReLU, stem_dim_out=64,
                                stem_conv_kernel_size=(1, 7, 7), stem_conv_stride=(1, 2, 2),
                                stem_pool=nn.MaxPool3d, stem_pool_kernel_size=(1, 3, 3),
                                stem_pool_stride=(1, 2, 2), stem=create_res_basic_stem, stage1_pool=None,
                                stage1_pool_kernel_size=(2, 1, 1),
                                stage_conv_a_kernel_size=((1, 1, 1), (1, 1, 1), (3, 1, 1), (3, 1, 1)),
                                stage_conv_b_kernel_size=((1, 3, 3), (1, 3, 3), (1, 3, 3), (1, 3, 3)),
                                stage_conv_b_num_groups=(1, 1, 1, 1),
                                stage_conv_b_dilation=((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 2, 2)),
                                stage_spatial_h_stride=(1, 2, 2, 1), stage_spatial_w_stride=(1, 2, 2, 1),
                                stage_temporal_stride=(1, 1, 1, 1), bottleneck=create_bottleneck_block,
                                head=create_res_roi_pooling_head, head_pool=nn.AvgPool3d,
                                head_pool_kernel_size=(4, 1, 1), head_output_size=(1, 1, 1),
                                head_activation=nn.Sigmoid, head_output_with_global_average=False,
                                head_spatial_resolution=(7, 7), head_spatial_scale=1.0 / 16.0,
                                head_sampling_ratio=0):
    """Slow-style ResNet for detection (reference: resnet.py:844-1019): the last stage keeps the 1/16 map
    (stride 1, conv_b dilated by 2) and the head pools every box with RoIAlign.  forward(x, bboxes)."""
    model = create_resnet(
        input_channel=input_channel, model_depth=model_depth, model_num_class=model_num_class,
        dropout_rate=dropout_rate, norm=norm, activation=activation, stem_dim_out=stem_dim_out,
        stem_conv_kernel_size=stem_conv_kernel_size, stem_conv_stride=stem_conv_stride, stem_pool=stem_pool,
        stem_pool_kernel_size=stem_pool_kernel_size, stem_pool_stride=stem_pool_stride, stem=stem,
        stage1_pool=stage1_pool, stage1_pool_kernel_size=stage1_pool_kernel_size,
        stage_conv_a_kernel_size=stage_conv_a_kernel_size, stage_conv_b_kernel_size=stage_conv_b_kernel_size,
        stage_conv_b_num_groups=stage_conv_b_num_groups, stage_conv_b_dilation=stage_conv_b_dilation,
        stage_spatial_h_stride=stage_spatial_h_stride, stage_spatial_w_stride=stage_spatial_w_stride,
        stage_temporal_stride=stage_temporal_stride, bottleneck=bottleneck, head=None)
    detection_head = head(
        in_features=stem_dim_out * 2 ** (len(_MODEL_STAGE_DEPTH[model_depth]) + 1),
        out_features=model_num_class, pool=head_pool, output_size=head_output_size,
        pool_kernel_size=head_pool_kernel_size, dropout_rate=dropout_rate, activation=head_activation,
        output_with_global_average=head_output_with_global_average, resolution=head_spatial_resolution,
        spatial_scale=head_spatial_scale, sampling_ratio=head_sampling_ratio)
    return DetectionBBoxNetwork(model, detection_head)
