"""SlowFast (reference: pytorchvideo/models/slowfast.py)."""
from typing import Callable, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ..layers.utils import set_attributes
from .head import create_res_basic_head, create_res_roi_pooling_head
from .net import DetectionBBoxNetwork, MultiPathWayWithFuse, Net
from .resnet import _conv_b_padding, _half, create_bottleneck_block, create_res_stage
from .stem import create_res_basic_stem

_MODEL_STAGE_DEPTH = {18: (1, 1, 1, 1), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


class FuseFastToSlow(nn.Module):
    """Lateral connection: time-strided conv on the fast pathway (+BN+ReLU), concatenated
    onto the slow pathway's channels (reference: slowfast.py:697-729)."""

    def __init__(self, conv_fast_to_slow, norm=None, activation=None) -> None:
        super().__init__()
        set_attributes(self, locals())

    def forward(self, x):
        slow, fast = x[0], x[1]
        fuse = self.conv_fast_to_slow(fast)
        if self.norm is not None:
            fuse = self.norm(fuse)
        if self.activation is not None:
            fuse = self.activation(fuse)
        return [torch.cat([slow, fuse], 1), fast]


class FastToSlowFusionBuilder:
    """Factory for the lateral connections (reference: slowfast.py:623-694); stages beyond
    `max_stage_idx` get nn.Identity."""

    def __init__(self, slowfast_channel_reduction_ratio, conv_fusion_channel_ratio, conv_kernel_size,
                 conv_stride, norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, activation=nn.ReLU,
                 max_stage_idx=3) -> None:
        set_attributes(self, locals())

    def create_module(self, fusion_dim_in: int, stage_idx: int) -> nn.Module:
        if stage_idx > self.max_stage_idx:
            return nn.Identity()
        c_fast = fusion_dim_in // self.slowfast_channel_reduction_ratio
        conv = nn.Conv3d(c_fast, int(c_fast * self.conv_fusion_channel_ratio),
                         kernel_size=self.conv_kernel_size, stride=self.conv_stride,
                         padding=[k // 2 for k in self.conv_kernel_size], bias=False)
        norm = None if self.norm is None else self.norm(
            num_features=c_fast * self.conv_fusion_channel_ratio, eps=self.norm_eps, momentum=self.norm_momentum)
        return FuseFastToSlow(conv_fast_to_slow=conv, norm=norm,
                              activation=None if self.activation is None else self.activation())


class PoolConcatPathway(nn.Module):
    """Pool every pathway and concatenate on channels (reference: slowfast.py:586-620)."""

    def __init__(self, retain_list: bool = False, pool: Optional[nn.ModuleList] = None, dim: int = 1) -> None:
        super().__init__()
        set_attributes(self, locals())

    def forward(self, x: List[torch.Tensor]):
        if self.pool is not None:
            assert len(x) == len(self.pool)
        kept = []
        for i in range(len(x)):
            if x[i] is not None:
                if self.pool is not None and self.pool[i] is not None:
                    x[i] = self.pool[i](x[i])
                kept.append(x[i])
        out = torch.cat(kept, 1)
        return [out] if self.retain_list else out


def create_slowfast(*, slowfast_channel_reduction_ratio: Union[Tuple[int], int] = (8,),
                    slowfast_conv_channel_fusion_ratio: int = 2,
                    slowfast_fusion_conv_kernel_size: Tuple[int] = (7, 1, 1),
                    slowfast_fusion_conv_stride: Tuple[int] = (4, 1, 1),
                    fusion_builder: Callable[[int, int], nn.Module] = None,
                    input_channels: Tuple[int] = (3, 3), model_depth: int = 50, model_num_class: int = 400,
                    dropout_rate: float = 0.5, norm: Callable = nn.BatchNorm3d, activation: Callable = nn.ReLU,
                    stem_function: Tuple[Callable] = (create_res_basic_stem, create_res_basic_stem),
                    stem_dim_outs: Tuple[int] = (64, 8),
                    stem_conv_kernel_sizes: Tuple[Tuple[int]] = ((1, 7, 7), (5, 7, 7)),
                    stem_conv_strides: Tuple[Tuple[int]] = ((1, 2, 2), (1, 2, 2)),
                    stem_pool: Union[Callable, Tuple[Callable]] = (nn.MaxPool3d, nn.MaxPool3d),
                    stem_pool_kernel_sizes: Tuple[Tuple[int]] = ((1, 3, 3), (1, 3, 3)),
                    stem_pool_strides: Tuple[Tuple[int]] = ((1, 2, 2), (1, 2, 2)),
                    stage_conv_a_kernel_sizes=(((1, 1, 1), (1, 1, 1), (3, 1, 1), (3, 1, 1)),
                                               ((3, 1, 1), (3, 1, 1), (3, 1, 1), (3, 1, 1))),
                    stage_conv_b_kernel_sizes=(((1, 3, 3), (1, 3, 3), (1, 3, 3), (1, 3, 3)),
                                               ((1, 3, 3), (1, 3, 3), (1, 3, 3), (1, 3, 3))),
                    stage_conv_b_num_groups=((1, 1, 1, 1), (1, 1, 1, 1)),
                    stage_conv_b_dilations=(((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
                                            ((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1))),
                    stage_spatial_strides=((1, 2, 2, 2), (1, 2, 2, 2)),
                    stage_temporal_strides=((1, 1, 1, 1), (1, 1, 1, 1)),
                    bottleneck: Union[Callable, Tuple[Tuple[Callable]]] = (
                        (create_bottleneck_block,) * 4, (create_bottleneck_block,) * 4),
                    head: Callable = create_res_basic_head, head_pool: Callable = nn.AvgPool3d,
                    head_pool_kernel_sizes: Tuple[Tuple[int]] = ((8, 7, 7), (32, 7, 7)),
                    head_output_size: Tuple[int] = (1, 1, 1), head_activation: Callable = None,
                    head_output_with_global_average: bool = True) -> nn.Module:
    """Two-pathway SlowFast builder (reference: slowfast.py:22-361).  Input: [slow, fast]."""
    torch._C._log_api_usage_once("PYTORCHVIDEO.model.create_slowfast")
    n_path = len(input_channels)
    assert model_depth in _MODEL_STAGE_DEPTH.keys(), f"{model_depth} is not in {_MODEL_STAGE_DEPTH.keys()}"
    depths = _MODEL_STAGE_DEPTH[model_depth]
    if isinstance(slowfast_channel_reduction_ratio, int):
        slowfast_channel_reduction_ratio = (slowfast_channel_reduction_ratio,)
    if isinstance(stem_pool, Callable):
        stem_pool = (stem_pool,) * n_path
    if isinstance(bottleneck, Callable):
        bottleneck = ((bottleneck,) * len(depths),) * n_path
    if fusion_builder is None:
        fusion_builder = FastToSlowFusionBuilder(
            slowfast_channel_reduction_ratio=slowfast_channel_reduction_ratio[0],
            conv_fusion_channel_ratio=slowfast_conv_channel_fusion_ratio,
            conv_kernel_size=slowfast_fusion_conv_kernel_size, conv_stride=slowfast_fusion_conv_stride,
            norm=norm, activation=activation, max_stage_idx=len(depths) - 1).create_module

    stems = [stem_function[p](
        in_channels=input_channels[p], out_channels=stem_dim_outs[p],
        conv_kernel_size=stem_conv_kernel_sizes[p], conv_stride=stem_conv_strides[p],
        conv_padding=_half(stem_conv_kernel_sizes[p]), pool=stem_pool[p],
        pool_kernel_size=stem_pool_kernel_sizes[p], pool_stride=stem_pool_strides[p],
        pool_padding=_half(stem_pool_kernel_sizes[p]), norm=norm, activation=activation)
        for p in range(n_path)]
    stages = [MultiPathWayWithFuse(multipathway_blocks=nn.ModuleList(stems),
                                   multipathway_fusion=fusion_builder(fusion_dim_in=stem_dim_outs[0], stage_idx=0))]

    dim_in, dim_out = stem_dim_outs[0], stem_dim_outs[0] * 4
    for i in range(len(depths)):
        # slow pathway also receives the fused fast channels
        fused = dim_in * slowfast_conv_channel_fusion_ratio // slowfast_channel_reduction_ratio[0]
        dims_in, dims_inner, dims_out = [dim_in + fused], [dim_out // 4], [dim_out]
        for r in slowfast_channel_reduction_ratio:
            dims_in.append(dim_in // r)
            dims_inner.append(dim_out // 4 // r)
            dims_out.append(dim_out // r)
        pathways = []
        for p in range(n_path):
            ka = stage_conv_a_kernel_sizes[p][i]
            pad_a = _half(ka) if isinstance(ka[0], int) else [_half(k) for k in ka]
            s = stage_spatial_strides[p][i]
            pathways.append(create_res_stage(
                depth=depths[i], dim_in=dims_in[p], dim_inner=dims_inner[p], dim_out=dims_out[p],
                bottleneck=bottleneck[p][i], conv_a_kernel_size=ka,
                conv_a_stride=(stage_temporal_strides[p][i], 1, 1), conv_a_padding=pad_a,
                conv_b_kernel_size=stage_conv_b_kernel_sizes[p][i], conv_b_stride=(1, s, s),
                conv_b_padding=_conv_b_padding(stage_conv_b_kernel_sizes[p][i], stage_conv_b_dilations[p][i]),
                conv_b_num_groups=stage_conv_b_num_groups[p][i],
                conv_b_dilation=stage_conv_b_dilations[p][i], norm=norm, activation=activation))
        stages.append(MultiPathWayWithFuse(
            multipathway_blocks=nn.ModuleList(pathways),
            multipathway_fusion=fusion_builder(fusion_dim_in=dim_out, stage_idx=i + 1)))
        dim_in, dim_out = dim_out, dim_out * 2

    if head_pool is None:
        pools = None
    elif head_pool == nn.AdaptiveAvgPool3d:
        pools = [head_pool(head_output_size[p]) for p in range(n_path)]
    elif head_pool == nn.AvgPool3d:
        pools = [head_pool(kernel_size=head_pool_kernel_sizes[p], stride=(1, 1, 1), padding=(0, 0, 0))
                 for p in range(n_path)]
    else:
        raise NotImplementedError(f"Unsupported pool_model type {head_pool}")
    stages.append(PoolConcatPathway(retain_list=False, pool=nn.ModuleList(pools)))
    head_in = dim_in + sum(dim_in // r for r in slowfast_channel_reduction_ratio)
    if head is not None:
        stages.append(head(in_features=head_in, out_features=model_num_class, pool=None,
                           output_size=head_output_size, dropout_rate=dropout_rate,
                           activation=head_activation,
                           output_with_global_average=head_output_with_global_average))
    return Net(blocks=nn.ModuleList(stages))


def create_slowfast_with_roi_head(*, slowfast_channel_reduction_ratio=(8,), slowfast_conv_channel_fusion_ratio=2,
                                  slowfast_fusion_conv_kernel_size=(7, 1, 1),
                                  slowfast_fusion_conv_stride=(4, 1, 1), fusion_builder=None,
                                  input_channels=(3, 3), model_depth=50, model_num_class=80, dropout_rate=0.5,
                                  norm=nn.BatchNorm3d, activation=nn.ReLU,
                                  stem_function=(create_res_basic_stem, create_res_basic_stem),
                                  stem_dim_outs=(64, 8), stem_conv_kernel_sizes=((1, 7, 7), (5, 7, 7)),
                                  stem_conv_strides=((1, 2, 2), (1, 2, 2)),
                                  stem_pool=(nn.MaxPool3d, nn.MaxPool3d),
                                  stem_pool_kernel_sizes=((1, 3, 3), (1, 3, 3)),
                                  stem_pool_strides=((1, 2, 2), (1, 2, 2)),
                                  stage_conv_a_kernel_sizes=(((1, 1, 1), (1, 1, 1), (3, 1, 1), (3, 1, 1)),
                                                             ((3, 1, 1), (3, 1, 1), (3, 1, 1), (3, 1, 1))),
                                  stage_conv_b_kernel_sizes=(((1, 3, 3), (1, 3, 3), (1, 3, 3), (1, 3, 3)),
                                                             ((1, 3, 3), (1, 3, 3), (1, 3, 3), (1, 3, 3))),
                                  stage_conv_b_num_groups=((1, 1, 1, 1), (1, 1, 1, 1)),
                                  stage_conv_b_dilations=(((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 2, 2)),
                                                          ((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 2, 2))),
                                  stage_spatial_strides=((1, 2, 2, 1), (1, 2, 2, 1)),
                                  stage_temporal_strides=((1, 1, 1, 1), (1, 1, 1, 1)),
                                  bottleneck=((create_bottleneck_block,) * 4, (create_bottleneck_block,) * 4),
                                  head=create_res_roi_pooling_head, head_pool=nn.AvgPool3d,
                                  head_pool_kernel_sizes=((8, 1, 1), (32, 1, 1)), head_output_size=(1, 1, 1),
                                  head_activation=nn.Sigmoid, head_output_with_global_average=False,
                                  head_spatial_resolution=(7, 7), head_spatial_scale=1.0 / 16.0,
                                  head_sampling_ratio=0) -> nn.Module:
    """SlowFast for detection (reference: slowfast.py:364-582): the backbone ends in PoolConcatPathway
    (temporal pooling only), the RoI head works on the 1/16 map.  forward([slow, fast], bboxes).
    Like the reference (slowfast.py:565), the stages are always built from create_bottleneck_block."""
    model = create_slowfast(
        slowfast_channel_reduction_ratio=slowfast_channel_reduction_ratio,
        slowfast_conv_channel_fusion_ratio=slowfast_conv_channel_fusion_ratio,
        slowfast_fusion_conv_kernel_size=slowfast_fusion_conv_kernel_size,
        slowfast_fusion_conv_stride=slowfast_fusion_conv_stride, fusion_builder=fusion_builder,
        input_channels=input_channels, model_depth=model_depth, model_num_class=model_num_class,
        dropout_rate=dropout_rate, norm=norm, activation=activation, stem_function=stem_function,
        stem_dim_outs=stem_dim_outs, stem_conv_kernel_sizes=stem_conv_kernel_sizes,
        stem_conv_strides=stem_conv_strides, stem_pool=stem_pool,
        stem_pool_kernel_sizes=stem_pool_kernel_sizes, stem_pool_strides=stem_pool_strides,
        stage_conv_a_kernel_sizes=stage_conv_a_kernel_sizes, stage_conv_b_kernel_sizes=stage_conv_b_kernel_sizes,
        stage_conv_b_num_groups=stage_conv_b_num_groups, stage_conv_b_dilations=stage_conv_b_dilations,
        stage_spatial_strides=stage_spatial_strides, stage_temporal_strides=stage_temporal_strides,
        bottleneck=create_bottleneck_block, head=None, head_pool=head_pool,
        head_pool_kernel_sizes=head_pool_kernel_sizes)
    stage_dim_out = stem_dim_outs[0] * 2 ** (len(_MODEL_STAGE_DEPTH[model_depth]) + 1)
    slow_fast_beta = stem_dim_outs[0] // stem_dim_outs[1]
    detection_head = create_res_roi_pooling_head(
        in_features=stage_dim_out + stage_dim_out // slow_fast_beta, out_features=model_num_class, pool=None,
        output_size=head_output_size, dropout_rate=dropout_rate, activation=head_activation,
        output_with_global_average=head_output_with_global_average, resolution=head_spatial_resolution,
        spatial_scale=head_spatial_scale, sampling_ratio=head_sampling_ratio)
    return DetectionBBoxNetwork(model, detection_head)
