"""Channel-Separated Network (reference: pytorchvideo/models/csn.py): a 3-D ResNet whose
conv_b is a depthwise 3x3x3 (`conv_b_num_groups = dim_inner`, csn.py:169)."""
import torch
import torch.nn as nn

from .head import create_res_basic_head
from .net import Net
from .resnet import _MODEL_STAGE_DEPTH, _half, create_bottleneck_block, create_res_stage
from .stem import create_res_basic_stem


def create_csn(*, input_channel=3, model_depth=50, model_num_class=400, dropout_rate=0, norm=nn.BatchNorm3d,
               activation=nn.ReLU, stem_dim_out=64, stem_conv_kernel_size=(3, 7, 7),
               stem_conv_stride=(1, 2, 2), stem_pool=None, stem_pool_kernel_size=(1, 3, 3),
               stem_pool_stride=(1, 2, 2), stage_conv_a_kernel_size=(1, 1, 1),
               stage_conv_b_kernel_size=(3, 3, 3), stage_conv_b_width_per_group=1,
               stage_spatial_stride=(1, 2, 2, 2), stage_temporal_stride=(1, 2, 2, 2),
               bottleneck=create_bottleneck_block, bottleneck_ratio=4, head_pool=nn.AvgPool3d,
               head_pool_kernel_size=(1, 7, 7), head_output_size=(1, 1, 1), head_activation=None,
               head_output_with_global_average=True):
    """(reference: csn.py:12-191)"""
    torch._C._log_api_usage_once("PYTORCHVIDEO.model.create_csn")
    assert model_depth in _MODEL_STAGE_DEPTH.keys(), f"{model_depth} is not in {_MODEL_STAGE_DEPTH.keys()}"
    depths = _MODEL_STAGE_DEPTH[model_depth]
    blocks = [create_res_basic_stem(
        in_channels=input_channel, out_channels=stem_dim_out, conv_kernel_size=stem_conv_kernel_size,
        conv_stride=stem_conv_stride, conv_padding=_half(stem_conv_kernel_size), pool=stem_pool,
        pool_kernel_size=stem_pool_kernel_size, pool_stride=stem_pool_stride,
        pool_padding=_half(stem_pool_kernel_size), norm=norm, activation=activation)]
    dim_in, dim_out = stem_dim_out, stem_dim_out * 4
    for i in range(len(depths)):
        dim_inner = dim_out // bottleneck_ratio
        blocks.append(create_res_stage(
            depth=depths[i], dim_in=dim_in, dim_inner=dim_inner, dim_out=dim_out, bottleneck=bottleneck,
            conv_a_kernel_size=stage_conv_a_kernel_size, conv_a_stride=(1, 1, 1),
            conv_a_padding=_half(stage_conv_a_kernel_size), conv_b_kernel_size=stage_conv_b_kernel_size,
            conv_b_stride=(stage_temporal_stride[i], stage_spatial_stride[i], stage_spatial_stride[i]),
            conv_b_padding=_half(stage_conv_b_kernel_size),
            conv_b_num_groups=dim_inner // stage_conv_b_width_per_group, conv_b_dilation=(1, 1, 1),
            norm=norm, activation=activation))
        dim_in, dim_out = dim_out, dim_out * 2
    blocks.append(create_res_basic_head(
        in_features=dim_in, out_features=model_num_class, pool=head_pool, output_size=head_output_size,
        pool_kernel_size=head_pool_kernel_size, dropout_rate=dropout_rate, activation=head_activation,
        output_with_global_average=head_output_with_global_average))
    return Net(blocks=nn.ModuleList(blocks))
