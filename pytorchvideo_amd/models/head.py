"""Heads (reference: pytorchvideo/models/head.py)."""
import torch
import torch.nn as nn

from ..layers.roi_align import RoIAlign
from ..layers.utils import set_attributes


def _head_activation(activation):
    if activation is None:
        return None
    return activation(dim=1) if activation == nn.Softmax else activation()


class SequencePool(nn.Module):
    """Token pooling: the cls token or the mean (reference: head.py:11-36)."""

    def __init__(self, mode: str) -> None:
        super().__init__()
        assert mode in ["cls", "mean"], "Unsupported mode for SequencePool."
        self.mode = mode

    def forward(self, x):
        if self.mode == "cls":
            return x[:, 0]
        if self.mode == "mean":
            return x.mean(1)
        raise NotImplementedError


class ResNetBasicHead(nn.Module):
    """pool -> dropout -> per-position Linear -> activation -> global mean
    (reference: head.py:330-391)."""

    def __init__(self, pool=None, dropout=None, proj=None, activation=None, output_pool=None) -> None:
        super().__init__()
        set_attributes(self, locals())
        assert self.proj is not None

    def forward(self, x):
        if self.pool is not None:
            x = self.pool(x)
        if self.dropout is not None:
            x = self.dropout(x)
        # the projection is applied channels-last on every remaining (t,h,w) position
        x = self.proj(x.permute((0, 2, 3, 4, 1))).permute((0, 4, 1, 2, 3))
        if self.activation is not None:
            x = self.activation(x)
        if self.output_pool is not None:
            x = self.output_pool(x)
            x = x.view(x.shape[0], -1)
        return x


def create_res_basic_head(*, in_features, out_features, pool=nn.AvgPool3d, output_size=(1, 1, 1),
                          pool_kernel_size=(1, 7, 7), pool_stride=(1, 1, 1), pool_padding=(0, 0, 0),
                          dropout_rate=0.5, activation=None, output_with_global_average=True):
    """(reference: head.py:39-131)"""
    if pool is None:
        pool_model = None
    elif pool == nn.AdaptiveAvgPool3d:
        pool_model = pool(output_size)
    else:
        pool_model = pool(kernel_size=pool_kernel_size, stride=pool_stride, padding=pool_padding)
    return ResNetBasicHead(
        proj=nn.Linear(in_features, out_features),
        activation=_head_activation(activation),
        pool=pool_model,
        dropout=nn.Dropout(dropout_rate) if dropout_rate > 0 else None,
        output_pool=nn.AdaptiveAvgPool3d(1) if output_with_global_average else None,
    )


class ResNetRoIHead(nn.Module):
    """Detection head: pool -> RoIAlign on the (T == 1) map -> 2-D pool -> dropout -> per-position Linear
    -> activation -> optional global mean (reference: head.py:394-482).  `bboxes` is [R, 5]:
    (batch index, x1, y1, x2, y2) in input-image pixels."""

    def __init__(self, pool=None, pool_spatial=None, roi_layer=None, dropout=None, proj=None,
                 activation=None, output_pool=None) -> None:
        super().__init__()
        set_attributes(self, locals())
        assert self.proj is not None

    def forward(self, x: torch.Tensor, bboxes: torch.Tensor) -> torch.Tensor:
        if self.pool is not None:
            x = self.pool(x)
        if self.roi_layer is not None:
            if x.shape[-3] != 1:
                raise Exception("Temporal dimension should be 1. Consider modifying the pool layer.")
            x = self.roi_layer(torch.squeeze(x, -3), bboxes)
            if self.pool_spatial is not None:
                x = self.pool_spatial(x)
            x = x.unsqueeze(-3)
        if self.dropout is not None:
            x = self.dropout(x)
        x = self.proj(x.permute((0, 2, 3, 4, 1))).permute((0, 4, 1, 2, 3))
        if self.activation is not None:
            x = self.activation(x)
        if self.output_pool is not None:
            x = self.output_pool(x)
            x = x.view(x.shape[0], -1)
        return x


def create_res_roi_pooling_head(*, in_features, out_features, resolution, spatial_scale, sampling_ratio=0,
                                roi=RoIAlign, pool=nn.AvgPool3d, output_size=(1, 1, 1),
                                pool_kernel_size=(1, 7, 7), pool_stride=(1, 1, 1), pool_padding=(0, 0, 0),
                                pool_spatial=nn.MaxPool2d, dropout_rate=0.5, activation=None,
                                output_with_global_average=True):
    """(reference: head.py:203-327)"""
    if pool is None:
        pool_model = None
    elif pool == nn.AdaptiveAvgPool3d:
        pool_model = pool(output_size)
    else:
        pool_model = pool(kernel_size=pool_kernel_size, stride=pool_stride, padding=pool_padding)
    return ResNetRoIHead(
        proj=nn.Linear(in_features, out_features),
        activation=_head_activation(activation),
        pool=pool_model,
        pool_spatial=pool_spatial(resolution, stride=1) if pool_spatial else None,
        roi_layer=roi(output_size=resolution, spatial_scale=spatial_scale, sampling_ratio=sampling_ratio),
        dropout=nn.Dropout(dropout_rate) if dropout_rate > 0 else None,
        output_pool=nn.AdaptiveAvgPool3d(1) if output_with_global_average else None,
    )


class VisionTransformerBasicHead(nn.Module):
    """sequence pool -> dropout -> Linear -> activation (reference: head.py:485-535)."""

    def __init__(self, sequence_pool=None, dropout=None, proj=None, activation=None) -> None:
        super().__init__()
        set_attributes(self, locals())
        assert self.proj is not None

    def forward(self, x):
        if self.sequence_pool is not None:
            x = self.sequence_pool(x)
        if self.dropout is not None:
            x = self.dropout(x)
        x = self.proj(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def create_vit_basic_head(*, in_features, out_features, seq_pool_type="cls", dropout_rate=0.5,
                          activation=None):
    """(reference: head.py:134-200)"""
    assert seq_pool_type in ["cls", "mean", "none"]
    return VisionTransformerBasicHead(
        sequence_pool=None if seq_pool_type == "none" else SequencePool(seq_pool_type),
        dropout=nn.Dropout(dropout_rate) if dropout_rate > 0.0 else None,
        proj=nn.Linear(in_features, out_features),
        activation=_head_activation(activation),
    )
