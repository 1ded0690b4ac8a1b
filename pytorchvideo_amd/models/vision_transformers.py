"""Multiscale Vision Transformers (reference: pytorchvideo/models/vision_transformers.py).
`fuse_bn` (:123-170) is not mirrored: it is broken in the reference snapshot."""
from functools import partial
from typing import Callable, List, Optional

import torch
import torch.nn as nn

from ..layers.attention import MultiScaleBlock
from ..layers.positional_encoding import SpatioTemporalClsPositionalEncoding
from ..layers.utils import round_width
from .head import create_vit_basic_head
from .stem import create_conv_patch_embed
from .weight_init import init_net_weights


class MultiscaleVisionTransformers(nn.Module):
    """patch embed -> cls/pos encoding -> MultiScaleBlocks -> norm -> head
    (reference: vision_transformers.py:18-182)."""

    def __init__(self, *, patch_embed, cls_positional_encoding, pos_drop, blocks, norm_embed, head) -> None:
        super().__init__()
        assert hasattr(cls_positional_encoding, "patch_embed_shape"), (
            "cls_positional_encoding should have method patch_embed_shape.")
        self.patch_embed = patch_embed or nn.Identity()
        self.cls_positional_encoding = cls_positional_encoding
        self.pos_drop = pos_drop or nn.Identity()
        self.blocks = blocks
        self.norm_embed = norm_embed or nn.Identity()
        self.head = head or nn.Identity()
        init_net_weights(self, init_std=0.02, style="vit")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.pos_drop(self.cls_positional_encoding(self.patch_embed(x)))
        thw = self.cls_positional_encoding.patch_embed_shape()
        for blk in self.blocks:
            x, thw = blk(x, thw)
        return self.head(self.norm_embed(x))


def _pool_kernel(strides, fixed_kernel):
    """Pooling kernel for a stride triple: the configured one, else stride+1 where strided."""
    return fixed_kernel if fixed_kernel is not None else [s + 1 if s > 1 else s for s in strides]


def create_multiscale_vision_transformers(
        *, spatial_size, temporal_size, cls_embed_on=True, sep_pos_embed=True, depth=16, norm="layernorm",
        enable_patch_embed=True, input_channels=3, patch_embed_dim=96, conv_patch_embed_kernel=(3, 7, 7),
        conv_patch_embed_stride=(2, 4, 4), conv_patch_embed_padding=(1, 3, 3), enable_patch_embed_norm=False,
        use_2d_patch=False, num_heads=1, mlp_ratio=4.0, qkv_bias=True, dropout_rate_block=0.0,
        droppath_rate_block=0.0, pooling_mode="conv", pool_first=False, residual_pool=False,
        depthwise_conv=True, bias_on=True, separate_qkv=True,
        embed_dim_mul: Optional[List[List[int]]] = None, atten_head_mul: Optional[List[List[int]]] = None,
        dim_mul_in_att=False, pool_q_stride_size: Optional[List[List[int]]] = None,
        pool_kv_stride_size: Optional[List[List[int]]] = None, pool_kv_stride_adaptive=None,
        pool_kvq_kernel=None, head: Optional[Callable] = create_vit_basic_head, head_dropout_rate=0.5,
        head_activation: Callable = None, head_num_classes=400, create_scriptable_model=False,
        multiscale_vit_class: Callable = MultiscaleVisionTransformers) -> nn.Module:
    """MViT builder (reference: vision_transformers.py:185-506).  MViT-B 32x3 =
    spatial 224, temporal 32, embed_dim_mul/atten_head_mul [[1,2],[3,2],[14,2]],
    pool_q_stride_size [[1,1,2,2],[3,1,2,2],[14,1,2,2]], pool_kv_stride_adaptive [1,8,8],
    pool_kvq_kernel [3,3,3] (models/hub/vision_transformers.py:31-39)."""
    if use_2d_patch:
        assert temporal_size == 1, "If use_2d_patch, temporal_size needs to be 1."
    if pool_kv_stride_adaptive is not None:
        assert pool_kv_stride_size is None, (
            "pool_kv_stride_size should be none if pool_kv_stride_adaptive is set.")
    if norm == "layernorm":
        norm_layer = block_norm_layer = attn_norm_layer = partial(nn.LayerNorm, eps=1e-6)
    elif norm == "batchnorm":
        norm_layer, block_norm_layer, attn_norm_layer = None, nn.BatchNorm1d, nn.BatchNorm3d
    else:
        raise NotImplementedError("Only supports layernorm.")
    if create_scriptable_model:
        assert norm == "batchnorm", "The scriptable model supports only the batchnorm-based model."
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)

    patch_embed = None
    if enable_patch_embed:
        patch_embed = create_conv_patch_embed(
            in_channels=input_channels, out_channels=patch_embed_dim,
            conv_kernel_size=conv_patch_embed_kernel, conv_stride=conv_patch_embed_stride,
            conv_padding=conv_patch_embed_padding, conv=nn.Conv2d if use_2d_patch else nn.Conv3d)
    dims = [temporal_size, spatial_size[0], spatial_size[1]]
    strides = (1,) + tuple(conv_patch_embed_stride) if use_2d_patch else conv_patch_embed_stride
    grid = [dims[i] // strides[i] for i in range(3)] if enable_patch_embed else dims
    cls_pos = SpatioTemporalClsPositionalEncoding(embed_dim=patch_embed_dim, patch_embed_shape=grid,
                                                  sep_pos_embed=sep_pos_embed, has_cls=cls_embed_on)
    drop_path_rates = [r.item() for r in torch.linspace(0, droppath_rate_block, depth)]

    dim_mul, head_mul = torch.ones(depth + 1), torch.ones(depth + 1)
    for idx, mul in embed_dim_mul or []:
        dim_mul[idx] = mul
    for idx, mul in atten_head_mul or []:
        head_mul[idx] = mul

    # per-block pooling schedule
    kernel_q = [[] for _ in range(depth)]
    kernel_kv = [[] for _ in range(depth)]
    stride_q = [[] for _ in range(depth)]
    stride_kv = [[] for _ in range(depth)]
    for entry in pool_q_stride_size or []:
        stride_q[entry[0]] = entry[1:]
        kernel_q[entry[0]] = _pool_kernel(entry[1:], pool_kvq_kernel)
    if pool_kv_stride_adaptive is not None:
        # the K/V stride shrinks wherever Q is strided, so that the K/V grid stays constant
        cur = pool_kv_stride_adaptive
        pool_kv_stride_size = []
        for i in range(depth):
            if len(stride_q[i]) > 0:
                cur = [max(cur[d] // stride_q[i][d], 1) for d in range(len(cur))]
            pool_kv_stride_size.append([i] + cur)
    for entry in pool_kv_stride_size or []:
        stride_kv[entry[0]] = entry[1:]
        kernel_kv[entry[0]] = _pool_kernel(entry[1:], pool_kvq_kernel)

    blocks = nn.ModuleList()
    dim_in = patch_embed_dim
    for i in range(depth):
        num_heads = round_width(num_heads, head_mul[i], min_width=1, divisor=1)
        j = i if dim_mul_in_att else i + 1
        dim_out = round_width(dim_in, dim_mul[j], divisor=round_width(num_heads, head_mul[j]))
        blocks.append(MultiScaleBlock(
            dim=dim_in, dim_out=dim_out, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
            dropout_rate=dropout_rate_block, droppath_rate=drop_path_rates[i], norm_layer=block_norm_layer,
            attn_norm_layer=attn_norm_layer, dim_mul_in_att=dim_mul_in_att, kernel_q=kernel_q[i],
            kernel_kv=kernel_kv[i], stride_q=stride_q[i], stride_kv=stride_kv[i], pool_mode=pooling_mode,
            has_cls_embed=cls_embed_on, pool_first=pool_first, residual_pool=residual_pool, bias_on=bias_on,
            depthwise_conv=depthwise_conv, separate_qkv=separate_qkv))
        dim_in = dim_out

    head_model = None
    if head is not None:
        head_model = head(in_features=dim_in, out_features=head_num_classes,
                          seq_pool_type="cls" if cls_embed_on else "mean", dropout_rate=head_dropout_rate,
                          activation=head_activation)
    return multiscale_vit_class(
        patch_embed=patch_embed, cls_positional_encoding=cls_pos,
        pos_drop=nn.Dropout(p=dropout_rate_block) if dropout_rate_block > 0.0 else None,
        blocks=blocks, norm_embed=None if norm_layer is None else norm_layer(dim_in), head=head_model)
