"""MViT pooling attention (reference: pytorchvideo/layers/attention.py).

Original form only: plain torch ops with the reference's module / parameter names
(`attn._attention_pool_k.pool.weight`, `mlp.fc1`, ...).  The MI355X deploy form replaces a
whole `MultiScaleBlock` (LN -> q/k/v GEMM -> depthwise pooling on the token layout -> LN ->
fused QK^T-softmax-PV kernel -> proj -> skip max-pool -> LN -> MLP) -- see
accelerator/mi355x/emit_mvit.py.
"""
from typing import Callable, List, Optional, Tuple

import numpy
import torch
import torch.nn as nn

from .drop_path import DropPath


class Mlp(nn.Module):
    """fc1 -> act -> [dropout] -> fc2 -> [dropout] (reference: attention.py:51-114)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer: Callable = nn.GELU,
                 dropout_rate: float = 0.0, bias_on: bool = True) -> None:
        super().__init__()
        self.dropout_rate = dropout_rate
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias_on)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias_on)
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0.0 else nn.Identity()

    def forward(self, x):
        x = self.act(self.fc1(x))
        if self.dropout_rate > 0.0:
            x = self.dropout(x)
        x = self.fc2(x)
        if self.dropout_rate > 0.0:
            x = self.dropout(x)
        return x


class _AttentionPool(nn.Module):
    """Pool a (B, heads, N, C) or (B, N, C) token tensor on its (T,H,W) grid, keeping the cls
    token aside; optional norm before (BatchNorm3d + GELU) or after (LayerNorm) the pool
    (reference: attention.py:117-212)."""

    def __init__(self, pool: Optional[nn.Module], has_cls_embed: bool, norm: Optional[nn.Module]) -> None:
        super().__init__()
        self.has_pool = pool is not None
        self.pool = pool if pool is not None else nn.Identity()
        self.has_cls_embed = has_cls_embed
        self.has_norm = norm is not None
        self.norm_before_pool = isinstance(norm, (nn.BatchNorm3d, nn.Identity)) if norm is not None else False
        self.norm = norm if norm is not None else nn.Identity()

    def forward(self, tensor: torch.Tensor, thw_shape: List[int]) -> Tuple[torch.Tensor, List[int]]:
        if not self.has_pool:
            return tensor, thw_shape
        ndim = tensor.ndim
        if ndim == 3:
            tensor = tensor.unsqueeze(1)
        elif ndim != 4:
            raise NotImplementedError("Unsupported input dimension (expected (B,N,C) or (B,heads,N,C) tokens)")
        cls_tok = tensor[:, :, :1, :]
        if self.has_cls_embed:
            tensor = tensor[:, :, 1:, :]
        B, N, C = tensor.shape[0], tensor.shape[1], tensor.shape[3]
        grid = tensor.reshape(B * N, thw_shape[0], thw_shape[1], thw_shape[2], C).permute(0, 4, 1, 2, 3).contiguous()
        if self.norm_before_pool:
            grid = nn.functional.gelu(self.norm(grid))
        grid = self.pool(grid)
        thw_shape = [grid.shape[2], grid.shape[3], grid.shape[4]]
        tensor = grid.reshape(B, N, C, thw_shape[0] * thw_shape[1] * thw_shape[2]).transpose(2, 3)
        if self.has_cls_embed:
            tensor = torch.cat((cls_tok, tensor), dim=2)
        if self.has_norm and not self.norm_before_pool:
            tensor = self.norm(tensor)
        if ndim == 3:
            tensor = tensor.squeeze(1)
        return tensor, thw_shape


def _prod(shape: List[int]) -> int:
    p = 1
    for d in shape:
        p *= d
    return p


class MultiScaleAttention(nn.Module):
    """Multi-head attention whose q / k / v are pooled on the token grid before the
    softmax (reference: attention.py:215-575)."""

    _version = 3

    def __init__(self, dim, dim_out=None, num_heads=8, qkv_bias=False, dropout_rate=0.0,
                 kernel_q=(1, 1, 1), kernel_kv=(1, 1, 1), stride_q=(1, 1, 1), stride_kv=(1, 1, 1),
                 norm_layer: Callable = nn.LayerNorm, has_cls_embed=True, pool_mode="conv", pool_first=False,
                 residual_pool=True, depthwise_conv=True, bias_on=True, separate_qkv=True) -> None:
        super().__init__()
        assert pool_mode in ["conv", "avg", "max"]
        self.pool_first = pool_first
        self.dropout_rate = dropout_rate
        self.num_heads = num_heads
        dim_out = dim if not dim_out else dim_out
        self.dim_out = dim_out
        head_dim = dim_out // num_heads
        self.scale = head_dim ** -0.5
        self.has_cls_embed = has_cls_embed
        self.residual_pool = residual_pool
        self.separate_qkv = separate_qkv
        pad_q = [int(k // 2) for k in kernel_q]
        pad_kv = [int(k // 2) for k in kernel_kv]

        self.q = self.k = self.v = self.qkv = nn.Identity()
        if pool_first or separate_qkv:
            self.q = nn.Linear(dim, dim_out, bias=qkv_bias)
            self.k = nn.Linear(dim, dim_out, bias=qkv_bias)
            self.v = nn.Linear(dim, dim_out, bias=qkv_bias)
        else:
            self.qkv = nn.Linear(dim, dim_out * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim_out, dim_out, bias=True if bias_on else False)
        self.proj_drop = nn.Dropout(dropout_rate) if dropout_rate > 0.0 else nn.Identity()

        # a 1x1x1 / stride-1 pool is no pool at all
        if kernel_q is not None and _prod(kernel_q) == 1 and _prod(stride_q) == 1:
            kernel_q = None
        if kernel_kv is not None and _prod(kernel_kv) == 1 and _prod(stride_kv) == 1:
            kernel_kv = None

        if pool_mode in ("avg", "max"):
            op = nn.MaxPool3d if pool_mode == "max" else nn.AvgPool3d
            self.pool_q = op(kernel_q, stride_q, pad_q, ceil_mode=False) if kernel_q is not None else None
            self.pool_k = op(kernel_kv, stride_kv, pad_kv, ceil_mode=False) if kernel_kv is not None else None
            self.pool_v = op(kernel_kv, stride_kv, pad_kv, ceil_mode=False) if kernel_kv is not None else None
        elif pool_mode == "conv":
            c = (dim if pool_first else dim_out) // num_heads  # pooling weights are shared over heads

            def pool_conv(kernel, stride, pad):
                if kernel is None:
                    return None
                return nn.Conv3d(c, c, kernel, stride=stride, padding=pad, groups=c if depthwise_conv else 1,
                                 bias=False)

            self.pool_q = pool_conv(kernel_q, stride_q, pad_q)
            self.norm_q = norm_layer(c) if kernel_q is not None else None
            self.pool_k = pool_conv(kernel_kv, stride_kv, pad_kv)
            self.norm_k = norm_layer(c) if kernel_kv is not None else None
            self.pool_v = pool_conv(kernel_kv, stride_kv, pad_kv)
            self.norm_v = norm_layer(c) if kernel_kv is not None else None
        else:
            raise NotImplementedError(f"Unsupported model {pool_mode}")

        # the modules actually executed hold *references* to the pools / norms above
        self._attention_pool_q = _AttentionPool(self.pool_q, has_cls_embed, getattr(self, "norm_q", None))
        self._attention_pool_k = _AttentionPool(self.pool_k, has_cls_embed, getattr(self, "norm_k", None))
        self._attention_pool_v = _AttentionPool(self.pool_v, has_cls_embed, getattr(self, "norm_v", None))

    def _heads(self, t: torch.Tensor, B: int, n: int) -> torch.Tensor:
        return t.reshape(B, n, self.num_heads, -1).permute(0, 2, 1, 3)

    def forward(self, x: torch.Tensor, thw_shape: List[int]) -> Tuple[torch.Tensor, List[int]]:
        B, N, C = x.shape
        if self.pool_first:
            xh = self._heads(x, B, N)
            q, q_shape = self._attention_pool_q(xh, thw_shape)
            k, k_shape = self._attention_pool_k(xh, thw_shape)
            v, v_shape = self._attention_pool_v(xh, thw_shape)
            extra = 1 if self.has_cls_embed else 0
            nq, nk, nv = _prod(q_shape) + extra, _prod(k_shape) + extra, _prod(v_shape) + extra
            q = self._heads(self.q(q.permute(0, 2, 1, 3).reshape(B, nq, C)), B, nq)
            k = self._heads(self.k(k.permute(0, 2, 1, 3).reshape(B, nk, C)), B, nk)
            v = self._heads(self.v(v.permute(0, 2, 1, 3).reshape(B, nv, C)), B, nv)
        else:
            if self.separate_qkv:
                q, k, v = self._heads(self.q(x), B, N), self._heads(self.k(x), B, N), self._heads(self.v(x), B, N)
            else:
                qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
                q, k, v = qkv[0], qkv[1], qkv[2]
            q, q_shape = self._attention_pool_q(q, thw_shape)
            k, _ = self._attention_pool_k(k, thw_shape)
            v, _ = self._attention_pool_v(v, thw_shape)

        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        out = attn @ v
        if self.residual_pool:
            out = out + q
        x = self.proj(out.transpose(1, 2).reshape(B, -1, self.dim_out))
        if self.dropout_rate > 0.0:
            x = self.proj_drop(x)
        return x, q_shape

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        # checkpoints older than version 2 store the pools under pool_q / norm_q ... only
        # (reference: attention.py:546-575)
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for layer in ("pool", "norm"):
                for which in ("q", "k", "v"):
                    for kind in ("weight", "bias"):
                        old = f"{prefix}{layer}_{which}.{kind}"
                        if old in state_dict:
                            state_dict[f"{prefix}_attention_pool_{which}.{layer}.{kind}"] = state_dict[old]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)


class MultiScaleBlock(nn.Module):
    """Pre-norm transformer block with pooling attention; the skip path is max-pooled when
    q is strided, and the width change happens in the MLP (or in the attention when
    `dim_mul_in_att`) with the residual taken from the *normed* tensor
    (reference: attention.py:578-757)."""

    def __init__(self, dim, dim_out, num_heads, mlp_ratio=4.0, qkv_bias=False, dropout_rate=0.0,
                 droppath_rate=0.0, act_layer: nn.Module = nn.GELU, norm_layer: nn.Module = nn.LayerNorm,
                 attn_norm_layer: nn.Module = nn.LayerNorm, dim_mul_in_att=False, kernel_q=(1, 1, 1),
                 kernel_kv=(1, 1, 1), stride_q=(1, 1, 1), stride_kv=(1, 1, 1), pool_mode="conv",
                 has_cls_embed=True, pool_first=False, residual_pool=False, depthwise_conv=True,
                 bias_on=True, separate_qkv=True) -> None:
        super().__init__()
        self.dim = dim
        self.dim_out = dim_out
        self.norm1 = norm_layer(dim)
        self.dim_mul_in_att = dim_mul_in_att
        self.norm1_is_batchnorm_1d = isinstance(self.norm1, nn.BatchNorm1d)
        kernel_skip = [s + 1 if s > 1 else s for s in stride_q]
        att_dim = dim_out if dim_mul_in_att else dim
        self.attn = MultiScaleAttention(
            dim=dim, dim_out=att_dim, num_heads=num_heads, qkv_bias=qkv_bias, dropout_rate=dropout_rate,
            kernel_q=kernel_q, kernel_kv=kernel_kv, stride_q=stride_q, stride_kv=stride_kv,
            norm_layer=attn_norm_layer, has_cls_embed=has_cls_embed, pool_mode=pool_mode,
            pool_first=pool_first, residual_pool=residual_pool, bias_on=bias_on,
            depthwise_conv=depthwise_conv, separate_qkv=separate_qkv)
        self.drop_path = DropPath(droppath_rate) if droppath_rate > 0.0 else nn.Identity()
        self.norm2 = norm_layer(att_dim)
        self.norm2_is_batchnorm_1d = isinstance(self.norm2, nn.BatchNorm1d)
        self.has_cls_embed = has_cls_embed
        self.mlp = Mlp(in_features=att_dim, hidden_features=int(att_dim * mlp_ratio), out_features=dim_out,
                       act_layer=act_layer, dropout_rate=dropout_rate, bias_on=bias_on)
        self.proj = nn.Linear(dim, dim_out, bias=bias_on) if dim != dim_out else nn.Identity()
        self.pool_skip = (
            nn.MaxPool3d(kernel_skip, stride_q, [int(k // 2) for k in kernel_skip], ceil_mode=False)
            if len(stride_q) > 0 and numpy.prod(stride_q) > 1 else None)
        self._attention_pool = _AttentionPool(self.pool_skip, has_cls_embed=self.has_cls_embed, norm=None)

    def forward(self, x: torch.Tensor, thw_shape: List[int]) -> Tuple[torch.Tensor, List[int]]:
        # BatchNorm1d normalises dim 1, the tokens' channel dim is the last one
        if self.norm1_is_batchnorm_1d:
            x_norm = self.norm1(x.permute(0, 2, 1)).permute(0, 2, 1)
        else:
            x_norm = self.norm1(x)
        x_block, thw_new = self.attn(x_norm, thw_shape)
        if self.dim_mul_in_att and self.dim != self.dim_out:
            x = self.proj(x_norm)
        x_res, _ = self._attention_pool(x, thw_shape)
        x = x_res + self.drop_path(x_block)
        if self.norm2_is_batchnorm_1d:
            x_norm = self.norm2(x.permute(0, 2, 1)).permute(0, 2, 1)
        else:
            x_norm = self.norm2(x)
        x_mlp = self.mlp(x_norm)
        if not self.dim_mul_in_att and self.dim != self.dim_out:
            x = self.proj(x_norm)
        return x + self.drop_path(x_mlp), thw_new
