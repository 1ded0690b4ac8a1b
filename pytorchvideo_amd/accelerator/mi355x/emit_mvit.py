"""Emitters for the MViT path: PatchEmbed, cls/pos encoding, MultiScaleBlock, final norm, head.

Token tensors (B, N, C) are channels-last activations with T=H=1, W=N, i.e. they already are
the NDHWC layout of their (T,H,W) grid (plus the cls row in front): the reference's
permute + contiguous around every pooling conv (layers/attention.py:185-200) and the
(B,N,h*d) <-> (B,h,N,d) head shuffles (:425-451, :537) cost nothing here.

Launch list of one MultiScaleBlock (reference layers/attention.py:729-757):
  LN(norm1) -> one GEMM for q|k|v (weights concatenated, reads x_norm once)
  -> depthwise pooling conv on the token grid per q/k/v (weights shared over heads, cls row
     copied through) -> LN(head_dim) per (token, head), in place
  -> fused softmax(q k^T) v -> proj GEMM with bias + (max-pooled) skip in its epilogue
  -> LN(norm2) -> fc1 GEMM + bias + GELU -> fc2 GEMM + bias + residual
     (residual = block.proj(x_norm) when the width changes, :754-755).

Precision of the bf16 deploy form: the residual stream (block inputs/outputs, the skip path)
is kept in fp32 -- LayerNorm reads fp32 and writes the bf16 GEMM operand, the proj / fc2
epilogues add an fp32 residual and write fp32 -- so that 16 blocks of bf16 roundings do not
pile up on the stream; every MFMA operand (x_norm, q/k/v, P, attention output, MLP hidden)
is bf16.
"""
import ctypes as C

from . import tuning

import torch
import torch.nn as nn

from ... import _lib as L
from . import emit as E
from .emit import Unsupported, _cls_name
from .session import pad8


# --------------------------------------------------------------------------- helpers
def _linear_from(weight, bias):
    lin = nn.Linear(weight.shape[1], weight.shape[0], bias=bias is not None)
    lin.weight.data = weight.detach().float()
    if bias is not None:
        lin.bias.data = bias.detach().float()
    return lin


def emit_linear(sess, lin, x, act=L.ACT_NONE, residual=None, y_f32=False, label="linear"):
    """nn.Linear over the channel dim of a token tensor (+bias +act +residual) -> pv_conv3d."""
    if not isinstance(lin, nn.Linear):
        raise Unsupported("%s is not nn.Linear" % _cls_name(lin))
    if lin.in_features != x.C:
        raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (Linear expects %d features, got %d)"
                           % (lin.in_features, x.C))
    conv = nn.Conv3d(lin.in_features, lin.out_features, 1, bias=lin.bias is not None)
    conv.weight.data = lin.weight.detach().reshape(lin.out_features, lin.in_features, 1, 1, 1)
    if lin.bias is not None:
        conv.bias.data = lin.bias.detach()
    y = E.emit_conv(sess, conv, x, None, act, residual=residual, y_f32=y_f32, label=label)
    y.thw, y.has_cls = x.thw, x.has_cls
    return y


def emit_layernorm(sess, norm, x, out=None, rows=None, ldx=None, label="layernorm"):
    """nn.LayerNorm over the last dim.  Default: every row of the token tensor; `rows`/`ldx`
    select a strided subset (e.g. only the cls row of every batch item)."""
    if not isinstance(norm, nn.LayerNorm):
        raise Unsupported("norm %s" % _cls_name(norm))
    if len(norm.normalized_shape) != 1 or norm.normalized_shape[0] != x.C:
        raise RuntimeError("LayerNorm over %s applied to %d channels" % (tuple(norm.normalized_shape), x.C))
    if x.bs != x.voxels * x.ld:
        raise Unsupported("LayerNorm on a non-dense token tensor")
    y = out if out is not None else sess.alloc_act(x.B, x.T, x.H, x.W, x.C)
    if y.f32 and sess.pv_dtype != L.PV_F32:
        raise Unsupported("LayerNorm with an fp32 output in a bf16 session")
    y.thw, y.has_cls = x.thw, x.has_cls
    n_rows = x.B * x.voxels if rows is None else rows
    f = dict(x=x.ptr, y=y.ptr,
             gamma=sess.add_weight(norm.weight.detach().float()) if norm.weight is not None else None,
             beta=sess.add_weight(norm.bias.detach().float()) if norm.bias is not None else None,
             rows=n_rows, C=x.C, ldx=x.ld if ldx is None else ldx, ldy=y.ld if ldx is None else y.ld,
             rows_per_batch=0, eps=float(norm.eps), dtype=sess.pv_dtype,
             x_f32=1 if (x.f32 and sess.pv_dtype != L.PV_F32) else 0)
    sess.add_op(L.OP_LAYERNORM, f, label=label, alg_bytes=(x.itemsize + sess.itemsize) * n_rows * pad8(x.C))
    return y


def _bn_affine(norm, C):
    """Folded eval-mode BatchNorm: per-channel (gamma, beta) with y = x * gamma + beta."""
    if norm.running_mean is None or norm.num_features != C:
        raise Unsupported("BatchNorm without running statistics / channel mismatch")
    g = (norm.running_var.detach().double() + norm.eps).rsqrt()
    if norm.weight is not None:
        g = g * norm.weight.detach().double()
    b = -norm.running_mean.detach().double() * g
    if norm.bias is not None:
        b = b + norm.bias.detach().double()
    return g.float(), b.float()


def emit_affine_rows(sess, x, gamma, beta, act=L.ACT_NONE, out=None, rows=None, ldx=None, rows_per_batch=0, n_prefix=0,
                     label="batchnorm"):
    """y = act(x * gamma + beta) on token rows (pv_affine_rows): BatchNorm in eval mode for the norm="batchnorm" MViT
    (reference models/vision_transformers.py:336-339).  `out=None`: a new bf16 operand tensor; `out is x`: in place."""
    y = out if out is not None else sess.alloc_act(x.B, x.T, x.H, x.W, x.C)
    if y is not x:
        y.thw, y.has_cls = x.thw, x.has_cls
    if y.f32 and sess.pv_dtype != L.PV_F32 and y is not x:
        raise Unsupported("fp32 output in a bf16 session")
    n_rows = x.B * x.voxels if rows is None else rows
    f = dict(x=x.ptr, y=y.ptr, gamma=sess.add_weight(gamma) if gamma is not None else None,
             beta=sess.add_weight(beta) if beta is not None else None,
             rows=n_rows, C=x.C, ldx=x.ld if ldx is None else ldx, ldy=y.ld, rows_per_batch=rows_per_batch, eps=0.0,
             dtype=sess.pv_dtype, x_f32=1 if (x.f32 and sess.pv_dtype != L.PV_F32) else 0, g_period=0,
             act=act, n_prefix=n_prefix)
    sess.add_op(L.OP_AFFINE_ROWS, f, label=label, alg_bytes=(x.itemsize + sess.itemsize) * n_rows * pad8(x.C))
    return y


def emit_block_norm(sess, norm, x, label):
    """norm1 / norm2 of a MultiScaleBlock: LayerNorm, or BatchNorm1d over the channel dim (attention.py:738-753)."""
    if isinstance(norm, nn.BatchNorm1d):
        if x.bs != x.voxels * x.ld:
            raise Unsupported("BatchNorm1d on a non-dense token tensor")
        g, b = _bn_affine(norm, x.C)
        return emit_affine_rows(sess, x, g, b, label=label)
    return emit_layernorm(sess, norm, x, label=label)


class _PoolAfterNorm:
    """View of an _AttentionPool whose norm (BatchNorm3d + GELU BEFORE the pool, attention.py:186-190) has already
    been applied to its input: what is left is the pool and the cls pass-through."""
    has_norm, norm_before_pool, norm = False, False, None

    def __init__(self, ap):
        self.has_pool, self.pool, self.has_cls_embed = ap.has_pool, ap.pool, ap.has_cls_embed


def emit_norm_before_pool(sess, ap, x, heads, label):
    """BatchNorm3d(head_dim) (or Identity) + GELU on the non-cls tokens of `x`, in place; returns the pool view."""
    if not ap.has_pool or not (ap.has_norm and ap.norm_before_pool):
        return ap
    hd = x.C // heads
    norm = ap.norm
    if isinstance(norm, nn.Identity):
        g = b = None
    elif isinstance(norm, nn.BatchNorm3d):
        g, b = _bn_affine(norm, hd)
        g, b = g.repeat(heads), b.repeat(heads)          # the same statistics for every head (tensor is (B*heads, hd, T, H, W))
    else:
        raise Unsupported("pre-pooling norm %s" % _cls_name(norm))
    n_prefix = 1 if ap.has_cls_embed else 0
    emit_affine_rows(sess, x, g, b, act=L.ACT_GELU, out=x, rows=x.B * x.voxels, rows_per_batch=x.voxels,
                     n_prefix=n_prefix, label=label + ".bn_gelu")
    return _PoolAfterNorm(ap)


def emit_head_layernorm(sess, norm, x, heads, label="pool.norm"):
    """LayerNorm(head_dim) applied to every (token, head) of a dense (B, N, heads*head_dim)
    tensor, in place (reference: _AttentionPool norm after pool, attention.py:202-205)."""
    if not isinstance(norm, nn.LayerNorm):
        raise Unsupported("attention norm %s" % _cls_name(norm))
    hd = x.C // heads
    if len(norm.normalized_shape) != 1 or norm.normalized_shape[0] != hd:
        raise RuntimeError("attention LayerNorm over %s, head dim %d" % (tuple(norm.normalized_shape), hd))
    if hd % 8 or x.ld != x.C or x.bs != x.voxels * x.ld:
        raise Unsupported("per-head LayerNorm needs a dense tensor and head_dim % 8 == 0")
    n_rows = x.B * x.voxels * heads
    f = dict(x=x.ptr, y=x.ptr,
             gamma=sess.add_weight(norm.weight.detach().float()) if norm.weight is not None else None,
             beta=sess.add_weight(norm.bias.detach().float()) if norm.bias is not None else None,
             rows=n_rows, C=hd, ldx=hd, ldy=hd, rows_per_batch=0, eps=float(norm.eps), dtype=sess.pv_dtype,
             x_f32=0)
    sess.add_op(L.OP_LAYERNORM, f, label=label, alg_bytes=2 * sess.itemsize * n_rows * hd)
    return x


def _pool_params(pool):
    if not isinstance(pool, (nn.MaxPool3d, nn.AvgPool3d)):
        raise Unsupported("pool %s" % _cls_name(pool))
    if getattr(pool, "ceil_mode", False):
        raise Unsupported("ceil_mode pooling")
    k = E._triple(pool.kernel_size)
    s = E._triple(pool.stride if pool.stride is not None else pool.kernel_size)
    p = E._triple(pool.padding)
    if isinstance(pool, nn.MaxPool3d):
        if E._triple(pool.dilation) != (1, 1, 1) or pool.return_indices:
            raise Unsupported("max pool options")
        return k, s, p, L.POOL_MAX
    if not pool.count_include_pad or pool.divisor_override is not None:
        raise Unsupported("avg pool options")
    return k, s, p, L.POOL_AVG


def emit_attention_pool(sess, ap, x, heads, label):
    """_AttentionPool.forward (attention.py:162-212) on a token tensor whose channel dim is
    heads*head_dim.  Returns a new dense tensor, or `x` itself when the module has no pool."""
    if not ap.has_pool:
        return x
    if x.thw is None:
        raise Unsupported("token tensor without a grid")
    if ap.has_norm and ap.norm_before_pool:
        raise Unsupported("BatchNorm3d+GELU before pooling")
    n_prefix = 1 if ap.has_cls_embed else 0
    pool = ap.pool
    if isinstance(pool, nn.Conv3d):
        hd = x.C // heads
        if pool.groups != pool.in_channels or pool.in_channels != pool.out_channels:
            raise Unsupported("dense pooling conv")
        if pool.in_channels != hd:
            raise RuntimeError("pooling conv has %d channels, head dim is %d" % (pool.in_channels, hd))
        if pool.bias is not None or tuple(pool.dilation) != (1, 1, 1) or pool.padding_mode != "zeros":
            raise Unsupported("pooling conv options")
        if hd % 8:
            raise Unsupported("head_dim % 8 != 0")
        y = E.emit_dwconv(sess, pool, x, None, L.ACT_NONE, w_mod=hd if heads > 1 else 0, grid=x.thw,
                          n_prefix=n_prefix, label=label)
    else:
        k, s, p, mode = _pool_params(pool)
        y = E.emit_pool_raw(sess, x, k, s, p, mode, n_prefix=n_prefix, grid=x.thw, label=label)
    if ap.has_norm:
        emit_head_layernorm(sess, ap.norm, y, heads, label=label + ".norm")
    return y


def _streams_well(ap, x):
    """Pooling convs the plane-streaming depthwise kernel takes (3x3x3, temporal stride 1, spatial
    stride 1 or 2) on a grid big enough to fill the chip: it reads every input voxel about once,
    where the fused per-token kernel re-reads the 27-voxel window from L2 for every output."""
    pool = ap.pool if ap.has_pool else None
    if not isinstance(pool, nn.Conv3d) or x.thw is None:
        return False
    if tuple(pool.kernel_size) != (3, 3, 3) or tuple(E._triple(pool.padding)) != (1, 1, 1):
        return False
    st = tuple(pool.stride)
    if st[0] != 1 or st[1] != st[2] or st[1] not in (1, 2):
        return False
    T, H, W = x.thw
    return x.B * T * H * W * x.C >= tuning.get("pool_stream_min_elems")


def emit_attention_pool_kv(sess, ap_k, ap_v, xkv, heads, label):
    """pool_k and pool_v of one MultiScaleAttention in ONE depthwise launch + ONE LayerNorm launch.
    k and v are adjacent channel slices of the fused qkv GEMM's output, so `xkv` (B, N, 2*dim) is a
    single token tensor whose depthwise filter table is [pool_k's filter x heads | pool_v's x heads] and
    whose per-head LayerNorm uses a (2*heads)-row gamma/beta table (pv_rows_desc.g_period).  Returns
    (k, v) as channel slices of the pooled tensor, or None when the two pools differ in geometry."""
    pk, pv = ap_k.pool, ap_v.pool
    if not (isinstance(pk, nn.Conv3d) and isinstance(pv, nn.Conv3d)) or xkv.thw is None:
        return None
    dim = xkv.C // 2
    hd = dim // heads
    same = ("kernel_size", "stride", "padding", "dilation", "groups", "in_channels", "out_channels", "padding_mode")
    if any(getattr(pk, a) != getattr(pv, a) for a in same) or pk.bias is not None or pv.bias is not None:
        return None
    if pk.groups != pk.in_channels or pk.in_channels != hd or hd % 8 or 2 * heads > 16 or 16 % (2 * heads):
        return None
    if bool(ap_k.has_cls_embed) != bool(ap_v.has_cls_embed) or bool(ap_k.has_norm) != bool(ap_v.has_norm):
        return None
    if ap_k.has_norm:
        nk, nv = ap_k.norm, ap_v.norm
        if ap_k.norm_before_pool or ap_v.norm_before_pool or not isinstance(nk, nn.LayerNorm) or not isinstance(nv, nn.LayerNorm):
            return None
        if tuple(nk.normalized_shape) != (hd,) or tuple(nv.normalized_shape) != (hd,) or nk.eps != nv.eps:
            return None
        if (nk.weight is None) != (nv.weight is None) or (nk.bias is None) != (nv.bias is None):
            return None
    # one depthwise conv over 2*dim channels: channel c uses pool_k's filter (c % hd) for c < dim, pool_v's after
    conv = nn.Conv3d(2 * dim, 2 * dim, pk.kernel_size, pk.stride, pk.padding, groups=2 * dim, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.cat([pk.weight.detach().float().repeat(heads, 1, 1, 1, 1),
                                     pv.weight.detach().float().repeat(heads, 1, 1, 1, 1)], 0))
    n_prefix = 1 if ap_k.has_cls_embed else 0
    y = E.emit_dwconv(sess, conv, xkv, None, L.ACT_NONE, grid=xkv.thw, n_prefix=n_prefix, label=label)
    if ap_k.has_norm:
        def table(pk_, pv_):
            if pk_ is None:
                return None
            return sess.add_weight(torch.cat([pk_.detach().float().repeat(heads), pv_.detach().float().repeat(heads)]))
        n_rows = y.B * y.voxels * 2 * heads
        f = dict(x=y.ptr, y=y.ptr, gamma=table(ap_k.norm.weight, ap_v.norm.weight), beta=table(ap_k.norm.bias, ap_v.norm.bias),
                 rows=n_rows, C=hd, ldx=hd, ldy=hd, rows_per_batch=0, eps=float(ap_k.norm.eps), dtype=sess.pv_dtype,
                 x_f32=0, g_period=2 * heads)
        sess.add_op(L.OP_LAYERNORM, f, label=label + ".norm", alg_bytes=2 * sess.itemsize * n_rows * hd)
    k = y.channel_slice(0, dim)
    v = y.channel_slice(dim, dim)
    for t in (k, v):
        t.thw, t.has_cls = y.thw, y.has_cls
    return y, k, v


def emit_attention_pools_fused(sess, pools, xs, heads, label, skip=()):
    """q / k / v pooling of one MultiScaleAttention as ONE launch (pv_token_pool): depthwise conv on
    the token grid + cls pass-through + LayerNorm(head_dim).  `pools` are the _AttentionPool modules,
    `xs` their inputs; tensors whose module has no pool are returned unchanged.  Returns None when
    the combination is not covered (the caller then emits the pools one by one)."""
    todo = [(i, ap, x) for i, (ap, x) in enumerate(zip(pools, xs)) if ap.has_pool and i not in skip]
    if not todo:
        return list(xs)
    x0 = todo[0][2]
    hd = x0.C // heads
    if x0.thw is None or hd % 8 or hd > 128:
        return None
    kernel = None
    for _, ap, x in todo:
        pool = ap.pool
        if not isinstance(pool, nn.Conv3d) or pool.groups != pool.in_channels or pool.in_channels != pool.out_channels:
            return None
        if pool.in_channels != hd or pool.bias is not None or tuple(pool.dilation) != (1, 1, 1) or pool.padding_mode != "zeros":
            return None
        if tuple(E._triple(pool.padding)) != tuple(k // 2 for k in pool.kernel_size):
            return None
        if ap.has_norm and (ap.norm_before_pool or not isinstance(ap.norm, nn.LayerNorm)
                            or tuple(ap.norm.normalized_shape) != (hd,)):
            return None
        if kernel is None:
            kernel, eps, prefix = tuple(pool.kernel_size), (float(ap.norm.eps) if ap.has_norm else 1e-6), ap.has_cls_embed
        elif tuple(pool.kernel_size) != kernel or bool(ap.has_cls_embed) != bool(prefix) or x.thw != x0.thw \
                or x.C != x0.C or (ap.has_norm and float(ap.norm.eps) != eps):
            return None
    if kernel[0] * kernel[1] * kernel[2] > 64:
        return None
    T, H, W = x0.thw
    n_prefix = 1 if prefix else 0
    out = list(xs)
    f = dict(x=[], y=[], w=[], gamma=[], beta=[], x_bs=[], y_bs=[], ldx=[], ldy=[], st=[], sh=[], sw=[],
             To=[], Ho=[], Wo=[], n=len(todo), B=x0.B, Ti=T, Hi=H, Wi=W, heads=heads, head_dim=hd,
             kt=kernel[0], kh=kernel[1], kw=kernel[2], n_prefix=n_prefix, eps=eps, dtype=sess.pv_dtype)
    alg = flops = 0
    for i, ap, x in todo:
        pool = ap.pool
        st = tuple(pool.stride)
        To, Ho, Wo = [E._conv_out(v, k, s_, k // 2) for v, k, s_ in zip((T, H, W), kernel, st)]
        if min(To, Ho, Wo) <= 0:
            raise RuntimeError("pool output would be empty")
        y = sess.alloc_act(x.B, 1, 1, n_prefix + To * Ho * Wo, x.C)
        y.thw, y.has_cls = (To, Ho, Wo), bool(n_prefix)
        f["x"].append(x.ptr), f["y"].append(y.ptr)
        f["w"].append(sess.add_weight(pool.weight.detach().float().reshape(hd, -1).t().contiguous()))
        f["gamma"].append(sess.add_weight(ap.norm.weight.detach().float()) if ap.has_norm and ap.norm.weight is not None else None)
        f["beta"].append(sess.add_weight(ap.norm.bias.detach().float()) if ap.has_norm and ap.norm.bias is not None else None)
        f["x_bs"].append(x.bs), f["y_bs"].append(y.bs), f["ldx"].append(x.ld), f["ldy"].append(y.ld)
        f["st"].append(st[0]), f["sh"].append(st[1]), f["sw"].append(st[2])
        f["To"].append(To), f["Ho"].append(Ho), f["Wo"].append(Wo)
        vin, vout = x.B * T * H * W, x.B * To * Ho * Wo
        alg += sess.itemsize * (min(vin, vout * kernel[0] * kernel[1] * kernel[2]) + vout) * x.C
        flops += 2 * vout * x.C * kernel[0] * kernel[1] * kernel[2]
        out[i] = y
    sess.add_op(L.OP_TOKEN_POOL, f, label=label, alg_bytes=alg, flops=flops)
    return out


def emit_attention_core(sess, q, k, v, heads, scale, residual_q, label="attention"):
    hd = q.C // heads
    if hd not in (32, 64, 96, 128):
        raise Unsupported("head_dim %d" % hd)
    o = sess.alloc_act(q.B, 1, 1, q.voxels, q.C)
    o.thw, o.has_cls = q.thw, q.has_cls
    f = dict(q=q.ptr, k=k.ptr, v=v.ptr, o=o.ptr, q_bs=q.bs, k_bs=k.bs, v_bs=v.bs, o_bs=o.bs,
             ldq=q.ld, ldk=k.ld, ldv=v.ld, ldo=o.ld, B=q.B, heads=heads, head_dim=hd,
             Nq=q.voxels, Nk=k.voxels, scale=float(scale), residual_q=1 if residual_q else 0,
             dtype=sess.pv_dtype)
    alg = sess.itemsize * q.B * q.C * (2 * q.voxels + 2 * k.voxels)
    flops = 4 * q.B * heads * q.voxels * k.voxels * hd
    sess.add_op(L.OP_ATTENTION, f, label=label, alg_bytes=alg, flops=flops)
    return o


def _chi(rho):
    """Row permutation of a 32-row MFMA tile that makes a lane's 16 accumulator registers 16 consecutive channels."""
    return 16 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 3) + (rho & 3)


def pack_mlp_weights(w1, b1, w2):
    """The per-hidden-block LDS image pv_mlp_rows streams (layout: include/pv_mi355x.h, pv_mlp_desc).
    w1 [H, C], b1 [H] or None, w2 [Cout, H] (fp32, host) -> uint8 [(H/32 + 3) * (C/16*1024 + Cout/32*2048 + 256)]."""
    H, Cin = w1.shape
    Cout = w2.shape[0]
    NH, KS, NOB = H // 32, Cin // 16, Cout // 32
    w1 = w1.detach().float().cpu()
    w2 = w2.detach().float().cpu()
    # W1: fragment f = 2 ks + uh, lane l = 16 g + m, element j8:  W1[32 hb + 16 uh + m][32 ks + 8 g + j8]
    #     [hb, uh, m, ks, g, j8] -> [hb, ks, uh, g, m, j8]
    w1p = w1.reshape(NH, 2, 16, Cin // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(NH, KS * 512)
    # W2: fragment ob (16 output channels), lane l = 16 g + m, element j8:
    #     W2[32 (ob>>1) + 8 (m>>2) + 4 (ob&1) + (m&3)][32 hb + (j8 < 4 ? 4 g + j8 : 16 + 4 g + j8 - 4)]
    ob = torch.arange(Cout // 16)
    mrow = torch.arange(16)
    rows = (32 * (ob >> 1) + 4 * (ob & 1))[:, None] + (8 * (mrow >> 2) + (mrow & 3))[None, :]               # [ob, m]
    gq, j8 = torch.arange(4), torch.arange(8)
    unit = torch.where(j8[None, :] < 4, 4 * gq[:, None] + j8[None, :], 16 + 4 * gq[:, None] + j8[None, :] - 4)   # [g, j8]
    cols = 32 * torch.arange(NH)[:, None, None] + unit[None, :, :]                                         # [hb, g, j8]
    w2p = w2[rows[None, :, None, :, None], cols[:, None, :, None, :]].reshape(NH, NOB * 1024)              # [hb, ob, g, m, j8]
    b1p = torch.zeros(NH, 64, dtype=torch.float32)
    if b1 is not None:
        b1p[:, :32] = b1.detach().float().cpu().reshape(NH, 32)
    # the kernel's software pipeline multiplies phase B of hidden block j - 1 in iteration j: image block j = [W1(j) | W2(j-1) | b1(j)],
    # j = 0 .. NH with W2(-1) = W1(NH) = b1(NH) = 0, followed by two blocks of padding (prefetched, never used)
    z = lambda t: torch.zeros(1, t.shape[1], dtype=t.dtype)
    w1b = torch.cat([w1p, z(w1p)]).to(torch.bfloat16).view(torch.int16).view(torch.uint8).reshape(NH + 1, -1)
    w2b = torch.cat([z(w2p), w2p]).to(torch.bfloat16).view(torch.int16).view(torch.uint8).reshape(NH + 1, -1)
    b1b = torch.cat([b1p, z(b1p)]).view(torch.uint8).reshape(NH + 1, -1)
    img = torch.cat([w1b, w2b, b1b], dim=1)
    assert img.shape[1] == KS * 1024 + NOB * 2048 + 256
    img = torch.cat([img, torch.zeros(2, img.shape[1], dtype=torch.uint8)])
    return img.reshape(-1).contiguous()


def pack_ln_linear_weights(w, b):
    """The per-output-block LDS image pv_ln_linear_rows streams (layout: include/pv_mi355x.h, pv_ln_linear_desc).
    w [N, C], b [N] or None (fp32, host) -> uint8 [N/32 * (C/16*1024 + 256)]."""
    N, Cin = w.shape
    NB, KS = N // 32, Cin // 16
    w = w.detach().float().cpu()
    rows = 32 * torch.arange(NB)[:, None] + torch.tensor([_chi(r) for r in range(32)])[None, :]                # [nb, rho]
    wr = w[rows]                                                                                              # [nb, rho, C]
    # [nb, rho, q, hi, e, j] -> [nb, ks = (q, e), hi, rho, j];  channel = 32 q + 16 hi + 8 e + j
    wp = wr.reshape(NB, 32, Cin // 32, 2, 2, 8).permute(0, 2, 4, 3, 1, 5).reshape(NB, KS * 512)
    bp = torch.zeros(NB, 64, dtype=torch.float32)
    if b is not None:
        bp[:, :32] = b.detach().float().cpu().reshape(NB, 32)          # [hi][r] = b[32 nb + 16 hi + r]
    img = torch.cat([wp.to(torch.bfloat16).view(torch.int16).view(torch.uint8).reshape(NB, -1),
                     bp.view(torch.uint8).reshape(NB, -1)], dim=1)
    assert img.shape[1] == KS * 1024 + 256
    img = torch.cat([img, torch.zeros(2, img.shape[1], dtype=torch.uint8)])      # two blocks of padding (prefetched, never used)
    return img.reshape(-1).contiguous()


def can_fuse_ln_linear(sess, norm, lin, x):
    """norm (LayerNorm over the fp32 stream) -> lin as ONE pv_ln_linear_rows launch (csrc/pv_mlp.hip)?"""
    if not tuning.get("fuse_ln_qkv") or sess.pv_dtype != L.PV_BF16 or not x.f32 or not isinstance(norm, nn.LayerNorm):
        return False
    if not isinstance(lin, nn.Linear) or norm.weight is None or norm.bias is None:
        return False
    if tuple(norm.normalized_shape) != (x.C,) or lin.in_features != x.C or x.bs != x.voxels * x.ld:
        return False
    # Measured per layer on MViT-B (b = 8, same box, norm1 + qkv GEMM vs the fused launch): 96 ch 150 -> 119 us, 192 ch
    # 335 -> 271 us (401 k rows) and 75 -> 72 us (100 k rows); 384 ch 48 -> 55 us, 768 ch 42 -> 140 us (25 k / 6 k rows: a
    # row-resident workgroup per 128 rows leaves the chip half empty and the 128 x 128-tile GEMM wins).
    if x.C > tuning.get("fuse_ln_qkv_max_c"):
        return False
    d = L.LnLinearDesc()
    d.x = d.wb = d.y = d.ln_gamma = d.ln_beta = 1
    d.M, d.C, d.N, d.ldx, d.ldy, d.dtype = x.B * x.voxels, x.C, lin.out_features, x.ld, pad8(lin.out_features), L.PV_BF16
    return L.lib().pv_ln_linear_rows_supported(C.byref(d)) == 1


def emit_ln_linear(sess, norm, lin, x, label):
    y = sess.alloc_act(x.B, 1, 1, x.voxels, lin.out_features)
    y.thw, y.has_cls = x.thw, x.has_cls
    M = x.B * x.voxels
    f = dict(x=x.ptr, wb=sess.add_weight(pack_ln_linear_weights(lin.weight, lin.bias)), y=y.ptr,
             ln_gamma=sess.add_weight(norm.weight.detach().float()), ln_beta=sess.add_weight(norm.bias.detach().float()),
             M=M, C=x.C, N=lin.out_features, ldx=x.ld, ldy=y.ld, act=L.ACT_NONE, dtype=L.PV_BF16, ln_eps=float(norm.eps))
    sess.add_op(L.OP_LN_LINEAR, f, label="%s|%dx%d c%d->%d ln" % (label, x.B, x.voxels, x.C, lin.out_features),
                alg_bytes=M * (4 * pad8(x.C) + 2 * pad8(lin.out_features)) + 2 * x.C * lin.out_features,
                flops=2 * M * x.C * lin.out_features)
    return y


def can_fuse_mlp(sess, blk, x1):
    """norm2 -> fc1 -> act -> fc2 -> + residual of a MultiScaleBlock as ONE pv_mlp_rows launch (csrc/pv_mlp.hip)?"""
    if not tuning.get("fuse_mlp") or sess.pv_dtype != L.PV_BF16 or not x1.f32:
        return False
    mlp = blk.mlp
    if not isinstance(getattr(mlp, "fc1", None), nn.Linear) or not isinstance(getattr(mlp, "fc2", None), nn.Linear):
        return False
    if not isinstance(getattr(mlp, "dropout", nn.Identity()), (nn.Identity, nn.Dropout)):
        return False
    if x1.bs != x1.voxels * x1.ld or mlp.fc1.in_features != x1.C or mlp.fc2.in_features != mlp.fc1.out_features:
        return False
    d = L.MlpDesc()
    d.x = d.w12 = d.y = 1      # non-null placeholders: only the geometry is judged
    d.M, d.C, d.H, d.Cout = x1.B * x1.voxels, mlp.fc1.in_features, mlp.fc1.out_features, mlp.fc2.out_features
    d.ldx, d.ldr, d.ldy, d.dtype = pad8(d.C), pad8(d.Cout), pad8(d.Cout), L.PV_BF16
    return L.lib().pv_mlp_rows_supported(C.byref(d)) == 1


def _next_norm1_standalone(sess, nxt, y):
    """Will the NEXT MultiScaleBlock evaluate its norm1 as a LayerNorm launch of its own over `y` (this block's output)?
    Mirrors the first lines of emit_multiscale_block: not when norm1 is fused into the q|k|v projection."""
    if nxt is None or not tuning.get("fuse_next_norm") or sess.pv_dtype != L.PV_BF16:
        return False
    n1 = getattr(nxt, "norm1", None)
    if not isinstance(n1, nn.LayerNorm) or n1.weight is None or n1.bias is None or tuple(n1.normalized_shape) != (y.C,):
        return False
    if y.C % 32 or y.bs != y.voxels * y.ld:
        return False
    widen = nxt.dim != nxt.dim_out
    if not nxt.attn.pool_first and not (nxt.dim_mul_in_att and widen):
        try:
            if can_fuse_ln_linear(sess, n1, _qkv_linear(nxt.attn), y):
                return False
        except E.Unsupported:
            return False
    return True


def emit_mlp_fused(sess, blk, x1):
    """Second half of MultiScaleBlock.forward (layers/attention.py:750-757, Mlp.forward :102-114).  With a LayerNorm
    norm2 and an unchanged width the kernel normalises the fp32 stream itself and uses it as the residual (one read);
    otherwise (width change: the residual is blk.proj(norm2(x)); BatchNorm norm2) norm2 stays its own launch."""
    mlp = blk.mlp
    widen = blk.dim != blk.dim_out
    act = E.act_code(mlp.act)
    Cin, H, Cout = mlp.fc1.in_features, mlp.fc1.out_features, mlp.fc2.out_features
    w12 = sess.add_weight(pack_mlp_weights(mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight))
    b2 = mlp.fc2.bias.detach().float() if mlp.fc2.bias is not None else torch.zeros(Cout)
    y = sess.alloc_act(x1.B, 1, 1, x1.voxels, Cout, f32=True)
    y.thw, y.has_cls = x1.thw, x1.has_cls
    M = x1.B * x1.voxels
    f = dict(w12=w12, y=y.ptr, b2=sess.add_weight(b2), M=M, C=Cin, H=H, Cout=Cout, ldy=y.ld, act=act, dtype=L.PV_BF16,
             residual=None, ln_gamma=None, ln_beta=None, ln_eps=0.0, ldr=0, yn=None, nn_gamma=None, nn_beta=None, ldyn=0, nn_eps=0.0)
    nxt = blk.__dict__.get("_pv_next_block")
    if _next_norm1_standalone(sess, nxt, y):
        # norm1 of the NEXT block from the rows this kernel still holds in registers (round 4): the next block finds the
        # bf16 operand on its input (`prenorm`) and emits no LayerNorm launch
        yn = sess.alloc_act(y.B, 1, 1, y.voxels, Cout)
        yn.thw, yn.has_cls = y.thw, y.has_cls
        f.update(yn=yn.ptr, ldyn=yn.ld, nn_gamma=sess.add_weight(nxt.norm1.weight.detach().float()),
                 nn_beta=sess.add_weight(nxt.norm1.bias.detach().float()), nn_eps=float(nxt.norm1.eps))
        y.prenorm = (yn, nxt.norm1)
    in_kernel_ln = isinstance(blk.norm2, nn.LayerNorm) and not ((not blk.dim_mul_in_att) and widen) and Cin == Cout \
        and blk.norm2.weight is not None and blk.norm2.bias is not None and tuple(blk.norm2.normalized_shape) == (Cin,)
    flops = 2 * M * H * (Cin + Cout)
    wbytes = 2 * H * (Cin + Cout)
    nn_tag, nn_bytes = (" +norm1'", 2 * M * pad8(Cout)) if f["yn"] is not None else ("", 0)
    if in_kernel_ln:
        f.update(x=x1.ptr, ldx=x1.ld, ln_gamma=sess.add_weight(blk.norm2.weight.detach().float()),
                 ln_beta=sess.add_weight(blk.norm2.bias.detach().float()), ln_eps=float(blk.norm2.eps))
        sess.add_op(L.OP_MLP_ROWS, f, label="mlp.fused|%dx%d c%d->%d->%d ln%s" % (x1.B, x1.voxels, Cin, H, Cout, nn_tag),
                    alg_bytes=4 * M * (pad8(Cin) + pad8(Cout)) + wbytes + nn_bytes, flops=flops)
        return y
    xn2 = emit_block_norm(sess, blk.norm2, x1, label="norm2")
    if (not blk.dim_mul_in_att) and widen:
        res2 = emit_linear(sess, blk.proj, xn2, y_f32=True, label="proj_dim")
    else:
        res2 = x1
    f.update(x=xn2.ptr, ldx=xn2.ld, residual=res2.ptr, ldr=res2.ld)
    sess.add_op(L.OP_MLP_ROWS, f, label="mlp.fused|%dx%d c%d->%d->%d%s" % (x1.B, x1.voxels, Cin, H, Cout, nn_tag),
                alg_bytes=2 * M * pad8(Cin) + 8 * M * pad8(Cout) + wbytes + nn_bytes, flops=flops)
    sess.release(xn2)
    if res2 is not x1:
        sess.release(res2)
    return y


# --------------------------------------------------------------------------- modules
def _qkv_linear(attn):
    """The q | k | v projection of a MultiScaleAttention as ONE nn.Linear (weights concatenated along the output dim)."""
    if attn.separate_qkv:
        ws = [attn.q.weight, attn.k.weight, attn.v.weight]
        bs = [attn.q.bias, attn.k.bias, attn.v.bias]
        bias = None if bs[0] is None else torch.cat([b.detach() for b in bs])
        lin = _linear_from(torch.cat([w.detach() for w in ws]), bias)
    else:
        lin = attn.qkv
    if lin.out_features != 3 * attn.dim_out or attn.dim_out % 8:
        raise Unsupported("qkv width")
    return lin


def emit_multiscale_attention(sess, attn, xn, residual, label="attn", qkv=None):
    """MultiScaleAttention.forward (attention.py:501-544) + the block's residual join fused into
    proj's epilogue.  Returns the (B, Nq, dim_out) token tensor.  `qkv`: the q|k|v tensor when the caller has
    already produced it (norm1 fused into the projection, emit_ln_linear); `xn` is then not needed."""
    heads = attn.num_heads
    if attn.dropout_rate > 0.0 and not isinstance(attn.proj_drop, (nn.Dropout, nn.Identity)):
        raise Unsupported("proj_drop")
    owned = []
    if attn.pool_first:
        if xn.C % heads:
            raise RuntimeError("dim %d not divisible by %d heads" % (xn.C, heads))
        qp = emit_attention_pool(sess, attn._attention_pool_q, xn, heads, label + ".pool_q")
        kp = emit_attention_pool(sess, attn._attention_pool_k, xn, heads, label + ".pool_k")
        vp = emit_attention_pool(sess, attn._attention_pool_v, xn, heads, label + ".pool_v")
        q = emit_linear(sess, attn.q, qp, label=label + ".q")
        k = emit_linear(sess, attn.k, kp, label=label + ".k")
        v = emit_linear(sess, attn.v, vp, label=label + ".v")
        for t in (qp, kp, vp):
            if t is not xn:
                sess.release(t)
        owned += [q, k, v]
    else:
        if qkv is None:
            qkv = emit_linear(sess, _qkv_linear(attn), xn, label=label + ".qkv")
        xn = qkv          # (only its grid / cls flag are used below)
        parts = []
        for i in range(3):
            t = qkv.channel_slice(i * attn.dim_out, attn.dim_out)
            t.thw, t.has_cls = xn.thw, xn.has_cls
            parts.append(t)
        names = (".pool_q", ".pool_k", ".pool_v")
        pools = tuple(emit_norm_before_pool(sess, ap, t, heads, label + n) for ap, t, n in
                      zip((attn._attention_pool_q, attn._attention_pool_k, attn._attention_pool_v), parts, names))
        outs = list(parts)
        big = [i for i in range(3) if _streams_well(pools[i], parts[i])]
        kv = None
        if 1 in big and 2 in big and tuning.get("fuse_kv_pool"):
            xkv = qkv.channel_slice(attn.dim_out, 2 * attn.dim_out)
            xkv.thw, xkv.has_cls = xn.thw, xn.has_cls
            kv = emit_attention_pool_kv(sess, pools[1], pools[2], xkv, heads, label + ".pool_kv")
        if kv is not None:
            outs[1], outs[2] = kv[1], kv[2]
            owned.append(kv[0])
        for i in big:
            if kv is not None and i in (1, 2):
                continue
            outs[i] = emit_attention_pool(sess, pools[i], parts[i], heads, label + names[i])
        fused = emit_attention_pools_fused(sess, pools, parts, heads, label + ".pool_qkv", skip=big)
        if fused is not None:
            for i in range(3):
                if i not in big:
                    outs[i] = fused[i]
        else:
            for i in range(3):
                if i not in big:
                    outs[i] = emit_attention_pool(sess, pools[i], parts[i], heads, label + names[i])
        q, k, v = outs
        owned += [t for t, p in zip((q, k, v), parts) if t is not p and (kv is None or (t is not kv[1] and t is not kv[2]))]
        owned.append(qkv)
    o = emit_attention_core(sess, q, k, v, heads, attn.scale, attn.residual_pool, label=label + ".core")
    q_thw = q.thw
    for t in owned:
        sess.release(t)
    y = emit_linear(sess, attn.proj, o, residual=residual, y_f32=True, label=label + ".proj")
    sess.release(o)
    y.thw = q_thw
    return y


def emit_multiscale_block(sess, blk, x):
    """MultiScaleBlock.forward (attention.py:729-757).  Consumes nothing: the caller releases x."""
    if not isinstance(blk.drop_path, nn.Identity) and _cls_name(blk.drop_path) != "DropPath":
        raise Unsupported("drop_path %s" % _cls_name(blk.drop_path))
    act = E.act_code(blk.mlp.act)
    widen = blk.dim != blk.dim_out
    qkv = xn = None
    if not blk.attn.pool_first and not (blk.dim_mul_in_att and widen):
        lin = _qkv_linear(blk.attn)
        if can_fuse_ln_linear(sess, blk.norm1, lin, x):
            # norm1 has no other consumer: LayerNorm + q|k|v projection in one launch, the bf16 operand tensor is never written
            qkv = emit_ln_linear(sess, blk.norm1, lin, x, label="attn.qkv")
    pre = getattr(x, "prenorm", None)        # norm1(x) already written by the previous block's fused MLP (emit_mlp_fused)
    if pre is not None:
        x.prenorm = None
        if qkv is None and pre[1] is blk.norm1:
            xn = pre[0]
        else:
            sess.release(pre[0])
    if qkv is None and xn is None:
        xn = emit_block_norm(sess, blk.norm1, x, label="norm1")
    skip_src = x
    if blk.dim_mul_in_att and widen:
        skip_src = emit_linear(sess, blk.proj, xn, y_f32=True, label="proj_dim")
    if blk._attention_pool.has_pool:
        k, s, p, mode = _pool_params(blk._attention_pool.pool)
        x_res = E.emit_pool_raw(sess, skip_src, k, s, p, mode, n_prefix=1 if blk.has_cls_embed else 0,
                                grid=skip_src.thw, label="pool_skip")
    else:
        x_res = skip_src
    x1 = emit_multiscale_attention(sess, blk.attn, xn, residual=x_res, qkv=qkv)
    if xn is not None:
        sess.release(xn)
    if x_res is not skip_src:
        sess.release(x_res)
    if skip_src is not x:
        sess.release(skip_src)
    if can_fuse_mlp(sess, blk, x1):
        y = emit_mlp_fused(sess, blk, x1)
        sess.release(x1)
        return y
    xn2 = emit_block_norm(sess, blk.norm2, x1, label="norm2")
    hmid = emit_linear(sess, blk.mlp.fc1, xn2, act=act, label="mlp.fc1")
    if (not blk.dim_mul_in_att) and widen:
        res2 = emit_linear(sess, blk.proj, xn2, y_f32=True, label="proj_dim")
    else:
        res2 = x1
    sess.release(xn2)
    y = emit_linear(sess, blk.mlp.fc2, hmid, residual=res2, y_f32=True, label="mlp.fc2")
    sess.release(hmid)
    if res2 is not x1:
        sess.release(res2)
    sess.release(x1)
    return y


def emit_patch_embed_and_pos(sess, patch_embed, cls_pos, x):
    """PatchEmbed.forward (models/stem.py:289-292) + SpatioTemporalClsPositionalEncoding.forward
    (layers/positional_encoding.py:112-136): the conv writes straight into rows [1, 1+T*H*W) of
    the token buffer; one kernel then writes the cls row and adds the position tables."""
    if _cls_name(patch_embed) != "PatchEmbed":
        raise Unsupported("patch embed %s" % _cls_name(patch_embed))
    conv = patch_embed.patch_model
    if isinstance(conv, nn.Conv2d):     # use_2d_patch: the same conv on a one-frame clip (x is [B,1,H,W,C] here)
        if conv.padding_mode != "zeros" or isinstance(conv.padding, str):
            raise Unsupported("2-D patch embedding padding %s" % (conv.padding,))
        c3 = nn.Conv3d(conv.in_channels, conv.out_channels, (1,) + tuple(conv.kernel_size), (1,) + tuple(conv.stride),
                       (0,) + tuple(conv.padding), (1,) + tuple(conv.dilation), conv.groups, conv.bias is not None)
        c3.weight.data = conv.weight.detach().unsqueeze(2)
        if conv.bias is not None:
            c3.bias.data = conv.bias.detach()
        conv = c3
    if not isinstance(conv, nn.Conv3d):
        raise Unsupported("patch embedding %s" % _cls_name(conv))
    if E.check_conv3d(conv):
        raise Unsupported("depthwise patch embedding")
    T, H, W = cls_pos.patch_embed_shape()
    has_cls = bool(cls_pos.cls_embed_on)
    Cc = conv.out_channels
    tok = sess.alloc_act(x.B, 1, 1, T * H * W + (1 if has_cls else 0), Cc, f32=True)  # residual stream
    tok.thw, tok.has_cls = (T, H, W), has_cls
    grid = (tok.row_offset(1) if has_cls else tok).as_grid(T, H, W)
    grid.C = Cc
    sep = bool(cls_pos.sep_pos_embed)

    def flat(p):
        return sess.add_weight(p.detach().float().reshape(-1, Cc))

    # In the bf16 plan the conv runs on the first-layer kernel, whose fp32 epilogue adds the position tables
    # (the full table minus its cls row when they are not separable); only the cls row is left to pos_encoding.
    in_conv = x.ld == 4 and tuning.get("fuse_posenc")
    pos = None
    if in_conv:
        full = cls_pos.pos_embed_spatial if sep else cls_pos.pos_embed
        if not sep and has_cls:
            full = full.detach().float().reshape(-1, Cc)[1:]
        pos = (flat(full), flat(cls_pos.pos_embed_temporal) if sep else None)
    E.emit_conv(sess, conv, x, None, L.ACT_NONE, out=grid, y_f32=True, label="patch_embed", pos=pos)
    if in_conv and not has_cls:
        return tok
    f = dict(x=tok.ptr, cls_token=flat(cls_pos.cls_token) if has_cls else None,
             pos_spatial=flat(cls_pos.pos_embed_spatial if sep else cls_pos.pos_embed),
             pos_temporal=flat(cls_pos.pos_embed_temporal) if sep else None,
             pos_class=flat(cls_pos.pos_embed_class) if (sep and has_cls) else None,
             B=x.B, T=T, HW=H * W, C=Cc, ld=tok.ld, dtype=L.PV_F32, cls_only=1 if in_conv else 0)
    rows = x.B if in_conv else x.B * tok.voxels
    sess.add_op(L.OP_POSENC, f, label="pos_encoding", alg_bytes=2 * 4 * rows * pad8(Cc))
    return tok


def emit_vit_head(sess, norm_embed, head, x):
    """final LayerNorm + VisionTransformerBasicHead.forward (models/head.py:521-535).  With cls
    pooling only the cls row of each clip is normalised (LayerNorm is row-wise)."""
    if _cls_name(head) != "VisionTransformerBasicHead":
        raise Unsupported("head %s" % _cls_name(head))
    if head.dropout is not None and not isinstance(head.dropout, nn.Dropout):
        raise Unsupported("head dropout")
    sp = head.sequence_pool
    if sp is None or _cls_name(sp) != "SequencePool":
        raise Unsupported("head without sequence pooling")
    if sp.mode == "cls":
        pooled = sess.alloc_act(x.B, 1, 1, 1, x.C)
        if isinstance(norm_embed, nn.Identity):   # norm="batchnorm" models have no final norm: the cls rows as they are
            emit_affine_rows(sess, x, None, None, out=pooled, rows=x.B, ldx=x.bs, label="cls_rows")
        else:
            emit_layernorm(sess, norm_embed, x, out=pooled, rows=x.B, ldx=x.bs, label="norm_embed")
    elif sp.mode == "mean":
        if not isinstance(norm_embed, nn.Identity):
            xn = emit_layernorm(sess, norm_embed, x, label="norm_embed")
        elif x.f32 and sess.pv_dtype != L.PV_F32:      # no final norm: the fp32 stream as a bf16 operand of the mean
            xn = emit_affine_rows(sess, x, None, None, label="stream_rows")
        else:
            xn = x
        pooled32 = sess.alloc_act(x.B, 1, 1, 1, x.C, f32=True)
        f = dict(x=xn.ptr, y=pooled32.ptr, gamma=None, beta=None, rows=x.B * x.voxels, C=x.C, ldx=xn.ld,
                 ldy=pooled32.ld, rows_per_batch=x.voxels, eps=0.0, dtype=sess.pv_dtype, x_f32=0)
        sess.add_op(L.OP_MEAN_ROWS, f, label="head.seq_mean")
        if xn is not x:
            sess.release(xn)
        pooled = pooled32     # fp32 in every plan: the head Linear then runs as an fp32 op (emit_conv)
        if sess.pv_dtype == L.PV_F32:
            pooled.f32 = False
    else:
        raise Unsupported("sequence pool %s" % sp.mode)
    logits = emit_linear(sess, head.proj, pooled, y_f32=True, label="head.proj")
    sess.release(pooled)
    a = head.activation
    if a is not None:
        if isinstance(a, nn.Softmax) and a.dim == 1:
            f = dict(x=logits.ptr, y=logits.ptr, gamma=None, beta=None, rows=logits.B, C=logits.C,
                     ldx=logits.ld, ldy=logits.ld, rows_per_batch=0, eps=0.0, dtype=L.PV_F32)
            sess.add_op(L.OP_SOFTMAX_ROWS, f, label="head.softmax")
        else:
            raise Unsupported("head activation %s" % _cls_name(a))
    return logits


def emit_mvit(sess, model, x):
    """MultiscaleVisionTransformers.forward (models/vision_transformers.py:172-182)."""
    if not isinstance(model.pos_drop, (nn.Identity, nn.Dropout)):
        raise Unsupported("pos_drop")
    cur = emit_patch_embed_and_pos(sess, model.patch_embed, model.cls_positional_encoding, x)
    for blk in model.blocks:
        if _cls_name(blk) != "MultiScaleBlock":
            raise Unsupported("block %s" % _cls_name(blk))
        nxt = emit_multiscale_block(sess, blk, cur)
        sess.release(cur)
        cur = nxt
    out = emit_vit_head(sess, model.norm_embed, model.head, cur)
    sess.release(cur)
    return out
