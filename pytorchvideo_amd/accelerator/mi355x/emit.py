"""Emitters: turn (reference-shaped) nn.Modules into kernel launches of a deploy Session.

Each `emit_*` takes the session, a module in its *original form* (the reference module
tree, weights in fp32) and the input activation reference(s), appends ops and returns the
output reference.  BatchNorm is folded in fp64 on the host into per-channel scale/shift
that ride in the producing kernel's epilogue; activations, residual adds, squeeze-excite
scaling and Swish are fused into conv epilogues / prologues as described in DESIGN.md.

`Unsupported` is raised for anything the kernels do not implement; transmuters probe with
`supports(module)` and *decline* (return None) in that case, leaving the reference module
in place (reference convention: transmuter_mobile_cpu.py:21-22).
"""
import ctypes as C

import torch
import torch.nn as nn

from ... import _lib as L
from . import tuning
from .session import pad8


class Unsupported(Exception):
    pass


# --------------------------------------------------------------------------- helpers
def _cls_name(m):
    """Class name of a module; an Mi355x block answers with the reference class it adopted."""
    return getattr(m, "_orig_cls", type(m)).__name__


def _is_a(m, types):
    """isinstance that sees through an Mi355x block to the class it adopted (a bare nn.MaxPool3d
    between stages -- create_resnet's stage1_pool -- is wrapped like any other block)."""
    return issubclass(getattr(m, "_orig_cls", type(m)), types)


def act_code(act):
    """Map an activation module to the kernel's enum (None/Identity -> NONE)."""
    if act is None or isinstance(act, nn.Identity):
        return L.ACT_NONE
    if isinstance(act, nn.ReLU):
        return L.ACT_RELU
    if _cls_name(act) == "Swish" or isinstance(act, nn.SiLU):
        return L.ACT_SWISH
    if isinstance(act, nn.GELU):
        if getattr(act, "approximate", "none") != "none":
            raise Unsupported("tanh GELU")
        return L.ACT_GELU
    if isinstance(act, nn.Sigmoid):
        return L.ACT_SIGMOID
    raise Unsupported("activation %s" % _cls_name(act))


def fold_norm(norm, channels, conv_bias=None):
    """(scale, shift) fp32 vectors equivalent to `norm(conv + bias)` in eval mode.
    BatchNorm eval: (x-mean)/sqrt(var+eps)*gamma+beta (reference numerics)."""
    scale = torch.ones(channels, dtype=torch.float64)
    shift = torch.zeros(channels, dtype=torch.float64)
    if conv_bias is not None:
        shift = conv_bias.detach().double().cpu().clone()
    if norm is None or isinstance(norm, nn.Identity):
        pass
    elif isinstance(norm, nn.modules.batchnorm._BatchNorm):
        if norm.running_mean is None or norm.running_var is None:
            raise Unsupported("BatchNorm without running stats")
        if norm.num_features != channels:
            raise RuntimeError("BatchNorm has %d features, conv produces %d" % (norm.num_features, channels))
        inv = 1.0 / torch.sqrt(norm.running_var.detach().double().cpu() + norm.eps)
        g = norm.weight.detach().double().cpu() if norm.weight is not None else torch.ones(channels, dtype=torch.float64)
        b = norm.bias.detach().double().cpu() if norm.bias is not None else torch.zeros(channels, dtype=torch.float64)
        s = g * inv
        shift = (shift - norm.running_mean.detach().double().cpu()) * s + b
        scale = s
    else:
        raise Unsupported("norm %s" % _cls_name(norm))
    return scale.float(), shift.float()


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def _conv_out(i, k, s, p, d=1):
    return (i + 2 * p - ((k - 1) * d + 1)) // s + 1


def check_conv3d(conv):
    if not isinstance(conv, nn.Conv3d):
        raise Unsupported("%s is not nn.Conv3d" % _cls_name(conv))
    if conv.padding_mode != "zeros" or isinstance(conv.padding, str):
        raise Unsupported("padding mode")
    depthwise = conv.groups == conv.in_channels == conv.out_channels and conv.groups > 1
    if depthwise and tuple(conv.dilation) != (1, 1, 1):
        raise Unsupported("dilated depthwise conv")
    # channel-wise grouped conv with 2 / 4 / 8 channels per group (create_csn(stage_conv_b_width_per_group=...),
    # reference models/csn.py:34,169) rides in the depthwise kernel (pv_dwconv3d_desc.gw); every other grouping is
    # evaluated as a dense conv with block-diagonal weights (emit_conv) -- slower than a grouped kernel would be,
    # but on the HIP path like the rest of the deployed model
    return depthwise or group_width(conv) > 1


def group_width(conv):
    """Input channels per group if `conv` can run as a channel-wise grouped conv in the depthwise kernel, else 0/1."""
    if not isinstance(conv, nn.Conv3d) or conv.groups <= 1 or conv.in_channels != conv.out_channels:
        return 0
    gw = conv.in_channels // conv.groups
    if gw == 1:
        return 1
    return gw if gw in (2, 4, 8) and tuple(conv.dilation) == (1, 1, 1) else 0


def is_add_fusion(fn):
    """The reference passes `lambda x, y: x + y` / `_trivial_sum`; probe it."""
    if fn is None:
        return False
    try:
        a, b = torch.tensor([1.5, -2.0]), torch.tensor([0.25, 4.0])
        return bool(torch.equal(fn(a, b), a + b))
    except Exception:
        return False


# --------------------------------------------------------------------------- conv family
def _shortcut_geometry(conv_s, x2, out_thw):
    """Strides of a projection shortcut (1x1x1, no padding) if it maps x2 onto the `out_thw` grid, else None."""
    if not isinstance(conv_s, nn.Conv3d) or conv_s.kernel_size != (1, 1, 1) or _triple(conv_s.padding) != (0, 0, 0):
        return None
    if conv_s.groups != 1 or conv_s.bias is not None or tuple(conv_s.dilation) != (1, 1, 1) or conv_s.in_channels != x2.C:
        return None
    st = tuple(int(v) for v in conv_s.stride)
    if tuple((i - 1) // s + 1 for i, s in zip((x2.T, x2.H, x2.W), st)) != tuple(out_thw):
        return None
    return st


def can_fold_shortcut(sess, conv_c, conv_s, x2, cin_c):
    """True when the projection shortcut `conv_s(x2)` can ride in the pointwise conv_c as a second K operand
    (csrc/pv_pwconv.hip); decided by the library from the geometry.  `cin_c`: conv_c's input channels."""
    if not tuning.get("fuse_shortcut") or sess.itemsize != 2 or x2.f32:
        return False
    if not isinstance(conv_c, nn.Conv3d) or not isinstance(conv_s, nn.Conv3d):
        return False
    if conv_c.kernel_size != (1, 1, 1) or conv_c.stride != (1, 1, 1) or _triple(conv_c.padding) != (0, 0, 0) \
            or conv_c.groups != 1 or conv_c.out_channels != conv_s.out_channels:
        return False
    st = tuple(int(v) for v in conv_s.stride)
    out_thw = tuple((i - 1) // s + 1 for i, s in zip((x2.T, x2.H, x2.W), st))
    if _shortcut_geometry(conv_s, x2, out_thw) is None:
        return False
    d = L.Conv3dDesc()
    d.B, d.To, d.Ho, d.Wo = x2.B, out_thw[0], out_thw[1], out_thw[2]
    d.Ti, d.Hi, d.Wi = out_thw
    d.cin, d.cout, d.dtype = pad8(cin_c), conv_c.out_channels, sess.pv_dtype
    d.kt = d.kh = d.kw = d.st = d.sh = d.sw = 1
    d.x2_cin, d.x2_ld, d.x2_st, d.x2_sh, d.x2_sw = pad8(x2.C), x2.ld, st[0], st[1], st[2]
    return L.lib().pv_conv3d_x2_supported(C.byref(d)) == 1


def emit_conv(sess, conv, x, norm=None, act=L.ACT_NONE, residual=None, a_gate=None, a_act=L.ACT_NONE,
              out=None, y_f32=False, label="conv", dwt=None, pos=None, shortcut=None):
    """Dense Conv3d (+BN +bias +residual +act) -> pv_conv3d, or depthwise -> pv_dwconv3d.
    `dwt`: a depthwise temporal Conv3d (k,1,1) applied to the conv's output before norm/act inside the
    same launch (X3D stem; only where can_fuse_temporal_dw said so)."""
    depthwise = check_conv3d(conv)
    if depthwise:
        if residual is not None or a_gate is not None or a_act != L.ACT_NONE or y_f32:
            raise Unsupported("fusion on a depthwise conv")
        return emit_dwconv(sess, conv, x, norm, act, out=out, label=label)
    if conv.in_channels != x.C:
        raise RuntimeError("conv expects %d input channels, got %d" % (conv.in_channels, x.C))
    kt, kh, kw = conv.kernel_size
    st, sh, sw = conv.stride
    pt, ph, pw = _triple(conv.padding)
    dil = tuple(int(v) for v in conv.dilation)
    To, Ho, Wo = _conv_out(x.T, kt, st, pt, dil[0]), _conv_out(x.H, kh, sh, ph, dil[1]), _conv_out(x.W, kw, sw, pw, dil[2])
    if min(To, Ho, Wo) <= 0:
        raise RuntimeError("conv output would be empty")
    cout, cin_p = conv.out_channels, pad8(x.C)
    c4 = x.ld == 4   # 4-channel first-layer layout -> stem kernel (csrc/pv_stem.hip)
    if c4:
        if x.C > 4 or residual is not None or a_gate is not None or a_act != L.ACT_NONE or sess.itemsize != 2:
            raise Unsupported("4-channel input layout is only consumed by a plain first-layer conv")
        cin_p = 4
    elif x.ld < cin_p:
        raise Unsupported("input row narrower than padded channels")
    y = out if out is not None else sess.alloc_act(x.B, To, Ho, Wo, cout, f32=y_f32)
    if (y.B, y.T, y.H, y.W) != (x.B, To, Ho, Wo) or y.C != cout:
        raise RuntimeError("conv output buffer geometry mismatch")
    # pack [cout][taps][cin_p]
    w = conv.weight.detach().float().cpu()  # [cout, cin / groups, kt, kh, kw]
    if conv.groups > 1:   # grouped, not channel-wise: block-diagonal dense weights
        cg_in, cg_out = conv.in_channels // conv.groups, cout // conv.groups
        wd = torch.zeros((cout, conv.in_channels) + tuple(w.shape[2:]), dtype=w.dtype)
        for g in range(conv.groups):
            wd[g * cg_out:(g + 1) * cg_out, g * cg_in:(g + 1) * cg_in] = w[g * cg_out:(g + 1) * cg_out]
        w = wd
    wpair = 0
    if c4 and pad8(cout) <= 8 and dwt is None and not y_f32 and tuning.get("stem_wpair"):
        # <= 8 output channels (SlowFast's fast stem): two W-adjacent outputs per MFMA column; row (j, co) of the
        # 16-row filter tile is co's filter shifted right by j*sw voxels
        wpair, cp8 = 2, pad8(cout)
        wp = torch.zeros(2, cp8, kt, kh, (kw + sw + 1) // 2 * 2, 4, dtype=torch.float32)
        for j in range(2):
            wp[j, :cout, :, :, j * sw: j * sw + kw, : x.C] = w.permute(0, 2, 3, 4, 1)
        wp = wp.reshape(2 * cp8, -1)
    elif c4:   # [cout][kt][kh][round_up(kw,2)][4], zeros in the padding
        wp = torch.zeros(cout, kt, kh, (kw + 1) // 2 * 2, 4, dtype=torch.float32)
        wp[:, :, :, :kw, : x.C] = w.permute(0, 2, 3, 4, 1)
        wp = wp.reshape(cout, -1)
    else:
        wp = torch.zeros(cout, kt * kh * kw, cin_p, dtype=torch.float32)
        wp[:, :, : x.C] = w.permute(0, 2, 3, 4, 1).reshape(cout, kt * kh * kw, x.C)
    # an fp32 operand inside a bf16 plan (e.g. the mean-pooled tokens in front of MViT's head Linear) makes
    # this one op an fp32 op: fp32 weights, fp32 output
    f32_op = bool(x.f32) and sess.pv_dtype != L.PV_F32
    if f32_op and (not y_f32 or c4 or a_gate is not None or (residual is not None and not residual.f32)):
        raise Unsupported("fp32 operand in a bf16 plan needs an fp32 output")
    wp = wp.to(torch.float32 if f32_op else sess.dtype)
    bias = conv.bias if dwt is None else dwt.bias
    scale, shift = fold_norm(norm, cout, bias)
    has_affine = norm is not None and not isinstance(norm, nn.Identity)
    x2 = None
    if shortcut is not None:
        # (conv_s, norm_s, x2): the projection shortcut as a second K operand.  The kernel accumulates the two
        # products separately and joins them with their own fp32 BatchNorm scales; the shifts add up.
        conv_s, norm_s, x2 = shortcut
        st2 = _shortcut_geometry(conv_s, x2, (To, Ho, Wo))
        if st2 is None or c4 or residual is not None or y_f32 or f32_op or (kt, kh, kw, st, sh, sw) != (1,) * 6:
            raise Unsupported("shortcut folding")
        scale2, shift2 = fold_norm(norm_s, cout, None)
        k1, k2 = (x.C + 31) // 32 * 32, (x2.C + 31) // 32 * 32
        wcat = torch.zeros(cout, k1 + k2, dtype=torch.float32)
        wcat[:, : x.C] = w.reshape(cout, x.C)
        wcat[:, k1: k1 + x2.C] = conv_s.weight.detach().float().cpu().reshape(cout, x2.C)
        wp = wcat.to(sess.dtype)
        shift = shift + shift2
        has_affine2 = norm_s is not None and not isinstance(norm_s, nn.Identity)
    f = dict(
        x=x.ptr, w=sess.add_weight(wp), y=y.ptr,
        scale=sess.add_weight(scale) if has_affine else None,
        shift=sess.add_weight(shift) if (has_affine or bias is not None) else None,
        residual=residual.ptr if residual is not None else None,
        a_gate=a_gate,
        x_bs=x.bs, y_bs=y.bs, r_bs=residual.bs if residual is not None else 0,
        ldx=x.ld, ldy=y.ld, ldr=residual.ld if residual is not None else 0,
        B=x.B, Ti=x.T, Hi=x.H, Wi=x.W, cin=cin_p, To=To, Ho=Ho, Wo=Wo, cout=cout,
        kt=kt, kh=kh, kw=kw, st=st, sh=sh, sw=sw, pt=pt, ph=ph, pw=pw,
        act=act, a_act=a_act, dtype=L.PV_F32 if f32_op else sess.pv_dtype, y_f32=1 if (y_f32 and not f32_op) else 0,
        r_f32=1 if (residual is not None and residual.f32 and not f32_op) else 0, c4_wpair=wpair,
        dil_t=dil[0], dil_h=dil[1], dil_w=dil[2],
    )
    if x2 is not None:
        f.update(shift=sess.add_weight(shift), x2_scale=sess.add_weight(scale2) if has_affine2 else None,
                 x2=x2.ptr, x2_bs=x2.bs, x2_ld=x2.ld, x2_cin=pad8(x2.C),
                 x2_Hi=x2.H, x2_Wi=x2.W, x2_st=st2[0], x2_sh=st2[1], x2_sw=st2[2])
    if dil != (1, 1, 1) and (c4 or dwt is not None):
        raise Unsupported("dilated first-layer conv")
    if residual is not None and ((residual.B, residual.T, residual.H, residual.W) != (y.B, y.T, y.H, y.W)
                                 or residual.C != cout):
        raise RuntimeError("residual geometry mismatch")
    if pos is not None:   # (spatial table, temporal table or None): added in the first-layer kernel's fp32 epilogue
        if not c4 or not y_f32 or wpair:
            raise Unsupported("position tables ride only in the first-layer conv's fp32 epilogue")
        f.update(pos_spatial=pos[0], pos_temporal=pos[1])
    if dwt is not None:
        dk = dwt.kernel_size[0]
        taps_t = torch.zeros(dk, pad8(cout), dtype=torch.float32)
        taps_t[:, :cout] = dwt.weight.detach().float().cpu().reshape(cout, dk).t()
        f.update(dwt_w=sess.add_weight(taps_t), dwt_k=dk)
    vox_in, vox_out = x.B * x.voxels, y.B * y.voxels
    taps = kt * kh * kw
    reads = vox_out * cin_p if taps == 1 else vox_in * cin_p  # each input voxel once
    alg = sess.itemsize * (reads + cout * taps * cin_p) \
        + (residual.itemsize * vox_out * pad8(cout) if residual is not None else 0) \
        + (4 if y_f32 else sess.itemsize) * vox_out * pad8(cout)
    flops = 2 * vox_out * cout * taps * x.C
    detail = "|%dx%dx%dx%d c%d->%d k%dx%dx%d s%d%d%d%s" % (x.B, To, Ho, Wo, x.C, cout, kt, kh, kw, st, sh, sw,
                                                          " gate" if a_gate is not None else "")
    if dwt is not None:
        flops += 2 * vox_out * cout * dwt.kernel_size[0]
        detail += "+k%dx1x1" % dwt.kernel_size[0]
    if x2 is not None:
        alg += sess.itemsize * (vox_out * pad8(x2.C) + cout * x2.C)
        flops += 2 * vox_out * cout * x2.C
        detail += " +shortcut c%d" % x2.C
    sess.add_op(L.OP_CONV3D, f, label=label + detail, alg_bytes=alg, flops=flops)
    return y


def _pointwise_producer_fields(sess, producer, x, Cc):
    """Descriptor fields of pv_dwconv3d's fused pointwise producer: `producer` = (conv, norm, act)
    of the 1x1x1 conv whose output (Cc channels) the depthwise conv consumes."""
    conv, norm, act = producer
    if conv.in_channels != x.C:
        raise RuntimeError("conv expects %d input channels, got %d" % (conv.in_channels, x.C))
    cin = x.C
    wp = torch.zeros((Cc + 31) // 32 * 32, (cin + 31) // 32 * 32, dtype=torch.float32)
    wp[:Cc, :cin] = conv.weight.detach().float().cpu().reshape(Cc, cin)
    scale, shift = fold_norm(norm, Cc, conv.bias)
    has_affine = norm is not None and not isinstance(norm, nn.Identity)
    return dict(pw_w=sess.add_weight(wp.to(sess.dtype)),
                pw_scale=sess.add_weight(scale) if has_affine else None,
                pw_shift=sess.add_weight(shift) if (has_affine or conv.bias is not None) else None,
                pw_cin=cin, pw_act=act)


def can_fuse_pointwise_into_dw(sess, conv_a, conv_b, x):
    """True when conv_a (1x1x1) -> conv_b (depthwise 3x3x3) can run as one pv_dwconv3d launch with
    the fused pointwise producer (csrc/pv_pwdw.hip); decided by the library from the geometry."""
    if not tuning.get("fuse_ab") or sess.itemsize != 2 or x.f32:
        return False
    # Every 32-channel slab of the expanded tensor is a workgroup that reads the whole block input, so
    # the fusion pays while the input is narrow (X3D res2/res3: measured 1.3-1.8x over the pair of
    # launches); with >= 96 input channels the slabs' re-reads cost more than the round trip saved.
    if x.C > tuning.get("fuse_ab_max_cin"):
        return False
    if not isinstance(conv_a, nn.Conv3d) or not isinstance(conv_b, nn.Conv3d):
        return False
    if conv_a.kernel_size != (1, 1, 1) or conv_a.stride != (1, 1, 1) or _triple(conv_a.padding) != (0, 0, 0) \
            or conv_a.groups != 1 or conv_a.in_channels != x.C or conv_a.out_channels != conv_b.in_channels:
        return False
    try:
        if check_conv3d(conv_a) or not check_conv3d(conv_b) or group_width(conv_b) != 1:
            return False
    except Unsupported:
        return False
    kt, kh, kw = conv_b.kernel_size
    st, sh, sw = conv_b.stride
    pt, ph, pw = _triple(conv_b.padding)
    d = L.DwConv3dDesc()
    d.ldx, d.ldy = x.ld, pad8(conv_b.out_channels)
    d.B, d.Ti, d.Hi, d.Wi, d.C = x.B, x.T, x.H, x.W, conv_b.out_channels
    d.To, d.Ho, d.Wo = _conv_out(x.T, kt, st, pt), _conv_out(x.H, kh, sh, ph), _conv_out(x.W, kw, sw, pw)
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, kh, kw, st, sh, sw, pt, ph, pw
    d.dtype, d.pw_cin = sess.pv_dtype, x.C
    return min(d.To, d.Ho, d.Wo) > 0 and L.lib().pv_dwconv3d_pw_supported(C.byref(d)) == 1


def _se_parts(se, Cc):
    """(fc1, fc2) of an fvcore-style SqueezeExcitation over Cc channels, or Unsupported."""
    blk = getattr(se, "block", None)
    if blk is None or len(blk) != 4 or not isinstance(blk[0], nn.Conv3d) or not isinstance(blk[2], nn.Conv3d) \
            or not isinstance(blk[1], nn.ReLU) or not isinstance(blk[3], nn.Sigmoid):
        raise Unsupported("squeeze-excitation structure")
    c1, c2 = blk[0], blk[2]
    if c1.in_channels != Cc or c2.out_channels != Cc or c1.kernel_size != (1, 1, 1) or c2.kernel_size != (1, 1, 1):
        raise Unsupported("squeeze-excitation shape")
    return c1, c2


def emit_dwconv(sess, conv, x, norm=None, act=L.ACT_NONE, want_psum=False, out=None, w_mod=0, label="dwconv",
                grid=None, n_prefix=0, producer=None):
    """Depthwise Conv3d (+BN +act [+SE partial sums]) -> pv_dwconv3d.  Returns y or (y, psum, nblk).
    With `grid=(T,H,W)` the input is a token tensor (B, n_prefix + T*H*W, C) convolved on its
    grid, the n_prefix leading rows (cls token) being copied through (attention.py:185-200).
    With `producer=(conv_a, norm_a, act_a)` x is the INPUT of that 1x1x1 conv, which is evaluated
    inside the depthwise kernel (only where can_fuse_pointwise_into_dw said so).
    (Round 4's variant that computed the squeeze-excitation gate in the SAME launch -- the clip's last workgroup -- measured 12 %
    slower than the gate launches it removed and was deleted in round 5.)"""
    if producer is not None:
        if not check_conv3d(conv) or w_mod or grid is not None:
            raise Unsupported("fused producer on a token pooling conv")
        if group_width(conv) != 1:
            raise Unsupported("fused producer on a channel-wise grouped conv")
    elif not w_mod:
        if not check_conv3d(conv):
            raise Unsupported("not depthwise")
        if conv.in_channels != x.C:
            raise RuntimeError("depthwise conv expects %d channels, got %d" % (conv.in_channels, x.C))
    kt, kh, kw = conv.kernel_size
    st, sh, sw = conv.stride
    pt, ph, pw = _triple(conv.padding)
    Ti, Hi, Wi = grid if grid is not None else (x.T, x.H, x.W)
    if grid is not None and x.voxels != n_prefix + Ti * Hi * Wi:
        raise RuntimeError("token count %d does not match grid %s" % (x.voxels, (Ti, Hi, Wi)))
    To, Ho, Wo = _conv_out(Ti, kt, st, pt), _conv_out(Hi, kh, sh, ph), _conv_out(Wi, kw, sw, pw)
    if min(To, Ho, Wo) <= 0:
        raise RuntimeError("conv output would be empty")
    Cc = x.C if producer is None else conv.in_channels
    wc = w_mod if w_mod else Cc
    gw = 0 if (w_mod or producer is not None) else group_width(conv)
    if gw > 1:    # [taps][gw][C]: w[t][j][c] multiplies input channel (c // gw) * gw + j
        w = conv.weight.detach().float().cpu().reshape(Cc, gw, kt * kh * kw)
        wp = torch.zeros(kt * kh * kw, gw, pad8(Cc), dtype=torch.float32)
        wp[:, :, :Cc] = w.permute(2, 1, 0)
    else:
        w = conv.weight.detach().float().cpu().reshape(wc, kt * kh * kw)
        wp = torch.zeros(kt * kh * kw, pad8(wc), dtype=torch.float32)
        wp[:, :wc] = w.t()
    if out is not None:
        y = out
    elif grid is not None:
        y = sess.alloc_act(x.B, 1, 1, n_prefix + To * Ho * Wo, Cc)
        y.thw, y.has_cls = (To, Ho, Wo), n_prefix > 0
    else:
        y = sess.alloc_act(x.B, To, Ho, Wo, Cc)
    has_affine = norm is not None and not isinstance(norm, nn.Identity)
    scale, shift = fold_norm(norm, Cc, conv.bias if not w_mod else None)
    f = dict(
        x=x.ptr, w=sess.add_weight(wp), y=y.ptr,
        scale=sess.add_weight(scale) if has_affine else None,
        shift=sess.add_weight(shift) if (has_affine or conv.bias is not None) else None,
        psum=None, x_bs=x.bs, y_bs=y.bs, ldx=x.ld, ldy=y.ld,
        B=x.B, Ti=Ti, Hi=Hi, Wi=Wi, C=Cc, To=To, Ho=Ho, Wo=Wo,
        kt=kt, kh=kh, kw=kw, st=st, sh=sh, sw=sw, pt=pt, ph=ph, pw=pw,
        w_mod=w_mod, act=act, dtype=sess.pv_dtype, n_prefix=n_prefix, gw=gw if gw > 1 else 0,
    )
    if producer is not None:
        f.update(_pointwise_producer_fields(sess, producer, x, Cc))
    psum = nblk = None
    if want_psum:
        d = L.DwConv3dDesc()
        for k, v in f.items():
            if not hasattr(v, "space") and v is not None:
                setattr(d, k, v)
        nblk = L.lib().pv_dwconv3d_psum_blocks(C.byref(d))
        if nblk <= 0:
            raise Unsupported("depthwise geometry")
        psum = sess.alloc_raw(4 * x.B * nblk * pad8(Cc))
        f["psum"] = psum
    vin, vout = x.B * Ti * Hi * Wi, x.B * To * Ho * Wo
    flops = 2 * vout * Cc * kt * kh * kw * max(gw, 1)
    if producer is None:
        alg = sess.itemsize * (min(vin, vout * kt * kh * kw) + vout) * pad8(Cc)
        detail = "|%dx%dx%dx%d c%d k%dx%dx%d s%d%d%d%s" % (x.B, To, Ho, Wo, Cc, kt, kh, kw, st, sh, sw, " psum" if want_psum else "")
    else:   # the block input once, the depthwise output once; conv_a's flops at the input resolution
        alg = sess.itemsize * (vin * pad8(x.C) + vout * pad8(Cc) + Cc * x.C)
        flops += 2 * vin * x.C * Cc
        detail = "|%dx%dx%dx%d c%d->%d k1x1x1+k%dx%dx%d s%d%d%d%s" % (x.B, To, Ho, Wo, x.C, Cc, kt, kh, kw, st, sh, sw,
                                                                     " psum" if want_psum else "")
    sess.add_op(L.OP_DWCONV3D, f, label=label + detail, alg_bytes=alg, flops=flops)
    if want_psum:
        return y, psum, nblk
    return y


def emit_se_gate(sess, se, psum, nblk, B, Cc, count):
    """fvcore-style SqueezeExcitation -> gate[B][pad8(C)] (the multiply happens in the consumer)."""
    c1, c2 = _se_parts(se, Cc)
    cr = c1.out_channels
    gate = sess.alloc_raw(4 * B * pad8(Cc))
    f = dict(
        psum=psum, gate=gate,
        w1=sess.add_weight(c1.weight.detach().float().reshape(cr, Cc)),
        b1=sess.add_weight(c1.bias.detach().float()) if c1.bias is not None else None,
        w2=sess.add_weight(c2.weight.detach().float().reshape(Cc, cr)),
        b2=sess.add_weight(c2.bias.detach().float()) if c2.bias is not None else None,
        B=B, C=Cc, c_p=pad8(Cc), cr=cr, nblk=nblk, inv_count=1.0 / float(count),
    )
    sess.add_op(L.OP_SE_GATE, f, label="se_gate", alg_bytes=4 * B * nblk * pad8(Cc))
    return gate


def emit_pool(sess, pool, x, n_prefix=0, out=None, label="pool"):
    """nn.MaxPool3d / nn.AvgPool3d / nn.AdaptiveAvgPool3d(1) -> pv_pool3d."""
    if _is_a(pool, nn.AdaptiveAvgPool3d):
        osz = _triple(pool.output_size)
        if any(o not in (1, None) for o in osz):
            raise Unsupported("adaptive pool to %s" % (osz,))
        k = (x.T if osz[0] == 1 else 1, x.H if osz[1] == 1 else 1, x.W if osz[2] == 1 else 1)
        s, p, mode = k, (0, 0, 0), L.POOL_AVG
    elif _is_a(pool, (nn.MaxPool3d, nn.AvgPool3d)):
        k = _triple(pool.kernel_size)
        s = _triple(pool.stride if pool.stride is not None else pool.kernel_size)
        p = _triple(pool.padding)
        if getattr(pool, "ceil_mode", False):
            raise Unsupported("ceil_mode pooling")
        if _is_a(pool, nn.MaxPool3d):
            if _triple(pool.dilation) != (1, 1, 1) or pool.return_indices:
                raise Unsupported("max pool options")
            mode = L.POOL_MAX
        else:
            if not pool.count_include_pad or pool.divisor_override is not None:
                raise Unsupported("avg pool options")
            mode = L.POOL_AVG
    else:
        raise Unsupported("pool %s" % _cls_name(pool))
    k = tuple(int(v) for v in k)
    To, Ho, Wo = (_conv_out(x.T, k[0], s[0], p[0]), _conv_out(x.H, k[1], s[1], p[1]),
                  _conv_out(x.W, k[2], s[2], p[2]))
    if min(To, Ho, Wo) <= 0:
        raise RuntimeError("pool output would be empty")
    return emit_pool_raw(sess, x, k, s, p, mode, n_prefix=n_prefix, out=out, label=label)


def emit_pool_raw(sess, x, k, s, p, mode, n_prefix=0, out=None, label="pool", grid=None):
    """`grid=(T,H,W)`: x is a token tensor (B, n_prefix + T*H*W, C) pooled on its grid."""
    Ti, Hi, Wi = grid if grid is not None else (x.T, x.H, x.W)
    if grid is not None and x.voxels != n_prefix + Ti * Hi * Wi:
        raise RuntimeError("token count %d does not match grid %s" % (x.voxels, (Ti, Hi, Wi)))
    To, Ho, Wo = (_conv_out(Ti, k[0], s[0], p[0]), _conv_out(Hi, k[1], s[1], p[1]),
                  _conv_out(Wi, k[2], s[2], p[2]))
    if min(To, Ho, Wo) <= 0:
        raise RuntimeError("pool output would be empty")
    if out is None:
        if grid is not None:
            y = sess.alloc_act(x.B, 1, 1, To * Ho * Wo + n_prefix, x.C, f32=x.f32)
            y.thw, y.has_cls = (To, Ho, Wo), n_prefix > 0
        else:
            y = sess.alloc_act(x.B, To, Ho, Wo, x.C, f32=x.f32)
    else:
        y = out
        if grid is None and ((y.B, y.T, y.H, y.W) != (x.B, To, Ho, Wo) or y.C != x.C):
            raise RuntimeError("pool output buffer geometry mismatch")
    f = dict(x=x.ptr, y=y.ptr, x_bs=x.bs, y_bs=y.bs, ldx=x.ld, ldy=y.ld,
             B=x.B, Ti=Ti, Hi=Hi, Wi=Wi, C=x.C, To=To, Ho=Ho, Wo=Wo,
             kt=k[0], kh=k[1], kw=k[2], st=s[0], sh=s[1], sw=s[2], pt=p[0], ph=p[1], pw=p[2],
             mode=mode, n_prefix=n_prefix, dtype=L.PV_F32 if x.f32 else sess.pv_dtype)
    if y.f32 != x.f32:
        raise Unsupported("pool between buffers of different precision")
    alg = x.itemsize * pad8(x.C) * x.B * (Ti * Hi * Wi + To * Ho * Wo)
    sess.add_op(L.OP_POOL3D, f, label=label, alg_bytes=alg)
    return y


# --------------------------------------------------------------------------- blocks
def _split_norm_b(norm_b):
    """X3D wraps conv_b's norm as Sequential(BN|Identity, SE|Identity) (x3d.py:199-206)."""
    if isinstance(norm_b, nn.Sequential):
        if len(norm_b) != 2:
            raise Unsupported("norm_b Sequential of length %d" % len(norm_b))
        bn, se = norm_b[0], norm_b[1]
        if isinstance(se, nn.Identity):
            se = None
        elif _cls_name(se) != "SqueezeExcitation":
            raise Unsupported("norm_b[1] is %s" % _cls_name(se))
        return (None if isinstance(bn, nn.Identity) else bn), se
    return norm_b, None


def emit_conv_b(sess, conv_b, x, norm_b, act_b, producer=None):
    """conv_b + norm_b + act_b of a bottleneck.  Returns (y, gate_ptr_or_None, deferred_act):
    with squeeze-excitation the activation is deferred to the consumer's load.  `producer`:
    conv_a fused into a depthwise conv_b (see emit_dwconv); x is then the block input."""
    bn, se = _split_norm_b(norm_b)
    act = act_code(act_b)
    if producer is not None:
        if se is not None:
            if getattr(se, "is_3d", True) is not True:
                raise Unsupported("2-D squeeze-excitation")
            _se_parts(se, conv_b.out_channels)
            y, psum, nblk = emit_dwconv(sess, conv_b, x, bn, L.ACT_NONE, want_psum=True, label="conv_ab.dw+se",
                                        producer=producer)
            gate = emit_se_gate(sess, se, psum, nblk, y.B, y.C, y.voxels)
            sess.release(psum)
            return y, gate, act
        return emit_dwconv(sess, conv_b, x, bn, act, label="conv_ab", producer=producer), None, L.ACT_NONE
    if _cls_name(conv_b) == "Conv2plus1d":
        first, second = (conv_b.conv_xy, conv_b.conv_t) if conv_b.conv_xy_first else (conv_b.conv_t, conv_b.conv_xy)
        if se is not None:
            raise Unsupported("SE after Conv2plus1d")
        mid = emit_conv(sess, first, x, conv_b.norm, act_code(conv_b.activation), label="conv_b.0")
        y = emit_conv(sess, second, mid, bn, act, label="conv_b.1")
        sess.release(mid)
        return y, None, L.ACT_NONE
    depthwise = check_conv3d(conv_b)
    if se is not None:
        if not depthwise:
            raise Unsupported("SE after a dense conv_b")
        if getattr(se, "is_3d", True) is not True:
            raise Unsupported("2-D squeeze-excitation")
        _se_parts(se, x.C)
        y, psum, nblk = emit_dwconv(sess, conv_b, x, bn, L.ACT_NONE, want_psum=True, label="conv_b.dw+se")
        gate = emit_se_gate(sess, se, psum, nblk, x.B, x.C, y.voxels)
        sess.release(psum)
        return y, gate, act
    y = emit_conv(sess, conv_b, x, bn, act, label="conv_b")
    return y, None, L.ACT_NONE


def _conv_bc_desc(sess, bb, a, residual, out):
    """Geometry of conv_b -> conv_c as ONE pv_conv3d (pw2_* fields), or None when the pair is not of that shape."""
    conv_b, conv_c = bb.conv_b, bb.conv_c
    if not isinstance(conv_b, nn.Conv3d) or not isinstance(conv_c, nn.Conv3d) or sess.itemsize != 2 or a.f32:
        return None
    if conv_b.padding_mode != "zeros" or isinstance(conv_b.padding, str) or conv_c.padding_mode != "zeros" \
            or isinstance(conv_c.padding, str):
        return None
    bn, se = _split_norm_b(bb.norm_b)
    if se is not None or conv_b.groups != 1 or tuple(conv_b.dilation) != (1, 1, 1) or conv_b.in_channels != a.C:
        return None
    if conv_c.kernel_size != (1, 1, 1) or tuple(conv_c.stride) != (1, 1, 1) or _triple(conv_c.padding) != (0, 0, 0) \
            or conv_c.groups != 1 or conv_c.in_channels != conv_b.out_channels:
        return None
    if a.ld < pad8(a.C) or (residual is not None and residual.f32):
        return None
    kt, kh, kw = conv_b.kernel_size
    st, sh, sw = conv_b.stride
    pt, ph, pw = _triple(conv_b.padding)
    To, Ho, Wo = _conv_out(a.T, kt, st, pt), _conv_out(a.H, kh, sh, ph), _conv_out(a.W, kw, sw, pw)
    if min(To, Ho, Wo) <= 0 or kt * kh * kw == 1:
        return None
    cout2 = conv_c.out_channels
    if out is not None:
        if (out.B, out.T, out.H, out.W, out.C) != (a.B, To, Ho, Wo, cout2) or out.f32:
            return None
        ldy, y_bs = out.ld, out.bs
    else:
        ldy, y_bs = pad8(cout2), To * Ho * Wo * pad8(cout2)
    if residual is not None and ((residual.B, residual.T, residual.H, residual.W, residual.C) != (a.B, To, Ho, Wo, cout2)
                                 or residual.ld != ldy or residual.bs != y_bs):
        return None
    return dict(x_bs=a.bs, y_bs=y_bs, r_bs=y_bs if residual is not None else 0, ldx=a.ld, ldy=ldy,
                ldr=ldy if residual is not None else 0, B=a.B, Ti=a.T, Hi=a.H, Wi=a.W, cin=pad8(a.C), To=To, Ho=Ho, Wo=Wo,
                cout=conv_b.out_channels, kt=kt, kh=kh, kw=kw, st=st, sh=sh, sw=sw, pt=pt, ph=ph, pw=pw,
                dtype=sess.pv_dtype, pw2_cout=cout2)


def can_fuse_conv_bc(sess, bb, a, residual, out, shortcut):
    """True when conv_b (a narrow dense conv: the tap-streaming kernel's range) and the pointwise conv_c behind it can
    run as one launch (csrc/pv_lateral.hip, PW2 mode): conv_b's output stays in registers as conv_c's MFMA operand.
    Decided by the library from the geometry."""
    if not tuning.get("fuse_bc") or shortcut is not None:
        return False
    g = _conv_bc_desc(sess, bb, a, residual, out)
    if g is None:
        return False
    d = L.Conv3dDesc()
    for k, v in g.items():
        setattr(d, k, v)
    return L.lib().pv_conv3d_pw2_supported(C.byref(d)) == 1


def emit_fused_conv_bc(sess, bb, a, residual, final_act, out=None):
    """conv_b + norm_b + act_b -> conv_c + norm_c + residual + final activation as one pv_conv3d (pw2_* fields).  The
    inner tensor is rounded to bf16 exactly where the unfused pair stores it, so the arithmetic is the pair's."""
    g = _conv_bc_desc(sess, bb, a, residual, out)
    conv_b, conv_c = bb.conv_b, bb.conv_c
    bn, _ = _split_norm_b(bb.norm_b)
    cb, cc, cin_p = conv_b.out_channels, conv_c.out_channels, pad8(a.C)
    taps = g["kt"] * g["kh"] * g["kw"]
    y = out if out is not None else sess.alloc_act(a.B, g["To"], g["Ho"], g["Wo"], cc)
    wb = torch.zeros(cb, taps, cin_p, dtype=torch.float32)
    wb[:, :, : a.C] = conv_b.weight.detach().float().cpu().permute(0, 2, 3, 4, 1).reshape(cb, taps, a.C)
    wc = torch.zeros(cc, pad8(cb), dtype=torch.float32)
    wc[:, :cb] = conv_c.weight.detach().float().cpu().reshape(cc, cb)
    scale_b, shift_b = fold_norm(bn, cb, conv_b.bias)
    scale_c, shift_c = fold_norm(bb.norm_c, cc, conv_c.bias)
    aff_b = bn is not None and not isinstance(bn, nn.Identity)
    aff_c = bb.norm_c is not None and not isinstance(bb.norm_c, nn.Identity)
    f = dict(g)
    f.update(x=a.ptr, w=sess.add_weight(wb.to(sess.dtype)), y=y.ptr,
             scale=sess.add_weight(scale_b) if aff_b else None,
             shift=sess.add_weight(shift_b) if (aff_b or conv_b.bias is not None) else None,
             residual=residual.ptr if residual is not None else None,
             act=act_code(bb.act_b), a_act=L.ACT_NONE,
             pw2_w=sess.add_weight(wc.to(sess.dtype)),
             pw2_scale=sess.add_weight(scale_c) if aff_c else None,
             pw2_shift=sess.add_weight(shift_c) if (aff_c or conv_c.bias is not None) else None,
             pw2_act=final_act)
    vox_in, vox_out = a.B * a.voxels, y.B * y.voxels
    alg = sess.itemsize * (vox_in * cin_p + cb * taps * cin_p + cc * pad8(cb) + vox_out * pad8(cc)
                           + (vox_out * pad8(cc) if residual is not None else 0))
    flops = 2 * vox_out * (cb * taps * a.C + cc * cb)
    sess.add_op(L.OP_CONV3D, f, label="conv_bc|%dx%dx%dx%d c%d->%d->%d k%dx%dx%d s%d%d%d+k1x1x1" % (
        a.B, g["To"], g["Ho"], g["Wo"], a.C, cb, cc, g["kt"], g["kh"], g["kw"], g["st"], g["sh"], g["sw"]),
        alg_bytes=alg, flops=flops)
    return y


def emit_bottleneck(sess, bb, x, residual=None, final_act=L.ACT_NONE, out=None, shortcut=None):
    """BottleneckBlock.forward (resnet.py:1345-1365) with the block's residual join and
    final activation fused into conv_c's epilogue."""
    for name in ("conv_a", "conv_b", "conv_c"):
        if getattr(bb, name, None) is None:
            raise Unsupported("bottleneck without %s" % name)
    if can_fuse_bottleneck_ab(sess, bb, x):
        # conv_a -> conv_b -> squeeze sums on the matrix + stencil waves of csrc/pv_block.hip (14 x 14 maps: X3D res4)
        b, gate, deferred = emit_fused_conv_ab_se(sess, bb, x)
    elif can_fuse_pointwise_into_dw(sess, bb.conv_a, bb.conv_b, x):
        # conv_a -> conv_b in one pass: the expanded tensor is never written (csrc/pv_pwdw.hip)
        b, gate, deferred = emit_conv_b(sess, bb.conv_b, x, bb.norm_b, bb.act_b,
                                        producer=(bb.conv_a, bb.norm_a, act_code(bb.act_a)))
    else:
        a = emit_conv(sess, bb.conv_a, x, bb.norm_a, act_code(bb.act_a), label="conv_a")
        if can_fuse_conv_bc(sess, bb, a, residual, out, shortcut):
            # conv_b -> conv_c in one pass: the inner tensor never leaves the registers (csrc/pv_lateral.hip, PW2 mode)
            c = emit_fused_conv_bc(sess, bb, a, residual, final_act, out=out)
            sess.release(a)
            return c
        b, gate, deferred = emit_conv_b(sess, bb.conv_b, a, bb.norm_b, bb.act_b)
        sess.release(a)
    check_conv3d(bb.conv_c)
    if (gate is not None or deferred != L.ACT_NONE) and \
            (bb.conv_c.kernel_size != (1, 1, 1) or bb.conv_c.stride != (1, 1, 1) or _triple(bb.conv_c.padding) != (0, 0, 0)):
        raise Unsupported("SE block whose conv_c is not pointwise")
    c = emit_conv(sess, bb.conv_c, b, bb.norm_c, final_act, residual=residual, a_gate=gate, a_act=deferred,
                  out=out, label="conv_c", shortcut=shortcut)
    sess.release(b)
    if gate is not None:
        sess.release(gate)
    return c


def pack_bottleneck_operands(conv_a, norm_a, conv_b, bn_b, conv_c, norm_c):
    """Host-packed operands of pv_bottleneck (layout: include/pv_mi355x.h, pv_bottleneck_desc): dict of CPU tensors
    wa / wc (bf16 MFMA A-fragment images), wb (fp32 taps), sa ha sb hb sc hc (folded BatchNorms, fp32)."""
    cin, Cc = conv_a.in_channels, conv_a.out_channels
    cout = conv_c.out_channels if conv_c is not None else 16
    Cp, cin_p = (Cc + 31) // 32 * 32, (cin + 31) // 32 * 32

    def frag_image(w, rows_p, cols_p):      # [rows_p/16][cols_p/32][lane = 16 q + m][j] = w[16 mt + m][32 ks + 8 q + j]
        wp = torch.zeros(rows_p, cols_p, dtype=torch.float32)
        wp[:w.shape[0], :w.shape[1]] = w
        return wp.reshape(rows_p // 16, 16, cols_p // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16).reshape(-1)

    wa = frag_image(conv_a.weight.detach().float().cpu().reshape(Cc, cin), Cp, cin_p)
    wc = frag_image(conv_c.weight.detach().float().cpu().reshape(cout, Cc), (cout + 15) // 16 * 16, Cp) if conv_c is not None else None
    wb = torch.zeros(27, Cp, dtype=torch.float32)
    wb[:, :Cc] = conv_b.weight.detach().float().cpu().reshape(Cc, 27).t()

    def padded(v, n):
        o = torch.zeros(n, dtype=torch.float32)
        o[:v.numel()] = v
        return o

    sa, ha = fold_norm(norm_a, Cc, conv_a.bias)
    sb, hb = fold_norm(bn_b, Cc, conv_b.bias)
    out = dict(wa=wa, wb=wb.reshape(-1), sa=padded(sa, Cp), ha=padded(ha, Cp), sb=padded(sb, Cp), hb=padded(hb, Cp))
    if conv_c is not None:
        sc, hc = fold_norm(norm_c, cout, conv_c.bias)
        out.update(wc=wc, sc=sc, hc=hc)
    return out


def _bottleneck_geometry_ok(sess, bb, x, want_se):
    """Shared structure test of the fused-bottleneck launches: conv_a 1x1x1, depthwise 3x3x3 conv_b with unit stride and padding 1,
    conv_c 1x1x1, with (want_se) or without squeeze-excitation behind norm_b.  Returns the (act_a, act_b) codes or None."""
    if not tuning.get("fuse_block") or sess.itemsize != 2 or x.f32:
        return None
    ca, cb, cc = getattr(bb, "conv_a", None), getattr(bb, "conv_b", None), getattr(bb, "conv_c", None)
    if not all(isinstance(c, nn.Conv3d) for c in (ca, cb, cc)):
        return None
    try:
        bn_b, se = _split_norm_b(bb.norm_b)
        if (se is not None) != want_se or check_conv3d(ca) or not check_conv3d(cb) or check_conv3d(cc) or group_width(cb) != 1:
            return None
        if se is not None:
            if getattr(se, "is_3d", True) is not True:
                return None
            _se_parts(se, cb.out_channels)
        acts = (act_code(bb.act_a), act_code(bb.act_b))
    except Unsupported:
        return None
    for c in (ca, cc):
        if c.kernel_size != (1, 1, 1) or c.stride != (1, 1, 1) or _triple(c.padding) != (0, 0, 0) or c.groups != 1:
            return None
    if cb.kernel_size != (3, 3, 3) or cb.stride != (1, 1, 1) or _triple(cb.padding) != (1, 1, 1) or _triple(cb.dilation) != (1, 1, 1):
        return None
    if ca.in_channels != x.C or ca.out_channels != cb.in_channels or cb.out_channels != cc.in_channels:
        return None
    if x.bs != x.voxels * x.ld:
        return None
    return acts


def can_fuse_bottleneck(sess, bb, x, residual, final_act=L.ACT_RELU):
    """A whole residual block without squeeze-excitation as ONE pv_bottleneck launch (csrc/pv_block.hip)?  Identity shortcut;
    the library decides the geometry."""
    if residual is not x:
        return False
    acts = _bottleneck_geometry_ok(sess, bb, x, want_se=False)
    if acts is None or bb.conv_c.out_channels != x.C:
        return False
    d = L.BottleneckDesc()
    d.B, d.T, d.H, d.W, d.cin, d.C, d.cout = x.B, x.T, x.H, x.W, x.C, bb.conv_a.out_channels, bb.conv_c.out_channels
    d.ldx, d.ldy, d.ldr, d.dtype, d.mode = x.ld, pad8(bb.conv_c.out_channels), x.ld, sess.pv_dtype, L.BLOCK_FULL
    d.act_a, d.act_b, d.act_out = acts[0], acts[1], final_act
    return L.lib().pv_bottleneck_supported(C.byref(d)) == 1


def can_fuse_bottleneck_ab(sess, bb, x):
    """conv_a + conv_b + the squeeze sums of a block WITH squeeze-excitation as one pv_bottleneck launch (mode PV_BLOCK_AB)?"""
    acts = _bottleneck_geometry_ok(sess, bb, x, want_se=True)
    if acts is None:
        return False
    d = L.BottleneckDesc()
    d.B, d.T, d.H, d.W, d.cin, d.C, d.cout = x.B, x.T, x.H, x.W, x.C, bb.conv_a.out_channels, bb.conv_c.out_channels
    d.ldx, d.ldy, d.dtype, d.mode, d.act_a = x.ld, pad8(bb.conv_a.out_channels), sess.pv_dtype, L.BLOCK_AB, acts[0]
    return L.lib().pv_bottleneck_supported(C.byref(d)) == 1


def emit_fused_conv_ab_se(sess, bb, x):
    """conv_a + norm_a + act_a + depthwise conv_b + BatchNorm of norm_b + the squeeze of its SqueezeExcitation in one launch; then the
    gate (pv_se_gate).  Returns (y, gate, deferred activation) like emit_conv_b."""
    bn_b, se = _split_norm_b(bb.norm_b)
    ops = pack_bottleneck_operands(bb.conv_a, bb.norm_a, bb.conv_b, bn_b, None, None)
    cin, Cc = bb.conv_a.in_channels, bb.conv_a.out_channels
    y = sess.alloc_act(x.B, x.T, x.H, x.W, Cc)
    f = dict(x=x.ptr, y=y.ptr, residual=None, wc=None, sc=None, hc=None, x_bs=x.bs, y_bs=y.bs, r_bs=0, ldx=x.ld, ldy=y.ld, ldr=0,
             B=x.B, T=x.T, H=x.H, W=x.W, cin=cin, C=Cc, cout=bb.conv_c.out_channels,
             act_a=act_code(bb.act_a), act_b=L.ACT_NONE, act_out=L.ACT_NONE, dtype=sess.pv_dtype, mode=L.BLOCK_AB)
    for k in ("wa", "wb", "sa", "ha", "sb", "hb"):
        f[k] = sess.add_weight(ops[k])
    d = L.BottleneckDesc()
    d.H, d.W, d.cin, d.C, d.cout, d.ldx, d.mode, d.dtype = x.H, x.W, cin, Cc, bb.conv_c.out_channels, x.ld, L.BLOCK_AB, sess.pv_dtype
    nblk = L.lib().pv_bottleneck_psum_blocks(C.byref(d))
    if nblk <= 0:
        raise Unsupported("bottleneck geometry")
    psum = sess.alloc_raw(4 * x.B * nblk * pad8(Cc))
    f["psum"] = psum
    vox = x.B * x.T * x.H * x.W
    sess.add_op(L.OP_BOTTLENECK, f, label="conv_ab.fused+se|%dx%dx%dx%d c%d->%d k1x1x1+k3x3x3 psum" % (x.B, x.T, x.H, x.W, cin, Cc),
                alg_bytes=sess.itemsize * vox * (pad8(cin) + pad8(Cc)) + 2 * Cc * cin + 108 * Cc, flops=2 * vox * Cc * (cin + 27))
    gate = emit_se_gate(sess, se, psum, nblk, x.B, Cc, x.T * x.H * x.W)
    sess.release(psum)
    return y, gate, act_code(bb.act_b)


def emit_fused_bottleneck(sess, bb, x, final_act, out=None):
    """ResBlock.forward (resnet.py:1179-1189) with an identity shortcut around BottleneckBlock.forward (resnet.py:1345-1365,
    built by x3d.py:169-212) as one launch."""
    bn_b, _ = _split_norm_b(bb.norm_b)
    ops = pack_bottleneck_operands(bb.conv_a, bb.norm_a, bb.conv_b, bn_b, bb.conv_c, bb.norm_c)
    cin, Cc, cout = bb.conv_a.in_channels, bb.conv_a.out_channels, bb.conv_c.out_channels
    y = out if out is not None else sess.alloc_act(x.B, x.T, x.H, x.W, cout)
    f = dict(x=x.ptr, y=y.ptr, residual=x.ptr, x_bs=x.bs, y_bs=y.bs, r_bs=x.bs, ldx=x.ld, ldy=y.ld, ldr=x.ld,
             B=x.B, T=x.T, H=x.H, W=x.W, cin=cin, C=Cc, cout=cout,
             act_a=act_code(bb.act_a), act_b=act_code(bb.act_b), act_out=final_act, dtype=sess.pv_dtype, mode=L.BLOCK_FULL, psum=None)
    for k in ("wa", "wb", "wc", "sa", "ha", "sb", "hb", "sc", "hc"):
        f[k] = sess.add_weight(ops[k])
    vox = x.B * x.T * x.H * x.W
    sess.add_op(L.OP_BOTTLENECK, f, label="block.fused|%dx%dx%dx%d c%d->%d->%d k1x1x1+k3x3x3+k1x1x1" % (x.B, x.T, x.H, x.W, cin, Cc, cout),
                alg_bytes=sess.itemsize * vox * (pad8(cin) + pad8(cout)) + 2 * Cc * (cin + cout) + 108 * Cc,
                flops=2 * vox * Cc * (cin + cout + 27))
    return y


def emit_res_block(sess, rb, x, out=None):
    """ResBlock.forward (resnet.py:1179-1189): act(shortcut + branch2(x))."""
    if not is_add_fusion(rb.branch_fusion):
        raise Unsupported("branch_fusion is not a sum")
    if _cls_name(rb.branch2) != "BottleneckBlock":
        raise Unsupported("branch2 is %s" % _cls_name(rb.branch2))
    bb = rb.branch2
    if rb.branch1_conv is not None and isinstance(getattr(bb, "conv_c", None), nn.Conv3d) \
            and can_fold_shortcut(sess, bb.conv_c, rb.branch1_conv, x, bb.conv_c.in_channels):
        # the projection shortcut rides in conv_c as a second K operand: no launch, no round trip of its output
        return emit_bottleneck(sess, bb, x, residual=None, final_act=act_code(rb.activation), out=out,
                               shortcut=(rb.branch1_conv, rb.branch1_norm, x))
    if rb.branch1_conv is None and can_fuse_bottleneck(sess, bb, x, x, act_code(rb.activation)):
        return emit_fused_bottleneck(sess, bb, x, act_code(rb.activation), out=out)
    if rb.branch1_conv is None:
        shortcut = x
    else:
        shortcut = emit_conv(sess, rb.branch1_conv, x, rb.branch1_norm, L.ACT_NONE, label="shortcut")
    y = emit_bottleneck(sess, rb.branch2, x, residual=shortcut, final_act=act_code(rb.activation), out=out)
    if shortcut is not x:
        sess.release(shortcut)
    return y


def emit_res_stage(sess, stage, x, out=None):
    """ResStage.forward; `out` (optional) receives the last block's output, e.g. a channel
    slice of a wider buffer so that a later torch.cat costs nothing."""
    cur = x
    n = len(stage.res_blocks)
    for i, blk in enumerate(stage.res_blocks):
        if _cls_name(blk) != "ResBlock":
            raise Unsupported("stage element %s" % _cls_name(blk))
        nxt = emit_res_block(sess, blk, cur, out=out if i == n - 1 else None)
        if cur is not x:
            sess.release(cur)
        cur = nxt
    return cur


def can_fuse_temporal_dw(sess, first, second, mid_norm, mid_act, x, act):
    """True when `first` (dense 1 x kh x kw conv on the 4-channel first-layer layout) and `second`
    (depthwise k x 1 x 1 temporal conv) with nothing in between can run as one pv_conv3d launch
    (csrc/pv_stem.hip, X3D stem); decided by the library from the geometry."""
    if not tuning.get("fuse_stem") or x.ld != 4 or sess.itemsize != 2:
        return False
    if mid_norm is not None and not isinstance(mid_norm, nn.Identity):
        return False
    if mid_act != L.ACT_NONE or not isinstance(first, nn.Conv3d) or not isinstance(second, nn.Conv3d):
        return False
    try:
        if check_conv3d(first) or not check_conv3d(second):
            return False
    except Unsupported:
        return False
    if group_width(second) != 1:      # a channel-wise GROUPED temporal conv passes check_conv3d too: not this fusion
        return False
    k = second.kernel_size
    if first.bias is not None or first.out_channels != second.in_channels or second.in_channels != second.out_channels:
        return False
    if k[1:] != (1, 1) or second.stride != (1, 1, 1) or _triple(second.padding) != (k[0] // 2, 0, 0) or k[0] % 2 == 0:
        return False
    kt, kh, kw = first.kernel_size
    st, sh, sw = first.stride
    pt, ph, pw = _triple(first.padding)
    d = L.Conv3dDesc()
    d.ldx, d.cin, d.cout, d.dtype, d.act, d.dwt_k = 4, 4, first.out_channels, sess.pv_dtype, act, k[0]
    d.B, d.Ti, d.Hi, d.Wi = x.B, x.T, x.H, x.W
    d.To, d.Ho, d.Wo = _conv_out(x.T, kt, st, pt), _conv_out(x.H, kh, sh, ph), _conv_out(x.W, kw, sw, pw)
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, kh, kw, st, sh, sw, pt, ph, pw
    return min(d.To, d.Ho, d.Wo) > 0 and L.lib().pv_conv3d_dwt_supported(C.byref(d)) == 1


def emit_stem(sess, stem, x, out=None):
    """ResNetBasicStem.forward (stem.py:252-260); conv may be Conv3d or X3D's Conv2plus1d."""
    act = act_code(stem.activation)
    conv = stem.conv
    if _cls_name(conv) == "Conv2plus1d":
        first, second = (conv.conv_xy, conv.conv_t) if conv.conv_xy_first else (conv.conv_t, conv.conv_xy)
        if can_fuse_temporal_dw(sess, first, second, conv.norm, act_code(conv.activation), x, act):
            y = emit_conv(sess, first, x, stem.norm, act, out=out if stem.pool is None else None,
                          label="stem.conv01", dwt=second)
        else:
            mid = emit_conv(sess, first, x, conv.norm, act_code(conv.activation), label="stem.conv0")
            y = emit_conv(sess, second, mid, stem.norm, act, out=out if stem.pool is None else None,
                          label="stem.conv1")
            sess.release(mid)
    else:
        y = emit_conv(sess, conv, x, stem.norm, act, out=out if stem.pool is None else None, label="stem.conv")
    if stem.pool is not None:
        p = emit_pool(sess, stem.pool, y, out=out, label="stem.pool")
        sess.release(y)
        y = p
    return y


def _emit_linear_rows(sess, lin, x, act=L.ACT_NONE, y_f32=False, label="linear"):
    """nn.Linear on the channel dim of a channels-last activation = pointwise conv."""
    if not isinstance(lin, nn.Linear):
        raise Unsupported("%s is not nn.Linear" % _cls_name(lin))
    if lin.in_features != x.C:
        raise RuntimeError("Linear expects %d features, got %d" % (lin.in_features, x.C))
    conv = nn.Conv3d(lin.in_features, lin.out_features, 1, bias=lin.bias is not None)
    conv.weight.data = lin.weight.detach().reshape(lin.out_features, lin.in_features, 1, 1, 1)
    if lin.bias is not None:
        conv.bias.data = lin.bias.detach()
    return emit_conv(sess, conv, x, None, act, y_f32=y_f32, label=label)


def emit_projected_pool(sess, pp, x):
    """ProjectedPool.forward (x3d.py:791-806)."""
    a = emit_conv(sess, pp.pre_conv, x, pp.pre_norm, act_code(pp.pre_act), label="head.pre_conv")
    p = emit_pool(sess, pp.pool, a, label="head.pool")
    sess.release(a)
    y = emit_conv(sess, pp.post_conv, p, pp.post_norm, act_code(pp.post_act), label="head.post_conv")
    sess.release(p)
    return y


def emit_res_head(sess, head, x):
    """ResNetBasicHead.forward (head.py:371-391).  Returns an fp32 [B, classes] row tensor
    when the head ends in the global average, else the (B,T,H,W,classes) activation."""
    cur = x
    if head.pool is not None:
        if _cls_name(head.pool) == "ProjectedPool":
            cur = emit_projected_pool(sess, head.pool, x)
        else:
            cur = emit_pool(sess, head.pool, x, label="head.pool")
    # dropout is the identity in eval
    if head.dropout is not None and not isinstance(head.dropout, nn.Dropout):
        raise Unsupported("head dropout %s" % _cls_name(head.dropout))
    a = head.activation
    if a is not None and not isinstance(a, (nn.Softmax, nn.Sigmoid, nn.ReLU)):
        raise Unsupported("head activation %s" % _cls_name(a))
    # element-wise activations (Sigmoid: the multi-label heads, ReLU) ride in the projection's epilogue
    act = L.ACT_SIGMOID if isinstance(a, nn.Sigmoid) else (L.ACT_RELU if isinstance(a, nn.ReLU) else L.ACT_NONE)
    logits = _emit_linear_rows(sess, head.proj, cur, act=act, y_f32=True, label="head.proj")
    if cur is not x:
        sess.release(cur)
    if isinstance(a, nn.Softmax):
        if a.dim != 1:
            raise Unsupported("softmax over dim %s" % a.dim)
        f = dict(x=logits.ptr, y=logits.ptr, gamma=None, beta=None, rows=logits.B * logits.voxels, C=logits.C,
                 ldx=logits.ld, ldy=logits.ld, rows_per_batch=0, eps=0.0, dtype=L.PV_F32)
        sess.add_op(L.OP_SOFTMAX_ROWS, f, label="head.softmax")
    if head.output_pool is None:
        return logits
    if not isinstance(head.output_pool, nn.AdaptiveAvgPool3d) or _triple(head.output_pool.output_size) != (1, 1, 1):
        raise Unsupported("head output pool")
    out = sess.alloc_act(logits.B, 1, 1, 1, logits.C, f32=True)
    f = dict(x=logits.ptr, y=out.ptr, gamma=None, beta=None, rows=logits.B * logits.voxels, C=logits.C,
             ldx=logits.ld, ldy=out.ld, rows_per_batch=logits.voxels, eps=0.0, dtype=L.PV_F32)
    sess.add_op(L.OP_MEAN_ROWS, f, label="head.mean")
    sess.release(logits)
    return out


# --------------------------------------------------------------------------- detection head
def _pool2d_geometry(pool):
    """(kernel, stride, padding, mode) of the nn.MaxPool2d / nn.AvgPool2d that follows the RoI layer."""
    def pair(v):
        return (int(v), int(v)) if isinstance(v, int) else tuple(int(e) for e in v)
    if getattr(pool, "ceil_mode", False):
        raise Unsupported("ceil_mode pooling")
    k = pair(pool.kernel_size)
    s = pair(pool.stride if pool.stride is not None else pool.kernel_size)
    p = pair(pool.padding)
    if isinstance(pool, nn.MaxPool2d):
        if pair(pool.dilation) != (1, 1) or pool.return_indices:
            raise Unsupported("max pool options")
        return k, s, p, L.POOL_MAX
    if isinstance(pool, nn.AvgPool2d):
        if not pool.count_include_pad or pool.divisor_override is not None:
            raise Unsupported("avg pool options")
        return k, s, p, L.POOL_AVG
    raise Unsupported("spatial pool %s" % _cls_name(pool))


def check_roi_head(head):
    """Structural support check of a ResNetRoIHead (models/head.py:394-482); raises Unsupported."""
    roi = head.roi_layer
    if roi is None or _cls_name(roi) != "RoIAlign":
        raise Unsupported("roi layer %s" % (None if roi is None else _cls_name(roi)))
    for a in ("output_size", "spatial_scale", "sampling_ratio"):
        if not hasattr(roi, a):
            raise Unsupported("RoIAlign without %s" % a)
    if head.pool is not None and not _is_a(head.pool, (nn.AvgPool3d, nn.MaxPool3d, nn.AdaptiveAvgPool3d)):
        raise Unsupported("head pool %s" % _cls_name(head.pool))
    if head.pool_spatial is not None:
        _pool2d_geometry(head.pool_spatial)
    if not isinstance(head.proj, nn.Linear):
        raise Unsupported("head proj %s" % _cls_name(head.proj))
    if head.dropout is not None and not isinstance(head.dropout, nn.Dropout):
        raise Unsupported("head dropout %s" % _cls_name(head.dropout))
    a = head.activation
    if a is not None and not isinstance(a, (nn.Sigmoid, nn.ReLU)) and not (isinstance(a, nn.Softmax) and a.dim == 1):
        raise Unsupported("head activation %s" % _cls_name(a))
    if head.output_pool is not None and not (isinstance(head.output_pool, nn.AdaptiveAvgPool3d)
                                            and _triple(head.output_pool.output_size) == (1, 1, 1)):
        raise Unsupported("head output pool")


def emit_roi_head(sess, head, x, boxes, n_boxes):
    """ResNetRoIHead.forward (models/head.py:437-482): pool -> squeeze T -> RoIAlign -> pool_spatial ->
    dropout (identity) -> proj -> activation [-> global mean].  `boxes` is the arena pointer of the
    [n_boxes, 5] fp32 box list.  RoIAlign and a MaxPool2d over its whole output (what
    create_res_roi_pooling_head builds, head.py:317) are one launch whose [R,C,ph,pw] tensor is never
    stored; the sigmoid rides in the projection's epilogue.  Returns fp32 [R,1,h,w,classes] (or the
    [R,classes] mean when the head averages)."""
    check_roi_head(head)
    cur = x
    if head.pool is not None:
        cur = emit_pool(sess, head.pool, x, label="det.pool")
    if cur.T != 1:
        raise Exception("Temporal dimension should be 1. Consider modifying the pool layer.")  # head.py:452-455
    roi = head.roi_layer
    osz = roi.output_size
    ph, pw = (int(osz), int(osz)) if isinstance(osz, int) else (int(osz[0]), int(osz[1]))
    fuse_max, tail = False, None
    if head.pool_spatial is not None:
        k, s, p, mode = _pool2d_geometry(head.pool_spatial)
        if mode == L.POOL_MAX and k == (ph, pw) and p == (0, 0):
            fuse_max = True          # one window covers the whole RoI output: stride is irrelevant
        else:
            tail = (k, s, p, mode)
    y = sess.alloc_act(n_boxes, 1, 1 if fuse_max else ph, 1 if fuse_max else pw, cur.C, f32=cur.f32)
    f = dict(x=cur.ptr, boxes=boxes, y=y.ptr, x_bs=cur.bs, ldx=cur.ld, ldy=y.ld, B=cur.B, H=cur.H, W=cur.W,
             C=cur.C, R=n_boxes, ph=ph, pw=pw, sampling_ratio=int(roi.sampling_ratio),
             aligned=1 if getattr(roi, "aligned", False) else 0, pool_max=1 if fuse_max else 0,
             spatial_scale=float(roi.spatial_scale), dtype=L.PV_F32 if cur.f32 else sess.pv_dtype)
    alg = cur.itemsize * pad8(cur.C) * (cur.B * cur.H * cur.W + n_boxes * y.voxels)
    sess.add_op(L.OP_ROI_ALIGN, f, label="det.roi_align", alg_bytes=alg)
    if cur is not x:
        sess.release(cur)
    if tail is not None:
        k, s, p, mode = tail
        z = emit_pool_raw(sess, y, (1,) + k, (1,) + s, (0,) + p, mode, label="det.pool_spatial")
        sess.release(y)
        y = z
    a = head.activation
    act = L.ACT_SIGMOID if isinstance(a, nn.Sigmoid) else (L.ACT_RELU if isinstance(a, nn.ReLU) else L.ACT_NONE)
    logits = _emit_linear_rows(sess, head.proj, y, act=act, y_f32=True, label="det.proj")
    sess.release(y)
    if isinstance(a, nn.Softmax):
        f = dict(x=logits.ptr, y=logits.ptr, gamma=None, beta=None, rows=logits.B * logits.voxels, C=logits.C,
                 ldx=logits.ld, ldy=logits.ld, rows_per_batch=0, eps=0.0, dtype=L.PV_F32)
        sess.add_op(L.OP_SOFTMAX_ROWS, f, label="det.softmax")
    if head.output_pool is None:
        return logits
    out = sess.alloc_act(logits.B, 1, 1, 1, logits.C, f32=True)
    f = dict(x=logits.ptr, y=out.ptr, gamma=None, beta=None, rows=logits.B * logits.voxels, C=logits.C,
             ldx=logits.ld, ldy=out.ld, rows_per_batch=logits.voxels, eps=0.0, dtype=L.PV_F32)
    sess.add_op(L.OP_MEAN_ROWS, f, label="det.mean")
    sess.release(logits)
    return out


# --------------------------------------------------------------------------- SlowFast
def out_shape(m, x):
    """(T,H,W,C) a single-IO module will produce for input `x`, by emitting into a scratch session."""
    from .session import Session
    scratch = Session(dtype=torch.bfloat16)
    xin = scratch.alloc_act(x.B, x.T, x.H, x.W, x.C)
    y = emit_module(scratch, m, xin)
    return y.T, y.H, y.W, y.C


def emit_multipathway(sess, mp, xs):
    """MultiPathWayWithFuse.forward (models/net.py:107-122) with FuseFastToSlow
    (models/slowfast.py:720-729): the lateral conv (7x1x1, temporal stride 4) + BN + ReLU
    writes straight into the channel slice [C_slow, C_slow + 2*beta*C) of a slow-pathway
    buffer that was allocated wide enough, and the slow pathway's own last kernel writes
    the slice [0, C_slow) -- `torch.cat` (a full copy of the slow tensor in the reference)
    costs nothing.  Returns ([outputs], [pre-fusion outputs])."""
    blocks = list(mp.multipathway_blocks)
    if len(xs) != len(blocks):
        raise RuntimeError("pathway count mismatch")
    fusion = mp.multipathway_fusion
    fuse_kind = None
    if fusion is None or isinstance(fusion, nn.Identity):
        fuse_kind = "none"
    elif _cls_name(fusion) == "FuseFastToSlow":
        if len(blocks) != 2 or blocks[0] is None or blocks[1] is None:
            raise Unsupported("FuseFastToSlow needs two live pathways")
        fuse_kind = "fast_to_slow"
    else:
        raise Unsupported("fusion %s" % _cls_name(fusion))

    if fuse_kind == "none":
        outs = [emit_module(sess, b, x) if b is not None else x for b, x in zip(blocks, xs)]
        return outs, outs

    fast = emit_module(sess, blocks[1], xs[1])
    conv = fusion.conv_fast_to_slow
    check_conv3d(conv)
    t_s, h_s, w_s, c_s = out_shape(blocks[0], xs[0])
    c_f = conv.out_channels
    if c_s % 8:
        raise Unsupported("slow width %d not a multiple of 8 (channel-slice concat)" % c_s)
    wide = sess.alloc_act(xs[0].B, t_s, h_s, w_s, c_s + c_f)
    slow_slice = wide.channel_slice(0, c_s)
    emit_module_out(sess, blocks[0], xs[0], slow_slice)
    emit_lateral(sess, conv, fast, fusion.norm, act_code(fusion.activation), wide.channel_slice(c_s, c_f))
    return [wide, fast], [slow_slice, fast]


def emit_lateral(sess, conv, fast, norm, act, out):
    """FuseFastToSlow's conv + norm + activation (models/slowfast.py:661-694, :720-729) -> pv_lateral_fuse, the
    dedicated time-strided kernel (csrc/pv_lateral.hip) that stores into the slow buffer's channel slice `out`.
    Anything that is not the builder's (kt,1,1) / (alpha,1,1) / (kt//2,0,0) bias-free conv goes through the
    general conv emitter (same result, generic kernel)."""
    kt, kh, kw = conv.kernel_size
    st, sh, sw = conv.stride
    pt, ph, pw = _triple(conv.padding)
    plain = (kh, kw, sh, sw, ph, pw) == (1, 1, 1, 1, 0, 0) and tuple(conv.dilation) == (1, 1, 1) and conv.groups == 1 \
        and conv.bias is None and fast.ld != 4 and not fast.f32
    if not plain:
        return emit_conv(sess, conv, fast, norm, act, out=out, label="lateral_fuse")
    if conv.in_channels != fast.C:
        raise RuntimeError("conv expects %d input channels, got %d" % (conv.in_channels, fast.C))
    To = _conv_out(fast.T, kt, st, pt)
    cout, cin_p = conv.out_channels, pad8(fast.C)
    if To <= 0:
        raise RuntimeError("conv output would be empty")
    if (out.B, out.T, out.H, out.W) != (fast.B, To, fast.H, fast.W) or out.C != cout:
        raise RuntimeError("lateral fusion output geometry mismatch: %s vs %s" % (
            (out.B, out.T, out.H, out.W, out.C), (fast.B, To, fast.H, fast.W, cout)))
    wp = torch.zeros(cout, kt, cin_p, dtype=torch.float32)
    wp[:, :, : fast.C] = conv.weight.detach().float().cpu().reshape(cout, fast.C, kt).permute(0, 2, 1)
    scale, shift = fold_norm(norm, cout, None)
    has_affine = norm is not None and not isinstance(norm, nn.Identity)
    f = dict(x=fast.ptr, w=sess.add_weight(wp.to(sess.dtype)), y=out.ptr,
             scale=sess.add_weight(scale) if has_affine else None, shift=sess.add_weight(shift) if has_affine else None,
             x_bs=fast.bs, y_bs=out.bs, ldx=fast.ld, ldy=out.ld, B=fast.B, Ti=fast.T, H=fast.H, W=fast.W, cin=cin_p,
             To=To, cout=cout, kt=kt, st=st, pt=pt, act=act, dtype=sess.pv_dtype)
    vox_out = fast.B * To * fast.H * fast.W
    alg = sess.itemsize * (fast.B * fast.voxels * cin_p + cout * kt * cin_p + vox_out * pad8(cout))
    detail = "|%dx%dx%dx%d c%d->%d k%dx1x1 s%d11" % (fast.B, To, fast.H, fast.W, fast.C, cout, kt, st)
    sess.add_op(L.OP_LATERAL, f, label="lateral_fuse" + detail, alg_bytes=alg, flops=2 * vox_out * cout * kt * fast.C)
    return out


def emit_module_out(sess, m, x, out):
    n = _cls_name(m)
    if n == "ResNetBasicStem":
        return emit_stem(sess, m, x, out=out)
    if n == "ResStage":
        return emit_res_stage(sess, m, x, out=out)
    if n == "ResBlock":
        return emit_res_block(sess, m, x, out=out)
    raise Unsupported("no out= emitter for %s" % n)


def emit_pool_concat(sess, pc, xs):
    """PoolConcatPathway.forward (models/slowfast.py:608-620): every pathway's pool writes its
    channel slice of one output buffer."""
    if pc.dim != 1:
        raise Unsupported("concat dim %d" % pc.dim)
    live = [(i, x) for i, x in enumerate(xs) if x is not None]
    shapes = []
    for i, x in live:
        pool = pc.pool[i] if pc.pool is not None else None
        if pool is None:
            shapes.append((x.T, x.H, x.W, x.C))
        else:
            shapes.append(out_shape(pool, x))
    if len({s[:3] for s in shapes}) != 1:
        raise RuntimeError("Sizes of tensors must match except in dimension 1")
    total = sum(s[3] for s in shapes)
    T, H, W = shapes[0][:3]
    y = sess.alloc_act(live[0][1].B, T, H, W, total)
    c0 = 0
    for (i, x), shp in zip(live, shapes):
        if c0 % 8:
            raise Unsupported("concat offset %d not a multiple of 8" % c0)
        pool = pc.pool[i] if pc.pool is not None else None
        dst = y.channel_slice(c0, shp[3])
        if pool is None:
            raise Unsupported("concat without pooling")
        emit_pool(sess, pool, x, out=dst, label="head.pool")
        c0 += shp[3]
    return y


# --------------------------------------------------------------------------- dispatch
def emit_module(sess, m, x):
    """Emit any supported single-input/single-output conv-family module."""
    n = _cls_name(m)
    if n == "ResNetBasicStem":
        return emit_stem(sess, m, x)
    if n == "ResStage":
        return emit_res_stage(sess, m, x)
    if n == "ResBlock":
        return emit_res_block(sess, m, x)
    if n == "ResNetBasicHead":
        return emit_res_head(sess, m, x)
    if _is_a(m, (nn.MaxPool3d, nn.AvgPool3d, nn.AdaptiveAvgPool3d)):
        return emit_pool(sess, m, x)
    raise Unsupported("no emitter for %s" % n)
