"""CPU restatement of the reference forward path (TEST INFRASTRUCTURE -- the checker, never
the product).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module.

Plain functional fp32 (or fp64) PyTorch-CPU arithmetic driven directly by a reference
`state_dict` -- no nn.Module of this package is involved, so it is an independent check of
the host mirror, of BN folding / weight packing and of the HIP kernels.  Every function
cites the reference code it follows.  Pinned against golden vectors produced by the real
reference (tests/golden/make_golden.py, run where /root/reference exists).
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- 16-bit storage emulation
# Default (None): the exact fp32 restatement of the reference -- the north-star comparator.
# Inside `with storage_emulation(torch.bfloat16):` the SAME restatement additionally rounds every tensor the bf16
# deploy form holds in 16 bits, at the point where it is stored / becomes an MFMA operand (DESIGN.md section 2:
# conv outputs after the folded BN + activation (+ residual), squeeze-excited operands, LayerNorm outputs, q/k/v,
# softmax probabilities, the attention output, the MLP hidden; the MViT token stream, logits, SE gates, BN / LN
# statistics stay fp32).  Together with `oracle.weights.quantize_like_kernels` this is "the reference evaluated
# with bf16 storage": what is left between it and the HIP path is the kernels' own arithmetic (accumulation order,
# exp / erf / sigmoid approximations).  It is also the kernel-independent measure of what 16-bit storage alone costs
# against the fp32 reference (tools/storage_floor.py).
_STORE = None
_BATCH_HINT = 1     # clips in the deploy form's batch: the MViT plan routes pooling convs by tensor size
POOL_STREAM_MIN_ELEMS_DEFAULT = 1 << 22     # the deploy form's default routing threshold (tuning.OPTIONS["pool_stream_min_elems"]);
_POOL_STREAM_MIN_ELEMS = POOL_STREAM_MIN_ELEMS_DEFAULT   # tests/test_host.py pins the two to each other


class storage_emulation:
    """`pool_stream_min_elems`: the deploy form's routing threshold for MViT pooling convs (a caller that changed the
    knob passes the same value; tests/test_host.py checks that the default here is the emitters' default)."""

    def __init__(self, dtype=torch.bfloat16, batch=1, pool_stream_min_elems=None):
        self.dtype, self.batch = dtype, batch
        self.pool_min = POOL_STREAM_MIN_ELEMS_DEFAULT if pool_stream_min_elems is None else int(pool_stream_min_elems)

    def __enter__(self):
        global _STORE, _BATCH_HINT, _POOL_STREAM_MIN_ELEMS
        dt = self.dtype
        _STORE, _BATCH_HINT, _POOL_STREAM_MIN_ELEMS = (lambda t: t.to(dt).to(torch.float32)), self.batch, self.pool_min
        return self

    def __exit__(self, *exc):
        global _STORE, _BATCH_HINT, _POOL_STREAM_MIN_ELEMS
        _STORE, _BATCH_HINT, _POOL_STREAM_MIN_ELEMS = None, 1, POOL_STREAM_MIN_ELEMS_DEFAULT
        return False


def _st(x):
    """A tensor the 16-bit deploy form stores (identity unless inside `storage_emulation`)."""
    return x if _STORE is None else _STORE(x)


def _shortcut_is_second_operand(cin_c, cin_x):
    """Under emulation: does the projection shortcut ride in conv_c as a second K operand (no stored copy)?
    Same rule as csrc/pv_pwconv.hip::pv_pwconv_x2_supported (both operands within 8 K-steps of 32)."""
    return _STORE is not None and (cin_c + 31) // 32 + (cin_x + 31) // 32 <= 8


def _bn(x, sd, p, eps=1e-5):
    """BatchNorm3d in eval mode: (x-mean)/sqrt(var+eps)*gamma+beta (torch semantics)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd.get(p + ".weight"),
                        sd.get(p + ".bias"), training=False, eps=eps)


def _has(sd, p):
    return (p + ".weight") in sd


def swish(x):
    """x*sigmoid(x) (pytorchvideo/layers/swish.py:21-25)."""
    return x * torch.sigmoid(x)


def round_width(width, multiplier, min_width=8, divisor=8):
    """pytorchvideo/layers/utils.py:19-39 (ceil=False branch)."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


# ----------------------------------------------------------------------------- X3D
def x3d_stem(sd, x, p="blocks.0"):
    """create_x3d_stem (models/x3d.py:19-102) -> Conv2plus1d.forward (layers/convolutions.py:232-237)
    -> ResNetBasicStem.forward (models/stem.py:252-260).  The slot called conv_t holds the
    1x3x3 *spatial* conv and conv_xy the depthwise 5x1x1 *temporal* conv (x3d.py:83-88)."""
    w = sd[p + ".conv.conv_t.weight"]
    x = F.conv3d(x, w, stride=(1, 2, 2), padding=(0, w.shape[3] // 2, w.shape[4] // 2))
    w = sd[p + ".conv.conv_xy.weight"]
    x = F.conv3d(x, w, stride=1, padding=(w.shape[2] // 2, 0, 0), groups=w.shape[0])
    return _st(F.relu(_bn(x, sd, p + ".norm")))


def squeeze_excite(sd, x, p):
    """fvcore SqueezeExcitation (restated; see pytorchvideo_amd/layers/squeeze_excitation.py):
    x * sigmoid(W2 relu(W1 mean_{T,H,W}(x) + b1) + b2)."""
    m = x.mean(dim=[2, 3, 4], keepdim=True)
    h = F.relu(F.conv3d(m, sd[p + ".block.0.weight"], sd[p + ".block.0.bias"]))
    return _st(x) * torch.sigmoid(F.conv3d(h, sd[p + ".block.2.weight"], sd[p + ".block.2.bias"]))


def x3d_res_block(sd, x, p, stride):
    """create_x3d_res_block (x3d.py:231-324) + BottleneckBlock.forward (resnet.py:1345-1365)
    + ResBlock.forward (resnet.py:1179-1189)."""
    b = p + ".branch2"
    if _has(sd, p + ".branch1_conv"):
        sc = F.conv3d(x, sd[p + ".branch1_conv.weight"], stride=stride)
        if _has(sd, p + ".branch1_norm"):
            sc = _bn(sc, sd, p + ".branch1_norm")
        if not _shortcut_is_second_operand(sd[b + ".conv_c.weight"].shape[1], x.shape[1]):
            sc = _st(sc)
    else:
        sc = x
    y = _st(F.relu(_bn(F.conv3d(x, sd[b + ".conv_a.weight"]), sd, b + ".norm_a")))
    w = sd[b + ".conv_b.weight"]
    y = F.conv3d(y, w, stride=stride, padding=[k // 2 for k in w.shape[2:]], groups=w.shape[0])
    y = _bn(y, sd, b + ".norm_b.0")
    if (b + ".norm_b.1.block.0.weight") in sd:
        y = squeeze_excite(sd, y, b + ".norm_b.1")     # (stored before the gate, gated + swished on conv_c's load)
    y = _st(swish(y))
    y = _bn(F.conv3d(y, sd[b + ".conv_c.weight"]), sd, b + ".norm_c")
    return _st(F.relu(sc + y))


def x3d_head(sd, x, p, pool_kernel):
    """ProjectedPool.forward (x3d.py:791-806) + ResNetBasicHead.forward (models/head.py:371-391),
    head_activation=None (create_x3d default, x3d.py:577)."""
    x = _st(F.relu(_bn(F.conv3d(x, sd[p + ".pool.pre_conv.weight"]), sd, p + ".pool.pre_norm")))
    x = _st(F.avg_pool3d(x, pool_kernel, stride=1))
    x = _st(F.relu(F.conv3d(x, sd[p + ".pool.post_conv.weight"])))
    x = F.linear(x.permute(0, 2, 3, 4, 1), sd[p + ".proj.weight"], sd[p + ".proj.bias"]).permute(0, 4, 1, 2, 3)
    return x.mean(dim=[2, 3, 4])


def x3d_forward(sd, x, input_clip_length, input_crop_size, return_blocks=False):
    """create_x3d (x3d.py:539-739) with default strides; stage depths are read off the
    state_dict.  Returns logits (and the six block outputs)."""
    outs = []
    x = x3d_stem(sd, x)
    outs.append(x)
    for s in range(1, 5):
        i = 0
        while ("blocks.%d.res_blocks.%d.branch2.conv_a.weight" % (s, i)) in sd:
            x = x3d_res_block(sd, x, "blocks.%d.res_blocks.%d" % (s, i), (1, 2, 2) if i == 0 else (1, 1, 1))
            i += 1
        outs.append(x)
    side = int(math.ceil(input_crop_size / 32))
    x = x3d_head(sd, x, "blocks.5", (input_clip_length, side, side))
    outs.append(x)
    return (x, outs) if return_blocks else x


# ----------------------------------------------------------------------------- ResNet family
def _conv_same(x, w, stride, groups=1, bias=None):
    """Conv3d with padding = kernel//2 per dim (how every builder on the path pads:
    resnet.py:744-791, csn.py:160-170, r2plus1d.py:280-290, slowfast.py:209-229,672-694)."""
    return F.conv3d(x, w, bias, stride=stride, padding=[k // 2 for k in w.shape[2:]], groups=groups)


def res_basic_stem(sd, x, p, stride=(1, 2, 2), pool=True):
    """create_res_basic_stem (models/stem.py:11-107) + ResNetBasicStem.forward (:252-260)."""
    x = _st(F.relu(_bn(_conv_same(x, sd[p + ".conv.weight"], stride), sd, p + ".norm")))
    if pool:
        x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    return x


def bottleneck_res_block(sd, x, p, stride_a, stride_b, dilation_b=(1, 1, 1)):
    """create_res_block (resnet.py:326-462) + create_bottleneck_block (:17-148) +
    BottleneckBlock.forward (:1345-1365) + ResBlock.forward (:1179-1189).  conv_b may be
    dense, depthwise (CSN, csn.py:169) or a Conv2plus1d (R(2+1)D, convolutions.py:232-237)."""
    b = p + ".branch2"
    if _has(sd, p + ".branch1_conv"):
        s = tuple(a * c for a, c in zip(stride_a, stride_b))
        sc = F.conv3d(x, sd[p + ".branch1_conv.weight"], stride=s)
        if _has(sd, p + ".branch1_norm"):
            sc = _bn(sc, sd, p + ".branch1_norm")
        wc = sd[b + ".conv_c.weight"]
        if not (_shortcut_is_second_operand(wc.shape[1], x.shape[1]) and wc.shape[1] <= 128):
            sc = _st(sc)
    else:
        sc = x
    y = _st(F.relu(_bn(_conv_same(x, sd[b + ".conv_a.weight"], stride_a), sd, b + ".norm_a")))
    if (b + ".conv_b.conv_t.weight") in sd:  # (2+1)D: temporal conv -> BN -> ReLU -> spatial conv
        y = _conv_same(y, sd[b + ".conv_b.conv_t.weight"], (stride_b[0], 1, 1))
        y = _st(F.relu(_bn(y, sd, b + ".conv_b.norm")))
        y = _conv_same(y, sd[b + ".conv_b.conv_xy.weight"], (1, stride_b[1], stride_b[2]))
    else:
        w = sd[b + ".conv_b.weight"]
        groups = y.shape[1] // w.shape[1]
        if tuple(dilation_b) == (1, 1, 1):
            y = _conv_same(y, w, stride_b, groups=groups)
        else:  # dilated conv_b: the spatial padding follows the dilation (resnet.py:797-811)
            k = w.shape[2:]
            pad = (k[0] // 2, dilation_b[1] if dilation_b[1] > 1 else k[1] // 2,
                   dilation_b[2] if dilation_b[2] > 1 else k[2] // 2)
            y = F.conv3d(y, w, None, stride=stride_b, padding=pad, dilation=tuple(dilation_b), groups=groups)
    y = _st(F.relu(_bn(y, sd, b + ".norm_b")))
    y = _bn(F.conv3d(y, sd[b + ".conv_c.weight"]), sd, b + ".norm_c")
    return _st(F.relu(sc + y))


def res_stage(sd, x, p, stride_a, stride_b, dilation_b=(1, 1, 1)):
    i = 0
    while (p + ".res_blocks.%d.branch2.conv_a.weight" % i) in sd:
        first = i == 0
        x = bottleneck_res_block(sd, x, p + ".res_blocks.%d" % i, stride_a if first else (1, 1, 1),
                                 stride_b if first else (1, 1, 1), dilation_b)
        i += 1
    return x


def res_basic_head(sd, x, p, pool_kernel=None, softmax=False):
    """create_res_basic_head (models/head.py:39-131) + ResNetBasicHead.forward (:371-391)."""
    if pool_kernel is not None:
        x = _st(F.avg_pool3d(x, pool_kernel, stride=1))
    x = F.linear(x.permute(0, 2, 3, 4, 1), sd[p + ".proj.weight"], sd[p + ".proj.bias"]).permute(0, 4, 1, 2, 3)
    if softmax:
        x = torch.softmax(x, dim=1)
    return x.mean(dim=[2, 3, 4])


def resnet_forward(sd, x, head_pool_kernel=(4, 7, 7), stage1_pool_kernel=None, spatial=(1, 2, 2, 2), temporal=(1, 1, 1, 1)):
    """create_resnet (models/resnet.py:601-1003) as the hub uses it for slow_r50 / c2d_r50 / i3d_r50
    (models/hub/resnet.py:41-160): stem with pool, four bottleneck stages whose conv_a kernels differ per
    block (read off the weights), the temporal stride on conv_a and the spatial one on conv_b
    (resnet.py:905-925), an optional MaxPool3d block after stage 1 (`stage1_pool`, :927-934) and the head."""
    x = res_basic_stem(sd, x, "blocks.0")
    b = 1
    for i in range(4):
        x = res_stage(sd, x, "blocks.%d" % b, (temporal[i], 1, 1), (1, spatial[i], spatial[i]))
        b += 1
        if i == 0 and stage1_pool_kernel is not None:
            x = F.max_pool3d(x, stage1_pool_kernel, stage1_pool_kernel)
            b += 1
    return res_basic_head(sd, x, "blocks.%d" % b, head_pool_kernel)


def csn_forward(sd, x, head_pool_kernel=(1, 7, 7), spatial=(1, 2, 2, 2), temporal=(1, 2, 2, 2)):
    """create_csn (models/csn.py:12-191): stem without pool, depthwise 3x3x3 conv_b."""
    x = res_basic_stem(sd, x, "blocks.0", pool=False)
    for i in range(4):
        x = res_stage(sd, x, "blocks.%d" % (i + 1), (1, 1, 1), (temporal[i], spatial[i], spatial[i]))
    return res_basic_head(sd, x, "blocks.5", head_pool_kernel)


def r2plus1d_forward(sd, x, head_pool_kernel=(4, 7, 7), spatial=(2, 2, 2, 2), temporal=(1, 1, 2, 2)):
    """create_r2plus1d (models/r2plus1d.py:123-313): stem without pool, head softmax."""
    x = res_basic_stem(sd, x, "blocks.0", pool=False)
    for i in range(4):
        x = res_stage(sd, x, "blocks.%d" % (i + 1), (1, 1, 1), (temporal[i], spatial[i], spatial[i]))
    return res_basic_head(sd, x, "blocks.5", head_pool_kernel, softmax=True)


def slowfast_forward(sd, slow, fast, head_pool_kernels=((8, 7, 7), (32, 7, 7)), return_blocks=False):
    """create_slowfast (models/slowfast.py:22-361) with default strides: per-pathway stems and
    stages (MultiPathWayWithFuse.forward, models/net.py:107-122), FuseFastToSlow after the stem
    and res2..res4 (slowfast.py:720-729; conv 7x1x1 stride (4,1,1) -> BN -> ReLU -> cat),
    PoolConcatPathway (slowfast.py:608-620) and the basic head."""
    outs = []

    def fuse(s, f, p):
        if (p + ".conv_fast_to_slow.weight") not in sd:
            return s
        w = sd[p + ".conv_fast_to_slow.weight"]
        z = F.conv3d(f, w, stride=(4, 1, 1), padding=(w.shape[2] // 2, 0, 0))
        z = _st(F.relu(_bn(z, sd, p + ".norm")))
        return torch.cat([s, z], 1)

    s = res_basic_stem(sd, slow, "blocks.0.multipathway_blocks.0")
    f = res_basic_stem(sd, fast, "blocks.0.multipathway_blocks.1")
    s = fuse(s, f, "blocks.0.multipathway_fusion")
    outs.append((s, f))
    spatial = (1, 2, 2, 2)
    for i in range(4):
        p = "blocks.%d" % (i + 1)
        s = res_stage(sd, s, p + ".multipathway_blocks.0", (1, 1, 1), (1, spatial[i], spatial[i]))
        f = res_stage(sd, f, p + ".multipathway_blocks.1", (1, 1, 1), (1, spatial[i], spatial[i]))
        s = fuse(s, f, p + ".multipathway_fusion")
        outs.append((s, f))
    x = _st(torch.cat([F.avg_pool3d(s, head_pool_kernels[0], stride=1),
                       F.avg_pool3d(f, head_pool_kernels[1], stride=1)], 1))
    outs.append(x)
    y = res_basic_head(sd, x, "blocks.6")
    outs.append(y)
    return (y, outs) if return_blocks else y


# ----------------------------------------------------------------------------- detection (RoI head)
def roi_align(x, boxes, output_size, spatial_scale, sampling_ratio=0, aligned=False):
    """torchvision.ops.roi_align restated sample by sample (third-party: the reference imports it at
    models/head.py:8 and builds it at head.py:318-322 with torchvision's default aligned=False; torchvision
    is un-pinned in the reference's setup.py and absent here -- PARITY UNPINNED for this op).
    Published definition (torchvision/csrc/ops/cpu/roi_align_kernel.cpp, roi_align_common.h): box
    (batch, x1, y1, x2, y2) is scaled by spatial_scale (minus 0.5 if aligned); its size is clamped to >= 1
    when not aligned; every output bin averages grid_h x grid_w bilinear samples at
    start + bin*bin_size + (i + .5) * bin_size / grid, grid = sampling_ratio or ceil(roi / bins); a sample
    with y < -1 or y > H (same for x) contributes 0; coordinates are clamped to >= 0 and, at the last
    row/column, both neighbours collapse onto it.  x [B,C,H,W] fp32, boxes [R,5] -> [R,C,ph,pw]."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    _, C, H, W = x.shape
    out = torch.zeros((boxes.shape[0], C, ph, pw), dtype=x.dtype)
    off = 0.5 if aligned else 0.0
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)  # keep the coordinate arithmetic in fp32 like the kernel
    for n in range(boxes.shape[0]):
        bi = int(boxes[n, 0])
        bx = boxes[n].to(torch.float32)
        sc = f32(spatial_scale)
        start_w, start_h = bx[1] * sc - off, bx[2] * sc - off
        end_w, end_h = bx[3] * sc - off, bx[4] * sc - off
        roi_w, roi_h = end_w - start_w, end_h - start_h
        if not aligned:
            roi_w, roi_h = torch.clamp(roi_w, min=1.0), torch.clamp(roi_h, min=1.0)
        bin_h, bin_w = roi_h / ph, roi_w / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(roi_h / ph)))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(roi_w / pw)))
        count = max(gh * gw, 1)
        feat = x[bi]
        for i in range(ph):
            for j in range(pw):
                acc = torch.zeros(C, dtype=x.dtype)
                for iy in range(gh):
                    y = float(start_h + i * bin_h + (iy + 0.5) * bin_h / gh)
                    for ix in range(gw):
                        xx = float(start_w + j * bin_w + (ix + 0.5) * bin_w / gw)
                        if y < -1.0 or y > H or xx < -1.0 or xx > W:
                            continue
                        yy, xc = max(y, 0.0), max(xx, 0.0)
                        y_lo, x_lo = int(yy), int(xc)
                        if y_lo >= H - 1:
                            y_hi = y_lo = H - 1
                            yy = float(y_lo)
                        else:
                            y_hi = y_lo + 1
                        if x_lo >= W - 1:
                            x_hi = x_lo = W - 1
                            xc = float(x_lo)
                        else:
                            x_hi = x_lo + 1
                        ly, lx = yy - y_lo, xc - x_lo
                        hy, hx = 1.0 - ly, 1.0 - lx
                        acc += (hy * hx * feat[:, y_lo, x_lo] + hy * lx * feat[:, y_lo, x_hi]
                                + ly * hx * feat[:, y_hi, x_lo] + ly * lx * feat[:, y_hi, x_hi])
                out[n, :, i, j] = acc / count
    return out


def res_roi_head(sd, x, boxes, p, pool_kernel=None, resolution=(7, 7), spatial_scale=1.0 / 16.0,
                 sampling_ratio=0, sigmoid=True):
    """create_res_roi_pooling_head (models/head.py:203-327) + ResNetRoIHead.forward (:437-482): temporal
    AvgPool3d -> squeeze T -> RoIAlign -> MaxPool2d(resolution, stride 1) -> Linear -> Sigmoid; no global
    average (head_output_with_global_average=False in both detection builders)."""
    if pool_kernel is not None:
        x = F.avg_pool3d(x, pool_kernel, stride=1)
    if x.shape[2] != 1:
        raise Exception("Temporal dimension should be 1. Consider modifying the pool layer.")
    r = roi_align(x[:, :, 0], boxes, resolution, spatial_scale, sampling_ratio)
    r = F.max_pool2d(r, resolution, stride=1).unsqueeze(-3)
    y = F.linear(r.permute(0, 2, 3, 4, 1), sd[p + ".proj.weight"], sd[p + ".proj.bias"]).permute(0, 4, 1, 2, 3)
    if sigmoid:
        y = torch.sigmoid(y)
    return y.reshape(y.shape[0], -1)   # DetectionBBoxNetwork.forward (models/net.py:72-74)


def resnet_detection_forward(sd, x, boxes, head_pool_kernel=(4, 1, 1), resolution=(7, 7), spatial_scale=1.0 / 16.0,
                             sampling_ratio=0):
    """create_resnet_with_roi_head (models/resnet.py:844-1019): stem (1,7,7), stages with spatial strides
    (1,2,2,1) and conv_b of the last stage dilated (1,2,2), then the RoI head.  state_dict prefixes
    `model.` / `detection_head.` (DetectionBBoxNetwork, models/net.py:47-74)."""
    bb = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    hd = {k[len("detection_head."):]: v for k, v in sd.items() if k.startswith("detection_head.")}
    y = res_basic_stem(bb, x, "blocks.0")
    spatial, dil = (1, 2, 2, 1), ((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 2, 2))
    for i in range(4):
        y = res_stage(bb, y, "blocks.%d" % (i + 1), (1, 1, 1), (1, spatial[i], spatial[i]), dil[i])
    return res_roi_head({"h." + k: v for k, v in hd.items()}, y, boxes, "h", head_pool_kernel, resolution,
                        spatial_scale, sampling_ratio)


def slowfast_detection_forward(sd, slow, fast, boxes, head_pool_kernels=((8, 1, 1), (32, 1, 1)), resolution=(7, 7),
                               spatial_scale=1.0 / 16.0, sampling_ratio=0):
    """create_slowfast_with_roi_head (models/slowfast.py:364-582): SlowFast backbone with spatial strides
    (1,2,2,1), last-stage conv_b dilated (1,2,2), PoolConcatPathway pooling time only, then the RoI head."""
    bb = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    hd = {"h." + k[len("detection_head."):]: v for k, v in sd.items() if k.startswith("detection_head.")}

    def fuse(s, f, p):
        if (p + ".conv_fast_to_slow.weight") not in bb:
            return s
        w = bb[p + ".conv_fast_to_slow.weight"]
        z = F.conv3d(f, w, stride=(4, 1, 1), padding=(w.shape[2] // 2, 0, 0))
        return torch.cat([s, F.relu(_bn(z, bb, p + ".norm"))], 1)

    s = res_basic_stem(bb, slow, "blocks.0.multipathway_blocks.0")
    f = res_basic_stem(bb, fast, "blocks.0.multipathway_blocks.1")
    s = fuse(s, f, "blocks.0.multipathway_fusion")
    spatial, dil = (1, 2, 2, 1), ((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 2, 2))
    for i in range(4):
        p = "blocks.%d" % (i + 1)
        s = res_stage(bb, s, p + ".multipathway_blocks.0", (1, 1, 1), (1, spatial[i], spatial[i]), dil[i])
        f = res_stage(bb, f, p + ".multipathway_blocks.1", (1, 1, 1), (1, spatial[i], spatial[i]), dil[i])
        s = fuse(s, f, p + ".multipathway_fusion")
    x = torch.cat([F.avg_pool3d(s, head_pool_kernels[0], stride=1),
                   F.avg_pool3d(f, head_pool_kernels[1], stride=1)], 1)
    return res_roi_head(hd, x, boxes, "h", None, resolution, spatial_scale, sampling_ratio)


# ----------------------------------------------------------------------------- MViT
def _ln(x, sd, p, eps=1e-6):
    """nn.LayerNorm(eps=1e-6) as built at models/vision_transformers.py:333-335."""
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _block_norm(x, sd, p):
    """norm1 / norm2 of a MultiScaleBlock: LayerNorm, or -- norm="batchnorm", models/vision_transformers.py:336-339 --
    nn.BatchNorm1d over the channel dim in eval mode (layers/attention.py:738-753: permute, BN, permute back)."""
    if (p + ".running_mean") in sd:
        return _bn(x.permute(0, 2, 1), sd, p).permute(0, 2, 1)
    return _ln(x, sd, p)


def mvit_schedule(cfg):
    """Per-block (heads, kernel_q, stride_q, kernel_kv, stride_kv) of
    create_multiscale_vision_transformers (models/vision_transformers.py:394-445)."""
    depth = cfg.get("depth", 16)
    head_mul = [1.0] * (depth + 1)
    for i, m in cfg.get("atten_head_mul") or []:
        head_mul[i] = m
    kq = [None] * depth
    sq = [None] * depth
    kkv = [None] * depth
    skv = [None] * depth
    fixed = cfg.get("pool_kvq_kernel")
    for e in cfg.get("pool_q_stride_size") or []:
        sq[e[0]] = list(e[1:])
        kq[e[0]] = list(fixed) if fixed is not None else [s + 1 if s > 1 else s for s in e[1:]]
    kv_sizes = cfg.get("pool_kv_stride_size")
    if cfg.get("pool_kv_stride_adaptive") is not None:
        cur = list(cfg["pool_kv_stride_adaptive"])
        kv_sizes = []
        for i in range(depth):
            if sq[i]:
                cur = [max(cur[d] // sq[i][d], 1) for d in range(3)]
            kv_sizes.append([i] + cur)
    for e in kv_sizes or []:
        skv[e[0]] = list(e[1:])
        kkv[e[0]] = list(fixed) if fixed is not None else [s + 1 if s > 1 else s for s in e[1:]]
    heads, out = cfg.get("num_heads", 1), []
    for i in range(depth):
        heads = round_width(heads, head_mul[i], min_width=1, divisor=1)
        out.append((heads, kq[i], sq[i], kkv[i], skv[i]))
    return out


def _attention_pool(sd, t, thw, p, kernel, stride, has_cls, norm_p=None, pool_fn=None):
    """_AttentionPool.forward (layers/attention.py:162-212) for (B, heads, N, C) tokens and a
    depthwise conv pool shared over heads (or `pool_fn` for the skip-path max pool)."""
    if kernel is None:
        return t, thw
    if kernel is not None and stride is not None and math.prod(kernel) == 1 and math.prod(stride) == 1:
        return t, thw
    cls_tok = None
    if has_cls:
        cls_tok, t = t[:, :, :1, :], t[:, :, 1:, :]
    B, N, L, C = t.shape
    T, H, W = thw
    g = t.reshape(B * N, T, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
    bn_first = norm_p is not None and (norm_p + ".running_mean") in sd
    if bn_first:   # BatchNorm3d(head_dim) + GELU BEFORE the pool, cls token aside (layers/attention.py:186-190)
        g = _st(F.gelu(_bn(g, sd, norm_p)))
        norm_p = None
    if pool_fn is not None:
        g = pool_fn(g)
    else:
        w = sd[p + ".weight"]
        g = F.conv3d(g, w, None, stride=stride, padding=[k // 2 for k in kernel], groups=w.shape[0])
        # the deploy form pools big grids with the plane-streaming depthwise kernel (its output is stored, the per-head
        # LayerNorm is a second launch) and small ones with the fused pool + LayerNorm kernel (emit_mvit._streams_well)
        if norm_p is None or (tuple(kernel) == (3, 3, 3) and stride[0] == 1 and stride[1] == stride[2] and stride[1] in (1, 2)
                              and _BATCH_HINT * T * H * W * N * C >= _POOL_STREAM_MIN_ELEMS):
            g = _st(g)
    thw = [g.shape[2], g.shape[3], g.shape[4]]
    t = g.reshape(B, N, C, -1).transpose(2, 3)
    if cls_tok is not None:
        t = torch.cat((cls_tok, t), dim=2)
    if norm_p is not None:
        t = _st(_ln(t, sd, norm_p))
    return t, thw


def multiscale_block(sd, x, thw, p, heads, kq, sq, kkv, skv, has_cls=True, residual_pool=False,
                     dim_mul_in_att=False):
    """MultiScaleBlock.forward (layers/attention.py:729-757) with MultiScaleAttention.forward
    (:501-544), pool_mode="conv", depthwise, separate q/k/v, layernorm."""
    B, N, _ = x.shape
    xn = _st(_block_norm(x, sd, p + ".norm1"))
    a = p + ".attn"

    def heads_of(t):
        return _st(t).reshape(B, N, heads, -1).permute(0, 2, 1, 3)

    q = heads_of(F.linear(xn, sd[a + ".q.weight"], sd.get(a + ".q.bias")))
    k = heads_of(F.linear(xn, sd[a + ".k.weight"], sd.get(a + ".k.bias")))
    v = heads_of(F.linear(xn, sd[a + ".v.weight"], sd.get(a + ".v.bias")))
    q, q_thw = _attention_pool(sd, q, thw, a + ".pool_q", kq, sq, has_cls, a + ".norm_q")
    k, _ = _attention_pool(sd, k, thw, a + ".pool_k", kkv, skv, has_cls, a + ".norm_k")
    v, _ = _attention_pool(sd, v, thw, a + ".pool_v", kkv, skv, has_cls, a + ".norm_v")
    hd = q.shape[-1]
    if _STORE is None:
        attn = torch.softmax((q * hd ** -0.5) @ k.transpose(-2, -1), dim=-1)
        o = attn @ v
    else:   # the fused kernel: fp32 scores, un-normalised probabilities as a 16-bit MFMA operand, fp32 row sums
        sc = (q @ k.transpose(-2, -1)) * hd ** -0.5
        e = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
        o = (_st(e) @ v) / e.sum(dim=-1, keepdim=True)
    if residual_pool:
        o = o + q
    o = _st(o).transpose(1, 2).reshape(B, -1, heads * hd)
    x_block = F.linear(o, sd[a + ".proj.weight"], sd.get(a + ".proj.bias"))
    widen = (p + ".proj.weight") in sd
    if dim_mul_in_att and widen:
        x = F.linear(xn, sd[p + ".proj.weight"], sd.get(p + ".proj.bias"))
    if sq is not None and math.prod(sq) > 1:
        ks = [s + 1 if s > 1 else s for s in sq]
        x_res, _ = _attention_pool(sd, x.unsqueeze(1), thw, None, ks, sq, has_cls,
                                   pool_fn=lambda g: F.max_pool3d(g, ks, sq, [k_ // 2 for k_ in ks]))
        x_res = x_res.squeeze(1)
    else:
        x_res = x
    x = x_res + x_block
    xn = _st(_block_norm(x, sd, p + ".norm2"))
    h = _st(F.gelu(F.linear(xn, sd[p + ".mlp.fc1.weight"], sd.get(p + ".mlp.fc1.bias"))))
    x_mlp = F.linear(h, sd[p + ".mlp.fc2.weight"], sd.get(p + ".mlp.fc2.bias"))
    if (not dim_mul_in_att) and widen:
        x = F.linear(xn, sd[p + ".proj.weight"], sd.get(p + ".proj.bias"))
    return x + x_mlp, q_thw


def mvit_forward(sd, x, cfg, return_blocks=False):
    """MultiscaleVisionTransformers.forward (models/vision_transformers.py:172-182) as built by
    create_multiscale_vision_transformers(**cfg) with layernorm, conv pooling, cls token."""
    stride = cfg.get("conv_patch_embed_stride", (2, 4, 4))
    pad = cfg.get("conv_patch_embed_padding", (1, 3, 3))
    has_cls = cfg.get("cls_embed_on", True)
    # PatchEmbed.forward (models/stem.py:289-292)
    y = F.conv3d(x, sd["patch_embed.patch_model.weight"], sd.get("patch_embed.patch_model.bias"),
                 stride=stride, padding=pad)
    thw = [y.shape[2], y.shape[3], y.shape[4]]
    y = y.flatten(2).transpose(1, 2)
    # SpatioTemporalClsPositionalEncoding.forward (layers/positional_encoding.py:112-136)
    c = "cls_positional_encoding"
    if has_cls:
        y = torch.cat((sd[c + ".cls_token"].expand(y.shape[0], -1, -1), y), dim=1)
    if cfg.get("sep_pos_embed", True):
        pos = sd[c + ".pos_embed_spatial"].repeat(1, thw[0], 1) + torch.repeat_interleave(
            sd[c + ".pos_embed_temporal"], thw[1] * thw[2], dim=1)
        if has_cls:
            pos = torch.cat([sd[c + ".pos_embed_class"], pos], 1)
        y = y + pos
    else:
        y = y + sd[c + ".pos_embed"]
    outs = []
    for i, (heads, kq, sq, kkv, skv) in enumerate(mvit_schedule(cfg)):
        y, thw = multiscale_block(sd, y, thw, "blocks.%d" % i, heads, kq, sq, kkv, skv, has_cls,
                                  cfg.get("residual_pool", False), cfg.get("dim_mul_in_att", False))
        outs.append(y)
    if "norm_embed.weight" in sd:   # (the norm="batchnorm" model has no final norm: vision_transformers.py:337,486)
        y = _st(_ln(y, sd, "norm_embed"))
    # VisionTransformerBasicHead.forward (models/head.py:521-535), dropout = identity in eval
    y = y[:, 0] if has_cls else y.mean(1)
    y = F.linear(y, sd["head.proj.weight"], sd["head.proj.bias"])
    return (y, outs) if return_blocks else y
