"""Deterministic, key-addressed weight fill (TEST INFRASTRUCTURE).

The reference's default init is degenerate for parity purposes: the gamma of every
block-final BN is zero (pytorchvideo/models/weight_init.py:34-35), so every `branch2`
contributes exactly 0 and conv_a/b/c are never exercised.  This fill
follows the reference's own `rand_init_bn` recipe (tests/test_fuse_bn.py:58-63) for norm
statistics and a variance-preserving normal for everything else.

Each tensor is generated from a CPU generator seeded by crc32(key) ^ seed, so the values
depend only on (key, shape, seed) -- NOT on construction order.  The same call on the
reference model (in the survey container) and on this package's model (on the GPU box)
therefore produces identical weights provided the state_dict keys and shapes agree, which
is itself part of the drop-in claim.
"""
import zlib

import torch


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _uniform(shape, lo, hi, g):
    return torch.rand(shape, generator=g) * (hi - lo) + lo


@torch.no_grad()
def deterministic_fill(model, seed=0):
    sd = model.state_dict()
    for key, t in sd.items():
        if key.endswith("num_batches_tracked") or not t.is_floating_point():
            continue
        g = _gen(key, seed)
        shape = tuple(t.shape)
        if key.endswith("running_var"):
            v = _uniform(shape, 0.5, 1.5, g)
        elif key.endswith("running_mean"):
            v = _uniform(shape, -0.5, 0.5, g)
        elif "pos_embed" in key or "cls_token" in key:
            v = torch.randn(shape, generator=g) * 0.5
        elif t.dim() == 1 and key.endswith("weight"):
            v = _uniform(shape, 0.5, 1.5, g)         # norm gammas
        elif t.dim() == 1:
            v = _uniform(shape, -0.3, 0.3, g)        # biases / betas
        else:
            fan_in = max(1, t.numel() // shape[0])
            v = torch.randn(shape, generator=g) * (1.5 / fan_in) ** 0.5
        t.copy_(v.to(t.dtype))
    return model


DETECTION_PROJ_SCALE = 0.01


@torch.no_grad()
def detection_fill(model, seed=0):
    """deterministic_fill for a DetectionBBoxNetwork: the same key-addressed values, with the head projection
    scaled down so that the scores do not sit in the saturated tails of the sigmoid (the variance-preserving
    fill drives the pooled features to |x| ~ 100, i.e. logits ~ 150: every score would be exactly 0 or 1 and
    a comparison of scores would be vacuous)."""
    deterministic_fill(model, seed)
    model.detection_head.proj.weight.mul_(DETECTION_PROJ_SCALE)
    model.detection_head.proj.bias.mul_(DETECTION_PROJ_SCALE)
    return model


def seeded_input(shape, seed=0, dist="randn"):
    g = torch.Generator()
    g.manual_seed(1000 + seed)
    if dist == "randn":
        return torch.randn(shape, generator=g)
    return torch.rand(shape, generator=g)


@torch.no_grad()
def reference_style_fill(model, seed=0):
    """The reference tests' own de-degeneration: keep the factory's default conv/linear init,
    randomise every BatchNorm with the `rand_init_bn` recipe of reference
    tests/test_fuse_bn.py:58-63 (weight~U(0.5,1.5), bias~U(-0.5,0.5), running_var~U(0.5,1.5),
    running_mean~U(-0.5,0.5)) and give the final Linear a non-trivial scale (std 0.05; the
    default 0.01 makes logits ~5e-3 and an absolute tolerance vacuous).
    Key-addressed like deterministic_fill, so it reproduces across processes."""
    import torch.nn as nn
    for name, mod in model.named_modules():
        if isinstance(mod, nn.modules.batchnorm._BatchNorm):
            g = _gen(name, seed)
            mod.weight.copy_(_uniform(mod.weight.shape, 0.5, 1.5, g))
            mod.bias.copy_(_uniform(mod.bias.shape, -0.5, 0.5, g))
            mod.running_var.copy_(_uniform(mod.running_var.shape, 0.5, 1.5, g))
            mod.running_mean.copy_(_uniform(mod.running_mean.shape, -0.5, 0.5, g))
        elif isinstance(mod, (nn.Conv3d, nn.Linear)):
            g = _gen(name, seed)
            fan_out_std = None
            if isinstance(mod, nn.Linear) and name.endswith("proj") and mod.out_features >= 100:
                fan_out_std = 0.05
            if isinstance(mod, nn.Conv3d):
                # c2_msra_fill: N(0, sqrt(2/fan_out))
                fan_out = mod.out_channels * mod.kernel_size[0] * mod.kernel_size[1] * mod.kernel_size[2] // mod.groups
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / fan_out) ** 0.5)
            elif fan_out_std is not None:
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * fan_out_std)
            if mod.bias is not None and isinstance(mod, nn.Conv3d):
                mod.bias.copy_(_uniform(mod.bias.shape, -0.1, 0.1, g))
    return model


def quantize_like_kernels(sd, x=None):
    """The bf16 deploy form stores dense-conv / linear weights and activations as bf16 while
    depthwise filters, squeeze-excitation FCs, folded-BN vectors, LayerNorm parameters and
    positional tables stay fp32.  Returns (sd', x') with exactly those tensors rounded to
    bf16 (values kept in fp32 containers) so that the oracle evaluates *the same quantised
    model*; what remains is the kernels' own arithmetic / activation-storage error."""
    out = {}
    for k, v in sd.items():
        dense = v.dim() >= 2 and k.endswith("weight")
        if dense and v.dim() == 5 and v.shape[1] == 1 and v.shape[0] > 1:
            dense = False  # depthwise filter [C,1,kt,kh,kw]
        if ".block." in k or "pos_embed" in k or "cls_token" in k:
            dense = False  # SE FCs / positional tables stay fp32
        out[k] = v.bfloat16().float() if dense else v
    return (out, x.bfloat16().float()) if x is not None else out


@torch.no_grad()
def calibrated_fill(model, calib_input, seed=0):
    """reference_style_fill, then the running statistics of every BatchNorm are set to the ACTUAL batch
    statistics of its input on `calib_input` -- what training leaves in a checkpoint (running_mean /
    running_var track the data, torch BatchNorm with momentum=None = cumulative average, one batch).
    With `rand_init_bn`'s arbitrary running statistics a deep stack is not normalised at all: X3D-L
    (55 blocks) grows its logits to ~1e9 and amplifies a 2^-9 perturbation of the weights to 27 % of the
    output (measured, tools/parity_full.py) -- a property of that random instance, not of the arithmetic
    under test.  Calibrated statistics give the activation scales a trained network has (unit variance
    after every BatchNorm, residual stream growing like sqrt(depth))."""
    reference_style_fill(model, seed)
    return _calibrate_running_stats(model, calib_input)


@torch.no_grad()
def _calibrate_running_stats(model, calib_input):
    """Set every BatchNorm's running statistics to the batch statistics of its input on `calib_input`."""
    import torch.nn as nn
    bns = [m for m in model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
    saved = [m.momentum for m in bns]
    model.eval()
    for m in bns:
        m.reset_running_stats()
        m.momentum = None
        m.train()
    model(list(calib_input) if isinstance(calib_input, (list, tuple)) else calib_input)
    for m, mom in zip(bns, saved):
        m.momentum = mom
    model.eval()
    return model


TRAINED_LIKE_FINAL_GAMMA = (0.05, 0.2)


@torch.no_grad()
def trained_like_fill(model, calib_input, seed=0, final_gamma=TRAINED_LIKE_FINAL_GAMMA):
    """calibrated_fill with the gamma of every BLOCK-FINAL BatchNorm (the modules the reference flags
    `block_final_bn`, pytorchvideo/models/resnet.py `norm_c.block_final_bn = True`) drawn from U(0.05, 0.2) instead
    of `rand_init_bn`'s U(0.5, 1.5); the running statistics of EVERY BatchNorm are calibrated on the data afterwards
    (a checkpoint's running statistics are the data's).

    Why: the reference's own init sets that gamma to ZERO (pytorchvideo/models/weight_init.py:34-35: a residual
    branch starts as the identity) and training moves it off zero; it is never the O(1) that
    tests/test_fuse_bn.py:58-63 draws for a *single* fused layer.  With gamma ~ 1 on every block-final BatchNorm each
    residual branch is as large as the trunk, 26-55 blocks compound every rounding, and bf16 STORAGE alone (exact
    arithmetic, no kernel) moves the logits by 3-7e-2 (`calibrated_fill`, kept as the stress instance).

    Measured on the CPU, no kernel involved (tools/storage_floor.py, profiles/r4/storage_floor.json; metric
    max|d| / max|logits|, one clip at the BASELINE geometry; "5 %" = the answer of the fp32 oracle to a 5 % scaling
    of ONE res3 conv_c filter bank, i.e. how loudly a kernel defect in one layer speaks in the logits; "clips" = the
    difference between the logits of two different clips):

        X3D-M, block-final gamma                  bf16 weights   bf16 storage   5 % defect   clips
        U(0.5,1.5)  calibrated (stress)              2.6e-2         4.2e-2        5.9e-2     7.0e-2
        U(0.2,0.6)  calibrated                       8.3e-3         9.4e-3        2.4e-2     6.5e-2
        U(0.1,0.4)  calibrated                       7.5e-3         9.1e-3        9.1e-3     7.2e-2
        U(0.05,0.2) calibrated  (THIS fill)          6.3e-3         6.3e-3        3.8e-3     6.6e-2
        U(0.5,1.5) x 0.25 AFTER calibration,         3.1e-3         3.6e-3        6.6e-4     2.0e-3
                   statistics left stale
        X3D-L:  U(0.1,0.4) 9.9e-3 / 1.2e-2;  U(0.05,0.2) 6.7e-3 / 8.2e-3;  U(0.03,0.12) 7.4e-3 / 8.6e-3;  stale 2.7e-3 / 3.5e-3
        SlowFast-R50:  U(0.1,0.4) 4.5e-3 / 5.0e-3;  U(0.05,0.2) 3.7e-3 / 3.6e-3;  stale 2.0e-3 / 2.0e-3

    Two things follow.  (1) With calibrated statistics the floor of the conv stacks stops falling at ~6e-3 however small
    the branches are made: it is set by the ~8 layers EVERY signal passes in series (stem, the four projection shortcuts,
    the three head layers; X3D-M: head 4.5e-3, shortcuts 3.4e-3, all 26 branches together 2.2e-3): a bf16 filter's
    rounding error times the non-negative (post-ReLU) mean of its input is a per-channel constant that the global
    average pool does not average away.  (2) The last row -- scaling gamma after the calibration and leaving every
    downstream running statistic stale, which is how the 3.6e-3 / 3.5e-3 / 2.0e-3 of the round-3 review were obtained
    -- reaches its low floor by switching the signal path off: the logits of two different clips then differ by 2e-3
    and a 5 % defect in a conv_c moves them by 6.6e-4, so a 1e-2 gate on that instance cannot fail for any kernel
    defect below ~75 % in a layer.  This fill keeps the statistics calibrated (input-sensitive logits, clips differ by
    6.6e-2) and takes the smallest branch scale at which every BASELINE conv stack sits below 1e-2 in bf16 storage.
    Models without flagged norms (MViT) get exactly calibrated_fill."""
    import torch.nn as nn
    reference_style_fill(model, seed)
    lo, hi = final_gamma
    for name, mod in model.named_modules():
        if isinstance(mod, nn.modules.batchnorm._BatchNorm) and getattr(mod, "block_final_bn", False):
            mod.weight.copy_(_uniform(mod.weight.shape, lo, hi, _gen(name + "/final_gamma", seed)))
    return _calibrate_running_stats(model, calib_input)
