"""Host-side helpers for synthetic benchmarking."""
import zlib

import torch
import torch.nn as nn


@torch.no_grad()
def randomize_norm_stats(model, seed=0):
    """A freshly constructed ResNet-family model is degenerate (block-final BN gamma = 0,
    reference models/weight_init.py:34-35): every residual branch outputs exactly 0.  For
    synthetic benchmarks give every BatchNorm non-trivial statistics (the reference tests'
    `rand_init_bn` recipe, tests/test_fuse_bn.py:58-63) so that all kernels see real data."""
    g = torch.Generator().manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, nn.modules.batchnorm._BatchNorm):
            for t, lo, hi in ((mod.weight, 0.5, 1.5), (mod.bias, -0.5, 0.5),
                              (mod.running_var, 0.5, 1.5), (mod.running_mean, -0.5, 0.5)):
                if t is not None:
                    t.copy_(torch.rand(t.shape, generator=g) * (hi - lo) + lo)
    return model


def _keyed(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


@torch.no_grad()
def synthetic_trained_like_weights(model, calib_input, seed=0, final_gamma=(0.05, 0.2)):
    """Random weights conditioned like a trained checkpoint, for synthetic benchmarking (no network: no checkpoints).

    Conv3d: MSRA normal (the factories' own init, models/weight_init.py); every BatchNorm: gamma ~ U(0.5, 1.5),
    beta ~ U(-0.5, 0.5) (reference tests/test_fuse_bn.py:58-63) except the block-final BatchNorms the factories flag
    `block_final_bn` (zero-initialised by the reference, models/weight_init.py:34-35): gamma ~ U(0.05, 0.2); the final
    `proj` Linear gets std 0.05 so the logits are O(1); then one forward of `calib_input` in training mode sets every
    running mean / variance to the data's, as training would have.  Every tensor is drawn from a generator keyed by
    its module name, so the values do not depend on construction order -- the parity tests build the SAME instance
    (tests/test_host.py pins this function to the test infrastructure's fill bit for bit) and
    tests/test_gpu_full_geometry.py checks it against the fp32 reference arithmetic at the bench batch."""
    def uni(shape, lo, hi, g):
        return torch.rand(shape, generator=g) * (hi - lo) + lo

    for name, mod in model.named_modules():
        if isinstance(mod, nn.modules.batchnorm._BatchNorm):
            g = _keyed(name, seed)
            mod.weight.copy_(uni(mod.weight.shape, 0.5, 1.5, g))
            mod.bias.copy_(uni(mod.bias.shape, -0.5, 0.5, g))
            mod.running_var.copy_(uni(mod.running_var.shape, 0.5, 1.5, g))
            mod.running_mean.copy_(uni(mod.running_mean.shape, -0.5, 0.5, g))
        elif isinstance(mod, nn.Conv3d):
            g = _keyed(name, seed)
            k = mod.kernel_size
            fan_out = mod.out_channels * k[0] * k[1] * k[2] // mod.groups
            mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / fan_out) ** 0.5)
            if mod.bias is not None:
                mod.bias.copy_(uni(mod.bias.shape, -0.1, 0.1, g))
        elif isinstance(mod, nn.Linear) and name.endswith("proj") and mod.out_features >= 100:
            mod.weight.copy_(torch.randn(mod.weight.shape, generator=_keyed(name, seed)) * 0.05)
    for name, mod in model.named_modules():
        if isinstance(mod, nn.modules.batchnorm._BatchNorm) and getattr(mod, "block_final_bn", False):
            mod.weight.copy_(uni(mod.weight.shape, final_gamma[0], final_gamma[1], _keyed(name + "/final_gamma", seed)))
    bns = [m for m in model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
    if bns:
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
