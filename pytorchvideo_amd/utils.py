"""Host-side helpers for synthetic benchmarking."""
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
