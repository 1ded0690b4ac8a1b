"""Import shim for the *reference* package (TEST INFRASTRUCTURE ONLY, this container only).

`/root/reference` is not installed and three of its third-party imports are absent
(torchvision.ops.RoIAlign, fvcore.nn.weight_init, fvcore.nn.squeeze_excitation).  This
module injects stand-ins for exactly those symbols so that `import pytorchvideo.models`
runs the reference's own model code (the RoIAlign stand-in evaluates the restatement in
oracle/functional.py: detection goldens pin the reference's backbone, dilation and head
wiring, NOT the third-party op itself).  It is used by `tests/golden/make_golden.py` to
generate golden vectors; nothing in the product imports it and it never travels to the
GPU box (the reference tree does not exist there).

Restated third-party pieces (fvcore is un-pinned in reference `setup.py:54`):
  * SqueezeExcitation -- structure corroborated in-repo by
    pytorchvideo/layers/accelerator/mobile_cpu/attention.py:74-91 and by the state-dict
    keys `norm_b.1.block.{0,2}.{weight,bias}` used by the model zoo.
  * c2_msra_fill / c2_xavier_fill -- kaiming_normal_(fan_out, relu) / kaiming_uniform_(a=1), bias 0.
"""
import os
import sys
import types

import torch.nn as nn

REFERENCE_ROOT = os.environ.get("PV_REFERENCE_ROOT", "/root/reference")


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "pytorchvideo")):
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if "fvcore.nn.squeeze_excitation" in sys.modules and "pytorchvideo" in sys.modules:
        return
    tv = _mod("torchvision")
    tv.ops = _mod("torchvision.ops")

    class RoIAlign(nn.Module):  # the detection head's roi layer (models/head.py:8,212,318-322)
        def __init__(self, output_size=None, spatial_scale=1.0, sampling_ratio=0, aligned=False):
            super().__init__()
            self.output_size, self.spatial_scale = output_size, spatial_scale
            self.sampling_ratio, self.aligned = sampling_ratio, aligned

        def forward(self, x, boxes):
            from oracle.functional import roi_align
            return roi_align(x, boxes, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)

    tv.ops.RoIAlign = RoIAlign
    fv = _mod("fvcore")
    fv.nn = _mod("fvcore.nn")
    se = _mod("fvcore.nn.squeeze_excitation")
    wi = _mod("fvcore.nn.weight_init")
    di = _mod("fvcore.nn.distributed")

    class SqueezeExcitation(nn.Module):
        def __init__(self, num_channels, num_channels_reduced=None, reduction_ratio=2.0,
                     is_3d=False, activation=None):
            super().__init__()
            r = num_channels_reduced if num_channels_reduced is not None else int(
                num_channels // reduction_ratio)
            conv = nn.Conv3d if is_3d else nn.Conv2d
            self.is_3d = is_3d
            self.block = nn.Sequential(
                conv(num_channels, r, 1, bias=True),
                activation or nn.ReLU(),
                conv(r, num_channels, 1, bias=True),
                nn.Sigmoid(),
            )

        def forward(self, x):
            dims = [2, 3, 4] if self.is_3d else [2, 3]
            return x * self.block(x.mean(dim=dims, keepdim=True))

    se.SqueezeExcitation = SqueezeExcitation

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    def c2_xavier_fill(m):
        nn.init.kaiming_uniform_(m.weight, a=1)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    wi.c2_msra_fill, wi.c2_xavier_fill = c2_msra_fill, c2_xavier_fill
    di.differentiable_all_reduce = lambda x: x
    di.differentiable_all_gather = lambda x: [x]
    fv.nn.squeeze_excitation, fv.nn.weight_init, fv.nn.distributed = se, wi, di
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
