"""torch.hub entry points with the names of the reference's hubconf.py (`torch.hub.load(<this repo>, "x3d_m")`).
`pretrained=True` needs `checkpoint_path=` (a local copy of the model-zoo file): there is no download here."""
dependencies = ["torch"]
from pytorchvideo_amd.models.hub import (  # noqa: F401, E402
    c2d_r50, csn_r101, i3d_r50, mvit_base_16, mvit_base_16x4, mvit_base_32x3, r2plus1d_r50, slow_r50, slow_r50_detection,
    slowfast_16x8_r101_50_50, slowfast_r101, slowfast_r50, slowfast_r50_detection, x3d_l, x3d_m, x3d_s, x3d_xs)
