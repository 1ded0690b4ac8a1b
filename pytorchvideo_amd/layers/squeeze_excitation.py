"""3-D squeeze-excitation as used by X3D.

The reference imports `fvcore.nn.squeeze_excitation.SqueezeExcitation`
(pytorchvideo/models/x3d.py:9,190-198); fvcore is a third-party package, un-pinned in
reference setup.py:54 and absent from /root/reference.  Its published behaviour, restated:
`block = Sequential(Conv(C,Cr,1,bias), act, Conv(Cr,C,1,bias), Sigmoid())` and
`forward(x) = x * block(mean over the spatial(-temporal) dims)`.  The sub-module indices
(block.0 / block.2) are part of the model-zoo state_dict keys
(`...norm_b.1.block.{0,2}.{weight,bias}`) and are corroborated in-repo by
pytorchvideo/layers/accelerator/mobile_cpu/attention.py:74-91.
"""
import torch.nn as nn


class SqueezeExcitation(nn.Module):
    def __init__(self, num_channels, num_channels_reduced=None, reduction_ratio=2.0,
                 is_3d=False, activation=None):
        super().__init__()
        if num_channels_reduced is None:
            num_channels_reduced = int(num_channels // reduction_ratio)
        conv = nn.Conv3d if is_3d else nn.Conv2d
        self.is_3d = is_3d
        self.block = nn.Sequential(
            conv(num_channels, num_channels_reduced, kernel_size=1, stride=1, bias=True),
            nn.ReLU() if activation is None else activation,
            conv(num_channels_reduced, num_channels, kernel_size=1, stride=1, bias=True),
            nn.Sigmoid(),
        )

    def forward(self, x):
        dims = [2, 3, 4] if self.is_3d else [2, 3]
        return x * self.block(x.mean(dim=dims, keepdim=True))
