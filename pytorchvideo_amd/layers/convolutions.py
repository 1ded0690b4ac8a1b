"""Convolution building blocks (reference: pytorchvideo/layers/convolutions.py)."""
import torch
import torch.nn as nn

from .utils import set_attributes


class ConvReduce3D(nn.Module):
    """N parallel Conv3d over the same input, combined by sum or channel-concat
    (reference: layers/convolutions.py:11-85).  Per-branch options are tuples aligned with
    `kernel_size`; a `None` entry leaves the nn.Conv3d default."""

    _OPTIONAL = ("stride", "padding", "dilation", "groups", "bias", "padding_mode")

    def __init__(self, *, in_channels, out_channels, kernel_size, stride=None, padding=None,
                 padding_mode=None, dilation=None, groups=None, bias=None, reduction_method="sum"):
        super().__init__()
        assert reduction_method in ("sum", "cat")
        self.reduction_method = reduction_method
        given = dict(stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias,
                     padding_mode=padding_mode)
        branches = []
        for i, ks in enumerate(kernel_size):
            kwargs = {k: given[k][i] for k in self._OPTIONAL
                      if given[k] is not None and given[k][i] is not None}
            branches.append(nn.Conv3d(in_channels, out_channels, ks, **kwargs))
        self.convs = nn.ModuleList(branches)

    def forward(self, x):
        outs = [conv(x) for conv in self.convs]
        if self.reduction_method == "sum":
            return torch.stack(outs, dim=0).sum(dim=0, keepdim=False)
        return torch.cat(outs, dim=1)


class Conv2plus1d(nn.Module):
    """(2+1)D convolution: conv_t -> norm -> activation -> conv_xy, or with the two convs
    swapped when `conv_xy_first` (reference: layers/convolutions.py:191-237)."""

    def __init__(self, *, conv_t=None, norm=None, activation=None, conv_xy=None, conv_xy_first=False):
        super().__init__()
        set_attributes(self, locals())
        assert self.conv_t is not None
        assert self.conv_xy is not None

    def forward(self, x):
        first, second = (self.conv_xy, self.conv_t) if self.conv_xy_first else (self.conv_t, self.conv_xy)
        x = first(x)
        if self.norm:
            x = self.norm(x)
        if self.activation:
            x = self.activation(x)
        return second(x)


def create_conv_2plus1d(*, in_channels, out_channels, inner_channels=None, conv_xy_first=False,
                        kernel_size=(3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1), bias=False,
                        dilation=(1, 1, 1), groups=1, norm=nn.BatchNorm3d, norm_eps=1e-5,
                        norm_momentum=0.1, activation=nn.ReLU):
    """Factory for Conv2plus1d (reference: layers/convolutions.py:88-188)."""
    if inner_channels is None:
        inner_channels = out_channels
    assert groups == 1, "Support for groups is not implemented in R2+1 convolution layer"
    assert max(dilation) == 1 and min(dilation) == 1, (
        "Support for dillaiton is not implemented in R2+1 convolution layer")
    # channel plan: in -> inner -> out, in execution order
    t_io = (inner_channels, out_channels) if conv_xy_first else (in_channels, inner_channels)
    xy_io = (in_channels, inner_channels) if conv_xy_first else (inner_channels, out_channels)
    conv_t = nn.Conv3d(t_io[0], t_io[1], kernel_size=(kernel_size[0], 1, 1), stride=(stride[0], 1, 1),
                       padding=(padding[0], 0, 0), bias=bias)
    conv_xy = nn.Conv3d(xy_io[0], xy_io[1], kernel_size=(1, kernel_size[1], kernel_size[2]),
                        stride=(1, stride[1], stride[2]), padding=(0, padding[1], padding[2]), bias=bias)
    return Conv2plus1d(
        conv_t=conv_t,
        norm=None if norm is None else norm(num_features=inner_channels, eps=norm_eps, momentum=norm_momentum),
        activation=None if activation is None else activation(),
        conv_xy=conv_xy,
        conv_xy_first=conv_xy_first,
    )
