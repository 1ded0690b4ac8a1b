"""Stems (reference: pytorchvideo/models/stem.py)."""
import torch.nn as nn

from ..layers.utils import set_attributes


def _norm(norm, n, eps, momentum):
    return None if norm is None else norm(num_features=n, eps=eps, momentum=momentum)


def _act(act):
    return None if act is None else act()


class ResNetBasicStem(nn.Module):
    """conv -> norm -> activation -> pool, each optional but conv (reference: stem.py:215-260)."""

    def __init__(self, *, conv=None, norm=None, activation=None, pool=None) -> None:
        super().__init__()
        set_attributes(self, locals())
        assert self.conv is not None

    def forward(self, x):        # spelled out stage by stage: TorchScript compiles absent (None) stages away
        x = self.conv(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        if self.pool is not None:
            x = self.pool(x)
        return x


def create_res_basic_stem(*, in_channels, out_channels, conv_kernel_size=(3, 7, 7), conv_stride=(1, 2, 2),
                          conv_padding=(1, 3, 3), conv_bias=False, conv=nn.Conv3d, pool=nn.MaxPool3d,
                          pool_kernel_size=(1, 3, 3), pool_stride=(1, 2, 2), pool_padding=(0, 1, 1),
                          norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, activation=nn.ReLU):
    """Conv3d + BN + ReLU + MaxPool3d stem (reference: stem.py:11-107)."""
    return ResNetBasicStem(
        conv=conv(in_channels=in_channels, out_channels=out_channels, kernel_size=conv_kernel_size,
                  stride=conv_stride, padding=conv_padding, bias=conv_bias),
        norm=_norm(norm, out_channels, norm_eps, norm_momentum),
        activation=_act(activation),
        pool=None if pool is None else pool(kernel_size=pool_kernel_size, stride=pool_stride,
                                            padding=pool_padding),
    )


class PatchEmbed(nn.Module):
    """Patch embedding: conv then (B,C,T,H,W) -> (B, T*H*W, C) (reference: stem.py:263-292)."""

    def __init__(self, *, patch_model=None) -> None:
        super().__init__()
        set_attributes(self, locals())
        assert self.patch_model is not None

    def forward(self, x):
        return self.patch_model(x).flatten(2).transpose(1, 2)


def create_conv_patch_embed(*, in_channels, out_channels, conv_kernel_size=(1, 16, 16),
                            conv_stride=(1, 4, 4), conv_padding=(1, 7, 7), conv_bias=True, conv=nn.Conv3d):
    """(reference: stem.py:295-338)"""
    return PatchEmbed(patch_model=conv(in_channels=in_channels, out_channels=out_channels,
                                       kernel_size=conv_kernel_size, stride=conv_stride,
                                       padding=conv_padding, bias=conv_bias))
