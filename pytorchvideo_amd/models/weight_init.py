"""Weight initialisation run by every factory (reference: pytorchvideo/models/weight_init.py).

fvcore's `c2_msra_fill` / `c2_xavier_fill` (third-party, absent) are restated from their
published definition: kaiming_normal_(fan_out, relu) / kaiming_uniform_(a=1), bias <- 0.
"""
import torch.nn as nn


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def _init_resnet_weights(model, fc_init_std=0.01):
    """Conv: MSRA; BN: gamma 1 (0 for the block-final BN flagged `block_final_bn`), beta 0;
    Linear: N(0, std) or xavier when flagged (reference: weight_init.py:8-47)."""
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            c2_msra_fill(m)
        elif isinstance(m, nn.modules.batchnorm._NormBase):
            if m.weight is not None:
                m.weight.data.fill_(0.0 if getattr(m, "block_final_bn", False) else 1.0)
            if m.bias is not None:
                m.bias.data.zero_()
        if isinstance(m, nn.Linear):
            if getattr(m, "xavier_init", False):
                c2_xavier_fill(m)
            else:
                m.weight.data.normal_(mean=0.0, std=fc_init_std)
            if m.bias is not None:
                m.bias.data.zero_()
    return model


def _init_vit_weights(model, trunc_normal_std=0.02):
    """Linear / positional tables: trunc-normal; LayerNorm: (1, 0) (reference: weight_init.py:50-69)."""
    from ..layers.positional_encoding import SpatioTemporalClsPositionalEncoding

    for m in model.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=trunc_normal_std)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, SpatioTemporalClsPositionalEncoding):
            for p in m.parameters():
                nn.init.trunc_normal_(p, std=trunc_normal_std)


def init_net_weights(model, init_std=0.01, style="resnet"):
    assert style in ["resnet", "vit"]
    if style == "resnet":
        return _init_resnet_weights(model, init_std)
    return _init_vit_weights(model, init_std)
