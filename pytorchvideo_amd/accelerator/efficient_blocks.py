"""Efficient-block contract of the accelerator plugin system (reference:
pytorchvideo/accelerator/efficient_blocks/efficient_block_base.py:8-35 and
no_op_convert_block.py:8-26).

An efficient block has two forms: the *original form* (identical maths to the vanilla
module, trainable, loads reference checkpoints) and the *deployable form* reached through
`convert()`, which is specialised to one input size and a target device.
"""
from abc import abstractmethod

import torch.nn as nn


class EfficientBlockBase(nn.Module):
    @abstractmethod
    def convert(self):
        pass

    @abstractmethod
    def forward(self):
        pass


class NoOpConvertBlock(EfficientBlockBase):
    """Wraps a module that needs no conversion so that the convert driver skips it."""

    def __init__(self, model: nn.Module):
        super().__init__()
        self.model = model

    def convert(self, *args, **kwargs):
        pass

    def forward(self, x):
        return self.model(x)
