"""Efficient-block contract of the accelerator plugin system (reference:
pytorchvideo/accelerator/efficient_blocks/efficient_block_base.py:8-35 and
no_op_convert_block.py:8-26).

An efficient block has two forms: the *original form* (identical maths to the vanilla
module, trainable, loads reference checkpoints) and the *deployable form* reached through
`convert()`, which is specialised to one input size and a target device.

Inside a PyTorchVideo installation the classes below ARE the reference's own: its convert
driver finds efficient blocks with `isinstance(module, EfficientBlockBase)`
(deployment/mobile_cpu/utils/model_conversion.py:66), so MI355X blocks have to derive from
that very class to be seen by it.  Without PyTorchVideo the same contract is defined here.
"""
from abc import abstractmethod

import torch.nn as nn

try:
    from pytorchvideo.accelerator.efficient_blocks.efficient_block_base import EfficientBlockBase
    from pytorchvideo.accelerator.efficient_blocks.no_op_convert_block import NoOpConvertBlock
    INSIDE_PYTORCHVIDEO = True
except ImportError:
    INSIDE_PYTORCHVIDEO = False

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
