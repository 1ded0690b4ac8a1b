"""Containers (reference: pytorchvideo/models/net.py)."""
from typing import List

import torch
import torch.nn as nn

from ..layers.utils import set_attributes
from .weight_init import init_net_weights


class Net(nn.Module):
    """A sequence of blocks, `model.blocks[i]` (reference: net.py:11-44).  Construction
    runs the ResNet-style initialiser over the whole tree."""

    def __init__(self, *, blocks: nn.ModuleList) -> None:
        super().__init__()
        assert blocks is not None
        self.blocks = blocks
        init_net_weights(self)

    def forward(self, x):
        for block in self.blocks:
            x = block(x)
        return x


class DetectionBBoxNetwork(nn.Module):
    """A backbone followed by a head that also takes bounding boxes (reference: net.py:47-74)."""

    def __init__(self, model: nn.Module, detection_head: nn.Module):
        super().__init__()
        self.model = model
        self.detection_head = detection_head

    def forward(self, x, bboxes: torch.Tensor):
        features = self.model(x)
        out = self.detection_head(features, bboxes)
        return out.view(out.shape[0], -1)


class MultiPathWayWithFuse(nn.Module):
    """Per-pathway blocks followed by a cross-pathway fusion (reference: net.py:77-122).
    With `inplace=True` the caller's list is updated in place, exactly like the reference."""

    def __init__(self, *, multipathway_blocks, multipathway_fusion, inplace=True) -> None:
        super().__init__()
        set_attributes(self, locals())

    def forward(self, x: List[torch.Tensor]):
        assert isinstance(x, list), "input for MultiPathWayWithFuse needs to be a list of tensors"
        out = x if self.inplace else [None] * len(x)
        for i, block in enumerate(self.multipathway_blocks):
            if block is not None:
                out[i] = block(x[i])
        if self.multipathway_fusion is not None:
            out = self.multipathway_fusion(out)
        return out
