"""Learned cls token + (separable) spatio-temporal position embedding for MViT
(reference: pytorchvideo/layers/positional_encoding.py:47-136).  The sin-cos helpers of the
reference (:139-244) are not reachable from the factories on the path."""
from typing import Tuple

import torch
from torch import nn


class SpatioTemporalClsPositionalEncoding(nn.Module):
    def __init__(self, embed_dim: int, patch_embed_shape: Tuple[int, int, int], sep_pos_embed: bool = False,
                 has_cls: bool = True) -> None:
        super().__init__()
        assert len(patch_embed_shape) == 3, "Patch_embed_shape should be in the form of (T, H, W)."
        self.cls_embed_on = has_cls
        self.sep_pos_embed = sep_pos_embed
        self._patch_embed_shape = tuple(patch_embed_shape)
        self.num_spatial_patch = patch_embed_shape[1] * patch_embed_shape[2]
        self.num_temporal_patch = patch_embed_shape[0]
        n_tokens = self.num_spatial_patch * self.num_temporal_patch
        # parameters are created in the reference's order (state_dict key order)
        if has_cls:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            n_tokens += 1
        else:
            self.cls_token = torch.tensor(0)
        empty = torch.tensor([])
        if sep_pos_embed:
            self.pos_embed_spatial = nn.Parameter(torch.zeros(1, self.num_spatial_patch, embed_dim))
            self.pos_embed_temporal = nn.Parameter(torch.zeros(1, self.num_temporal_patch, embed_dim))
            self.pos_embed_class = nn.Parameter(torch.zeros(1, 1, embed_dim)) if has_cls else empty
            self.pos_embed = empty
        else:
            self.pos_embed = nn.Parameter(torch.zeros(1, n_tokens, embed_dim))
            self.pos_embed_spatial = self.pos_embed_temporal = self.pos_embed_class = empty

    @torch.jit.export
    def patch_embed_shape(self) -> Tuple[int, int, int]:
        return self._patch_embed_shape

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        if self.cls_embed_on:
            x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        if not self.sep_pos_embed:
            return x + self.pos_embed
        # token (t, s) gets spatial[s] + temporal[t]
        pos = self.pos_embed_spatial.repeat(1, self.num_temporal_patch, 1) + torch.repeat_interleave(
            self.pos_embed_temporal, self.num_spatial_patch, dim=1)
        if self.cls_embed_on:
            pos = torch.cat([self.pos_embed_class, pos], 1)
        return x + pos
