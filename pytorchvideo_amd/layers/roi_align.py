"""RoIAlign, the region-of-interest layer of the detection head.

The reference takes it from torchvision (`from torchvision.ops import RoIAlign`, pytorchvideo/models/head.py:8;
default `roi=` of create_res_roi_pooling_head, head.py:212).  torchvision is a third-party dependency that
is not part of the reference tree (un-pinned in its setup.py), so the op is restated here from its published
definition (torchvision.ops.roi_align, Mask R-CNN section 3): for box (batch index, x1, y1, x2, y2) and output
bin (i, j), the mean of grid_h x grid_w bilinearly interpolated samples placed regularly inside the bin, with
grid = sampling_ratio, or ceil(roi size / output size) when sampling_ratio <= 0.  `aligned=False` (torchvision's
default, and what the reference builds) scales the box without the half-pixel shift and clamps the roi size
to at least one feature pixel.  A sample further than one pixel outside the map contributes zero and still
counts in the mean.

This module is the *original form* (plain differentiable torch ops, any device).  The MI355X deploy form
replaces it -- together with the MaxPool2d that follows it -- by `pv_roi_align` (csrc/pv_roi.hip).
"""
import math
from typing import Tuple, Union

import torch
import torch.nn as nn


def _axis_samples(start, bin_size, bins, grid, size, dtype, device):
    """Sample coordinates of one axis: for every (bin, grid point) the two neighbour indices, their
    weights and whether the sample counts.  Returns (lo, hi, w_lo, w_hi) of length bins*grid."""
    b = torch.arange(bins, dtype=dtype, device=device).repeat_interleave(grid)
    g = torch.arange(grid, dtype=dtype, device=device).repeat(bins)
    c = start + b * bin_size + (g + 0.5) * bin_size / grid
    inside = (c >= -1.0) & (c <= size)
    c = c.clamp(min=0.0)
    lo = c.floor().long()
    edge = lo >= size - 1
    lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
    hi = torch.where(edge, lo, lo + 1)
    c = torch.where(edge, lo.to(dtype), c)
    w_hi = c - lo.to(dtype)
    w_lo = 1.0 - w_hi
    zero = torch.zeros_like(w_lo)
    return lo, hi, torch.where(inside, w_lo, zero), torch.where(inside, w_hi, zero)


def roi_align(input: torch.Tensor, boxes: torch.Tensor, output_size: Union[int, Tuple[int, int]],
              spatial_scale: float = 1.0, sampling_ratio: int = -1, aligned: bool = False) -> torch.Tensor:
    """input [B,C,H,W], boxes [R,5] -> [R,C,ph,pw] (same contract as torchvision.ops.roi_align)."""
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    ph, pw = int(output_size[0]), int(output_size[1])
    if input.dim() != 4 or boxes.dim() != 2 or boxes.shape[1] != 5:
        raise RuntimeError("roi_align expects input [B,C,H,W] and boxes [R,5]")
    B, C, H, W = input.shape
    out = input.new_zeros((boxes.shape[0], C, ph, pw))
    off = 0.5 if aligned else 0.0
    dt, dev = input.dtype if input.is_floating_point() else torch.float32, input.device
    for n in range(boxes.shape[0]):
        bi = int(boxes[n, 0])
        x1, y1, x2, y2 = [float(v) * spatial_scale - off for v in boxes[n, 1:].to(torch.float32)]
        roi_w, roi_h = x2 - x1, y2 - y1
        if not aligned:
            roi_w, roi_h = max(roi_w, 1.0), max(roi_h, 1.0)
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_h / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_w / pw))
        if gh <= 0 or gw <= 0:
            continue
        ylo, yhi, wylo, wyhi = _axis_samples(y1, roi_h / ph, ph, gh, H, dt, dev)
        xlo, xhi, wxlo, wxhi = _axis_samples(x1, roi_w / pw, pw, gw, W, dt, dev)
        f = input[bi].to(dt)                                    # [C,H,W]
        rows_lo, rows_hi = f.index_select(1, ylo), f.index_select(1, yhi)
        v = (rows_lo.index_select(2, xlo) * (wylo[:, None] * wxlo[None, :])
             + rows_lo.index_select(2, xhi) * (wylo[:, None] * wxhi[None, :])
             + rows_hi.index_select(2, xlo) * (wyhi[:, None] * wxlo[None, :])
             + rows_hi.index_select(2, xhi) * (wyhi[:, None] * wxhi[None, :]))   # [C, ph*gh, pw*gw]
        out[n] = v.view(C, ph, gh, pw, gw).sum(dim=(2, 4)).div(gh * gw).to(out.dtype)
    return out


class RoIAlign(nn.Module):
    """Same constructor and attributes as torchvision.ops.RoIAlign."""

    def __init__(self, output_size, spatial_scale: float, sampling_ratio: int, aligned: bool = False):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.aligned = aligned

    def forward(self, input: torch.Tensor, rois: torch.Tensor) -> torch.Tensor:
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)

    def extra_repr(self) -> str:
        return "output_size=%s, spatial_scale=%s, sampling_ratio=%s, aligned=%s" % (
            self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)
