"""Swish activation x*sigmoid(x) (reference: pytorchvideo/layers/swish.py:7-34)."""
import torch
import torch.nn as nn


class _SwishFn(torch.autograd.Function):
    """Memory-lean autograd form: only the input is saved; sigmoid is recomputed."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x * torch.sigmoid(x)

    @staticmethod
    def backward(ctx, grad_out):
        (x,) = ctx.saved_tensors
        s = torch.sigmoid(x)
        return grad_out * (s * (1 + x * (1 - s)))


SwishFunction = _SwishFn


class Swish(nn.Module):
    def forward(self, x):
        return _SwishFn.apply(x)
