"""Helpers for the -m gpu tests: raw C-ABI calls on torch-owned device memory."""
import ctypes as C

import torch

from pytorchvideo_amd import _lib as L


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pv_dtype(t):
    return L.PV_BF16 if t.dtype == torch.bfloat16 else L.PV_F32


def call(fn_name, desc):
    lib = L.lib()
    L.check(getattr(lib, fn_name)(C.byref(desc), stream()), fn_name)
    torch.cuda.synchronize()


def rel_err(got, want):
    want = want.float().cpu()
    return (got.float().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
