"""Micro-benchmark of pv_bottleneck on the X3D stage shapes (16 clips x 16 frames: one sub-batch branch of the bench), both modes,
and -- with the development library (PV_MI355X_LIB=pytorchvideo_amd/_lib/dev/libpv_mi355x.so; timing only) -- the ablation builds
of the res4 kernel.  Run on the GPU box."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from bench_mlp import timed  # noqa: E402
from pytorchvideo_amd import _lib as L  # noqa: E402
from pytorchvideo_amd.accelerator.mi355x.emit import pack_bottleneck_operands  # noqa: E402

lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (cin, Cc, cout, HW, name) in ((96, 216, 96, 14, "res4"), (48, 108, 48, 28, "res3"), (24, 54, 24, 56, "res2")):
    ca, cb, cc = nn.Conv3d(cin, Cc, 1, bias=False), nn.Conv3d(Cc, Cc, 3, padding=1, groups=Cc, bias=False), nn.Conv3d(Cc, cout, 1, bias=False)
    ops = {k: v.cuda() for k, v in pack_bottleneck_operands(ca, None, cb, None, cc, None).items()}
    for B in (16, 32):
        T, H, W = 16, HW, HW
        x = torch.randn(B, T, H, W, cin).bfloat16().cuda()
        y = torch.empty(B, T, H, W, cout, dtype=torch.bfloat16, device="cuda")
        cp8 = (Cc + 7) // 8 * 8
        ym = torch.empty(B, T, H, W, cp8, dtype=torch.bfloat16, device="cuda")
        d = L.BottleneckDesc()
        d.x, d.y, d.residual = x.data_ptr(), y.data_ptr(), x.data_ptr()
        for k, v in ops.items():
            setattr(d, k, v.data_ptr())
        d.x_bs, d.y_bs, d.r_bs, d.ldx, d.ldy, d.ldr = T * H * W * cin, T * H * W * cout, T * H * W * cin, cin, cout, cin
        d.B, d.T, d.H, d.W, d.cin, d.C, d.cout = B, T, H, W, cin, Cc, cout
        d.act_a, d.act_b, d.act_out, d.dtype, d.mode = L.ACT_RELU, L.ACT_SWISH, L.ACT_RELU, L.PV_BF16, L.BLOCK_FULL
        for abl in ((0, 1, 2, 3, 4, 6, 0) if (name == "res4" and B == 16) else (0, 0)):
            L.tune(block_abl=abl)
            us = timed(lambda: L.check(lib.pv_bottleneck(C.byref(d), st)), 30)
            print("pv_bottleneck %s full B=%d abl=%d: %7.1f us" % (name, B, abl, us), flush=True)
        L.tune(block_abl=0)
        nblk = lib.pv_bottleneck_psum_blocks(C.byref(d))
        psum = torch.empty(B, nblk, cp8, device="cuda")
        d.y, d.y_bs, d.ldy, d.mode, d.psum = ym.data_ptr(), T * H * W * cp8, cp8, L.BLOCK_AB, psum.data_ptr()
        us = timed(lambda: L.check(lib.pv_bottleneck(C.byref(d), st)), 30)
        print("pv_bottleneck %s conv_ab + squeeze sums B=%d: %7.1f us" % (name, B, us), flush=True)
