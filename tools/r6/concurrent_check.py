"""Is pv_bottleneck bit-reproducible while another kernel runs beside it?  Two streams, each running the kernel on its own buffers,
repeated; every repeat of stream A's output is compared with a quiet (single-stream) run.  Run on the GPU box."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from pytorchvideo_amd import _lib as L  # noqa: E402
from pytorchvideo_amd.accelerator.mi355x.emit import pack_bottleneck_operands  # noqa: E402

lib = L.lib()
L.tune(block_stages=0x1c, block_stages_ab=0x1c)


def desc(cin, Cc, cout, B, T, H, W, mode, ops, x, y, psum, gate):
    d = L.BottleneckDesc()
    d.x, d.y, d.residual = x.data_ptr(), y.data_ptr(), (x.data_ptr() if mode == L.BLOCK_FULL else None)
    for k, v in ops.items():
        setattr(d, k, v.data_ptr())
    ldy = cout if mode != L.BLOCK_AB else (Cc + 7) // 8 * 8
    d.x_bs, d.y_bs, d.r_bs, d.ldx, d.ldy, d.ldr = T * H * W * x.shape[-1], T * H * W * ldy, T * H * W * x.shape[-1], x.shape[-1], ldy, x.shape[-1]
    d.B, d.T, d.H, d.W, d.cin, d.C, d.cout = B, T, H, W, cin, Cc, cout
    d.act_a, d.act_b, d.act_out, d.dtype, d.mode = L.ACT_RELU, L.ACT_SWISH, L.ACT_RELU, L.PV_BF16, mode
    if psum is not None:
        d.psum = psum.data_ptr()
    return d


for (cin, Cc, cout, HW, name) in ((24, 54, 24, 56, "res2"), (48, 108, 48, 28, "res3"), (96, 216, 96, 14, "res4")):
    ca, cb, cc = nn.Conv3d(cin, Cc, 1, bias=False), nn.Conv3d(Cc, Cc, 3, padding=1, groups=Cc, bias=False), nn.Conv3d(Cc, cout, 1, bias=False)
    ops = {k: v.cuda() for k, v in pack_bottleneck_operands(ca, None, cb, None, cc, None).items()}
    B, T, H, W = 16, 16, HW, HW
    for mode, mname in ((L.BLOCK_FULL, "full"), (L.BLOCK_AB, "conv_ab")):
        cy = cout if mode == L.BLOCK_FULL else (Cc + 7) // 8 * 8
        xs = [torch.randn(B, T, H, W, cin).bfloat16().cuda() for _ in range(2)]
        ys = [torch.zeros(B, T, H, W, cy, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        d0 = desc(cin, Cc, cout, B, T, H, W, mode, ops, xs[0], ys[0], None, None)
        nblk = lib.pv_bottleneck_psum_blocks(C.byref(d0))
        ps = [torch.zeros(B, nblk, (Cc + 7) // 8 * 8, device="cuda") for _ in range(2)]
        ds = [desc(cin, Cc, cout, B, T, H, W, mode, ops, xs[i], ys[i], ps[i], None) for i in range(2)]
        st = [torch.cuda.Stream() for _ in range(2)]
        L.check(lib.pv_bottleneck(C.byref(ds[0]), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        quiet = ys[0].clone()
        bad = []
        for rep in range(12):
            ys[0].zero_()
            torch.cuda.synchronize()
            for k in range(3):
                for i in range(2):
                    L.check(lib.pv_bottleneck(C.byref(ds[i]), C.c_void_p(st[i].cuda_stream)))
            torch.cuda.synchronize()
            bad.append(int((ys[0] != quiet).sum().item()))
        print("%s %s: elements differing from the quiet run, per repeat: %s" % (name, mname, bad), flush=True)
