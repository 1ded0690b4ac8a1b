"""Micro-benchmark of the fused row kernels (pv_mlp_rows, pv_ln_linear_rows) on the MViT-B shapes, with the ablation
builds of csrc/pv_mlp.hip (pv_tune "mlp_abl": timing only, wrong results) -- run on the GPU box.

    python tools/bench_mlp.py [--iters 50]

The ablation / A-B builds exist only in the development variant of the library:
    python -m pytorchvideo_amd.csrc.build --variant dev && PV_MI355X_LIB=pytorchvideo_amd/_lib/dev/libpv_mi355x.so python tools/bench_mlp.py
(with the product library every `abl` line times the same, shipped kernel).
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from pytorchvideo_amd import _lib as L  # noqa: E402
from pytorchvideo_amd.accelerator.mi355x.emit_mvit import pack_ln_linear_weights, pack_mlp_weights  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    for M, Cin, H, Cout in ((25096, 384, 1536, 384), (100360, 192, 768, 192), (401416, 96, 384, 192)):
        ln = Cin == Cout
        w1, w2 = torch.randn(H, Cin, generator=g) * Cin ** -0.5, torch.randn(Cout, H, generator=g) * H ** -0.5
        img = pack_mlp_weights(w1, torch.randn(H, generator=g), w2).cuda()
        x = torch.randn(M, Cin, generator=g).cuda()
        xb = x.bfloat16()
        y = torch.empty(M, Cout, device="cuda")
        r = torch.randn(M, Cout, generator=g).cuda()
        b2, gam, bet = torch.zeros(Cout, device="cuda"), torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        d = L.MlpDesc()
        d.x, d.w12, d.y, d.b2 = (x if ln else xb).data_ptr(), img.data_ptr(), y.data_ptr(), b2.data_ptr()
        d.residual = None if ln else r.data_ptr()
        d.ln_gamma, d.ln_beta, d.ln_eps = (gam.data_ptr(), bet.data_ptr(), 1e-6) if ln else (None, None, 0.0)
        d.M, d.C, d.H, d.Cout, d.ldx, d.ldr, d.ldy, d.act, d.dtype = M, Cin, H, Cout, Cin, Cout, Cout, L.ACT_GELU, L.PV_BF16
        flops = 2.0 * M * H * (Cin + Cout)
        # ablations (development variant of the library only; timing builds with WRONG results): 1 no activation, 2 no weight
        # streaming, 3 no phase B, 4 no phase A, 5 no barrier, 6 no LDS fragment reads, 7 no MFMA at all
        for abl in ((0, 1, 2, 3, 4, 5, 6, 7, 0) if (Cin, Cout) == (384, 384) else (0, 0, 0)):
            L.tune(mlp_abl=abl)
            us = timed(lambda: L.check(lib.pv_mlp_rows(C.byref(d), st)), a.iters)
            print("mlp_rows M=%d %d->%d->%d ln=%d abl=%d: %8.1f us  %7.1f TF/s" % (M, Cin, H, Cout, ln, abl, us, flops / us / 1e6), flush=True)
        L.tune(mlp_abl=0)
    for M, Cin, N in ((25096, 384, 1152), (401416, 192, 576), (401416, 96, 288), (6280, 768, 2304)):
        w = torch.randn(N, Cin, generator=g) * Cin ** -0.5
        img = pack_ln_linear_weights(w, torch.randn(N, generator=g)).cuda()
        x = torch.randn(M, Cin, generator=g).cuda()
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        gam, bet = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
        d = L.LnLinearDesc()
        d.x, d.wb, d.y, d.ln_gamma, d.ln_beta = x.data_ptr(), img.data_ptr(), y.data_ptr(), gam.data_ptr(), bet.data_ptr()
        d.M, d.C, d.N, d.ldx, d.ldy, d.act, d.dtype, d.ln_eps = M, Cin, N, Cin, N, L.ACT_NONE, L.PV_BF16, 1e-6
        us = timed(lambda: L.check(lib.pv_ln_linear_rows(C.byref(d), st)), a.iters)
        print("ln_linear M=%d %d->%d: %8.1f us  %7.1f TF/s  %6.0f GB/s" % (M, Cin, N, us, 2.0 * M * Cin * N / us / 1e6,
                                                                          (4.0 * M * Cin + 2.0 * M * N) / us / 1e3), flush=True)


if __name__ == "__main__":
    main()
