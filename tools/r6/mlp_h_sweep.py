"""Fixed cost vs per-hidden-block cost of the fused MLP kernels: time against H at M = 25096, C = 384 (run on the GPU box)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from bench_mlp import timed  # noqa: E402
from pytorchvideo_amd import _lib as L  # noqa: E402
from pytorchvideo_amd.accelerator.mi355x.emit_mvit import pack_mlp_weights  # noqa: E402

lib = L.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
M, Cin, Cout = 25096, 384, 384
x = torch.randn(M, Cin, generator=g).cuda()
y = torch.empty(M, Cout, device="cuda")
b2, gam, bet = torch.zeros(Cout, device="cuda"), torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
for H in (32, 64, 256, 768, 1536, 3072):
    w1, w2 = torch.randn(H, Cin, generator=g) * Cin ** -0.5, torch.randn(Cout, H, generator=g) * H ** -0.5
    b1 = torch.randn(H, generator=g)
    for layout in (16,):       # (the 32-rows-per-wave kernel this sweep was first run against is gone: profiles/r6/mlp_h_sweep_call5.txt)
        img = pack_mlp_weights(w1, b1, w2).cuda()
        d = L.MlpDesc()
        d.x, d.w12, d.y, d.b2 = x.data_ptr(), img.data_ptr(), y.data_ptr(), b2.data_ptr()
        d.ln_gamma, d.ln_beta, d.ln_eps = gam.data_ptr(), bet.data_ptr(), 1e-6
        d.M, d.C, d.H, d.Cout, d.ldx, d.ldr, d.ldy, d.act, d.dtype = M, Cin, H, Cout, Cin, Cout, Cout, L.ACT_GELU, L.PV_BF16
        us = timed(lambda: L.check(lib.pv_mlp_rows(C.byref(d), st)), 30)
        print("H=%5d rows/wave=%d: %7.1f us" % (H, layout, us), flush=True)
