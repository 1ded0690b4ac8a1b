"""Micro-benchmark of pv_attention on MViT-B 32x3 geometries (dev tool)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorchvideo_amd import _lib as L

SHAPES = [("b0  h1 Nq50177 Nk785", 8, 1, 50177, 785), ("b1  h2 Nq12545 Nk3137", 8, 2, 12545, 3137),
          ("b2  h2 Nq12545 Nk785", 8, 2, 12545, 785), ("b3  h4 Nq3137 Nk3137", 8, 4, 3137, 3137),
          ("b4  h4 Nq3137 Nk785", 8, 4, 3137, 785), ("b14 h8 Nq785 Nk3137", 8, 8, 785, 3137),
          ("b15 h8 Nq785 Nk785", 8, 8, 785, 785)]


def run(label, B, heads, Nq, Nk, hd=96, iters=20):
    Cw = heads * hd
    q = torch.randn(B, Nq, Cw, device="cuda").bfloat16()
    k = torch.randn(B, Nk, Cw, device="cuda").bfloat16()
    v = torch.randn(B, Nk, Cw, device="cuda").bfloat16()
    o = torch.empty_like(q)
    d = L.AttentionDesc()
    d.q, d.k, d.v, d.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    d.q_bs, d.k_bs, d.v_bs, d.o_bs = Nq * Cw, Nk * Cw, Nk * Cw, Nq * Cw
    d.ldq = d.ldk = d.ldv = d.ldo = Cw
    d.B, d.heads, d.head_dim, d.Nq, d.Nk = B, heads, hd, Nq, Nk
    d.scale, d.residual_q, d.dtype = hd ** -0.5, 0, L.PV_BF16
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        L.check(lib.pv_attention(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.pv_attention(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 4.0 * B * heads * Nq * Nk * hd
    print("%-26s %8.3f ms %8.1f TF/s" % (label, ms, flops / ms / 1e9), flush=True)
    return ms


if __name__ == "__main__":
    # A/B of the two bf16 kernels: attn_w64=0 routes to attn_pipe_kernel (pv_attn.hip), 1 to attn_w64_kernel (pv_attn64.hip)
    # (dev library, PV_MI355X_LIB=.../_lib/dev/libpv_mi355x.so: "abl<bits>" selects a timing-only ablation of attn_w64_kernel)
    for arg in (sys.argv[1:] or ["0", "1"]):
        if arg.startswith("abl"):      # abl<bits> (form 1) or abl<bits>f<form>
            bits, _, form = arg[3:].partition("f")
            L.tune(attn_w64=int(form or 1), attn_abl=int(bits))
        else:
            L.tune(attn_w64=int(arg))
        print("attn_w64 = %s" % arg)
        tot = sum(run(*s) * n for s, n in zip(SHAPES, (1, 1, 1, 1, 10, 1, 1)))
        print("MViT-B 32x3 attention total per step: %.3f ms" % tot)
