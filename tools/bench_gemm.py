"""Micro-benchmark of pv_conv3d as a plain GEMM / implicit-GEMM conv on the MI355X (dev tool).
    python tools/bench_gemm.py            # MViT-B / SlowFast / X3D shapes
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorchvideo_amd import _lib as L

ACT = L.ACT_RELU     # --act=gelu|none|relu

SHAPES = [  # (label, B, T,H,W in, cin, cout, k(t,h,w), s(t,h,w), p)
    ("mvit qkv b0   M401k K96  N288", 8, 1, 1, 50177, 96, 288, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit fc1 b0   M401k K96  N384", 8, 1, 1, 50177, 96, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit fc2 b0   M401k K384 N192", 8, 1, 1, 50177, 384, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit qkv b1   M401k K192 N576", 8, 1, 1, 50177, 192, 576, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit fc1 b2   M100k K192 N768", 8, 1, 1, 12545, 192, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit qkv b4   M25k  K384 N1152", 8, 1, 1, 3137, 384, 1152, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit fc1 b4   M25k  K384 N1536", 8, 1, 1, 3137, 384, 1536, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit fc2 b4   M25k  K1536 N384", 8, 1, 1, 3137, 1536, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit fc1 b15  M6k   K768 N3072", 8, 1, 1, 785, 768, 3072, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit fc2 b15  M6k   K3072 N768", 8, 1, 1, 785, 3072, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("ksweep M25k K64   N1152", 8, 1, 1, 3137, 64, 1152, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("ksweep M25k K768  N1152", 8, 1, 1, 3137, 768, 1152, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("ksweep M25k K1536 N1152", 8, 1, 1, 3137, 1536, 1152, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("ksweep M25k K3072 N1152", 8, 1, 1, 3137, 3072, 1152, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("ksweep M100k K384 N1152", 8, 1, 1, 12545, 384, 1152, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("big gemm      M32k  K4096 N4096", 1, 1, 1, 32768, 4096, 4096, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("sf conv_a res4 slow 3x1x1 1024->256", 16, 8, 16, 16, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("sf conv_b res4 slow 1x3x3 256->256", 16, 8, 16, 16, 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("sf conv_b res2 slow 1x3x3 64->64", 16, 8, 64, 64, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("sf conv_c res2 slow 64->256", 16, 8, 64, 64, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("sf conv_a res3 slow 512->128", 16, 8, 32, 32, 512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("sf conv_b res3 slow 1x3x3 128->128", 16, 8, 32, 32, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("sf conv_c res3 slow 128->512", 16, 8, 32, 32, 128, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("sf conv_c res4 slow 256->1024", 16, 8, 16, 16, 256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("sf conv_a res5 slow 3x1x1 2048->512", 16, 8, 8, 8, 2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("sf conv_b res5 slow 1x3x3 512->512", 16, 8, 8, 8, 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("sf conv_c res5 slow 512->2048", 16, 8, 8, 8, 512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("sf conv_a res4 first 3x1x1 640->256", 16, 8, 32, 32, 640, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("mvit proj b4  M25k  K384 N384", 8, 1, 1, 3137, 384, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("mvit qkv b14  M25k  K768 N2304", 8, 1, 1, 3137, 768, 2304, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("hb mvit qkv b4   M12k K384 N1152", 4, 1, 1, 3137, 384, 1152, (1, 1, 1), (1, 1, 1), (0, 0, 0)),   # hb: one of the two sub-batch branches
    ("hb mvit proj b4  M12k K384 N384", 4, 1, 1, 3137, 384, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("hb mvit qkv b14  M12k K768 N2304", 4, 1, 1, 3137, 768, 2304, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("hb mvit qkv b15  M3k  K768 N2304", 4, 1, 1, 785, 768, 2304, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("hb mvit proj b15 M3k  K768 N768", 4, 1, 1, 785, 768, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("hb mvit fc1 b15  M3k  K768 N3072", 4, 1, 1, 785, 768, 3072, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("hb mvit fc2 b15  M3k  K3072 N768", 4, 1, 1, 785, 3072, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("real sf shortcut res3 320->512 s122 in64", 16, 8, 64, 64, 320, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),   # real: the model's grids
    ("real sf conv_a res3 first 320->128 in64", 16, 8, 64, 64, 320, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("real sf shortcut res4 640->1024 s122 in32", 16, 8, 32, 32, 640, 1024, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("real sf shortcut res5 1280->2048 s122 in16", 16, 8, 16, 16, 1280, 2048, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("real sf conv_b res3 first 1x3x3 128->128 s122", 16, 8, 64, 64, 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("real sf conv_a res2 256->64 in64", 16, 8, 64, 64, 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("sf stem slow 1x7x7 3->64", 16, 8, 256, 256, 8, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
    ("sf stem fast 5x7x7 3->8", 16, 32, 256, 256, 8, 8, (5, 7, 7), (1, 2, 2), (2, 3, 3)),
    ("x3d stem 1x3x3 3->24", 32, 16, 224, 224, 8, 24, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("x3d conv_a s5 192->432", 32, 16, 7, 7, 192, 432, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("x3d conv_c s4 216->96", 32, 16, 14, 14, 216, 96, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("x3d conv_c s5 432->192", 32, 16, 7, 7, 432, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("x3d hb conv_a s5 192->432", 16, 16, 7, 7, 192, 432, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("x3d hb conv_c s5 432->192", 16, 16, 7, 7, 432, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("x3d pre_conv s5 192->432", 32, 16, 7, 7, 192, 432, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("sf shortcut res3 320->512 s122", 16, 8, 32, 32, 320, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("sf shortcut res4 640->1024 s122", 16, 8, 16, 16, 640, 1024, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("sf shortcut res5 1280->2048 s122", 16, 8, 8, 8, 1280, 2048, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ("sf shortcut res3 dense 320->512 s111", 16, 8, 16, 16, 320, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("nsweep M401k K192 N384", 8, 1, 1, 50177, 192, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("nsweep M401k K192 N512", 8, 1, 1, 50177, 192, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("nsweep M401k K192 N576", 8, 1, 1, 50177, 192, 576, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("nsweep M401k K192 N640", 8, 1, 1, 50177, 192, 640, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("nsweep M401k K192 N768", 8, 1, 1, 50177, 192, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("nsweep M200k K192 N576", 4, 1, 1, 50177, 192, 576, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("nsweep M100k K192 N576", 8, 1, 1, 12545, 192, 576, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
]


def run(label, B, T, H, W, cin, cout, k, s, p, iters=20):
    To, Ho, Wo = [(i + 2 * pp - kk) // ss + 1 for i, pp, kk, ss in zip((T, H, W), p, k, s)]
    taps = k[0] * k[1] * k[2]
    x = torch.randn(B, T, H, W, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, taps * cin, device="cuda") * 0.05).bfloat16()
    cp = (cout + 7) // 8 * 8
    y = torch.empty(B, To, Ho, Wo, cp, device="cuda", dtype=torch.bfloat16)
    d = L.Conv3dDesc()
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d.x_bs, d.y_bs, d.ldx, d.ldy = T * H * W * cin, To * Ho * Wo * cp, cin, cp
    d.B, d.Ti, d.Hi, d.Wi, d.cin, d.To, d.Ho, d.Wo, d.cout = B, T, H, W, cin, To, Ho, Wo, cout
    d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = (*k, *s, *p)
    d.act, d.a_act, d.dtype = ACT, L.ACT_NONE, L.PV_BF16
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        L.check(lib.pv_conv3d(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.pv_conv3d(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * To * Ho * Wo * cout * taps * cin
    byts = 2.0 * (x.numel() + w.numel() + y.numel())
    print("%-48s %8.4f ms %8.1f TF/s %8.1f GB/s" % (label, ms, flops / ms / 1e9, byts / ms / 1e6), flush=True)


if __name__ == "__main__":
    sel = [a for a in sys.argv[1:] if not a.startswith("--")]
    for a in sys.argv[1:]:
        if a.startswith("--act="):
            ACT = {"gelu": L.ACT_GELU, "none": L.ACT_NONE, "relu": L.ACT_RELU}[a[6:]]
    for a in sys.argv[1:]:
        if a.startswith("--tune="):      # e.g. --tune=gemm8=0,conv_route=2
            from pytorchvideo_amd.accelerator.mi355x import tuning
            tuning.apply(a[len("--tune="):])
            print("tune", a[len("--tune="):])
    for sh in SHAPES:
        if not sel or any(t in sh[0] for t in sel):
            run(*sh)
