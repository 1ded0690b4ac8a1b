"""Where a workgroup of attn_w64_kernel spends its time (dev library: clock stamps of wave 0 of every workgroup)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pytorchvideo_amd import _lib as L

NAMES = ["entry -> first tiles stored", "first barrier", "S(0) + second barrier + setup", "tile loop (all but the last 1-2 steps)",
         "last steps", "epilogue (stores drained)"]


def run(B, heads, Nq, Nk, hd=96, res=1):
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
    d.scale, d.residual_q, d.dtype = hd ** -0.5, res, L.PV_BF16
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        L.check(lib.pv_attention(C.byref(d), st))
    torch.cuda.synchronize()
    h = C.CDLL(L.LIB_PATH)
    nwg = B * heads * ((Nq + 255) // 256)
    buf = (C.c_ulonglong * (8 * nwg))()
    h.pv_dev_attn_stamps.argtypes = [C.c_void_p, C.c_int]
    assert h.pv_dev_attn_stamps(buf, 8 * nwg) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8).astype(np.float64)
    t0 = s[:, 0].min()
    print("B %d heads %d Nq %d Nk %d: %d workgroups; kernel span %.0f kcycles; workgroup entry times (kcycles after the first): p50 %.1f p90 %.1f max %.1f"
          % (B, heads, Nq, Nk, nwg, (s[:, 6].max() - t0) / 1e3, np.percentile(s[:, 0] - t0, 50) / 1e3, np.percentile(s[:, 0] - t0, 90) / 1e3, (s[:, 0].max() - t0) / 1e3))
    first = s[:, 0] - t0 < 2000   # the first round of workgroups
    for sel, nm in ((first, "first round"), (~first, "later rounds")):
        if sel.sum() == 0:
            continue
        dd = np.diff(s[sel, :7], axis=1)
        print("  %s (%d workgroups), cycles: " % (nm, sel.sum()) + "; ".join("%s %.0f" % (NAMES[i], dd[:, i].mean()) for i in range(6)) + "; total %.0f" % dd.sum(1).mean())


if __name__ == "__main__":
    L.tune(attn_w64=1)
    run(8, 4, 3137, 785)
    run(8, 4, 3137, 3137)
    run(8, 1, 50177, 785)
