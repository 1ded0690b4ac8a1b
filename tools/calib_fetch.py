"""Calibration of the PMC FETCH_SIZE figure of SlowFast's temporal (3,1,1) convs (round-3 verdict, weak #10 / next #6).

profiles/r3/slowfast_r50_pmc.md reports 206.6 MB fetched per launch of `conv_a 1024 -> 256 @ 8x16x16, b=16` against a
67 MB input: exactly 3x, one per temporal tap -- or an artefact of the x2 FETCH_SIZE correction / of Infinity-Cache hits
being counted.  This tool separates the two: the SAME kernel (pv_conv3d -> gemm_glds_kernel) on the SAME input with
  A  kt = 1  (1,1,1) 1024 -> 256                    a plain GEMM: every input row is needed once
  B  kt = 3  (3,1,1) 1024 -> 256                    the layer in question
  C  kt = 1, K = 3072 -> 256 (cin = 3072)           the same MFMA work and weight bytes as B, 3x the input bytes for real
each at the workload's batch (16 clips: 67 MB input, fits the 256 MB Infinity Cache) and at 4x the batch (268 MB input,
does not), launched N times in a fixed order:  A x N, B x N, C x N (batch 16), then A x N, B x N, C x N (batch 64).

  python tools/calib_fetch.py run [N]           # the launches (wrap in rocprofv3 --pmc ... --kernel-trace)
  python tools/calib_fetch.py fold <dir> [N]    # fold the counter CSVs of the passes under <dir> into a markdown table

Counters (one pass each, rocprofv3 --pmc <counter> --kernel-trace): FETCH_SIZE (KB; gfx950 reports half of the bytes of
wide coalesced reads: x2, MI355X_MICROARCH.md HBM section), TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum where available
(requests that actually leave L2 towards the fabric / Infinity Cache / HBM), TCC_HIT_sum, TCC_MISS_sum.
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [  # (tag, batch, cin, kt)
    ("A kt=1 1024->256", 16, 1024, 1), ("B kt=3 1024->256", 16, 1024, 3), ("C kt=1 3072->256", 16, 3072, 1),
    ("A kt=1 1024->256", 64, 1024, 1), ("B kt=3 1024->256", 64, 1024, 3), ("C kt=1 3072->256", 64, 3072, 1),
]
T, H, W, COUT = 8, 16, 16, 256


def run(n):
    import ctypes as C
    import torch
    from pytorchvideo_amd import _lib as L
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tag, B, cin, kt in CASES:
        x = torch.randn(B, T, H, W, cin, device="cuda").bfloat16()
        w = (torch.randn(COUT, kt * cin, device="cuda") * 0.02).bfloat16()
        y = torch.empty(B, T, H, W, COUT, device="cuda", dtype=torch.bfloat16)
        d = L.Conv3dDesc()
        d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
        d.x_bs, d.y_bs, d.ldx, d.ldy = T * H * W * cin, T * H * W * COUT, cin, COUT
        d.B, d.Ti, d.Hi, d.Wi, d.cin, d.To, d.Ho, d.Wo, d.cout = B, T, H, W, cin, T, H, W, COUT
        d.kt, d.kh, d.kw, d.st, d.sh, d.sw, d.pt, d.ph, d.pw = kt, 1, 1, 1, 1, 1, kt // 2, 0, 0
        d.act, d.a_act, d.dtype = L.ACT_RELU, L.ACT_NONE, L.PV_BF16
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            L.check(lib.pv_conv3d(C.byref(d), st))
        e1.record()
        torch.cuda.synchronize()
        inb, wb, outb = 2 * x.numel(), 2 * w.numel(), 2 * y.numel()
        print("CASE %-18s b=%-3d launches=%d  %.4f ms/launch  input %.1f MB  weights %.1f MB  output %.1f MB" % (
            tag, B, n, e0.elapsed_time(e1) / n, inb / 1e6, wb / 1e6, outb / 1e6), flush=True)
        del x, w, y


def fold(d, n):
    """Per case: counter totals of its n dispatches (the conv kernel dispatches in launch order; fills and copies of the
    tensor set-up are other kernels and are skipped by name)."""
    rows = {}
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        per = {}
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "gemm" not in k and "conv" not in k and "pw_stream" not in k and "tap_stream" not in k:
                continue
            per.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        ids = sorted(per)
        if len(ids) != n * len(CASES):
            print("# %s: %d conv dispatches, expected %d -- skipped" % (path, len(ids), n * len(CASES)))
            continue
        for ci, (tag, B, cin, kt) in enumerate(CASES):
            for i in ids[ci * n:(ci + 1) * n]:
                for name, v in per[i].items():
                    rows.setdefault((ci, name), []).append(v)
    names = sorted({k[1] for k in rows})
    print("| case | batch | input MB | " + " | ".join("%s / launch" % c for c in names) + " |")
    print("|---|---|---|" + "---|" * len(names))
    for ci, (tag, B, cin, kt) in enumerate(CASES):
        inb = 2.0 * B * T * H * W * cin / 1e6
        cells = []
        for c in names:
            v = rows.get((ci, c))
            if not v:
                cells.append("-")
            elif c == "FETCH_SIZE":
                kb = sum(v) / len(v)
                cells.append("%.1f KB = %.1f MB raw, x2 = %.1f MB (%.2fx input)" % (kb, kb * 1024 / 1e6, 2 * kb * 1024 / 1e6, 2 * kb * 1024 / 1e6 / inb))
            else:
                cells.append("%.0f" % (sum(v) / len(v)))
        print("| %s | %d | %.1f | %s |" % (tag, B, inb, " | ".join(cells)))


if __name__ == "__main__":
    n = int(sys.argv[3 if sys.argv[1] == "fold" else 2]) if len(sys.argv) > (3 if sys.argv[1] == "fold" else 2) else 6
    if sys.argv[1] == "run":
        run(n)
    else:
        fold(sys.argv[2], n)
