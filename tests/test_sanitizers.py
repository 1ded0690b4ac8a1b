"""Sanitizer / debug builds of the native code.

CPU: the AddressSanitizer build of the C-ABI shim (csrc/build.py --variant asan: host side of every translation unit
instrumented) goes through tests/helpers/asan_job.py -- plans, knobs from four threads, the communicator, 1 800 random
descriptors -- in a python started under the ASan runtime; any report fails the test.
GPU: launch plans converted with tuning.OPTIONS["arena_guards"] (every buffer in its own memory, canaries behind it):
whole-model forwards must leave every canary intact and still match the oracle; a deliberately overlong store must trip."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_address_sanitizer_build_of_the_c_abi_shim(tmp_path):
    from pytorchvideo_amd.csrc.build import asan_runtime, build_variant
    rt = asan_runtime()
    if rt is None:
        pytest.skip("no libclang_rt.asan in this ROCm")
    lib = build_variant("asan", verbose=False)
    stub = os.path.join(str(tmp_path), "librccl_stub.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", stub, os.path.join(ROOT, "tests", "helpers", "rccl_stub.c"), "-lrt"])
    env = dict(os.environ, LD_PRELOAD=rt, PV_MI355X_LIB=lib, PV_RCCL_LIB=stub, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:abort_on_error=0:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "asan_job.py")], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "asan job ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_the_sanitizer_runtime_does_catch_an_overrun(tmp_path):
    """The job above is only evidence if the preloaded runtime reports: a 4-line library with a heap overrun must fail."""
    from pytorchvideo_amd.csrc.build import HIPCC, asan_runtime
    rt = asan_runtime()
    if rt is None:
        pytest.skip("no libclang_rt.asan in this ROCm")
    src = os.path.join(str(tmp_path), "t.hip")
    open(src, "w").write('#include <hip/hip_runtime.h>\nextern "C" int f(int n) { int* a = new int[4]; int r = a[n]; delete[] a; return r; }\n')
    so = os.path.join(str(tmp_path), "libt.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-fsanitize=address", "-fno-gpu-sanitize", "-fPIC", "-shared", "-o", so, src])
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, "-c", "import ctypes; ctypes.CDLL(%r).f(7)" % so], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["x3d_xs", "slowfast_r50_small", "csn_r50_small", "r2plus1d_r50_small", "mvit_b_small", "mvit_v2ish_small"])
def test_whole_model_plans_write_only_inside_their_buffers(name):
    """Every model family of the path (the golden cases of tests/test_gpu_models.py), fp32 and bf16, as a debug plan."""
    from oracle import functional as OF
    from test_gpu_models import DTYPES, _deploy, _golden_case, _oracle
    from gpu_util import rel_err
    from pytorchvideo_amd.accelerator.mi355x import tuning
    from pytorchvideo_amd import models as M
    factory, fwd = {
        "x3d_xs": (M.create_x3d, lambda g: (lambda sd, x: OF.x3d_forward(sd, x, g["cfg"]["input_clip_length"], g["cfg"]["input_crop_size"]))),
        "slowfast_r50_small": (M.create_slowfast, lambda g: (lambda sd, x: OF.slowfast_forward(sd, x[0], x[1], head_pool_kernels=g["cfg"]["head_pool_kernel_sizes"]))),
        "csn_r50_small": (M.create_csn, None),
        "r2plus1d_r50_small": (M.create_r2plus1d, None),
        "mvit_b_small": (M.create_multiscale_vision_transformers, lambda g: (lambda sd, x: OF.mvit_forward(sd, x, g["cfg"]))),
        "mvit_v2ish_small": (M.create_multiscale_vision_transformers, lambda g: (lambda sd, x: OF.mvit_forward(sd, x, g["cfg"]))),
    }[name]
    tuning.OPTIONS["arena_guards"] = 4096
    try:
        for dtype, tol in DTYPES:
            g, m, x = _golden_case(name, factory)
            dm, xd = _deploy(m, x, dtype)
            for _ in range(2):
                got = dm(list(xd) if isinstance(xd, list) else xd)
            sess = dm._pv_session
            assert sess._arena.guard == 4096 and len(sess._arena.bands) > 10
            bad = sess.check_guards()
            assert bad == [], "kernels wrote past %d arena buffers, e.g. %s" % (len(bad), bad[:5])
            if fwd is None:
                continue      # (parity of these two families: tests/test_gpu_models.py)
            want = _oracle(m.state_dict(), x, dtype, fwd(g))
            assert rel_err(got, want) <= tol
    finally:
        tuning.OPTIONS["arena_guards"] = 0


@pytest.mark.gpu
def test_a_store_past_its_buffer_trips_the_canary():
    import ctypes as C
    from pytorchvideo_amd import _lib as L
    from pytorchvideo_amd.accelerator.mi355x import tuning
    from pytorchvideo_amd.accelerator.mi355x.session import Session
    tuning.OPTIONS["arena_guards"] = 4096
    try:
        s = Session(dtype=torch.float32)
        a, b, y = s.alloc_act(1, 1, 1, 64, 8), s.alloc_act(1, 1, 1, 64, 8), s.alloc_act(1, 1, 1, 64, 8)
        # y = a + b over 65 rows of a 64-row buffer: the last row lands in y's canary
        s.add_op(L.OP_ADD_ACT, dict(a=a.ptr, b=b.ptr, y=y.ptr, rows=65, C=8, lda=8, ldb=8, ldy=8, act=L.ACT_NONE, dtype=L.PV_F32), "overlong add")
        s.finalize()
        s.launch()
        bad = s.check_guards()
        assert len(bad) == 1 and bad[0][0] == y.off and bad[0][1] == 64 * 8 * 4 and bad[0][3] > 0, bad
    finally:
        tuning.OPTIONS["arena_guards"] = 0
