"""The C-ABI library loads and exports every symbol include/pv_mi355x.h declares; descriptor
layouts of the ctypes binding match the C structs (no compute, runs without a GPU)."""
import ctypes as C
import os
import re

from pytorchvideo_amd import _lib as L

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pv_mi355x.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pv_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(pv_lib):
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(pv_lib, n), "header declares %s but the library does not export it" % n
    assert sorted(L.EXPORTED_SYMBOLS) == names, "ctypes binding and header disagree on the symbol list"


def test_version_and_error_string(pv_lib):
    assert pv_lib.pv_version() == L.ABI_VERSION
    assert isinstance(pv_lib.pv_last_error(), (bytes, type(None)))


def test_descriptor_sizes_match_c_structs(pv_lib):
    """pv_plan_add rejects a descriptor whose size differs from the C sizeof."""
    plan = C.c_void_p(pv_lib.pv_plan_create())
    try:
        for kind, cls in L.DESC_FOR_OP.items():
            d = cls()
            idx = pv_lib.pv_plan_add(plan, kind, C.byref(d), C.sizeof(d))
            assert idx >= 0, "descriptor size mismatch for op kind %d (%s)" % (kind, cls.__name__)
        assert pv_lib.pv_plan_size(plan) == len(L.DESC_FOR_OP)
        d = L.Conv3dDesc()
        assert pv_lib.pv_plan_add(plan, L.OP_CONV3D, C.byref(d), C.sizeof(d) - 8) == L.PV_ERR_INVALID
    finally:
        pv_lib.pv_plan_destroy(plan)


def test_invalid_descriptors_are_rejected_without_a_gpu(pv_lib):
    d = L.Conv3dDesc()
    assert pv_lib.pv_conv3d(C.byref(d), None) == L.PV_ERR_INVALID
    p = L.Pool3dDesc()
    assert pv_lib.pv_pool3d(C.byref(p), None) == L.PV_ERR_INVALID
    a = L.AttentionDesc()
    assert pv_lib.pv_attention(C.byref(a), None) < 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        L.lib()
    except L.PvError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected PvError")


def test_random_descriptors_never_crash_the_library(pv_lib):
    """1500 random descriptors over all 15 compute entry points, then 600 geometrically consistent conv / depthwise
    descriptors that get past validation into the host-side routing: every call returns a non-positive status."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "abi_fuzz.py"), "7", "100"],
                       capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=root, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0 and r.stdout.startswith("ok 1500 calls") and "structured ok" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
