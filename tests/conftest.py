import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False,
                     help="also run the `slow` GPU cases (second / stress instances at full geometry); PV_RUN_SLOW=1 does the same")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-geometry cases on the second / stress weight instances; skipped unless --runslow "
                                       "or PV_RUN_SLOW=1 (tools/gpu_round.sh runs them)")
    # The GPU boxes have 128-256 hardware threads; torch's default (one thread per core) makes every CPU reference of the
    # parity tests SLOWER than 8-16 threads do (bench.py's cpu_baseline sweep: X3D-M 4.5 clips/s at 8 threads, 2.2 at 64)
    # and small F.conv3d references spend their time in the fork/join.  Cap the intra-op pool for the test session.
    import torch
    cap = int(os.environ.get("PV_TEST_THREADS", "16"))
    # (tools/parity_full.py::filled_model restores the default for the weight FILL only: the calibration forward of
    #  `trained_like` then gives the instance bench.py times, bit for bit -- on these hosts a 16-thread calibration moves the
    #  BatchNorm statistics in the last bit, and through bf16 storage X3D's logits by ~6e-4 of their range)
    os.environ.setdefault("PV_TORCH_DEFAULT_THREADS", str(torch.get_num_threads()))
    if torch.get_num_threads() > cap:
        torch.set_num_threads(cap)


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests skip (instead of failing with "No HIP GPUs") where there is no device."""
    import torch
    if not (config.getoption("--runslow") or os.environ.get("PV_RUN_SLOW") == "1"):
        skip_slow = pytest.mark.skip(reason="slow case: run with --runslow or PV_RUN_SLOW=1 (tools/gpu_round.sh does)")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(skip_slow)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pv_lib():
    """The C-ABI library, built on demand (hipcc cross-compiles without a GPU)."""
    from pytorchvideo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from pytorchvideo_amd.csrc.build import build
        build(verbose=False)
    return _lib.lib()
