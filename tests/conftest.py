import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests skip (instead of failing with "No HIP GPUs") where there is no device."""
    import torch
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
