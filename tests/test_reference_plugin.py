"""The drop-in boundary exercised with the REAL reference: `import pytorchvideo_amd.accelerator` inside a
PyTorchVideo installation registers "mi355x" in the reference's EFFICIENT_BLOCK_TRANSMUTER_REGISTRY, the reference's own
transmute_model replaces the reference's own modules by MI355X blocks (state_dict and original form untouched), the
launch plan emitted from the reference's module tree equals the host mirror's, and the reference's own convert driver
reaches Mi355xBlock.convert().  Needs the reference tree, so it runs in the build container only (skipped elsewhere)
and in a fresh interpreter (the binding to the reference's classes happens at import time)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("PV_REFERENCE_ROOT", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "pytorchvideo")), reason="reference tree not present")
def test_mi355x_target_inside_the_real_reference():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, REFERENCE]), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "reference_plugin_check.py")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("ok ")]
    assert r.stdout.strip().endswith("ALL OK") and len(lines) == 10, r.stdout


def test_standalone_package_defines_the_same_contract():
    """Without PyTorchVideo on the path (the GPU box) the package carries the plugin contract itself."""
    from pytorchvideo_amd import accelerator as A
    assert A.efficient_blocks.INSIDE_PYTORCHVIDEO is False
    assert "mi355x" in A.EFFICIENT_BLOCK_TRANSMUTER_REGISTRY and len(A.EFFICIENT_BLOCK_TRANSMUTER_REGISTRY["mi355x"]) == 5
    assert all(getattr(getattr(A.EfficientBlockBase, n), "__isabstractmethod__", False) for n in ("convert", "forward"))
