"""GPU parity at the FULL geometries BASELINE.json names (one clip each): X3D-M 16x224^2 (configs[1]), SlowFast-R50
8x8 at 256^2 (configs[2], reference models/hub/slowfast.py:59-66), MViT-B 32x3 at 224^2 (configs[3],
models/hub/vision_transformers.py:31-39) and X3D-L 16x224^2 (configs[4]) -- the same workloads bench.py times.

Three numbers per model, max|d| / max|oracle logits|, weights from the reference-style fill (the factory's own
conv / linear init + randomised BatchNorm statistics, reference tests/test_fuse_bn.py:58-63):
  * fp32 deploy form vs the CPU oracle                                  <= 1e-3  (north star)
  * bf16 deploy form vs the oracle on the same bf16-rounded weights/input  <= 1e-2  (kernel isolation)
  * bf16 deploy form vs the UNQUANTISED fp32 oracle                      <= 1e-2  (north star; the comparator the
    reference user sees: fp32 CPU forward on identical inputs)
"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

FP32_TOL, BF16_TOL = 1e-3, 1e-2


@pytest.mark.parametrize("workload", ["x3d_m", "x3d_l", "slowfast_r50", "mvit_b_32x3"])
def test_full_geometry_parity(workload):
    from parity_full import case
    r = case(workload, "reference_style")
    print("\n%s: fp32 %.2e | bf16 vs quantised oracle %.2e | bf16 vs fp32 oracle %.2e | bf16 weights alone %.2e" % (
        workload, r["fp32_vs_oracle"], r["bf16_vs_quantised_oracle"], r["bf16_vs_fp32_oracle"],
        r["quantised_oracle_vs_fp32_oracle"]))
    assert r["logit_std"] > 1e-3                       # non-degenerate logits
    assert r["fp32_vs_oracle"] <= FP32_TOL
    assert r["bf16_vs_quantised_oracle"] <= BF16_TOL
    assert r["bf16_vs_fp32_oracle"] <= BF16_TOL
    assert r["top1_agree"]
