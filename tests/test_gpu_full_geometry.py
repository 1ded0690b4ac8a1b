"""GPU parity at the FULL geometries BASELINE.json names: X3D-M 16x224^2 (configs[1]), SlowFast-R50 8x8 at 256^2
(configs[2], reference models/hub/slowfast.py:59-66), MViT-B 32x3 at 224^2 (configs[3],
models/hub/vision_transformers.py:31-39) and X3D-L 16x224^2 (configs[4]) -- the workloads bench.py times -- one clip
each, and at the bench batch with the bench's stream count (every row checked).

Weights: `calibrated_fill` -- the reference tests' BatchNorm randomisation (tests/test_fuse_bn.py:58-63) with the
running statistics then set to what a checkpoint holds (the data's), so the logits are O(1-10), not 1e4-1e10.

All bounds are FIXED numbers (nothing is computed from the run itself).  Metric: max|d| / max|oracle logits|.
  fp32 deploy form vs the fp32 oracle                          <= 1e-3                      (north star)
  bf16 deploy form vs the oracle evaluated with bf16 STORAGE   <= KERNEL_BF16[workload]     (the kernels' arithmetic)
  bf16 deploy form vs the UNQUANTISED fp32 oracle              <= NORTH_STAR_BF16[workload]
Where the last two come from -- CPU only, no kernel of this repo involved, tools/storage_floor.py ->
profiles/r3/storage_floor.json, re-checked by tests/test_storage_floor.py:
  * bf16 storage alone (same oracle, weights / input / every stored activation rounded where the deploy form rounds
    them -- oracle/functional.py::storage_emulation -- exact fp32 arithmetic in between) moves the logits of these
    random-weight instances by 4.2e-2 (X3D-M), 7.0e-2 (X3D-L), 3.1e-2 (SlowFast-R50), 7.9e-3 (MViT-B): 80-165 layers of
    2^-9 relative roundings.  No arithmetic holding bf16 weights and activations can be closer to the fp32 reference,
    so the north star's 1e-2 is attainable -- and asserted plainly -- only for MViT-B (LayerNorm renormalises every
    block); the conv stacks are held to their measured floor + 50 % (fp16 storage would give 6e-3 / 3e-2 / 6e-3 / 1e-3).
  * the conv stacks are also CHAOTIC at that precision: nudging every stored value of the bf16-storage oracle by one
    fp32 ulp before rounding moves ITS OWN logits by 1.6e-2 (X3D-M), 6.7e-2 (X3D-L), 9.3e-3 (SlowFast-R50), 3.2e-3
    (MViT-B).  An implementation whose fp32 arithmetic differs in the last bit (accumulation order, exp / sigmoid
    approximations) cannot agree with the emulation better than that; the kernels are held to 1.3-2x that figure
    (measured on the MI355X: 1.4e-2 / 5.5e-2 / 8.9e-3 / 2.4e-3 -- each BELOW the oracle's own sensitivity), and to the
    plain 1e-2 where the instance is well conditioned (MViT-B).  What isolates the kernels tightly is the fp32 deploy form (<= 3e-5 here) and the
    per-kernel bf16 tests of tests/test_gpu_kernels.py.
"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

FP32_TOL = 1e-3
KERNEL_BF16 = {"x3d_m": 3e-2, "x3d_l": 9e-2, "slowfast_r50": 2.5e-2, "mvit_b_32x3": 1e-2}
NORTH_STAR_BF16 = {"x3d_m": 6.5e-2, "x3d_l": 1.1e-1, "slowfast_r50": 4.7e-2, "mvit_b_32x3": 1e-2}


def _dump(r, what):
    """PV_PARITY_DUMP=<file>: every case's numbers as one JSON line (the evidence kept under profiles/)."""
    path = os.environ.get("PV_PARITY_DUMP")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps(dict(r, case=what, bounds={"fp32": FP32_TOL, "bf16_kernel": KERNEL_BF16[r["workload"]],
                                                          "bf16_north_star": NORTH_STAR_BF16[r["workload"]]}), default=str) + "\n")


def _check(r, bf16_only=False):
    w = r["workload"]
    assert 0.1 < r["logit_absmax"] < 100.0 and r["logit_std"] > 1e-2      # a well-scaled, non-degenerate instance
    if not bf16_only:
        assert r["fp32_vs_oracle"] <= FP32_TOL
        assert r["fp32_replay_equal"]
    assert r["bf16_replay_equal"]
    assert r["bf16_vs_emulated_oracle"] <= KERNEL_BF16[w] and r["bf16_rows_worst"] <= 1.5 * KERNEL_BF16[w]
    assert r["bf16_vs_fp32_oracle"] <= NORTH_STAR_BF16[w]


@pytest.mark.parametrize("workload", ["x3d_m", "x3d_l", "slowfast_r50", "mvit_b_32x3"])
def test_full_geometry_parity(workload):
    from parity_full import case
    r = case(workload, "calibrated")
    print("\n%s: fp32 %.2e | bf16 vs bf16-storage oracle %.2e | bf16 vs fp32 oracle %.2e (storage floor %.2e, weights "
          "alone %.2e)" % (workload, r["fp32_vs_oracle"], r["bf16_vs_emulated_oracle"], r["bf16_vs_fp32_oracle"],
                           r["storage_floor"], r["weights_floor"]))
    _dump(r, "one clip, single plan")
    _check(r)


@pytest.mark.parametrize("workload", ["x3d_m", "slowfast_r50", "mvit_b_32x3"])
def test_bench_batch_with_bench_streams_every_row(workload):
    """The deploy form exactly as bench.py builds it (per-GPU batch, sub-batch branches of one joint graph): every row
    of the batch against the oracle -- not just clip 0 of a batch-1 plan."""
    from bench import WORKLOADS
    from parity_full import case
    wl = WORKLOADS[workload]
    r = case(workload, "calibrated", batch=wl["batch"], streams=wl.get("streams", 1), dtypes=("bf16",))
    print("\n%s b=%d streams=%d: bf16 vs bf16-storage oracle %.2e (worst row %.2e) | vs fp32 oracle %.2e | top-1 %d/%d" % (
        workload, r["batch"], r["streams"], r["bf16_vs_emulated_oracle"], r["bf16_rows_worst"], r["bf16_vs_fp32_oracle"],
        r["top1_agree_emulated"], r["batch"]))
    _dump(r, "bench batch, bench streams, every row")
    _check(r, bf16_only=True)
    assert r["top1_agree_emulated"] >= r["batch"] - 1
