"""What 16-bit storage alone costs against the fp32 reference (CPU, no kernel involved): the evidence behind the
bounds of tests/test_gpu_full_geometry.py.  One clip of X3D-M at the full BASELINE geometry.

* `trained_like` (the instance the north-star 1e-2 is asserted on): bf16 weights + bf16 storage of every activation the
  deploy form stores, exact fp32 arithmetic otherwise, stays BELOW 1e-2 -- the bar is attainable by a bf16 deploy form.
* `calibrated` (block-final gamma ~ 1, the stress instance): rounding the dense weights and the input to bf16 alone
  already moves the logits by more than 1e-2; fp16 storage stays below it (it is the 8-bit mantissa, not the arithmetic).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_bf16_storage_alone_is_below_1e_2_on_the_trained_like_instance():
    from storage_floor import floors
    r = floors("x3d_m", "trained_like", formats=("bf16",), sensitivity=False)      # (the full record: profiles/r4/storage_floor.json)
    assert 0.1 < r["logit_absmax"] < 100.0
    assert r["bf16_weights_only"] < 8e-3 and r["bf16_storage"] < 8e-3


def test_bf16_storage_alone_exceeds_1e_2_on_the_stress_instance_and_fp16_does_not():
    from storage_floor import floors
    r = floors("x3d_m", "calibrated")
    assert 0.1 < r["logit_absmax"] < 100.0
    assert r["bf16_weights_only"] > 1e-2 and r["bf16_storage"] > r["bf16_weights_only"]
    assert r["bf16_storage"] < 6.5e-2          # ... and below the bound the GPU test holds the bf16 deploy form to
    assert r["fp16_storage"] < 1e-2
    # ... and the bf16-storage evaluation is chaotic on this instance: a one-ulp nudge of its fp32 values moves ITS OWN logits
    # by ~1.6e-2
    assert 5e-3 < r["bf16_self_sensitivity_1ulp"] < 3e-2
