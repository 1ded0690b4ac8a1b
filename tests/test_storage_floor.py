"""What 16-bit storage alone costs against the fp32 reference (CPU, no kernel involved): the evidence behind the
per-workload bound of tests/test_gpu_full_geometry.py.  One clip of X3D-M at the full BASELINE geometry with calibrated
weights: rounding the dense weights and the input to bf16 -- exact fp32 arithmetic otherwise -- already moves the logits
by more than the north star's 1e-2, and bf16 storage of the activations adds to it; fp16 storage stays below 1e-2."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_bf16_storage_alone_exceeds_1e_2_on_x3d_m_and_fp16_does_not():
    from storage_floor import floors
    r = floors("x3d_m", "calibrated")
    assert 0.1 < r["logit_absmax"] < 100.0
    assert r["bf16_weights_only"] > 1e-2 and r["bf16_storage"] > r["bf16_weights_only"]
    assert r["bf16_storage"] < 6.5e-2          # ... and below the bound the GPU test holds the bf16 deploy form to
    assert r["fp16_storage"] < 1e-2
    # ... and the bf16-storage evaluation is chaotic on this instance: a one-ulp nudge of its fp32 values moves ITS OWN logits
    # by ~1.6e-2 -- the band tests/test_gpu_full_geometry.py::KERNEL_BF16 gives the kernels (3e-2) is about twice that
    assert 5e-3 < r["bf16_self_sensitivity_1ulp"] < 3e-2
