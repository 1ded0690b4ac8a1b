"""The oracle (CPU restatement) reproduces the golden vectors produced by the real reference
(tests/golden/make_golden.py).  This is what pins the oracle; everything else is checked
against the oracle."""
import os

import pytest
import torch

from oracle import functional as OF
from oracle.weights import deterministic_fill, seeded_input

TOL = 1e-5  # relative to abs-max
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)


def _check_fingerprint(t, fp, tol=0.0):
    assert tuple(t.shape) == tuple(fp["shape"])
    got = t.detach().float().reshape(-1)[fp["sample_idx"]]
    scale = max(1.0, fp["absmax"])
    assert (got - fp["sample"]).abs().max().item() <= tol * scale
    assert abs(t.float().mean().item() - fp["mean"]) <= max(tol, 1e-6) * scale


def test_x3d_xs_oracle_matches_reference_golden():
    from pytorchvideo_amd.models import create_x3d
    g = _load("x3d_xs")
    m = create_x3d(**g["cfg"])
    deterministic_fill(m, g["seed"]).eval()
    # drop-in claim: identical state_dict keys and shapes as the reference model
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == g["state_keys"]
    x = seeded_input(g["input_shape"], g["seed"])
    logits, blocks = OF.x3d_forward(m.state_dict(), x, g["cfg"]["input_clip_length"],
                                    g["cfg"]["input_crop_size"], return_blocks=True)
    # same torch CPU kernels and op order as the reference: bit-exact on the machine that made
    # the fixture, last-ulp differences on a host with another CPU (oneDNN picks other kernels)
    assert (logits - g["logits"]).abs().max().item() <= TOL * g["logits"].abs().max().item()
    for t, fp in zip(blocks, g["blocks"]):
        _check_fingerprint(t, fp, TOL)
    # and the host mirror (original form) is the same function
    with torch.no_grad():
        assert (m(x) - g["logits"]).abs().max().item() <= TOL * g["logits"].abs().max().item()
