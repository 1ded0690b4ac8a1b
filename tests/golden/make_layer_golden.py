"""Record the outputs of the REAL reference's layers (imported through oracle/ref_shim.py) for the cases of
layer_cases.py -> tests/golden/layers.pt.  Runs only where /root/reference exists; the fixture is committed.

    python tests/golden/make_layer_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from layer_cases import LAYER_CASES, MVIT_CASES, build_case  # noqa: E402
from oracle import ref_shim  # noqa: E402
from oracle.weights import deterministic_fill, seeded_input  # noqa: E402


def run_case(module, spec, seed):
    """Shared with the test: fill by key, eval, run on the seeded input; returns a list of output tensors."""
    deterministic_fill(module, seed).eval()
    with torch.no_grad():
        if spec[0] == "tokens":
            y, thw = module(seeded_input(spec[1], seed), list(spec[2]))
            return [y, torch.tensor(list(thw))]
        if spec[0] == "list":
            out = module([seeded_input(s, seed + j) for j, s in enumerate(spec[1])])
            return list(out) if isinstance(out, (list, tuple)) else [out]
        return [module(seeded_input(spec[1], seed))]


def main():
    ref_shim.install()
    out = {}
    for i, (name, mod, attr, kwargs, spec) in enumerate(LAYER_CASES):
        m = build_case("pytorchvideo", mod, attr, kwargs)
        out[name] = {"outputs": run_case(m, spec, i), "state_keys": [(k, tuple(v.shape)) for k, v in m.state_dict().items()]}
    from pytorchvideo.models.vision_transformers import create_multiscale_vision_transformers
    for i, (name, cfg, shape) in enumerate(MVIT_CASES):
        m = create_multiscale_vision_transformers(**cfg)
        out[name] = {"outputs": run_case(m, ("tensor", shape), 100 + i),
                     "state_keys": [(k, tuple(v.shape)) for k, v in m.state_dict().items()]}
    path = os.path.join(HERE, "layers.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "cases")


if __name__ == "__main__":
    main()
