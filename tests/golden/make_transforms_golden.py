"""Record the REAL reference's pre-path transforms (pytorchvideo/transforms/functional.py, loaded by file path: the
package __init__ pulls in torchvision) -> tests/golden/transforms.pt.  Runs only where /root/reference exists.

    python tests/golden/make_transforms_golden.py
"""
import importlib.util
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("PV_REFERENCE_ROOT", "/root/reference")

CASES = [  # (clip shape, num_samples / frame ratios, temporal_dim)
    ((3, 32, 6, 5), 8, -3), ((3, 30, 4, 4), 7, -3), ((3, 5, 4, 4), 9, -3), ((2, 3, 16, 4, 4), 4, 2), ((10, 3, 2, 2), 3, 0),
]
REPEATED = [((3, 32, 4, 4), (4, 1), -3), ((3, 64, 2, 2), (8, 2, 1), -3), ((2, 3, 16, 2, 2), (4, 1), 2)]


def clip(shape, seed):
    return torch.randint(0, 256, shape, generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)


def main():
    spec = importlib.util.spec_from_file_location("pv_ref_functional", os.path.join(REFERENCE, "pytorchvideo/transforms/functional.py"))
    F = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(F)
    out = {"subsample": [], "repeated": [], "div_255": None}
    for i, (shape, n, dim) in enumerate(CASES):
        out["subsample"].append(F.uniform_temporal_subsample(clip(shape, i), n, dim))
    for i, (shape, ratios, dim) in enumerate(REPEATED):
        out["repeated"].append([t.clone() for t in F.uniform_temporal_subsample_repeated(clip(shape, 50 + i), ratios, dim)])
    out["div_255"] = F.div_255(clip((3, 4, 5, 6), 99))
    path = os.path.join(HERE, "transforms.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
