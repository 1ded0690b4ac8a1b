"""Record the call surface and the seeded-construction fingerprints of the REAL reference -> tests/golden/api_surface.json.
Runs only where /root/reference exists (imported through oracle/ref_shim.py); the fixture is committed.

    python tests/golden/make_api_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from api_cases import HUB, SEEDED, SYMBOLS, resolve, rounding_table, seeded_fingerprint, signature_of  # noqa: E402
from oracle import ref_shim  # noqa: E402


def main():
    ref_shim.install()
    out = {"signatures": {"%s.%s" % (m, n): signature_of(resolve("pytorchvideo", m, n)) for m, n in SYMBOLS},
           "seeded": {tag: seeded_fingerprint("pytorchvideo", m, n, cfg) for tag, m, n, cfg in SEEDED},
           "hub": {n: seeded_fingerprint("pytorchvideo", "models.hub", n, {}, seed=1) for n in HUB},
           "rounding": rounding_table("pytorchvideo")}
    path = os.path.join(HERE, "api_surface.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, len(out["signatures"]), "signatures,", len(out["seeded"]), "seeded constructions,", len(out["hub"]), "hub models")


if __name__ == "__main__":
    main()
