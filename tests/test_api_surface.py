"""Drop-in call surface: every factory / module class of the path has the reference's
parameter names, order, kinds and defaults, and draws the same random numbers in the same order -- under the same
torch seed the host mirror builds bit-identical weights.  Fixture: tests/golden/api_surface.json, recorded from the
real reference by tests/golden/make_api_golden.py."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from api_cases import HUB, SEEDED, SYMBOLS, resolve, rounding_table, seeded_fingerprint, signature_of  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "api_surface.json")))


@pytest.mark.parametrize("module,name", SYMBOLS, ids=["%s.%s" % s for s in SYMBOLS])
def test_signature_is_the_reference_signature(module, name):
    want = GOLD["signatures"]["%s.%s" % (module, name)]
    got = signature_of(resolve("pytorchvideo_amd", module, name))
    assert [p[0] for p in got] == [p[0] for p in want], "parameter names / order"
    assert got == want, "kinds or defaults"


@pytest.mark.parametrize("tag,module,name,cfg", SEEDED, ids=[s[0] for s in SEEDED])
def test_seeded_construction_reproduces_the_reference_weights(tag, module, name, cfg):
    want = GOLD["seeded"][tag]
    got = seeded_fingerprint("pytorchvideo_amd", module, name, cfg)
    assert got["n_tensors"] == want["n_tensors"]
    assert got["next_rand"] == want["next_rand"], "the factory consumed a different amount of randomness"
    assert got["sha256"] == want["sha256"], "state_dict differs from the reference's under the same seed"


@pytest.mark.parametrize("name", HUB)
def test_hub_entry_point_builds_the_reference_model(name):
    """Every model-zoo entry point of the reference's hubconf.py on the path (models/hub/*.py), with its default
    configuration: same state_dict keys / shapes (what a model-zoo checkpoint must fit) and, under the same seed,
    the same weights.  hubconf.py at the repo root exports the same names."""
    import hubconf
    from pytorchvideo_amd.models import hub
    assert getattr(hubconf, name) is getattr(hub, name) and name in hub.HUB_ENTRYPOINTS
    want = GOLD["hub"][name]
    got = seeded_fingerprint("pytorchvideo_amd", "models.hub", name, {}, seed=1)
    assert (got["n_tensors"], got["next_rand"], got["sha256"]) == (want["n_tensors"], want["next_rand"], want["sha256"])


def test_width_and_depth_rounding_tables_equal_the_reference():
    """round_width / round_repeats decide every channel count and block count of X3D and MViT (layers/utils.py:19-49)."""
    got = rounding_table("pytorchvideo_amd")
    assert got["round_width"] == GOLD["rounding"]["round_width"] and len(got["round_width"]) == 792
    assert got["round_repeats"] == GOLD["rounding"]["round_repeats"]
