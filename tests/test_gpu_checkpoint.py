"""Checkpoint ingest on the hardware: a model-zoo style {"model_state": ...} file goes through the hub
builder (reference models/hub/utils.py:12-45), the transmuter and the convert driver, and the deploy form must equal
the oracle evaluated on the LOADED state_dict -- including a version < 2 MViT file whose pooling keys are remapped at
load time (reference layers/attention.py:546-575).  Plus the hub image model mvit_base_16 at full size
(models/hub/vision_transformers.py:127): 1x3x3 pooling windows on the token grid of a 224^2 image."""
import pytest
import torch

from oracle import functional as OF
from oracle.weights import deterministic_fill, quantize_like_kernels, reference_style_fill, seeded_input, trained_like_fill
from gpu_util import rel_err

pytestmark = pytest.mark.gpu


def _deploy(model, x, dtype):
    from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
    transmute_model(model, "mi355x")
    xd = x.cuda().to(dtype)
    return convert_to_deployable_form(model, xd, dtype=dtype), xd


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-2)])
def test_x3d_s_from_a_model_zoo_file(tmp_path, dtype, tol):
    from pytorchvideo_amd.models import hub
    src = hub.x3d_s()
    # fp32: the reference's own initialisation (block-final gamma 1: every block as large as the trunk), held to 1e-3.  bf16: the
    # well-conditioned instance the north-star tests use -- on the reference-style one a last-bit difference in one squeeze-excitation
    # gate (another fp32 summation order of the same partial sums) grows to 3e-2 at the logits (profiles/r6/x3d_s_gate_order_call45.txt),
    # so a fixed 1e-2 there tests the summation order, not the checkpoint path
    if dtype == torch.float32:
        reference_style_fill(src, 3)
    else:
        trained_like_fill(src, seeded_input((2, 3, 13, 160, 160), 7), 3)
    path = tmp_path / "X3D_S.pyth"
    torch.save({"model_state": src.state_dict(), "cfg": "x3d_s"}, path)
    del src
    m = hub.x3d_s(pretrained=True, checkpoint_path=str(path)).eval()     # strict load, as the reference does
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = seeded_input((2, 3, 13, 160, 160), 3)                             # X3D-S: 13 frames at 160^2 (hub/x3d.py:104-111)
    want = OF.x3d_forward(*(quantize_like_kernels(sd, x) if dtype == torch.bfloat16 else (sd, x)), 13, 160)
    dm, xd = _deploy(m, x, dtype)
    assert dm._pv_session is not None
    assert rel_err(dm(xd), want) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-2)])
def test_mvit_base_16x4_from_a_version_1_file(tmp_path, dtype, tol):
    from pytorchvideo_amd.models import hub
    from pytorchvideo_amd.models.hub import mvit_video_base_config as cfg
    src = hub.mvit_base_16x4()
    deterministic_fill(src, 9)
    new = {k: v.clone() for k, v in src.state_dict().items()}
    old = {}
    for k, v in new.items():   # version 1: pools and their norms directly under the attention module
        for which in "qkv":
            k = k.replace("_attention_pool_%s.pool." % which, "pool_%s." % which).replace(
                "_attention_pool_%s.norm." % which, "norm_%s." % which)
        old[k] = v
    assert not any("_attention_pool_" in k for k in old)
    path = tmp_path / "MVIT_B_16x4.pyth"
    torch.save({"model_state": old}, path)
    del src
    m = hub.mvit_base_16x4()
    hub.load_checkpoint(m, str(path), strict=False).eval()   # the old names stay behind as unexpected keys
    assert all(torch.equal(v, new[k]) for k, v in m.state_dict().items())
    x = seeded_input((1, 3, 16, 224, 224), 9)
    want = OF.mvit_forward(*(quantize_like_kernels(new, x) if dtype == torch.bfloat16 else (new, x)), cfg)
    dm, xd = _deploy(m, x, dtype)
    assert all(type(b).__name__ == "Mi355xMViTBlock" and b.convert_flag for b in dm.blocks)
    assert rel_err(dm(xd), want) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-2)])
def test_mvit_base_16_image_model_at_full_size(dtype, tol):
    """hub mvit_base_16: an image [B,3,224,224], Conv2d patch embedding, (1,3,3) pooling kernels.  The comparator is
    the original-form forward of the same module tree (pinned to the reference by tests/test_oracle_golden.py and
    tests/test_reference_gates.py) on the kernels' quantisation of weights and input."""
    from pytorchvideo_amd.models import hub
    m = hub.mvit_base_16()
    deterministic_fill(m, 4).eval()
    x = seeded_input((2, 3, 224, 224), 4)
    ref = hub.mvit_base_16().eval()
    if dtype == torch.bfloat16:
        sd_q, x_q = quantize_like_kernels(m.state_dict(), x)
        ref.load_state_dict(sd_q)
    else:
        ref.load_state_dict(m.state_dict())
        x_q = x
    with torch.no_grad():
        want = ref(x_q)
    dm, xd = _deploy(m, x, dtype)
    assert all(type(b).__name__ == "Mi355xMViTBlock" and b.convert_flag for b in dm.blocks)
    assert getattr(dm, "_pv_inputs", None) is not None          # one launch plan
    got = dm(xd)
    assert got.shape == want.shape == (2, 400)
    assert rel_err(got, want) <= tol
