"""Checkpoint ingest: model-zoo {"model_state": ...} files, strict key checking
(reference hub/utils.py:39-44) and the version < 2 MViT key remap (layers/attention.py:546-575)."""
import pytest
import torch

from pytorchvideo_amd.models import hub
from oracle.weights import deterministic_fill


def test_model_zoo_style_checkpoint_round_trip(tmp_path):
    src = hub.x3d_xs(model_num_class=10)
    deterministic_fill(src, 5)
    path = tmp_path / "x3d_xs.pyth"
    torch.save({"model_state": src.state_dict(), "epoch": 3}, path)
    dst = hub.hub_model_builder(lambda **kw: hub.x3d_xs(model_num_class=10), pretrained=True, checkpoint_path=str(path))
    for (ka, a), (kb, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
    x = torch.randn(1, 3, 4, 160, 160)
    with torch.no_grad():
        assert torch.equal(src.eval()(x), dst.eval()(x))
    # strict: a missing or an unexpected key is a RuntimeError, as with the reference's load_state_dict
    sd = dict(src.state_dict())
    sd.pop(next(iter(sd)))
    with pytest.raises(RuntimeError):
        hub.load_checkpoint(hub.x3d_xs(model_num_class=10), {"model_state": sd})
    with pytest.raises(RuntimeError):
        hub.load_checkpoint(hub.x3d_xs(model_num_class=10), {"model_state": dict(src.state_dict(), extra=torch.zeros(1))})
    with pytest.raises(RuntimeError):
        hub.x3d_xs(pretrained=True)   # no local file given


def test_legacy_mvit_checkpoint_keys_are_remapped():
    cfg = dict(spatial_size=32, temporal_size=4, depth=2, patch_embed_dim=16, num_heads=1, head_num_classes=5,
               pool_q_stride_size=[[1, 1, 2, 2]], pool_kv_stride_adaptive=[1, 2, 2], pool_kvq_kernel=[3, 3, 3],
               embed_dim_mul=[[1, 2.0]], atten_head_mul=[[1, 2.0]])
    src = hub.mvit_base_16x4(**cfg)
    deterministic_fill(src, 9)
    new = src.state_dict()
    # a version-1 file: pools and their norms live directly under the attention module, no version metadata
    old = {}
    for k, v in new.items():
        for which in "qkv":
            k = k.replace("_attention_pool_%s.pool." % which, "pool_%s." % which).replace("_attention_pool_%s.norm." % which, "norm_%s." % which)
        old[k] = v
    assert any(".pool_k.weight" in k for k in old) and not any("_attention_pool_" in k for k in old)
    dst = hub.mvit_base_16x4(**cfg)
    hub.load_checkpoint(dst, {"model_state": old}, strict=False)   # the old names stay in the dict as unexpected keys
    for k, v in dst.state_dict().items():
        assert torch.equal(v, new[k]), k


def test_detection_hub_builders_load_reference_keyed_checkpoints(tmp_path):
    """slow_r50_detection / slowfast_r50_detection (reference hub/resnet.py:73-90, hub/slowfast.py:150-180): the AVA
    checkpoints are keyed `model.blocks...` / `detection_head.proj...` (DetectionBBoxNetwork, models/net.py:47-74)."""
    for build in (hub.slow_r50_detection, hub.slowfast_r50_detection):
        src = build()
        keys = list(src.state_dict().keys())
        assert keys[0].startswith("model.blocks.0.") and keys[-2:] == ["detection_head.proj.weight", "detection_head.proj.bias"]
        assert src.detection_head.proj.out_features == 80 and src.detection_head.roi_layer.spatial_scale == 1.0 / 16.0
        path = tmp_path / "det.pyth"
        torch.save({"model_state": src.state_dict()}, path)
        dst = build(pretrained=True, checkpoint_path=str(path))
        assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))


def test_torch_hub_load_from_the_local_repo():
    """The reference's hub tests (tests/test_models_x3d.py:82-119, test_models_slowfast.py:19-41,
    test_models_hub_vision_transformers.py:16-38): torch.hub.load(<repo root>, source="local", model=name,
    pretrained=False) then a forward."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, x in (("x3d_xs", torch.rand(1, 3, 4, 160, 160)), ("mvit_base_16", None)):
        m = torch.hub.load(repo_or_dir=root, source="local", model=name, pretrained=False).eval()
        if x is not None:
            with torch.no_grad():
                assert m(x).shape == (1, 400)
        else:
            assert type(m).__name__ == "MultiscaleVisionTransformers" and isinstance(m.patch_embed.patch_model, torch.nn.Conv2d)
    import hubconf
    assert all(callable(getattr(hubconf, n)) for n in ("x3d_m", "slowfast_r50", "slow_r50_detection", "mvit_base_32x3"))
