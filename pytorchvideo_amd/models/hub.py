"""Named configurations of the model zoo (reference: pytorchvideo/models/hub/*.py).  Only the
configs are mirrored; checkpoint download needs the network and is out of scope."""

x3d_configs = {  # hub/x3d.py:100-161
    "x3d_xs": dict(input_clip_length=4, input_crop_size=160, depth_factor=2.2),
    "x3d_s": dict(input_clip_length=13, input_crop_size=160, depth_factor=2.2),
    "x3d_m": dict(input_clip_length=16, input_crop_size=224, depth_factor=2.2),
    "x3d_l": dict(input_clip_length=16, input_crop_size=312, depth_factor=5.0),
}

mvit_video_base_config = {  # hub/vision_transformers.py:21-29 (16x4)
    "spatial_size": 224, "temporal_size": 16,
    "embed_dim_mul": [[1, 2.0], [3, 2.0], [14, 2.0]], "atten_head_mul": [[1, 2.0], [3, 2.0], [14, 2.0]],
    "pool_q_stride_size": [[1, 1, 2, 2], [3, 1, 2, 2], [14, 1, 2, 2]],
    "pool_kv_stride_adaptive": [1, 8, 8], "pool_kvq_kernel": [3, 3, 3],
}
mvit_video_base_32x3_config = dict(mvit_video_base_config, temporal_size=32)  # :31-39

slowfast_r50_config = dict(model_depth=50, slowfast_fusion_conv_kernel_size=(7, 1, 1))      # hub/slowfast.py:59-66
slowfast_r101_config = dict(model_depth=101, slowfast_fusion_conv_kernel_size=(5, 1, 1))    # hub/slowfast.py:92-99
csn_r101_config = dict(model_depth=101, stem_pool=None)  # hub/csn.py (torch nn.MaxPool3d -> see factory)
r2plus1d_r50_config = dict(model_depth=50, dropout_rate=0.5)  # hub/r2plus1d.py


# --------------------------------------------------------------------------- checkpoint ingest
# SURVEY 8f-3.  The model zoo stores {"model_state": state_dict, ...}; the reference builders load it
# with strict key checking (hub/utils.py:39-44, hub/x3d.py:31-32), and MViT checkpoints older than
# version 2 are remapped by MultiScaleAttention._load_from_state_dict (layers/attention.py:546-575,
# mirrored in pytorchvideo_amd/layers/attention.py).  There is no network here: `checkpoint` is a local
# file (torch.save format) or the already loaded dict.  The MI355X deploy form reads its weights from the
# module tree at convert time, so a loaded checkpoint needs nothing else.
def load_checkpoint(model, checkpoint, strict=True, trusted=False):
    """Load a model-zoo style checkpoint into a host-mirror (or reference) model; returns the model.
    Files are read with the tensors-only unpickler (the zoo files are plain tensor dicts); `trusted=True`
    allows full unpickling for a file that carries other Python objects and comes from a source you trust."""
    import torch
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        try:
            checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True)
        except Exception:
            if not trusted:
                raise
            checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    state = checkpoint["model_state"] if isinstance(checkpoint, dict) and "model_state" in checkpoint else checkpoint
    model.load_state_dict(state, strict=strict)     # RuntimeError on missing / unexpected keys, like the reference
    return model


def hub_model_builder(model_builder_func, pretrained=False, progress=True, checkpoint_path="", default_config=None, **kwargs):
    """hub/utils.py:11-45 with `checkpoint_path` a local file instead of a URL."""
    if pretrained:
        assert len(kwargs) == 0, "Do not change kwargs for pretrained model."
    if default_config is not None:
        for argument, value in default_config.items():
            if kwargs.get(argument) is None:
                kwargs[argument] = value
    model = model_builder_func(**kwargs)
    if pretrained:
        if not checkpoint_path:
            raise RuntimeError("pretrained=True needs checkpoint_path: a local copy of the model-zoo file (no network here)")
        load_checkpoint(model, checkpoint_path)
    return model


def _named(builder_name, module, config):
    def build(pretrained=False, progress=True, checkpoint_path="", **kwargs):
        import importlib
        fn = getattr(importlib.import_module("." + module, __package__), builder_name)
        return hub_model_builder(fn, pretrained, progress, checkpoint_path, default_config=config, **kwargs)
    return build


x3d_xs = _named("create_x3d", "x3d", x3d_configs["x3d_xs"])
x3d_s = _named("create_x3d", "x3d", x3d_configs["x3d_s"])
x3d_m = _named("create_x3d", "x3d", x3d_configs["x3d_m"])
x3d_l = _named("create_x3d", "x3d", x3d_configs["x3d_l"])
slowfast_r50 = _named("create_slowfast", "slowfast", slowfast_r50_config)
slowfast_r101 = _named("create_slowfast", "slowfast", slowfast_r101_config)
mvit_base_16x4 = _named("create_multiscale_vision_transformers", "vision_transformers", mvit_video_base_config)
mvit_base_32x3 = _named("create_multiscale_vision_transformers", "vision_transformers", mvit_video_base_32x3_config)


# hub/resnet.py:41-160 -- create_resnet variants
def slow_r50(pretrained=False, progress=True, checkpoint_path="", **kwargs):
    from .resnet import create_resnet
    return hub_model_builder(create_resnet, pretrained, progress, checkpoint_path,
                             default_config=dict(stem_conv_kernel_size=(1, 7, 7), head_pool_kernel_size=(8, 7, 7), model_depth=50), **kwargs)


def c2d_r50(pretrained=False, progress=True, checkpoint_path="", **kwargs):
    import torch.nn as nn
    from .resnet import create_resnet
    return hub_model_builder(create_resnet, pretrained, progress, checkpoint_path,
                             default_config=dict(stem_conv_kernel_size=(1, 7, 7), stage1_pool=nn.MaxPool3d,
                                                 stage_conv_a_kernel_size=((1, 1, 1),) * 4), **kwargs)


def i3d_r50(pretrained=False, progress=True, checkpoint_path="", **kwargs):
    import torch.nn as nn
    from .resnet import create_resnet
    return hub_model_builder(create_resnet, pretrained, progress, checkpoint_path,
                             default_config=dict(stem_conv_kernel_size=(5, 7, 7), stage1_pool=nn.MaxPool3d,
                                                 stage_conv_a_kernel_size=((3, 1, 1), [(3, 1, 1), (1, 1, 1)],
                                                                           [(3, 1, 1), (1, 1, 1)], [(1, 1, 1), (3, 1, 1)])), **kwargs)


# hub/resnet.py:73-90 and hub/slowfast.py:150-180 -- the AVA detection models (defaults of the *_with_roi_head builders)
def slow_r50_detection(pretrained=False, progress=True, checkpoint_path="", **kwargs):
    from .resnet import create_resnet_with_roi_head
    return hub_model_builder(create_resnet_with_roi_head, pretrained, progress, checkpoint_path, **kwargs)


def slowfast_r50_detection(pretrained=False, progress=True, checkpoint_path="", **kwargs):
    from .slowfast import create_slowfast_with_roi_head
    return hub_model_builder(create_slowfast_with_roi_head, pretrained, progress, checkpoint_path, **kwargs)


# hub/csn.py:20-58, hub/r2plus1d.py:20-56, hub/slowfast.py:101-147, hub/vision_transformers.py:41-54,127-158
def csn_r101(pretrained=False, progress=True, checkpoint_path="", **kwargs):
    import torch.nn as nn
    from .csn import create_csn
    return hub_model_builder(create_csn, pretrained, progress, checkpoint_path,
                             default_config=dict(model_depth=101, stem_pool=nn.MaxPool3d, head_pool_kernel_size=(4, 7, 7)), **kwargs)


def r2plus1d_r50(pretrained=False, progress=True, checkpoint_path="", **kwargs):
    from .r2plus1d import create_r2plus1d
    return hub_model_builder(create_r2plus1d, pretrained, progress, checkpoint_path, default_config=dict(dropout_rate=0.5), **kwargs)


def slowfast_16x8_r101_50_50(pretrained=False, progress=True, checkpoint_path="", **kwargs):
    """SlowFast R101 whose res4 has temporal conv_a kernels in its first 6 blocks only."""
    from .slowfast import create_slowfast
    res4 = ((3, 1, 1),) * 6 + ((1, 1, 1),) * (23 - 6)
    return hub_model_builder(create_slowfast, pretrained, progress, checkpoint_path, default_config=dict(
        model_depth=101, slowfast_fusion_conv_kernel_size=(5, 1, 1),
        stage_conv_a_kernel_sizes=(((1, 1, 1), (1, 1, 1), res4, (3, 1, 1)), ((3, 1, 1), (3, 1, 1), res4, (3, 1, 1))),
        head_pool_kernel_sizes=((16, 7, 7), (64, 7, 7))), **kwargs)


mvit_image_base_16_config = {
    "spatial_size": 224, "temporal_size": 1, "depth": 16, "conv_patch_embed_kernel": [7, 7],
    "conv_patch_embed_stride": [4, 4], "conv_patch_embed_padding": [3, 3], "use_2d_patch": True,
    "embed_dim_mul": [[1, 2.0], [3, 2.0], [14, 2.0]], "atten_head_mul": [[1, 2.0], [3, 2.0], [14, 2.0]],
    "pool_q_stride_size": [[1, 1, 2, 2], [3, 1, 2, 2], [14, 1, 2, 2]], "pool_kv_stride_adaptive": [1, 4, 4],
    "pool_kvq_kernel": [1, 3, 3],
}
mvit_base_16 = _named("create_multiscale_vision_transformers", "vision_transformers", mvit_image_base_16_config)

# the entry points of the reference's hubconf.py that lie on the path (efficient_x3d_* are the mobile_cpu models)
HUB_ENTRYPOINTS = ["x3d_xs", "x3d_s", "x3d_m", "x3d_l", "slow_r50", "c2d_r50", "i3d_r50", "slow_r50_detection",
                   "slowfast_r50", "slowfast_r101", "slowfast_16x8_r101_50_50", "slowfast_r50_detection",
                   "csn_r101", "r2plus1d_r50", "mvit_base_16", "mvit_base_16x4", "mvit_base_32x3"]
