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

slowfast_r50_config = dict(model_depth=50)      # hub/slowfast.py:59-66 (defaults)
slowfast_r101_config = dict(model_depth=101)    # hub/slowfast.py:69-100
csn_r101_config = dict(model_depth=101, stem_pool=None)  # hub/csn.py (torch nn.MaxPool3d -> see factory)
r2plus1d_r50_config = dict(model_depth=50, dropout_rate=0.5)  # hub/r2plus1d.py
