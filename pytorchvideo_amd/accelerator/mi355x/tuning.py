"""Fusion / routing decisions of the emitters that were settled by A/B measurements on the MI355X, kept
overridable for the tools that re-measure them (tools/*.py set OPTIONS[...] before converting a model).
Nothing here -- and nothing in the library (pv_tune_set) -- is read from the environment; none of the
options selects a non-HIP path."""

OPTIONS = {
    "fuse_shortcut": True,       # projection shortcut as a second K operand of conv_c            (emit.can_fold_shortcut)
    "stem_wpair": True,          # two W-adjacent outputs per MFMA column in the <= 8-channel stem (emit.emit_conv)
    "fuse_ab": True,             # conv_a evaluated inside the depthwise conv_b kernel             (emit.can_fuse_pointwise_into_dw)
    "fuse_ab_max_cin": 64,       # ... while the block input has at most this many channels
    "fuse_block": True,          # a whole X3D residual block without squeeze-excitation as ONE launch (pv_bottleneck, round 6)  (emit.can_fuse_bottleneck)
    "fuse_bc": True,             # conv_b (narrow dense conv) -> pointwise conv_c of a ResNet / SlowFast bottleneck as ONE launch (pv_conv3d pw2_*, round 6)  (emit.can_fuse_conv_bc)
    "fuse_stem": True,           # X3D stem (conv_xy + temporal depthwise + BN + ReLU) in one launch
    "fuse_kv_pool": True,        # MViT pool_k + pool_v as one depthwise launch                   (emit_mvit)
    "fuse_posenc": True,         # position tables added in the patch-embedding conv's epilogue   (emit_mvit)
    "split_joint_graph": True,   # sub-batches of SplitBatchDeployed as branches of ONE hipGraph  (conversion)
    "pool_stream_min_elems": 1 << 22,   # MViT pooling convs on grids at least this big use the plane-streaming kernel + a
                                        # separate per-head LayerNorm; smaller ones the fused pool + LayerNorm kernel  (emit_mvit)
    "fuse_ln_qkv": True,         # MViT norm1 + the q|k|v Linear as ONE launch (pv_ln_linear_rows)               (emit_mvit)
    "fuse_ln_qkv_max_c": 192,    # ... for token widths up to this (wider / shorter tensors: the LDS-DMA GEMM wins)
    "fuse_next_norm": True,      # MViT: norm1 of block i+1 written by block i's fused MLP from the rows it holds (emit_mvit.emit_mlp_fused)
    "fuse_mlp": True,            # MViT norm2 + fc1 + GELU + fc2 + residual as ONE launch (pv_mlp_rows)  (emit_mvit)
    "arena_margin": 0,           # bytes of zeros in front of and behind a plan's arena (diagnostics)
    "arena_guards": 0,           # debug build of the launch plan: every arena buffer gets its own memory (no re-use) followed
                                 # by this many bytes of canary; Session.check_guards() names the buffers a kernel wrote past
}


def get(name):
    return OPTIONS[name]


def apply(spec):
    """"k=v,k=v" (tools' --tune option): emitter options by name, everything else goes to the library (pv_tune_set)."""
    from ... import _lib as L
    for item in [t for t in (spec or "").split(",") if t]:
        k, v = item.split("=")
        if k in OPTIONS:
            OPTIONS[k] = type(OPTIONS[k])(int(v))
        else:
            L.tune(**{k: int(v)})
