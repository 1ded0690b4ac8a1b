"""Convert driver for the "mi355x" target (reference:
pytorchvideo/accelerator/deployment/mobile_cpu/utils/model_conversion.py:87-125).

Same contract as the reference driver: record every module's input size with forward
hooks during one eval forward, remove the hooks, deep-copy the model and call `convert()`
on every `EfficientBlockBase` top-down without descending into it.  Two MI355X additions:

  * the probing forward runs on a single clip (sizes are per-sample; the batch dimension
    is patched back), so converting for a batch of 32 does not need a 32-clip CPU forward;
  * all blocks of one model share a deploy `Session` (one arena, one launch plan).  When
    the converted model is a `Net` whose blocks are all MI355X blocks, consecutive blocks
    are chained zero-copy inside the plan and `model.forward` becomes one plan replay.
"""
import types
from copy import deepcopy

import torch
import torch.nn as nn

from ... import _lib as L
from ..efficient_blocks import EfficientBlockBase
from .session import Session


def _record_input_sizes(model, sample):
    lut, handles = {}, []

    def add(module, name):
        def hook(_m, _in, _out):
            if len(_in) > 0 and isinstance(_in[0], torch.Tensor):
                lut[name] = tuple(_in[0].size())
                if len(_in) > 1 and isinstance(_in[1], (list, tuple)) and len(_in[1]) == 3:
                    lut[name + "#thw"] = tuple(int(v) for v in _in[1])  # MultiScaleBlock(x, thw)
        handles.append(module.register_forward_hook(hook))
        for child_name, child in module.named_children():
            add(child, f"{name}.{child_name}")

    add(model, "")
    model.eval()
    with torch.no_grad():
        model(sample)
    for h in handles:
        h.remove()
    return lut


def _one_clip(x):
    if isinstance(x, torch.Tensor):
        return x[:1].detach().float().cpu()
    return [_one_clip(t) for t in x]


def _batch_of(x):
    return x.shape[0] if isinstance(x, torch.Tensor) else x[0].shape[0]


def _convert_children(module, lut, batch, name, sess, dtype, kwargs):
    if isinstance(module, EfficientBlockBase):
        size = lut.get(name)
        if size is not None:
            size = (batch,) + tuple(size[1:])
        extra = dict(kwargs)
        if (name + "#thw") in lut:
            extra["thw"] = lut[name + "#thw"]
        module.convert(size, session=sess, dtype=dtype, **extra)
        return
    for child_name, child in module.named_children():
        _convert_children(child, lut, batch, f"{name}.{child_name}", sess, dtype, kwargs)


def convert_to_deployable_form(model: nn.Module, input_tensor, convert_for_quantize: bool = False,
                               native_conv3d_op_qnnpack: bool = False, dtype=None, use_graph: bool = True):
    """Return a deploy-form copy of a transmuted `model`, specialised to `input_tensor`'s
    shape.  `dtype` (torch.bfloat16 | torch.float32) selects the kernels' storage type and
    defaults to the input tensor's dtype (fp32 input -> fp32 kernels)."""
    if dtype is None:
        t0 = input_tensor if isinstance(input_tensor, torch.Tensor) else input_tensor[0]
        dtype = torch.bfloat16 if t0.dtype == torch.bfloat16 else torch.float32
    L.lib()  # fail early and loudly when the HIP library is not built
    lut = {}
    if not _is_fusable_net(model) and not _is_fusable_mvit(model, input_tensor):
        lut = _record_input_sizes(model, _one_clip(input_tensor))
    converted = deepcopy(model)
    converted.eval()
    sess = Session(dtype=dtype)
    batch = _batch_of(input_tensor)
    fused = _try_fuse_net(converted, lut, batch, sess, dtype, input_tensor)
    if not fused and _is_fusable_mvit(converted, input_tensor):
        fused = _try_fuse_mvit(converted, sess, dtype, input_tensor)
        if not fused:  # a piece outside the blocks is unsupported: per-block conversion needs sizes
            sess = Session(dtype=dtype)
            converted = deepcopy(model)
            converted.eval()
            lut = _record_input_sizes(model, _one_clip(input_tensor))
    if not fused:
        _convert_children(converted, lut, batch, "", sess, dtype,
                          dict(convert_for_quantize=convert_for_quantize,
                               native_conv3d_op_qnnpack=native_conv3d_op_qnnpack))
    sess.finalize()
    if not fused:
        # Modules a transmuter declined stay in their original form (reference convention,
        # transmuter_mobile_cpu.py:21-22) and run as ordinary torch modules between the converted blocks:
        # they have to live where the activations do, in the plan's storage type.
        converted.to(device=sess.device, dtype=dtype)
    converted.__dict__["_pv_session"] = sess
    converted.__dict__["_pv_use_graph"] = use_graph
    return converted


def _is_fusable_net(model):
    from .blocks import Mi355xBlock
    blocks = getattr(model, "blocks", None)
    return (type(model).__name__ == "Net" and blocks is not None and len(blocks) > 0
            and all(isinstance(b, Mi355xBlock) for b in blocks))


# ------------------------------------------------------------------ whole-Net fusion
def _try_fuse_net(model, lut, batch, sess, dtype, input_tensor=None):
    """`Net` (models/net.py:11-44) whose blocks are all MI355X blocks: chain them inside the
    plan (block i+1 reads block i's arena buffers directly) and make forward one replay.
    Handles single-tensor nets (X3D, CSN, R(2+1)D, ResNet) and list-input nets (SlowFast)."""
    from .blocks import Mi355xBlock

    blocks = getattr(model, "blocks", None)
    if type(model).__name__ != "Net" or blocks is None or len(blocks) == 0:
        return False
    if not all(isinstance(b, Mi355xBlock) for b in blocks):
        return False
    multi = isinstance(input_tensor, (list, tuple))
    if multi:
        size = [tuple(t.shape) for t in input_tensor]
        if any(len(s) != 5 for s in size):
            return False
    else:
        size = tuple(input_tensor.shape)
        if len(size) != 5:
            return False
    cur, prev_owned = None, []
    for i, blk in enumerate(blocks):
        blk.convert(size if i == 0 else None, session=sess, input_ref=cur, dtype=dtype)
        if i > 0:
            for r in prev_owned:  # consumers of the previous block's outputs are emitted
                sess.release(r)
        cur = blk._out_ref
        prev_owned = list(cur) if isinstance(cur, list) else [cur]
    first_in, last = blocks[0]._in_ref, blocks[-1]
    model.__dict__["_pv_output"] = last._out_ref

    def fused_forward(self, x):
        s = self._pv_session
        if multi:
            assert isinstance(x, list), "input for MultiPathWayWithFuse needs to be a list of tensors"
            for t, ref in zip(x, first_in):
                if not s.matches(t, ref):
                    s.ingest(t, ref)
        elif not s.matches(x, first_in):
            s.ingest(x, first_in)
        s.launch(use_graph=self._pv_use_graph)
        out = last._out_ref
        if out.T == out.H == out.W == 1 and out.f32:
            return s.view_rows(out)[:, 0, :]
        return s.view(out)

    model.forward = types.MethodType(fused_forward, model)
    model.__dict__["_pv_inputs"] = first_in     # arena buffers a forward fills (transforms.DevicePacker)
    model.__dict__["_pv_result"] = lambda: fused_result(model)
    return True


def fused_result(model):
    s, out = model._pv_session, model._pv_output
    if out.T == out.H == out.W == 1 and out.f32:
        return s.view_rows(out)[:, 0, :]
    return s.view(out)


# ------------------------------------------------------------------ whole-MViT fusion
def _is_fusable_mvit(model, input_tensor):
    from .blocks import Mi355xMViTBlock
    blocks = getattr(model, "blocks", None)
    return (type(model).__name__ == "MultiscaleVisionTransformers" and blocks is not None and len(blocks) > 0
            and all(isinstance(b, Mi355xMViTBlock) for b in blocks)
            and isinstance(input_tensor, torch.Tensor) and input_tensor.dim() == 5
            and type(model.patch_embed).__name__ == "PatchEmbed")


def _try_fuse_mvit(model, sess, dtype, input_tensor):
    """MultiscaleVisionTransformers (models/vision_transformers.py:172-182) whose blocks are all
    MI355X blocks: patch embedding, cls/pos encoding, the blocks, the final norm and the head
    become one launch plan; forward = ingest + one replay."""
    from . import emit as E
    from . import emit_mvit as EM
    from ... import _lib as L

    B, Cc, T, H, W = [int(v) for v in input_tensor.shape]
    first_in = sess.alloc_input(B, T, H, W, Cc)
    n_ops = len(sess.ops)
    try:
        if not isinstance(model.pos_drop, (nn.Identity, nn.Dropout)):
            raise E.Unsupported("pos_drop")
        cur = EM.emit_patch_embed_and_pos(sess, model.patch_embed, model.cls_positional_encoding, first_in)
        # first_in stays live: it is re-filled by every forward
        for blk in model.blocks:
            blk.convert(None, session=sess, input_ref=cur, dtype=dtype)
            sess.release(cur)
            cur = blk._out_ref
        out = EM.emit_vit_head(sess, model.norm_embed, model.head, cur)
        sess.release(cur)
    except E.Unsupported:
        del sess.ops[n_ops:]
        return False

    def fused_forward(self, x):
        s = self._pv_session
        if not s.matches(x, first_in):
            s.ingest(x, first_in)
        s.launch(use_graph=self._pv_use_graph)
        return s.view_rows(out)[:, 0, :]

    model.forward = types.MethodType(fused_forward, model)
    model.__dict__["_pv_inputs"] = first_in
    model.__dict__["_pv_result"] = lambda: model._pv_session.view_rows(out)[:, 0, :]
    return True
