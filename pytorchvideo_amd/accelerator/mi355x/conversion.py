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
        def pre_hook(_m, _in):
            # list-of-tensors input (SlowFast's MultiPathWayWithFuse / PoolConcatPathway, net.py:107-122): the
            # reference's hook skips these (model_conversion.py:26-28); the MI355X multi-pathway blocks need them.
            # Recorded BEFORE the forward: MultiPathWayWithFuse overwrites the caller's list in place (net.py:111-118)
            if len(_in) > 0 and isinstance(_in[0], (list, tuple)) and len(_in[0]) > 0 and \
                    all(isinstance(t, torch.Tensor) for t in _in[0]):
                lut[name] = [tuple(t.size()) for t in _in[0]]

        def hook(_m, _in, _out):
            if len(_in) > 0 and isinstance(_in[0], torch.Tensor):
                lut[name] = tuple(_in[0].size())
                if len(_in) > 1 and isinstance(_in[1], (list, tuple)) and len(_in[1]) == 3:
                    lut[name + "#thw"] = tuple(int(v) for v in _in[1])  # MultiScaleBlock(x, thw)
                if len(_in) > 1 and isinstance(_in[1], torch.Tensor) and _in[1].dim() == 2:
                    lut[name + "#boxes"] = int(_in[1].shape[0])        # ResNetRoIHead(x, bboxes)
        handles.append(module.register_forward_pre_hook(pre_hook))
        handles.append(module.register_forward_hook(hook))
        for child_name, child in module.named_children():
            add(child, f"{name}.{child_name}")

    add(model, "")
    model.eval()
    with torch.no_grad():
        model(*sample) if isinstance(sample, tuple) else model(sample)
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
        # the probing forward ran on ONE clip: scale the leading dimension (it is B for the blocks of a Net and
        # B*heads for a pool inside a declined attention module) instead of assuming it is the batch
        if isinstance(size, list):
            size = [(s[0] * batch,) + tuple(s[1:]) for s in size]
        elif size is not None:
            size = (size[0] * batch,) + tuple(size[1:])
        extra = dict(kwargs)
        if (name + "#thw") in lut:
            extra["thw"] = lut[name + "#thw"]
        if (name + "#boxes") in lut:
            extra["num_boxes"] = lut[name + "#boxes"]
        module.convert(size, session=sess, dtype=dtype, **extra)
        return
    for child_name, child in module.named_children():
        _convert_children(child, lut, batch, f"{name}.{child_name}", sess, dtype, kwargs)


class SplitBatchDeployed(nn.Module):
    """`streams` > 1: the batch runs as `streams` independent sub-batches, each through its own deploy form
    (own arena, own hipGraph) on its own HIP stream.  Clips do not interact in an eval forward, so the result is
    the one-plan result row for row; what changes is the schedule: kernels of different sub-batches overlap, so
    the tail of one launch (its last, partial round of workgroups) and the ~2-4 us gap before its successor are
    filled with the other sub-batch's work.  Measured (tools/gpu_stream_sweep.sh, same box, 1 -> 2 sub-batches):
    X3D-M b=32 +10 %, X3D-L b=32 +10 %, MViT-B b=8 +7 %, SlowFast-R50 b=16 -2 % (its kernels already fill the
    chip); 3 and 4 sub-batches give nothing more.  By default the sub-plans are captured as parallel branches of
    ONE hipGraph (pv_joint_*: its own handle, untouched by whatever the parts' sessions do with their own graphs): a
    forward is the ingests, one graph launch and one concatenation.  Whether a graph is used is decided at FORWARD
    time from the parts' `_pv_use_graph` (bench.py --no-graph flips it after construction)."""

    def __init__(self, parts, splits, device, joint=True):
        super().__init__()
        self.parts = nn.ModuleList(parts)
        self._splits = splits
        # the last sub-batch runs on the caller's stream: k sub-batches occupy k hardware queues, not k + 1
        # (ROCm maps streams onto 4 hardware queues by default; a fifth stream shares one and serialises)
        self._streams = [torch.cuda.Stream(device=device) for _ in parts[:-1]]
        self._device = device
        self.__dict__["_pv_session"] = parts[0]._pv_session      # per-op profiling: the first sub-plan
        self.__dict__["_pv_sessions"] = [p._pv_session for p in parts]
        # one graph with the sub-plans as parallel branches (one launch per forward) when every part is a
        # whole-model plan with a known input buffer; else one graph per part on its own stream
        self._joint = joint and all(getattr(p, "_pv_inputs", None) is not None and hasattr(p, "_pv_result") for p in parts)
        self.__dict__["_joint_handle"] = None
        self.__dict__["_joint_ops"] = None       # (generation, op count) of the member plans the joint graph was captured from

    def __del__(self):
        h = self.__dict__.get("_joint_handle")
        if h:
            try:
                L.lib().pv_joint_destroy(h)
            except Exception:
                pass

    def _use_joint(self):
        return self._joint and all(getattr(p, "_pv_use_graph", False) for p in self.parts)

    def _launch_joint(self):
        """One launch of the joint graph (built on first use, rebuilt when a member plan changed)."""
        import ctypes as C
        lib = L.lib()
        s0 = self.parts[0]._pv_session
        with torch.cuda.device(self._device):
            # (generation of the plan build, op count) per member: a re-finalized plan with the same op count is a NEW plan
            ops = [(getattr(p._pv_session, "generation", 0), lib.pv_plan_size(p._pv_session.plan)) for p in self.parts]
            if self._joint_handle is None or ops != self._joint_ops:
                if self._joint_handle is None:
                    self.__dict__["_joint_handle"] = lib.pv_joint_create()
                arr = (C.c_void_p * len(self.parts))(*[p._pv_session.plan for p in self.parts])
                L.check(lib.pv_joint_build(self._joint_handle, arr, len(self.parts)), "joint graph build")
                self.__dict__["_joint_ops"] = ops
            L.check(lib.pv_joint_launch(self._joint_handle, s0._stream()), "joint graph launch")

    def _pv_launch(self):
        """Run every sub-plan on inputs that are already in the parts' input buffers (transforms.DevicePacker)."""
        if self._use_joint():
            self._launch_joint()
        else:
            for p in self.parts:
                p._pv_session.launch(use_graph=p._pv_use_graph)

    def _pv_result(self):
        return torch.cat([p._pv_result() for p in self.parts], dim=0)     # cat copies: a fresh tensor

    def _forward_joint(self, x, multi):
        lo = 0
        for part, b in zip(self.parts, self._splits):
            xc = [t[lo:lo + b] for t in x] if multi else x[lo:lo + b]
            lo += b
            if not multi and xc.dim() == 4:
                xc = xc.unsqueeze(2)                    # image model: a clip of one frame
            _ingest_inputs(part._pv_session, xc, part._pv_inputs, multi)
        self._launch_joint()
        return self._pv_result()

    def forward(self, x):
        multi = isinstance(x, (list, tuple))
        n = (x[0] if multi else x).shape[0]
        if n != sum(self._splits):
            raise L.PvError("deploy form was converted for a batch of %d, got %d" % (sum(self._splits), n))
        if self._use_joint():
            return self._forward_joint(x, multi)
        cs = torch.cuda.current_stream(self._device)
        outs, lo = [], 0
        for i, (part, b) in enumerate(zip(self.parts, self._splits)):
            xc = [t[lo:lo + b] for t in x] if multi else x[lo:lo + b]
            lo += b
            if i < len(self._streams):
                st = self._streams[i]
                st.wait_stream(cs)                 # the caller's input is ready
                with torch.cuda.stream(st):
                    o = part(xc)
                o.record_stream(cs)
            else:
                o = part(xc)
            outs.append(o)
        for st in self._streams:
            cs.wait_stream(st)
        return torch.cat(outs, dim=0)


def convert_to_deployable_form(model: nn.Module, input_tensor, convert_for_quantize: bool = False,
                               native_conv3d_op_qnnpack: bool = False, dtype=None, use_graph: bool = True,
                               streams: int = 1, emit_only: bool = False):
    """Return a deploy-form copy of a transmuted `model`, specialised to `input_tensor`'s
    shape.  `dtype` (torch.bfloat16 | torch.float32) selects the kernels' storage type and
    defaults to the input tensor's dtype (fp32 input -> fp32 kernels).  `streams` > 1: see SplitBatchDeployed.
    `emit_only=True` runs only the HOST half of the conversion (BatchNorm folding, weight packing, arena planning; no device
    is touched) and returns the plan's statistics as a dict -- {"fused", "ops", "arena_bytes", "weight_bytes"}, or with
    `streams` > 1 {"streams": [one such dict per sub-batch]} -- instead of a module; detection networks do not support it."""
    if emit_only and type(model).__name__ == "DetectionBBoxNetwork":
        raise NotImplementedError("emit_only is not implemented for DetectionBBoxNetwork (its conversion needs the box list on the device)")
    if streams > 1 and type(model).__name__ != "DetectionBBoxNetwork":
        n = _batch_of(input_tensor)
        k = min(int(streams), n)
        splits = [n // k + (1 if i < n % k else 0) for i in range(k)]
        parts, lo = [], 0
        for b in splits:
            xc = [t[lo:lo + b] for t in input_tensor] if isinstance(input_tensor, (list, tuple)) else input_tensor[lo:lo + b]
            lo += b
            parts.append(convert_to_deployable_form(model, xc, convert_for_quantize, native_conv3d_op_qnnpack,
                                                    dtype=dtype, use_graph=use_graph, emit_only=emit_only))
        if emit_only:
            return {"streams": parts}
        from . import tuning
        return SplitBatchDeployed(parts, splits, parts[0]._pv_session.device, joint=tuning.get("split_joint_graph"))
    if type(model).__name__ == "DetectionBBoxNetwork":
        return _convert_detection(model, input_tensor, dtype, use_graph,
                                  dict(convert_for_quantize=convert_for_quantize,
                                       native_conv3d_op_qnnpack=native_conv3d_op_qnnpack))
    if dtype is None:
        dtype = _default_dtype(input_tensor)
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
    if emit_only:
        # The host half of a conversion (BatchNorm folding in fp64, weight packing, arena planning, descriptors), no device:
        # what `bench.py --dry-host --dry-convert` times with one process per GPU of a node (input_tensor may be a stride-0 view,
        # only its shape and dtype are read).  Returns the plan's vital statistics instead of a module.
        return {"fused": bool(fused), "ops": len(sess.ops), "arena_bytes": int(sess._arena.peak), "weight_bytes": int(sess._wtop)}
    sess.finalize()
    if not fused:
        # Modules a transmuter declined stay in their original form (reference convention,
        # transmuter_mobile_cpu.py:21-22) and run as ordinary torch modules between the converted blocks:
        # they have to live where the activations do, in the plan's storage type -- except around MViT blocks,
        # whose token stream is fp32 in every plan (position encoding before them, final norm and head after them).
        from .blocks import Mi355xMViTBlock
        tokens = any(isinstance(b, Mi355xMViTBlock) for b in converted.modules())
        converted.to(device=sess.device, dtype=torch.float32 if tokens else dtype)
    converted.__dict__["_pv_session"] = sess
    converted.__dict__["_pv_use_graph"] = use_graph
    return converted


def _default_dtype(x):
    while not isinstance(x, torch.Tensor):
        x = x[0]
    return torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32


def _is_fusable_net(model):
    from .blocks import Mi355xBlock
    blocks = getattr(model, "blocks", None)
    return (type(model).__name__ == "Net" and blocks is not None and len(blocks) > 0
            and all(isinstance(b, Mi355xBlock) for b in blocks))


# ------------------------------------------------------------------ whole-Net fusion
def _chain_net_blocks(model, batch, sess, dtype, input_tensor):
    """Emit the blocks of a `Net` (models/net.py:11-44) whose blocks are all MI355X blocks back to back
    into `sess`: block i+1 reads block i's arena buffers directly.  Returns (input refs, multi-pathway?)
    or None when the model is not such a net.  Handles single-tensor nets (X3D, CSN, R(2+1)D, ResNet)
    and list-input nets (SlowFast)."""
    if not _is_fusable_net(model):
        return None
    blocks = model.blocks
    multi = isinstance(input_tensor, (list, tuple))
    if multi:
        size = [tuple(t.shape) for t in input_tensor]
        if any(len(s) != 5 for s in size):
            return None
    else:
        size = tuple(input_tensor.shape)
        if len(size) != 5:
            return None
    cur, prev_owned = None, []
    for i, blk in enumerate(blocks):
        blk.convert(size if i == 0 else None, session=sess, input_ref=cur, dtype=dtype)
        if i > 0:
            for r in prev_owned:  # consumers of the previous block's outputs are emitted
                sess.release(r)
        cur = blk._out_ref
        prev_owned = list(cur) if isinstance(cur, list) else [cur]
    model.__dict__["_pv_output"] = blocks[-1]._out_ref
    model.__dict__["_pv_inputs"] = blocks[0]._in_ref     # arena buffers a forward fills (transforms.DevicePacker)
    return blocks[0]._in_ref, multi


def _ingest_inputs(s, x, first_in, multi):
    if multi:
        assert isinstance(x, list), "input for MultiPathWayWithFuse needs to be a list of tensors"
        for t, ref in zip(x, first_in):
            if not s.matches(t, ref):
                s.ingest(t, ref)
    elif not s.matches(x, first_in):
        s.ingest(x, first_in)


def _try_fuse_net(model, lut, batch, sess, dtype, input_tensor=None):
    """A fusable `Net`: chain its blocks inside the plan and make forward one replay."""
    chained = _chain_net_blocks(model, batch, sess, dtype, input_tensor)
    if chained is None:
        return False
    first_in, multi = chained

    def fused_forward(self, x):
        s = self._pv_session
        _ingest_inputs(s, x, first_in, multi)
        s.launch(use_graph=self._pv_use_graph)
        return fused_result(self)

    model.forward = types.MethodType(fused_forward, model)
    model.__dict__["_pv_result"] = lambda: fused_result(model, zero_copy=True)
    return True


# ------------------------------------------------------------------ detection (backbone + RoI head)
def _convert_detection(model, inputs, dtype, use_graph, kwargs):
    """DetectionBBoxNetwork (models/net.py:47-74): `inputs` = (x, bboxes) exactly as forward takes them
    (x a clip, or [slow, fast] for SlowFast; bboxes [R,5]).  When the backbone is a fusable Net and the
    head an MI355X RoI head, forward(x, bboxes) is one replay of one plan: ingest, backbone, RoIAlign +
    max pool, projection + sigmoid.  Otherwise every converted block runs on its own, in the original
    module order."""
    from .blocks import Mi355xRoIHeadBlock
    if not (isinstance(inputs, (tuple, list)) and len(inputs) == 2 and isinstance(inputs[1], torch.Tensor)
            and inputs[1].dim() == 2 and inputs[1].shape[1] == 5):
        raise RuntimeError("a DetectionBBoxNetwork is converted for (x, bboxes) with bboxes of shape [R, 5]")
    x, bboxes = inputs
    n_boxes, batch = int(bboxes.shape[0]), _batch_of(x)
    if dtype is None:
        dtype = _default_dtype(x)
    L.lib()
    fusable = _is_fusable_net(model.model) and isinstance(model.detection_head, Mi355xRoIHeadBlock)
    lut = {}
    if not fusable:
        probe_boxes = bboxes.detach().float().cpu().clone()
        probe_boxes[:, 0] = 0                       # the probing forward runs on one clip
        lut = _record_input_sizes(model, (_one_clip(x), probe_boxes))
    converted = deepcopy(model)
    converted.eval()
    sess = Session(dtype=dtype)
    if fusable:
        first_in, multi = _chain_net_blocks(converted.model, batch, sess, dtype, x)
        features = converted.model._pv_output
        n_backbone = len(sess.ops)
        head = converted.detection_head
        head.convert(None, session=sess, input_ref=features, dtype=dtype, num_boxes=n_boxes)
        sess.release(features)

        def backbone_forward(self, x):              # model.model on its own: the backbone's share of the plan
            s = self._pv_session
            _ingest_inputs(s, x, first_in, multi)
            s.launch(0, n_backbone)
            return fused_result(self)

        def fused_forward(self, x, bboxes):
            s = self._pv_session
            _ingest_inputs(s, x, first_in, multi)
            s.load_boxes(bboxes, head._boxes, head._num_boxes)
            s.launch(use_graph=self._pv_use_graph)
            out = head._result()
            return out.reshape(out.shape[0], -1).clone()    # net.py:74; fresh tensor, not a view of the arena

        converted.model.forward = types.MethodType(backbone_forward, converted.model)
        converted.model.__dict__["_pv_session"] = sess
        converted.forward = types.MethodType(fused_forward, converted)
        converted.__dict__["_pv_inputs"] = first_in   # transforms.DevicePacker: packed clip + boxes -> one replay
        converted.__dict__["_pv_load_boxes"] = lambda b: sess.load_boxes(b, head._boxes, head._num_boxes)
        converted.__dict__["_pv_result"] = lambda: head._result().reshape(head._num_boxes, -1)
    else:
        _convert_children(converted, lut, batch, "", sess, dtype, kwargs)
    sess.finalize()
    if not fusable:
        converted.to(device=sess.device, dtype=dtype)
    converted.__dict__["_pv_session"] = sess
    converted.__dict__["_pv_use_graph"] = use_graph
    return converted


def fused_result(model, zero_copy=False):
    """Result of the last replay.  The logits row is returned as a FRESH tensor (the reference's deploy form
    returns fresh tensors; a view of the arena would be overwritten by the next forward).  `zero_copy=True`
    (what `model._pv_result()` hands out) is the explicit opt-in for the arena view, valid until the next forward."""
    s, out = model._pv_session, model._pv_output
    if out.T == out.H == out.W == 1 and out.f32:
        v = s.view_rows(out)[:, 0, :]
        return v if zero_copy else v.clone()
    return s.view(out)


# ------------------------------------------------------------------ whole-MViT fusion
def _is_fusable_mvit(model, input_tensor):
    from .blocks import Mi355xMViTBlock
    blocks = getattr(model, "blocks", None)
    return (type(model).__name__ == "MultiscaleVisionTransformers" and blocks is not None and len(blocks) > 0
            and all(isinstance(b, Mi355xMViTBlock) for b in blocks)
            and isinstance(input_tensor, torch.Tensor) and type(model.patch_embed).__name__ == "PatchEmbed"
            and input_tensor.dim() == (4 if isinstance(model.patch_embed.patch_model, nn.Conv2d) else 5))


def _try_fuse_mvit(model, sess, dtype, input_tensor):
    """MultiscaleVisionTransformers (models/vision_transformers.py:172-182) whose blocks are all
    MI355X blocks: patch embedding, cls/pos encoding, the blocks, the final norm and the head
    become one launch plan; forward = ingest + one replay."""
    from . import emit as E
    from . import emit_mvit as EM
    from ... import _lib as L

    # an image model (use_2d_patch, vision_transformers.py:301-312) takes [B,C,H,W]: a clip of one frame
    image = input_tensor.dim() == 4
    B, Cc, T, H, W = [int(v) for v in (input_tensor.unsqueeze(2) if image else input_tensor).shape]
    first_in = sess.alloc_input(B, T, H, W, Cc)
    n_ops = len(sess.ops)
    try:
        if not isinstance(model.pos_drop, (nn.Identity, nn.Dropout)):
            raise E.Unsupported("pos_drop")
        cur = EM.emit_patch_embed_and_pos(sess, model.patch_embed, model.cls_positional_encoding, first_in)
        # first_in stays live: it is re-filled by every forward
        for i, blk in enumerate(model.blocks):
            # the fused MLP of block i can write norm1 of block i + 1 (emit_mvit.emit_mlp_fused)
            blk.__dict__["_pv_next_block"] = model.blocks[i + 1] if i + 1 < len(model.blocks) else None
            try:
                blk.convert(None, session=sess, input_ref=cur, dtype=dtype)
            finally:       # never leave the hint behind: a block converted standalone later must not emit `yn` for a stranger
                blk.__dict__.pop("_pv_next_block", None)
            sess.release(cur)
            cur = blk._out_ref
        pre = getattr(cur, "prenorm", None)     # (a next-block operand nobody consumed would stay live in the arena)
        if pre is not None:
            cur.prenorm = None
            sess.release(pre[0])
        out = EM.emit_vit_head(sess, model.norm_embed, model.head, cur)
        sess.release(cur)
    except E.Unsupported:
        del sess.ops[n_ops:]
        return False

    def fused_forward(self, x):
        s = self._pv_session
        if image:
            if x.dim() != 4:
                raise L.PvError("deploy form was converted for images [B,C,H,W], got %s" % (tuple(x.shape),))
            x = x.unsqueeze(2)
        if not s.matches(x, first_in):
            s.ingest(x, first_in)
        s.launch(use_graph=self._pv_use_graph)
        return s.view_rows(out)[:, 0, :].clone()   # fresh tensor; `_pv_result()` is the zero-copy view

    model.forward = types.MethodType(fused_forward, model)
    model.__dict__["_pv_inputs"] = first_in
    model.__dict__["_pv_result"] = lambda: model._pv_session.view_rows(out)[:, 0, :]
    return True
