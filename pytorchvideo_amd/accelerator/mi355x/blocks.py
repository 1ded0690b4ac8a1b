"""MI355X efficient blocks and their transmuters.

A block *adopts* the reference module it replaces (same children, parameters, buffers and
attributes, so `state_dict()` keys and `model.blocks[i]` indexing are unchanged) and keeps
that module's own `forward` as its original form.  `convert()` emits the block's kernel
launches into a deploy `Session`; after that `forward` runs only HIP kernels through the
C ABI (it raises if the library or the GPU is missing -- there is no fallback).
"""
import torch
import torch.nn as nn

from ... import _lib as L
from ..efficient_blocks import EfficientBlockBase
from . import emit as E
from .session import Session


class Mi355xBlock(EfficientBlockBase):
    """Generic single-input / single-output block (stem, res stage, res block, head, pool)."""

    def __init__(self, original: nn.Module):
        super().__init__()
        # adopt the original module's state wholesale: children, params, buffers, hooks, plain attrs
        self.__dict__.update(original.__dict__)
        self.__dict__["_orig_cls"] = type(original)
        self.__dict__["convert_flag"] = False
        self.__dict__["_sess"] = None
        self.__dict__["_in_ref"] = None
        self.__dict__["_out_ref"] = None
        self.__dict__["_op_range"] = None
        self.__dict__["_owns_session"] = False

    # name shown in reprs / logs
    def _get_name(self):
        return "Mi355x[%s]" % self._orig_cls.__name__

    # -- original form ---------------------------------------------------------------
    def _original_forward(self, *args, **kwargs):
        return self._orig_cls.forward(self, *args, **kwargs)

    def __getattr__(self, name):
        # the adopted forward may use helpers defined on the original CLASS (static / class / plain methods,
        # properties, class constants): resolve them there once the instance lookup has failed
        try:
            return super().__getattr__(name)
        except AttributeError:
            oc = self.__dict__.get("_orig_cls")
            if oc is None or not hasattr(oc, name):
                raise
            import inspect
            attr = inspect.getattr_static(oc, name)
            if isinstance(attr, staticmethod):
                return attr.__func__
            if isinstance(attr, classmethod):
                return attr.__func__.__get__(oc, oc)
            if isinstance(attr, property):
                return attr.fget(self)
            if inspect.isfunction(attr):
                return attr.__get__(self, oc)
            return attr

    # -- conversion ------------------------------------------------------------------
    def _emit(self, sess, x_ref):
        return E.emit_module(sess, self, x_ref)

    def convert(self, input_blob_size, *args, session=None, input_ref=None, dtype=None, **kwargs):
        """Build the deploy form for inputs of `input_blob_size` (B,C,T,H,W).  Unknown
        kwargs of other devices (`convert_for_quantize`, `native_conv3d_op_qnnpack`) are
        accepted and ignored, as the reference's blocks do."""
        assert self.convert_flag is False, "already converted, cannot be converted again"
        self.eval()
        sess = session
        if sess is None:
            sess = Session(dtype=dtype or torch.bfloat16)
            self.__dict__["_owns_session"] = True
        if input_ref is None:
            B, Cc, T, H, W = [int(v) for v in input_blob_size]
            input_ref = sess.alloc_input(B, T, H, W, Cc)
        first = len(sess.ops)
        out_ref = self._emit(sess, input_ref)
        self.__dict__.update(_sess=sess, _in_ref=input_ref, _out_ref=out_ref, _op_range=(first, len(sess.ops)))
        if self._owns_session:
            sess.finalize()
        self.__dict__["convert_flag"] = True

    # -- deploy form -----------------------------------------------------------------
    def _deploy_forward(self, x):
        sess = self._sess
        sess.finalize()
        if not sess.matches(x, self._in_ref):
            sess.ingest(x, self._in_ref)
        sess.launch(*self._op_range)
        out = self._out_ref
        if out.T == out.H == out.W == 1 and out.f32:
            return sess.view_rows(out)[:, 0, :].clone()   # logits: a fresh tensor, like the reference's deploy form
        # a block-level activation is handed on as a zero-copy view of the session arena (the next block of the
        # same session recognises it and skips its ingest); it is valid until the next forward of this session
        return sess.view(out)

    def forward(self, *args, **kwargs):
        if self.convert_flag:
            return self._deploy_forward(*args, **kwargs)
        return self._original_forward(*args, **kwargs)


class Mi355xMultiPathBlock(Mi355xBlock):
    """MultiPathWayWithFuse / PoolConcatPathway: list-of-tensors input (SlowFast)."""

    def convert(self, input_blob_size, *args, session=None, input_ref=None, dtype=None, **kwargs):
        assert self.convert_flag is False, "already converted, cannot be converted again"
        self.eval()
        sess = session
        if sess is None:
            sess = Session(dtype=dtype or torch.bfloat16)
            self.__dict__["_owns_session"] = True
        if input_ref is None:
            # input_blob_size: list of (B,C,T,H,W), one per pathway
            input_ref = [sess.alloc_input(int(s[0]), int(s[2]), int(s[3]), int(s[4]), int(s[1])) for s in input_blob_size]
        first = len(sess.ops)
        pre = None
        if self._orig_cls.__name__ == "PoolConcatPathway":
            out_ref = E.emit_pool_concat(sess, self, list(input_ref))
            if self.retain_list:
                out_ref = [out_ref]
        else:
            out_ref, pre = E.emit_multipathway(sess, self, list(input_ref))
        self.__dict__.update(_sess=sess, _in_ref=list(input_ref), _out_ref=out_ref, _pre_ref=pre,
                             _op_range=(first, len(sess.ops)))
        if self._owns_session:
            sess.finalize()
        self.__dict__["convert_flag"] = True

    def _deploy_forward(self, x):
        assert isinstance(x, list), "input for MultiPathWayWithFuse needs to be a list of tensors"
        sess = self._sess
        sess.finalize()
        for t, ref in zip(x, self._in_ref):
            if not sess.matches(t, ref):
                sess.ingest(t, ref)
        sess.launch(*self._op_range)
        if getattr(self, "inplace", False) and self._pre_ref is not None:
            for i, r in enumerate(self._pre_ref):  # the reference overwrites the caller's list (net.py:111-118)
                x[i] = sess.view(r)
        out = self._out_ref
        return [sess.view(r) for r in out] if isinstance(out, list) else sess.view(out)


class Mi355xRoIHeadBlock(Mi355xBlock):
    """ResNetRoIHead (models/head.py:394-482): forward(x, bboxes).  The deploy form is specialised to the
    feature size AND the number of boxes; the box values are read on the device at every replay."""

    def convert(self, input_blob_size, *args, session=None, input_ref=None, dtype=None, num_boxes=None, **kwargs):
        assert self.convert_flag is False, "already converted, cannot be converted again"
        if num_boxes is None or int(num_boxes) <= 0:
            raise L.PvError("ResNetRoIHead.convert needs num_boxes (the deploy form is specialised to the box count)")
        self.eval()
        sess = session
        if sess is None:
            sess = Session(dtype=dtype or torch.bfloat16)
            self.__dict__["_owns_session"] = True
        if input_ref is None:
            B, Cc, T, H, W = [int(v) for v in input_blob_size]
            input_ref = sess.alloc_act(B, T, H, W, Cc)
        boxes = sess.alloc_boxes(int(num_boxes))
        first = len(sess.ops)
        out_ref = E.emit_roi_head(sess, self, input_ref, boxes, int(num_boxes))
        self.__dict__.update(_sess=sess, _in_ref=input_ref, _out_ref=out_ref, _op_range=(first, len(sess.ops)),
                             _boxes=boxes, _num_boxes=int(num_boxes))
        if self._owns_session:
            sess.finalize()
        self.__dict__["convert_flag"] = True

    def _result(self):
        sess, out = self._sess, self._out_ref
        if out.T == out.H == out.W == 1:
            return sess.view_rows(out)[:, 0, :]
        return sess.view(out)

    def _deploy_forward(self, x, bboxes):
        sess = self._sess
        sess.finalize()
        if not sess.matches(x, self._in_ref):
            sess.ingest(x, self._in_ref)
        sess.load_boxes(bboxes, self._boxes, self._num_boxes)
        sess.launch(*self._op_range)
        return self._result().clone()


# ------------------------------------------------------------------------- transmuters
_SINGLE_IO = ("ResNetBasicStem", "ResStage", "ResBlock", "ResNetBasicHead")


def _probe(module):
    """Structural support check without geometry: walk the module with the emitters' own
    predicates.  Returns True when every piece is something the kernels implement."""
    n = type(module).__name__
    try:
        if n == "ResNetBasicStem":
            conv = module.conv
            E.act_code(module.activation)
            if type(conv).__name__ == "Conv2plus1d":
                E.check_conv3d(conv.conv_t), E.check_conv3d(conv.conv_xy)
                E.act_code(conv.activation), E.fold_norm(conv.norm, _out_ch(conv, first=True))
            else:
                E.check_conv3d(conv)
            _check_norm(module.norm)
            if module.pool is not None and not isinstance(module.pool, (nn.MaxPool3d, nn.AvgPool3d)):
                return False
            return True
        if n == "ResStage":
            return all(type(b).__name__ == "ResBlock" and _probe(b) for b in module.res_blocks)
        if n == "ResBlock":
            if not E.is_add_fusion(module.branch_fusion) or type(module.branch2).__name__ != "BottleneckBlock":
                return False
            E.act_code(module.activation)
            if module.branch1_conv is not None:
                if E.check_conv3d(module.branch1_conv):
                    return False
                _check_norm(module.branch1_norm)
            bb = module.branch2
            if E.check_conv3d(bb.conv_a) or E.check_conv3d(bb.conv_c):
                return False
            E.act_code(bb.act_a), E.act_code(bb.act_b)
            _check_norm(bb.norm_a), _check_norm(bb.norm_c)
            bn, se = E._split_norm_b(bb.norm_b)
            _check_norm(bn)
            if type(bb.conv_b).__name__ == "Conv2plus1d":
                if se is not None:
                    return False
                E.check_conv3d(bb.conv_b.conv_t), E.check_conv3d(bb.conv_b.conv_xy)
                E.act_code(bb.conv_b.activation), _check_norm(bb.conv_b.norm)
            else:
                dw = E.check_conv3d(bb.conv_b)
                if se is not None and not dw:
                    return False
            return True
        if n == "ResNetBasicHead":
            if module.pool is not None and type(module.pool).__name__ != "ProjectedPool" and \
                    not isinstance(module.pool, (nn.AvgPool3d, nn.MaxPool3d, nn.AdaptiveAvgPool3d)):
                return False
            if not isinstance(module.proj, nn.Linear):
                return False
            if module.activation is not None and not isinstance(module.activation, (nn.Softmax, nn.Sigmoid, nn.ReLU)):
                return False
            if module.dropout is not None and not isinstance(module.dropout, nn.Dropout):
                return False
            return True
    except (E.Unsupported, AttributeError):
        return False
    return False


def _check_norm(norm):
    if norm is None or isinstance(norm, nn.Identity):
        return
    if not isinstance(norm, nn.modules.batchnorm._BatchNorm) or norm.running_mean is None:
        raise E.Unsupported("norm %s" % type(norm).__name__)


def _out_ch(conv2plus1d, first):
    c = conv2plus1d.conv_xy if (conv2plus1d.conv_xy_first == first) else conv2plus1d.conv_t
    return c.out_channels


def transmute_single_io(module: nn.Module):
    """Stem / stage / res-block / head -> Mi355xBlock, or None (decline)."""
    if isinstance(module, EfficientBlockBase):
        return None
    if type(module).__name__ not in _SINGLE_IO or not _probe(module):
        return None
    return Mi355xBlock(module)


def transmute_multipath(module: nn.Module):
    """SlowFast containers: MultiPathWayWithFuse (+FuseFastToSlow | Identity) and PoolConcatPathway."""
    if isinstance(module, EfficientBlockBase):
        return None
    n = type(module).__name__
    try:
        if n == "MultiPathWayWithFuse":
            for b in module.multipathway_blocks:
                if b is not None and not (type(b).__name__ in ("ResNetBasicStem", "ResStage") and _probe(b)):
                    return None
            f = module.multipathway_fusion
            if f is None or isinstance(f, nn.Identity):
                return Mi355xMultiPathBlock(module)
            if type(f).__name__ != "FuseFastToSlow" or len(module.multipathway_blocks) != 2:
                return None
            if E.check_conv3d(f.conv_fast_to_slow):
                return None
            _check_norm(f.norm), E.act_code(f.activation)
            return Mi355xMultiPathBlock(module)
        if n == "PoolConcatPathway":
            if module.pool is None or module.dim != 1:
                return None
            if not all(isinstance(p, (nn.AvgPool3d, nn.MaxPool3d, nn.AdaptiveAvgPool3d)) for p in module.pool):
                return None
            return Mi355xMultiPathBlock(module)
    except (E.Unsupported, AttributeError):
        return None
    return None


def transmute_pool(module: nn.Module):
    """Bare pooling layers that sit between blocks (e.g. create_resnet's stage1_pool)."""
    if isinstance(module, (nn.MaxPool3d, nn.AvgPool3d)) and not getattr(module, "ceil_mode", False):
        return Mi355xBlock(module)
    return None


def transmute_roi_head(module: nn.Module):
    """Detection head (ResNetRoIHead with RoIAlign) -> Mi355xRoIHeadBlock, or None (decline)."""
    if isinstance(module, EfficientBlockBase) or type(module).__name__ != "ResNetRoIHead":
        return None
    try:
        E.check_roi_head(module)
    except (E.Unsupported, AttributeError):
        return None
    return Mi355xRoIHeadBlock(module)


EFFICIENT_BLOCK_TRANSMUTER_MI355X = [transmute_single_io, transmute_multipath, transmute_pool, transmute_roi_head]


# ------------------------------------------------------------------------- MViT
class Mi355xMViTBlock(Mi355xBlock):
    """MultiScaleBlock (layers/attention.py:578-757): forward(x, thw) -> (x', thw').  The
    token tensor (B, N, C) lives in the arena as a channels-last activation with W = N."""

    def convert(self, input_blob_size, *args, session=None, input_ref=None, dtype=None, thw=None, **kwargs):
        from . import emit_mvit as EM
        assert self.convert_flag is False, "already converted, cannot be converted again"
        self.eval()
        if input_ref is None and thw is None:
            if session is not None:
                raise L.PvError("MultiScaleBlock.convert needs thw=(T,H,W) of the token grid")
            # The reference's convert driver records only the size of a module's first input
            # (model_conversion.py:13-43), not the grid: finish the conversion at the first forward, which
            # brings thw along (the deploy form is specialised to that grid from then on).
            self.__dict__.update(_pending=(tuple(int(v) for v in input_blob_size), dtype), convert_flag=True)
            return
        sess = session
        if sess is None:
            sess = Session(dtype=dtype or torch.bfloat16)
            self.__dict__["_owns_session"] = True
        if input_ref is None:
            B, N, Cc = [int(v) for v in input_blob_size]
            input_ref = sess.alloc_act(B, 1, 1, N, Cc, f32=True)  # the residual stream is fp32
            input_ref.thw, input_ref.has_cls = tuple(int(v) for v in thw), bool(self.has_cls_embed)
        first = len(sess.ops)
        out_ref = EM.emit_multiscale_block(sess, self, input_ref)
        self.__dict__.update(_sess=sess, _in_ref=input_ref, _out_ref=out_ref, _op_range=(first, len(sess.ops)))
        if self._owns_session:
            sess.finalize()
        self.__dict__["convert_flag"] = True

    def _deploy_forward(self, x, thw_shape=None):
        pending = self.__dict__.pop("_pending", None)
        if pending is not None:
            if thw_shape is None:
                raise L.PvError("the first forward of a lazily converted MultiScaleBlock needs thw_shape")
            size, dtype = pending
            self.__dict__["convert_flag"] = False
            self.convert(size, dtype=dtype, thw=thw_shape)
        sess = self._sess
        sess.finalize()
        ref = self._in_ref
        if thw_shape is not None and tuple(int(v) for v in thw_shape) != tuple(ref.thw):
            raise L.PvError("deploy form was converted for grid %s, got %s" % (tuple(ref.thw), tuple(thw_shape)))
        sess.ingest_rows(x, ref)
        sess.launch(*self._op_range)
        out = self._out_ref
        return sess.view_rows(out), list(out.thw)


def _probe_mvit_block(module):
    """Structural check of a MultiScaleBlock by dry-emitting it into a scratch session."""
    from . import emit_mvit as EM
    try:
        scratch = Session(dtype=torch.bfloat16)
        T, H, W = 2, 8, 8
        cls = 1 if module.has_cls_embed else 0
        x = scratch.alloc_act(1, 1, 1, T * H * W + cls, module.dim, f32=True)
        x.thw, x.has_cls = (T, H, W), bool(cls)
        EM.emit_multiscale_block(scratch, module, x)
        return True
    except (E.Unsupported, AttributeError, RuntimeError):
        return False


def transmute_mvit_block(module: nn.Module):
    if isinstance(module, EfficientBlockBase) or type(module).__name__ != "MultiScaleBlock":
        return None
    if not _probe_mvit_block(module):
        return None
    return Mi355xMViTBlock(module)


EFFICIENT_BLOCK_TRANSMUTER_MI355X.append(transmute_mvit_block)
