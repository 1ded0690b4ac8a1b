"""Deploy session: records the launch list of a converted model and owns its device memory.

A deploy-form model is specialised to one input size (reference contract,
pytorchvideo/accelerator/deployment/mobile_cpu/utils/model_conversion.py:100-103), so
everything is decided once at convert time:

  * activations live in one arena (a single device allocation) at offsets chosen by a
    first-fit allocator with explicit release -> buffers are re-used as soon as their
    consumers have been emitted, which keeps the working set small enough for producer ->
    consumer re-use through the 256 MiB Infinity Cache;
  * packed weights / folded-BN vectors live in one blob uploaded once;
  * the ops are appended to a C-side plan (`pv_plan_*`) and replayed with a single call per
    forward (optionally as a captured hipGraph).

No compute happens in Python and nothing here falls back to torch ops: without the HIP
library or without a GPU, `finalize()` raises.
"""
import ctypes as C

import torch

from ... import _lib as L

_ALIGN = 256
_CANARY = 0xA5


def _round_up(v, m):
    return (v + m - 1) // m * m


def pad8(c):
    return _round_up(c, 8)


class Ptr:
    """Symbolic device pointer: (space, byte offset), resolved at finalize()."""

    __slots__ = ("space", "off")

    def __init__(self, space, off):
        self.space, self.off = space, off


class TRef:
    """Channels-last activation inside the arena.  Voxel (b,t,h,w) lives at
    off + (b*bs + ((t*H+h)*W+w)*ld) * itemsize.  Token tensors use T=H=1, W=N and remember
    their (t,h,w) grid and cls-prefix separately."""

    __slots__ = ("off", "B", "T", "H", "W", "C", "ld", "bs", "itemsize", "nbytes", "owned", "f32",
                 "thw", "has_cls", "prenorm")

    def __init__(self, off, B, T, H, W, C, ld, bs, itemsize, nbytes, owned=True, f32=False):
        self.off, self.B, self.T, self.H, self.W, self.C = off, B, T, H, W, C
        self.ld, self.bs, self.itemsize, self.nbytes, self.owned, self.f32 = ld, bs, itemsize, nbytes, owned, f32
        self.thw, self.has_cls = None, False
        self.prenorm = None     # (TRef, norm module): LayerNorm of this stream already written by its producer (emit_mvit)

    @property
    def ptr(self):
        return Ptr("arena", self.off)

    @property
    def voxels(self):
        return self.T * self.H * self.W

    def channel_slice(self, c0, c):
        """View of channels [c0, c0+c) (c0 multiple of 8): same voxels, shifted base."""
        assert c0 % 8 == 0
        r = TRef(self.off + c0 * self.itemsize, self.B, self.T, self.H, self.W, c, self.ld, self.bs,
                 self.itemsize, 0, owned=False, f32=self.f32)
        return r

    def row_offset(self, rows):
        """View starting `rows` voxels later in every batch item (tokens after the cls row)."""
        assert self.T == 1 and self.H == 1
        return TRef(self.off + rows * self.ld * self.itemsize, self.B, 1, 1, self.W - rows, self.C,
                    self.ld, self.bs, self.itemsize, 0, owned=False, f32=self.f32)

    def as_grid(self, T, H, W):
        """Same memory seen as a (T,H,W) voxel grid (T*H*W rows per batch item)."""
        assert T * H * W == self.T * self.H * self.W
        return TRef(self.off, self.B, T, H, W, self.C, self.ld, self.bs, self.itemsize, 0,
                    owned=False, f32=self.f32)


class _Arena:
    """First-fit offset allocator with coalescing free list; tracks the peak."""

    def __init__(self, guard=0):
        self.free = []  # sorted (off, size)
        self.top = 0
        self.peak = 0
        self.live = {}
        self.guard = guard      # bytes of canary behind every allocation (debug plans)
        self.bands = []         # (allocation offset, requested bytes, band offset, band bytes)

    def alloc(self, nbytes):
        if self.guard:
            band = _round_up(max(nbytes, 1), 16)        # right behind the last 16-byte chunk a kernel may store
            total = _round_up(band + self.guard, _ALIGN)
            off = self.top
            self.top += total
            self.peak = max(self.peak, self.top)
            self.live[off] = total
            self.bands.append((off, nbytes, off + band, total - band))
            return off
        nbytes = _round_up(max(nbytes, 1), _ALIGN)
        for i, (off, size) in enumerate(self.free):
            if size >= nbytes:
                if size == nbytes:
                    self.free.pop(i)
                else:
                    self.free[i] = (off + nbytes, size - nbytes)
                self.live[off] = nbytes
                return off
        # grow; merge with a trailing free block if it touches the top
        if self.free and self.free[-1][0] + self.free[-1][1] == self.top:
            off, size = self.free.pop()
            self.top = off
        off = self.top
        self.top += nbytes
        self.peak = max(self.peak, self.top)
        self.live[off] = nbytes
        return off

    def release(self, off):
        if self.guard:
            return              # debug plans never re-use memory: a late write shows up in a canary, not in another tensor
        size = self.live.pop(off)
        self.free.append((off, size))
        self.free.sort()
        merged = []
        for o, s in self.free:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        self.free = merged


class Session:
    _generations = 0        # plans finalized in this process; see `generation`
    def __init__(self, dtype=torch.bfloat16, device=None, reuse_buffers=True):
        assert dtype in (torch.bfloat16, torch.float32), "deploy dtype must be bf16 or fp32"
        self.dtype = dtype
        self.pv_dtype = L.PV_BF16 if dtype == torch.bfloat16 else L.PV_F32
        self.itemsize = 2 if dtype == torch.bfloat16 else 4
        self.device = device
        self.reuse = reuse_buffers
        from . import tuning
        self._arena = _Arena(guard=int(tuning.get("arena_guards")))
        self._weights = []      # (byte_off, cpu tensor)
        self._wtop = 0
        self.ops = []           # (kind, desc_cls, fields dict, label, alg_bytes, flops)
        self.finalized = False
        self.plan = None
        self.arena_t = None
        self.weights_t = None
        self._graph_ready = False
        self._desc_keep = []

    # ------------------------------------------------------------------ memory
    def alloc_input(self, B, T, H, W, C):
        """Model-input activation.  In bf16 sessions an RGB-like input (C <= 4) uses the
        4-channel first-layer layout (8 bytes per voxel) consumed by the stem kernel."""
        if C <= 4 and self.itemsize == 2:
            return self.alloc_act(B, T, H, W, C, ld=4)
        return self.alloc_act(B, T, H, W, C)

    def alloc_act(self, B, T, H, W, C, f32=False, ld=None):
        """New activation [B,T,H,W,pad8(C)]; token tensors are (B,1,1,N,C)."""
        isz = 4 if f32 else self.itemsize
        ld = pad8(C) if ld is None else ld
        bs = T * H * W * ld
        nbytes = B * bs * isz
        off = self._arena.alloc(nbytes)
        return TRef(off, B, T, H, W, C, ld, bs, isz, nbytes, f32=f32)

    def alloc_raw(self, nbytes):
        return Ptr("arena", self._arena.alloc(nbytes))

    def release(self, ref):
        if not self.reuse or ref is None:
            return
        if isinstance(ref, TRef):
            if ref.owned:
                self._arena.release(ref.off)
        else:
            self._arena.release(ref.off)

    def add_weight(self, t):
        """Register a CPU tensor for upload; returns its symbolic pointer."""
        t = t.detach().contiguous().cpu()
        off = _round_up(self._wtop, _ALIGN)
        self._weights.append((off, t))
        self._wtop = off + t.numel() * t.element_size()
        return Ptr("weights", off)

    # ------------------------------------------------------------------ ops
    def add_op(self, kind, fields, label="", alg_bytes=0, flops=0):
        assert not self.finalized
        self._check_live(fields, label)
        self.ops.append((kind, L.DESC_FOR_OP[kind], fields, label, alg_bytes, flops))
        return len(self.ops) - 1

    def _check_live(self, fields, label):
        """Ops run in the order they are emitted, so a buffer an op touches must be allocated at the moment the
        op is emitted: an arena pointer outside every live allocation is a use-after-release (or a write into
        memory some later tensor will own) in the emitter that produced it."""
        if not self.reuse:
            return
        for name, v in fields.items():
            for p in (v if isinstance(v, (list, tuple)) else (v,)):
                if isinstance(p, Ptr) and p.space == "arena" and not any(
                        o <= p.off < o + n for o, n in self._arena.live.items()):
                    raise L.PvError("plan emitter bug: op '%s' field '%s' points at arena offset %d, which is not inside a "
                                    "live allocation (released too early?)" % (label, name, p.off))

    # ------------------------------------------------------------------ finalize / run
    def finalize(self):
        if self.finalized:
            return
        lib = L.lib()  # raises if the .so is missing
        if not torch.cuda.is_available() or lib.pv_device_count() <= 0:
            raise L.PvError("the mi355x deploy form needs a GPU (no CPU fallback); none is visible")
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        # a margin of zeros in front of and behind the arena (tuning "arena_margin"): what lies beside an allocation is somebody
        # else's memory -- with two sub-batch plans, the OTHER plan's arena, written while this one runs
        from . import tuning
        margin = int(tuning.get("arena_margin")) // _ALIGN * _ALIGN
        nbytes = max(self._arena.peak, _ALIGN)
        self._arena_alloc = torch.zeros(nbytes + 2 * margin, dtype=torch.uint8, device=dev)
        self.arena_t = self._arena_alloc[margin:margin + nbytes]
        for _, _, boff, blen in self._arena.bands:      # debug plans: canaries behind every buffer
            self.arena_t[boff:boff + blen] = _CANARY
        blob = torch.zeros(max(self._wtop, _ALIGN), dtype=torch.uint8)
        for off, t in self._weights:
            raw = t.view(torch.uint8).reshape(-1) if t.dtype != torch.bfloat16 else t.view(torch.int16).view(torch.uint8).reshape(-1)
            blob[off:off + raw.numel()] = raw
        self.weights_t = blob.to(dev)
        base = {"arena": self.arena_t.data_ptr(), "weights": self.weights_t.data_ptr()}
        self.plan = C.c_void_p(lib.pv_plan_create())
        Session._generations += 1
        self.generation = Session._generations       # identifies THIS build of the plan (a joint graph is captured from it)
        for kind, cls, fields, label, _, _ in self.ops:
            d = cls()
            def resolve(v):
                return base[v.space] + v.off if isinstance(v, Ptr) else v

            for k, v in fields.items():
                if isinstance(v, (list, tuple)):   # fixed-size array field, short lists are zero-padded
                    arr_t = dict(cls._fields_)[k]
                    vals = [resolve(e) for e in v]
                    vals += [None if arr_t._type_ is C.c_void_p else 0] * (arr_t._length_ - len(vals))
                    setattr(d, k, arr_t(*vals))
                else:
                    setattr(d, k, resolve(v))
            self._desc_keep.append(d)
            L.check(lib.pv_plan_add(self.plan, kind, C.byref(d), C.sizeof(d)), "pv_plan_add(%s)" % label)
        self.finalized = True

    def check_guards(self):
        """Debug plans (tuning "arena_guards" > 0): the arena buffers whose canary a kernel overwrote, as
        (buffer offset, buffer bytes, first bad byte behind the buffer, bad bytes).  Empty list = every write stayed
        inside the buffer it was addressed to.  Synchronises."""
        if not self._arena.guard:
            raise L.PvError("this plan was not converted with tuning.OPTIONS['arena_guards'] > 0")
        torch.cuda.synchronize(self.device)
        bad = []
        for off, nbytes, boff, blen in self._arena.bands:
            hit = (self.arena_t[boff:boff + blen] != _CANARY).nonzero()
            if hit.numel():
                bad.append((off, nbytes, int(hit[0].item()), int(hit.numel())))
        return bad

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def launch(self, first=0, last=None, use_graph=False):
        """Replay ops [first, last) on the session device's current torch stream (the session stays pinned to
        the device it was finalized on, whatever device is current in the caller)."""
        lib = L.lib()
        n = len(self.ops)
        last = n if last is None else last
        with torch.cuda.device(self.device):
            if use_graph and first == 0 and last == n:
                if not self._graph_ready:
                    L.check(lib.pv_plan_graph_build(self.plan, self._stream()), "graph build")
                    self._graph_ready = True
                L.check(lib.pv_plan_graph_launch(self.plan, self._stream()), "graph launch")
            else:
                L.check(lib.pv_plan_launch_range(self.plan, first, last, self._stream()), "plan launch")

    def profile(self, iters=5):
        """Per-op device milliseconds measured with HIP events on the launch stream."""
        lib = L.lib()
        n = len(self.ops)
        out = (C.c_float * n)()
        with torch.cuda.device(self.device):    # the session stays pinned to the device it was finalized on
            L.check(lib.pv_plan_profile(self.plan, self._stream(), iters, out), "profile")
        self.op_kernels = [(lib.pv_plan_op_kernel(self.plan, i) or b"").decode() for i in range(n)]   # kernel symbol per op
        return [(self.ops[i][3], self.ops[i][0], float(out[i]), self.ops[i][4], self.ops[i][5]) for i in range(n)]

    # ------------------------------------------------------------------ tensor views
    def view(self, ref):
        """torch view of an activation in the reference's logical layout: (B,C,T,H,W) with
        channels-last strides -- zero copy, consumable by any torch module."""
        dt = torch.float32 if ref.f32 else self.dtype
        isz = ref.itemsize
        n_el = ref.B * ref.bs
        flat = self.arena_t[ref.off: ref.off + n_el * isz].view(dt)
        return flat.as_strided((ref.B, ref.C, ref.T, ref.H, ref.W),
                               (ref.bs, 1, ref.H * ref.W * ref.ld, ref.W * ref.ld, ref.ld))

    def view_rows(self, ref):
        """(B, rows, C) view of a token / row tensor."""
        dt = torch.float32 if ref.f32 else self.dtype
        n_el = ref.B * ref.bs
        flat = self.arena_t[ref.off: ref.off + n_el * ref.itemsize].view(dt)
        rows = ref.T * ref.H * ref.W
        return flat.as_strided((ref.B, rows, ref.C), (ref.bs, ref.ld, 1))

    def matches(self, x, ref):
        """True when torch tensor `x` already *is* the arena buffer `ref` (zero-copy chaining)."""
        if not x.is_cuda or x.dim() != 5:
            return False
        v = self.view(ref)
        return x.data_ptr() == v.data_ptr() and x.dtype == v.dtype and x.stride() == v.stride() and x.shape == v.shape

    def ingest(self, x, ref, t_index=None, ch_scale=None, ch_shift=None):
        """Copy a user tensor (NCDHW, any strides; fp32, bf16 or uint8) into the channels-last arena
        buffer `ref`.  `t_index` (int32 device tensor, one source frame per destination frame) selects
        frames in the same pass; `ch_scale` / `ch_shift` (fp32 device tensors [C]) apply
        y = x*scale + shift -- see pytorchvideo_amd.transforms.DevicePacker."""
        lib = L.lib()
        if not x.is_cuda:
            x = x.to(self.device, non_blocking=True)
        if x.dtype not in (torch.float32, torch.bfloat16, torch.uint8):
            x = x.float()
        x = x.contiguous()
        B, Cc, Ts, H, W = x.shape
        T = Ts if t_index is None else int(t_index.numel())
        if (B, Cc, T, H, W) != (ref.B, ref.C, ref.T, ref.H, ref.W):
            raise L.PvError("deploy form was converted for input %s, got %s" %
                            ((ref.B, ref.C, ref.T, ref.H, ref.W), (B, Cc, T, H, W)))
        d = L.LayoutDesc()
        d.src, d.dst = x.data_ptr(), self.arena_t.data_ptr() + ref.off
        d.B, d.C, d.T, d.H, d.W = B, Cc, T, H, W
        d.c_p, d.ld, d.bs = (4 if ref.ld == 4 else pad8(ref.C)), ref.ld, ref.bs
        d.src_dtype = {torch.bfloat16: L.PV_BF16, torch.float32: L.PV_F32, torch.uint8: L.PV_U8}[x.dtype]
        d.dst_dtype = self.pv_dtype
        if t_index is not None:
            d.t_index, d.src_T = t_index.data_ptr(), Ts
        if ch_scale is not None:
            d.ch_scale = ch_scale.data_ptr()
            d.ch_shift = ch_shift.data_ptr() if ch_shift is not None else None
        with torch.cuda.device(self.device):
            L.check(lib.pv_ingest_ncdhw(C.byref(d), self._stream()), "ingest")

    def alloc_boxes(self, count):
        """Persistent [count, 5] fp32 buffer for the box list of a detection head.  It lives beside the
        weights, NOT in the arena: it is filled before the replay starts and read near its end, and every
        arena offset that is free when the head is emitted is one the backbone writes during the replay."""
        return self.add_weight(torch.zeros(count, 5, dtype=torch.float32))

    def load_boxes(self, bboxes, ptr, count):
        """Copy the [R,5] box list of a detection forward into its buffer (read by pv_roi_align at replay
        time; the deploy form is specialised to the box COUNT like it is to every other size)."""
        if bboxes.dim() != 2 or tuple(bboxes.shape) != (count, 5):
            raise L.PvError("deploy form was converted for %d boxes [R,5], got %s" % (count, tuple(bboxes.shape)))
        assert ptr.space == "weights"
        dst = self.weights_t[ptr.off: ptr.off + count * 20].view(torch.float32).view(count, 5)
        dst.copy_(bboxes.to(device=self.device, dtype=torch.float32, non_blocking=True))

    def ingest_rows(self, x, ref):
        """Copy a (B, N, C) token tensor into the arena buffer `ref` (no-op when `x` already is it)."""
        v = self.view_rows(ref)
        if tuple(x.shape) != tuple(v.shape):
            raise L.PvError("deploy form was converted for tokens %s, got %s" % (tuple(v.shape), tuple(x.shape)))
        if x.is_cuda and x.data_ptr() == v.data_ptr() and x.stride() == v.stride() and x.dtype == v.dtype:
            return
        v.copy_(x.to(self.device, non_blocking=True))
        if ref.ld > ref.C:  # padding channels of the row must stay zero
            full = self.arena_t[ref.off: ref.off + ref.B * ref.bs * ref.itemsize].view(v.dtype)
            full.view(ref.B, -1, ref.ld)[:, :, ref.C:].zero_()

    def __del__(self):
        try:
            if self.plan is not None:
                L.lib().pv_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    def __deepcopy__(self, memo):
        raise L.PvError("a converted mi355x model owns device memory and cannot be deep-copied; "
                        "convert a fresh copy of the original-form model instead")
