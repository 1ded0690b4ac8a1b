"""Batch-sharded multi-GPU forward: one process per GPU, weights replicated, clips split on
dim 0, and ONE collective on the classification head's logits.

The reference's forward path has no collective at all (multi-GPU inference = independent
replicas); the only exchange BASELINE.json prescribes is on the head.  It is
a single `all_gather` of [B_local, classes] fp32 rows (51 KB per rank for 32 clips x 400
classes), i.e. latency-bound: xGMI's 7 x ~153 GB/s links are irrelevant at this size, so it
is issued once per forward on the compute stream through torch.distributed (backend "nccl"
is RCCL on ROCm; "gloo" is used by the CPU tests).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(global_batch, rank, world_size):
    """Contiguous slice [lo, hi) of the global batch owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, rank=None, world_size=None):
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    if isinstance(x, (list, tuple)):
        return [shard_batch(t, rank, world_size) for t in x]
    lo, hi = shard_range(x.shape[0], rank, world_size)
    return x[lo:hi]


def gather_logits(local_logits, global_batch=None, group=None):
    """All ranks obtain the [global_batch, classes] logits.  Equal shards use one
    all_gather_into_tensor; ragged shards pad to the largest shard first."""
    rank, world_size = world()
    if world_size == 1:
        return local_logits
    local_logits = local_logits.contiguous()
    n_local, classes = local_logits.shape
    if global_batch is None:
        global_batch = n_local * world_size
    sizes = [shard_range(global_batch, r, world_size) for r in range(world_size)]
    max_n = max(hi - lo for lo, hi in sizes)
    if all(hi - lo == max_n for lo, hi in sizes):
        out = local_logits.new_empty((world_size * max_n, classes))
        dist.all_gather_into_tensor(out, local_logits, group=group)
        return out
    padded = local_logits.new_zeros((max_n, classes))
    padded[:n_local] = local_logits
    parts = [torch.empty_like(padded) for _ in range(world_size)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


# --------------------------------------------------------------------------- detection (boxes follow their clips)
def shard_boxes(bboxes, global_batch, rank=None, world_size=None, pad_to=None):
    """The rows of a [R,5] box list (clip index, x1, y1, x2, y2) whose clip lives on `rank` under
    `shard_range`, re-indexed to the rank's local clip numbers.  Returns (local boxes, rows): `rows[i]` is the
    position of local box i in the global list.  `pad_to` appends boxes with clip index -1 up to a fixed count
    -- the deploy form is specialised to the box count, and pv_roi_align answers zeros for a clip index outside
    the batch -- so every step of an evaluation can replay the same graph."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    lo, hi = shard_range(global_batch, rank, world_size)
    idx = bboxes[:, 0].long()
    rows = torch.nonzero((idx >= lo) & (idx < hi), as_tuple=False).flatten()
    local = bboxes[rows].clone()
    local[:, 0] -= lo
    if pad_to is not None:
        if rows.numel() > pad_to:
            raise RuntimeError("%d boxes on rank %d, deploy form converted for %d" % (rows.numel(), rank, pad_to))
        pad = local.new_zeros((pad_to - rows.numel(), 5))
        pad[:, 0] = -1
        local = torch.cat([local, pad], 0)
    return local, rows


def reduce_box_scores(local_scores, rows, total_boxes, group=None):
    """All ranks obtain the [total_boxes, classes] scores in the order of the global box list: every rank
    writes its rows into a zero buffer and the buffers are summed (one all_reduce of R x classes fp32 --
    41 KB for 128 boxes x 80 AVA classes; no ragged gather although the box counts differ per rank)."""
    out = local_scores.new_zeros((total_boxes, local_scores.shape[1]))
    out[rows.to(out.device)] = local_scores[: rows.numel()]
    _, world_size = world()
    if world_size > 1:
        dist.all_reduce(out, group=group)
    return out


# --------------------------------------------------------------------------- head collective driven from C
def rccl_candidates():
    """librccl candidates for pv_comm_create, in order: torch's own copy (already mapped into the process when torch
    is imported on ROCm, so ONE RCCL instance serves torch.distributed and the library), then the system one.
    PV_RCCL_LIB overrides (the CPU tests point it at a stub that exchanges host buffers through shared memory)."""
    import os
    if os.environ.get("PV_RCCL_LIB"):
        return os.environ["PV_RCCL_LIB"]
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "librccl.so.1",
             "/opt/rocm/lib/librccl.so"]
    return ":".join(cands)


class HeadComm:
    """An RCCL communicator owned by the C library (include/pv_mi355x.h, pv_comm_*): rank 0 draws the unique id, the
    process group carries its 128 bytes to the other ranks once, and from then on no torch.distributed call is made
    on the data path."""

    def __init__(self, group=None, lib_paths=None):
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        rank, world_size = world()
        paths = (lib_paths or rccl_candidates()).encode()
        # every rank first checks that it can bind an RCCL copy and the group agrees on the outcome: a rank that failed
        # here would otherwise leave the others waiting inside ncclCommInitRank
        ok = torch.tensor([1 if L.lib().pv_comm_probe(paths) == L.PV_OK else 0], dtype=torch.int32)
        why = "" if ok.item() else (L.lib().pv_last_error() or b"").decode()
        if world_size > 1:
            if dist.get_backend(group) == "nccl":
                ok = ok.cuda()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise L.PvError("pv_comm: librccl could not be bound on every rank (%s)" % (why or "another rank failed"))
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            L.check(L.lib().pv_comm_unique_id(buf, paths), "pv_comm_unique_id")
            ident = [bytes(buf.raw)]
        if world_size > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        h = C.c_void_p()
        L.check(L.lib().pv_comm_create(C.byref(h), ident[0], rank, world_size, paths), "pv_comm_create")
        self.handle, self.rank, self.world_size = h, rank, world_size
        self.library = (L.lib().pv_comm_library(h) or b"").decode()

    def all_gather(self, send, recv, stream=None):
        """recv[r] = rank r's `send` (contiguous tensors; device tensors on `stream`, default the current one)."""
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() == send.numel() * self.world_size
        if stream is None:
            stream = torch.cuda.current_stream(send.device).cuda_stream if send.is_cuda else 0
        self._L.check(self._L.lib().pv_comm_all_gather(self.handle, send.data_ptr(), recv.data_ptr(),
                                                       send.numel() * send.element_size(), self._C.c_void_p(stream)),
                      "pv_comm_all_gather")
        return recv

    def gather_ragged(self, local, global_batch):
        """[global_batch, classes] from shards whose sizes differ by at most one (shard_range): every rank pads its rows to the
        largest shard, ONE pv_comm_all_gather, the padding rows are dropped."""
        sizes = [hi - lo for lo, hi in (shard_range(global_batch, r, self.world_size) for r in range(self.world_size))]
        assert local.shape[0] == sizes[self.rank], (local.shape, sizes, self.rank)
        max_n = max(sizes)
        send = local.new_zeros((max_n, local.shape[1]))
        send[: local.shape[0]] = local
        recv = local.new_empty((self.world_size * max_n, local.shape[1]))
        self.all_gather(send.contiguous(), recv)
        return torch.cat([recv[r * max_n: r * max_n + n] for r, n in enumerate(sizes)], dim=0)

    def close(self):
        if getattr(self, "handle", None):
            self._L.lib().pv_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedForward:
    """One step of the batch-sharded forward as ONE C call behind the input ingest: replay of the rank's graph (a single
    plan, or the joint graph of its sub-batch plans), collection of the logits rows of every sub-plan, ncclAllGather
    into the [B_global, classes] buffer -- pv_forward_gather (include/pv_mi355x.h).  `deployed` is what
    `convert_to_deployable_form` returned (whole-model plan, streams >= 1).  The returned tensor is the library's
    receive buffer: valid until the next call."""

    def __init__(self, deployed, comm=None):
        import ctypes as C
        from . import _lib as L
        from .accelerator.mi355x.conversion import _ingest_inputs
        self._L, self._C, self._ingest = L, C, _ingest_inputs
        self.deployed, self.comm = deployed, comm
        self.parts = list(getattr(deployed, "parts", [deployed]))
        if any(getattr(p, "_pv_inputs", None) is None or not hasattr(p, "_pv_result") for p in self.parts):
            raise L.PvError("ShardedForward needs a whole-model deploy form (one launch plan per sub-batch)")
        self.splits = list(getattr(deployed, "_splits", [None]))
        views = [p._pv_result() for p in self.parts]
        if any(v.dim() != 2 or v.stride(1) != 1 for v in views):
            raise L.PvError("ShardedForward gathers [B, classes] logits rows")
        self.n_local, self.classes, self.dtype = sum(v.shape[0] for v in views), views[0].shape[1], views[0].dtype
        dev = views[0].device
        world_size = comm.world_size if comm is not None else 1
        isz = views[0].element_size()
        self._srcs = (L.GatherSrc * len(views))(*[
            L.GatherSrc(v.data_ptr(), v.shape[1] * isz, v.stride(0) * isz if v.shape[0] > 1 else v.shape[1] * isz, v.shape[0])
            for v in views])
        self._staging = torch.empty((self.n_local, self.classes), dtype=self.dtype, device=dev)
        self._recv = torch.empty((world_size * self.n_local, self.classes), dtype=self.dtype, device=dev)
        self._device = dev
        self._joint = len(self.parts) > 1
        self._ready = False

    def _handles(self):
        """(plan, joint) for pv_forward_gather; graphs are built by one ordinary forward-less launch on first use."""
        if not self._ready:
            if self._joint:
                if not self.deployed._use_joint():
                    raise self._L.PvError("ShardedForward needs the joint graph of the split-batch form")
                self.deployed._launch_joint()
            else:
                self.parts[0]._pv_session.launch(use_graph=True)
            self._ready = True
        if self._joint:
            return None, self.deployed._joint_handle
        return self.parts[0]._pv_session.plan, None

    def __call__(self, x):
        multi = isinstance(x, (list, tuple))
        n = (x[0] if multi else x).shape[0]
        if n != self.n_local:                      # before any ingest
            raise self._L.PvError("deploy form was converted for a batch of %d, got %d" % (self.n_local, n))
        if not multi and x.dim() == 4:
            x = x.unsqueeze(2)                     # image model: a clip of one frame (as SplitBatchDeployed.forward)
        lo = 0
        for part, b in zip(self.parts, self.splits):
            if b is None:
                xc = x
            else:
                xc = [t[lo:lo + b] for t in x] if multi else x[lo:lo + b]
                lo += b
            self._ingest(part._pv_session, list(xc) if multi else xc, part._pv_inputs, multi)
        plan, joint = self._handles()
        with torch.cuda.device(self._device):
            stream = self._C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)
            self._L.check(self._L.lib().pv_forward_gather(
                plan, joint, self.comm.handle if self.comm is not None else None, self._srcs, len(self.parts),
                self._staging.data_ptr(), self._recv.data_ptr(), stream), "pv_forward_gather")
        return self._recv
