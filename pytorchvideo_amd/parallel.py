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
