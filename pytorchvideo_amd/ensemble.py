"""Video-level ensembling of clip predictions -- the step right after the forward path in the
reference's test protocol (pytorchvideo_trainer/module/video_classification.py:244-311; the model zoo
evaluates 10 clips x 3 crops per video, docs/source/model_zoo.md:63) -- and the natural home of the one
collective of the batch-sharded forward: instead of gathering every rank's logits, each rank folds its
clips into per-video score rows on the device (`pv_ensemble_scores`) and the rows are reduced across
ranks once, at the end (sum / max over RCCL; a [videos, classes] fp32 buffer).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class VideoEnsembler:
    """`update(logits, video_index)` after every forward; `merge()` once on every rank; `result()` =
    per-video mean (or max / count, as the reference divides both by the clip count) of softmax scores."""

    def __init__(self, num_videos, num_classes, method="sum", device="cuda"):
        if method not in ("sum", "max"):
            raise NotImplementedError("ensemble method %r (the reference knows 'sum' and 'max')" % method)
        self.method = method
        self.accum = torch.zeros(num_videos, num_classes, dtype=torch.float32, device=device)
        self.counts = torch.zeros(num_videos, dtype=torch.int32, device=device)

    @torch.no_grad()
    def update(self, logits, video_index):
        """logits: [N, classes] on the GPU (any float dtype); video_index: N ints (list or tensor)."""
        if not logits.is_cuda or not self.accum.is_cuda:
            raise L.PvError("VideoEnsembler.update runs on the MI355X; there is no CPU fallback")
        logits = logits.float().contiguous()
        idx = torch.as_tensor(video_index, dtype=torch.int32).to(logits.device).contiguous()
        if idx.numel() != logits.shape[0] or logits.shape[1] != self.accum.shape[1]:
            raise RuntimeError("logits %s do not match %d indices / %d classes" %
                               (tuple(logits.shape), idx.numel(), self.accum.shape[1]))
        d = L.EnsembleDesc()
        d.logits, d.video_index, d.accum, d.counts = logits.data_ptr(), idx.data_ptr(), self.accum.data_ptr(), self.counts.data_ptr()
        d.N, d.C, d.ld, d.V = logits.shape[0], logits.shape[1], logits.stride(0), self.accum.shape[0]
        d.mode = 1 if self.method == "max" else 0
        stream = C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
        L.check(L.lib().pv_ensemble_scores(C.byref(d), stream), "ensemble")

    @torch.no_grad()
    def merge(self, group=None):
        """Reduce the per-video rows over all ranks (no-op without an initialised process group)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.accum, op=dist.ReduceOp.MAX if self.method == "max" else dist.ReduceOp.SUM, group=group)
            dist.all_reduce(self.counts, op=dist.ReduceOp.SUM, group=group)
        return self

    @torch.no_grad()
    def result(self):
        """[videos, classes] scores divided by the clip count (videos never seen stay 0)."""
        return self.accum / self.counts.clamp(min=1).unsqueeze(1).float()
