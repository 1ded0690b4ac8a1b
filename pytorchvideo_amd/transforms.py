"""The steps that sit immediately before the forward path in the reference's data pipeline
(SURVEY.md section 8f, rank 1), in two forms:

* host mirrors with the reference's names and semantics (`uniform_temporal_subsample`,
  `uniform_temporal_subsample_repeated`, `div_255`, `Normalize`, `Div255`: reference
  pytorchvideo/transforms/functional.py:19-41,134-160 and transforms/transforms.py:177-195,414-430),
  so existing pipelines keep working, and
* `DevicePacker`: the same arithmetic fused into the MI355X deploy form's ingest kernel
  (`pv_ingest_ncdhw`): a decoded uint8 (or float) clip [B,3,T,H,W] that is already on the device --
  or is uploaded once, as the fast-rate clip only -- is frame-subsampled per pathway, scaled,
  normalised, converted to bf16 and laid out channels-last in ONE pass per pathway.  The reference
  does this as index_select + div + sub + div on the host followed by an upload of every pathway.
"""
from typing import Sequence, Tuple

import torch


# --------------------------------------------------------------------------- host mirrors
def uniform_temporal_subsample(x: torch.Tensor, num_samples: int, temporal_dim: int = -3) -> torch.Tensor:
    """transforms/functional.py:19-41: `num_samples` equispaced frames (nearest neighbour when
    num_samples exceeds the clip length)."""
    t = x.shape[temporal_dim]
    assert num_samples > 0 and t > 0
    return torch.index_select(x, temporal_dim, temporal_indices(t, num_samples).to(x.device))


def temporal_indices(t: int, num_samples: int) -> torch.Tensor:
    """The frame indices uniform_temporal_subsample selects (int64)."""
    return torch.clamp(torch.linspace(0, t - 1, num_samples), 0, t - 1).long()


def uniform_temporal_subsample_repeated(frames: torch.Tensor, frame_ratios: Sequence[int],
                                        temporal_dim: int = -3) -> Tuple[torch.Tensor, ...]:
    """transforms/functional.py:134-160: one subsampled copy per pathway (SlowFast: ratios (4, 1))."""
    t = frames.shape[temporal_dim]
    return [uniform_temporal_subsample(frames, t // r, temporal_dim) for r in frame_ratios]


def div_255(x: torch.Tensor) -> torch.Tensor:
    """transforms/functional.py div_255: [0,255] -> [0,1]."""
    return x / 255.0


class Div255(torch.nn.Module):
    def forward(self, x):
        return div_255(x)


class Normalize(torch.nn.Module):
    """transforms/transforms.py:177-195: per-channel (x - mean) / std of a (C,T,H,W) or (B,C,T,H,W) clip."""

    def __init__(self, mean, std):
        super().__init__()
        self.mean, self.std = tuple(float(m) for m in mean), tuple(float(s) for s in std)

    def forward(self, x):
        shape = [1] * x.dim()
        shape[-4] = len(self.mean)
        mean = torch.tensor(self.mean, dtype=x.dtype, device=x.device).view(shape)
        std = torch.tensor(self.std, dtype=x.dtype, device=x.device).view(shape)
        return (x - mean) / std


# --------------------------------------------------------------------------- fused device path
class DevicePacker:
    """`DevicePacker(deployed, mean, std, div255=True, frame_ratios=(4, 1))(clip)` = the deploy form
    applied to `[Normalize(Div255(subsample_r(clip))) for r in frame_ratios]`, with everything before
    the first convolution done by the ingest kernel.

    `deployed` is what `convert_to_deployable_form` returned for a whole model (one graph replay per
    forward); `frame_ratios` must be given for multi-pathway models in the order of the model's
    input list (SlowFast: slow = T/4 frames, fast = T frames), and left None for single-input models.
    A deployed detection model (DetectionBBoxNetwork) is called as `packer(clip, bboxes)`."""

    def __init__(self, deployed, mean=None, std=None, div255=False, frame_ratios=None):
        self.subs = None
        if hasattr(deployed, "parts") and hasattr(deployed, "_pv_launch"):
            # split-batch deploy form (convert_to_deployable_form(..., streams=k)): one packer per sub-batch fills that
            # sub-plan's input buffers, then ONE launch of the joint graph
            self.model = deployed
            self.subs = [DevicePacker(p, mean, std, div255, frame_ratios) for p in deployed.parts]
            self.sess, self.refs = self.subs[0].sess, self.subs[0].refs
            self.frame_ratios = self.subs[0].frame_ratios
            return
        inputs = getattr(deployed, "_pv_inputs", None)
        if inputs is None:
            raise RuntimeError("DevicePacker needs a model converted as a whole by convert_to_deployable_form(model, x)")
        self.model = deployed
        self.sess = deployed._pv_session
        self.refs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        if frame_ratios is None:
            if len(self.refs) != 1:
                raise ValueError("frame_ratios is required for a model with %d input pathways" % len(self.refs))
            frame_ratios = (1,)
        if len(frame_ratios) != len(self.refs):
            raise ValueError("%d frame ratios for %d input pathways" % (len(frame_ratios), len(self.refs)))
        self.frame_ratios = tuple(int(r) for r in frame_ratios)
        dev = self.sess.device
        channels = self.refs[0].C
        self.scale = self.shift = None
        if mean is not None or std is not None or div255:
            mean_t = torch.tensor(mean if mean is not None else [0.0] * channels, dtype=torch.float64)
            std_t = torch.tensor(std if std is not None else [1.0] * channels, dtype=torch.float64)
            if mean_t.numel() != channels or std_t.numel() != channels:
                raise ValueError("mean/std must have %d entries" % channels)
            k = 255.0 if div255 else 1.0
            self.scale = (1.0 / (k * std_t)).float().to(dev)
            self.shift = (-mean_t / std_t).float().to(dev)
        self._index = {}

    def _t_index(self, t_src, ref):
        key = (t_src, ref.T)
        if key not in self._index:
            idx = temporal_indices(t_src, ref.T)
            self._index[key] = None if (t_src == ref.T and torch.equal(idx, torch.arange(t_src))) \
                else idx.to(torch.int32).to(self.sess.device)
        return self._index[key]

    @torch.no_grad()
    def __call__(self, clip, bboxes=None):
        load_boxes = getattr(self.model, "_pv_load_boxes", None)
        if (bboxes is None) != (load_boxes is None):
            raise RuntimeError("bboxes are given to a detection model and only to a detection model")
        if clip.dim() != 5:
            raise RuntimeError("expected a [B,C,T,H,W] clip, got %s" % (tuple(clip.shape),))
        if self.subs is not None:
            # validate BEFORE any ingest: a short batch must not reach the sub-plans' input buffers
            want = sum(self.model._splits)
            if clip.shape[0] != want:
                raise RuntimeError("deploy form was converted for a batch of %d, got %d" % (want, clip.shape[0]))
            if bboxes is not None:
                raise RuntimeError("bboxes cannot be given to a split-batch deploy form (detection models are converted as one plan)")
            clip = clip.to(self.sess.device, non_blocking=True)
            lo = 0
            for sub, b in zip(self.subs, self.model._splits):
                sub._fill(clip[lo:lo + b])
                lo += b
            self.model._pv_launch()
            return self.model._pv_result()
        clip = clip.to(self.sess.device, non_blocking=True)
        self._fill(clip)
        if load_boxes is not None:
            load_boxes(bboxes)
        self.sess.launch(use_graph=self.model._pv_use_graph)
        return self.model._pv_result()

    def _fill(self, clip):
        t_src = clip.shape[2]
        for ratio, ref in zip(self.frame_ratios, self.refs):
            if t_src // ratio != ref.T:
                raise RuntimeError("pathway with frame ratio %d expects %d frames, the clip gives %d" % (ratio, ref.T, t_src // ratio))
            self.sess.ingest(clip, ref, t_index=self._t_index(t_src, ref), ch_scale=self.scale, ch_shift=self.shift)
