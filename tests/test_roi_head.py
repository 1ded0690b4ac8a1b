"""The detection head: RoIAlign + ResNetRoIHead + DetectionBBoxNetwork.

CPU part: the oracle's RoIAlign against closed-form answers (the op is third-party -- torchvision is
neither vendored in the reference nor installed here, so these known answers are what pins it), the host
layer against the oracle, and both detection builders against golden vectors produced by the reference's
own backbone / head code (tests/golden/make_golden.py).  GPU part (-m gpu): pv_roi_align and the deploy
form of both detection models against the oracle."""
import ctypes as C
import os

import pytest
import torch

from oracle import functional as OF
from oracle.weights import detection_fill, quantize_like_kernels, seeded_input
from pytorchvideo_amd import _lib as L
from pytorchvideo_amd.layers.roi_align import RoIAlign, roi_align

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BOXES = torch.tensor([[0, 4.0, 6.0, 40.0, 50.0], [1, 0.0, 0.0, 175.0, 143.0], [1, -20.0, -30.0, 30.0, 20.0],
                      [0, 100.0, 100.0, 300.0, 300.0], [1, 50.0, 50.0, 50.5, 50.2], [0, 170.0, 140.0, 180.0, 150.0],
                      [1, 400.0, 400.0, 500.0, 500.0]])


# ----------------------------------------------------------------------------- oracle: known answers
def test_roi_align_hand_computed_case():
    x = torch.tensor([[[[1.0, 2.0], [3.0, 4.0]]]])                   # f(y, x) = 1 + x + 2y
    box = torch.tensor([[0, 0.0, 0.0, 1.0, 1.0]])
    # one bin, one sample at (0.5, 0.5)
    assert OF.roi_align(x, box, 1, 1.0, sampling_ratio=1).item() == pytest.approx(2.5)
    # 2x2 samples at 0.25 / 0.75: 1.75, 2.25, 2.75, 3.25
    assert OF.roi_align(x, box, 1, 1.0, sampling_ratio=2).item() == pytest.approx(2.5)
    # 2x2 bins of one sample each: centres at 0.25 / 0.75
    got = OF.roi_align(x, box, 2, 1.0, sampling_ratio=1)[0, 0]
    assert torch.allclose(got, torch.tensor([[1.75, 2.25], [2.75, 3.25]]))
    # aligned=True shifts the box by half a pixel: samples at (0, 0) of the one-bin case
    assert OF.roi_align(x, box, 1, 1.0, sampling_ratio=1, aligned=True).item() == pytest.approx(1.0)


def test_roi_align_published_known_answer():
    """The known-answer vectors of detectron2's public ROIAlign test (tests/layers/test_roi_align.py, `test_forward_output`):
    a 5x5 ramp arange(25), box (1, 1, 3, 3), 4x4 bins, scale 1, sampling_ratio 0.  detectron2.layers.ROIAlign wraps
    torchvision.ops.roi_align and is the other `roi` the reference's head names (models/head.py:245-247); `old_results`
    there is aligned=False (what the reference builds), `correct_results` aligned=True."""
    x = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5)
    box = torch.tensor([[0, 1.0, 1.0, 3.0, 3.0]])
    old = torch.tensor([[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14], [15, 15.5, 16, 16.5]])
    new = torch.tensor([[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]])
    for fn in (OF.roi_align, roi_align):
        assert torch.allclose(fn(x, box, (4, 4), 1.0, 0, False)[0, 0], old, atol=1e-6)
        assert torch.allclose(fn(x, box, (4, 4), 1.0, 0, True)[0, 0], new, atol=1e-6)


def test_roi_align_is_exact_on_affine_maps_and_zero_outside():
    H, W = 9, 11
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    x = torch.stack([torch.full((H, W), 3.0), 0.5 * xs - 0.25 * ys + 1.0])[None]     # a constant and an affine channel
    box = torch.tensor([[0, 16.0, 24.0, 120.0, 100.0]])                              # well inside the map at scale 1/16
    ph, pw, scale = 3, 5, 1.0 / 16.0
    for sr in (0, 1, 3):
        got = OF.roi_align(x, box, (ph, pw), scale, sampling_ratio=sr)[0]
        assert torch.allclose(got[0], torch.full((ph, pw), 3.0), atol=1e-6)
        # bilinear interpolation reproduces an affine map, and the regular sample grid of a bin averages to its centre
        x1, y1, x2, y2 = [v * scale for v in box[0, 1:].tolist()]
        cx = x1 + (torch.arange(pw) + 0.5) * (x2 - x1) / pw
        cy = y1 + (torch.arange(ph) + 0.5) * (y2 - y1) / ph
        want = 0.5 * cx[None, :] - 0.25 * cy[:, None] + 1.0
        assert torch.allclose(got[1], want, atol=1e-5)
    # a box entirely more than one pixel outside the map pools zeros
    far = torch.tensor([[0, 400.0, 400.0, 500.0, 500.0]])
    assert OF.roi_align(x, far, (ph, pw), scale).abs().max().item() == 0.0
    # a degenerate box is clamped to one feature pixel (aligned=False): a single sample per bin around its corner
    tiny = torch.tensor([[0, 48.0, 32.0, 48.0, 32.0]])
    got = OF.roi_align(x, tiny, (1, 1), scale)[0]
    assert got[1].item() == pytest.approx(0.5 * 3.5 - 0.25 * 2.5 + 1.0, abs=1e-6)


# ----------------------------------------------------------------------------- host layer == oracle
@pytest.mark.parametrize("out,scale,sr,aligned", [((7, 7), 1 / 16.0, 0, False), ((7, 7), 1 / 16.0, 2, False),
                                                  ((3, 5), 0.5, 0, False), ((2, 2), 1 / 16.0, 0, True), (4, 0.25, 3, True)])
def test_host_roi_align_matches_oracle(out, scale, sr, aligned):
    x = seeded_input((2, 12, 9, 11), 3)
    want = OF.roi_align(x, BOXES, out, scale, sr, aligned)
    got = roi_align(x, BOXES, out, scale, sr, aligned)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= 1e-5
    layer = RoIAlign(output_size=out, spatial_scale=scale, sampling_ratio=sr, aligned=aligned)
    assert torch.equal(layer(x, BOXES), got)


def test_roi_head_builder_shapes_follow_the_reference_tests():
    """reference tests/test_models_head.py:182-360 (shape-only there): output (R, classes) with and without the
    global average, Exception when the temporal dimension is not pooled to 1."""
    import torch.nn as nn
    from pytorchvideo_amd.models import create_res_roi_pooling_head
    x = torch.rand(2, 16, 4, 8, 8)
    boxes = torch.tensor([[0, 1.0, 1.0, 100.0, 100.0], [1, 10.0, 20.0, 60.0, 90.0], [1, 0.0, 0.0, 127.0, 127.0]])
    for avg in (True, False):
        head = create_res_roi_pooling_head(in_features=16, out_features=5, resolution=(3, 3), spatial_scale=1 / 16.0,
                                           pool_kernel_size=(4, 1, 1), activation=nn.Softmax,
                                           output_with_global_average=avg).eval()
        y = head(x, boxes)
        assert tuple(y.shape) == ((3, 5) if avg else (3, 5, 1, 1, 1))
    head = create_res_roi_pooling_head(in_features=16, out_features=5, resolution=(3, 3), spatial_scale=1 / 16.0,
                                       pool_kernel_size=(2, 1, 1)).eval()
    with pytest.raises(Exception, match="Temporal dimension should be 1"):
        head(x, boxes)


# ----------------------------------------------------------------------------- goldens (reference backbone + head)
def _golden(name):
    from pytorchvideo_amd.models import create_resnet_with_roi_head, create_slowfast_with_roi_head
    g = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    factory = create_slowfast_with_roi_head if "slowfast" in name else create_resnet_with_roi_head
    m = factory(**g["cfg"])
    detection_fill(m, g["seed"]).eval()
    shapes = g["input_shape"]
    if isinstance(shapes[0], (tuple, list)):
        fast = seeded_input(shapes[1], g["seed"])
        idx = torch.linspace(0, shapes[1][2] - 1, shapes[0][2]).long()
        x = [fast[:, :, idx].clone(), fast]
    else:
        x = seeded_input(shapes, g["seed"])
    return g, m, x


def _oracle_scores(g, sd, x):
    if isinstance(x, list):
        return OF.slowfast_detection_forward(sd, x[0], x[1], g["boxes"], head_pool_kernels=g["cfg"]["head_pool_kernel_sizes"])
    return OF.resnet_detection_forward(sd, x, g["boxes"])


@pytest.mark.parametrize("name", ["resnet_det_r50_small", "slowfast_det_r50_small"])
def test_detection_oracle_and_host_mirror_match_reference_golden(name):
    g, m, x = _golden(name)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == g["state_keys"]   # drop-in state_dict
    want = g["logits"]
    assert 0.0 < want.min().item() and want.max().item() < 1.0 and want.std().item() > 0.2   # scores are not saturated
    got = _oracle_scores(g, m.state_dict(), x)
    assert (got - want).abs().max().item() <= 1e-5
    with torch.no_grad():
        feats = m.model(list(x) if isinstance(x, list) else x)
        host = m(list(x) if isinstance(x, list) else x, g["boxes"])
    fp = g["blocks"][0]
    assert tuple(feats.shape) == tuple(fp["shape"])
    assert (feats.reshape(-1)[fp["sample_idx"]] - fp["sample"]).abs().max().item() <= 1e-5 * max(1.0, fp["absmax"])
    assert (host - want).abs().max().item() <= 1e-5


# ----------------------------------------------------------------------------- plugin boundary (no GPU needed)
def _transmuted(name):
    from pytorchvideo_amd.accelerator import transmute_model
    g, m, x = _golden(name)
    keys = list(m.state_dict().keys())
    transmute_model(m, "mi355x")
    assert list(m.state_dict().keys()) == keys
    return g, m, x


def test_detection_plan_is_one_chain_with_fused_roi_pool_and_sigmoid_epilogue():
    from pytorchvideo_amd.accelerator.mi355x import conversion as CV
    from pytorchvideo_amd.accelerator.mi355x.session import Session
    g, m, x = _transmuted("resnet_det_r50_small")
    assert type(m.detection_head).__name__ == "Mi355xRoIHeadBlock" and CV._is_fusable_net(m.model)
    with torch.no_grad():   # original form of the transmuted model = the same function
        assert (m(x, g["boxes"]) - g["logits"]).abs().max().item() <= 1e-5
    sess = Session(dtype=torch.bfloat16)
    CV._chain_net_blocks(m.model, 2, sess, torch.bfloat16, x)
    feats = m.model._pv_output
    assert (feats.B, feats.T, feats.H, feats.W, feats.C) == (2, 4, 4, 4, 2048)           # 1/16 map: last stage not strided
    assert sum(1 for o in sess.ops if o[2].get("dil_h", 0) == 2) == 3                    # its conv_b is dilated instead
    n = len(sess.ops)
    m.detection_head.convert(None, session=sess, input_ref=feats, num_boxes=6)
    tail = [(o[3].split("|")[0], o[2]) for o in sess.ops[n:]]
    assert [l for l, _ in tail] == ["det.pool", "det.roi_align", "det.proj"]
    roi, proj = tail[1][1], tail[2][1]
    assert (roi["pool_max"], roi["ph"], roi["pw"], roi["R"], roi["sampling_ratio"], roi["aligned"]) == (1, 7, 7, 6, 0, 0)
    assert roi["spatial_scale"] == pytest.approx(1 / 16.0)
    assert proj["act"] == L.ACT_SIGMOID and proj["y_f32"] == 1 and proj["B"] == 6
    out = m.detection_head._out_ref
    assert (out.B, out.voxels, out.C, out.f32) == (6, 1, 16, True)
    # the box list is filled before the replay and read at its end: it must not share the arena with the backbone
    assert roi["boxes"].space == "weights"
    with pytest.raises(AssertionError):
        m.detection_head.convert(None, session=sess, input_ref=feats, num_boxes=6)       # no double convert


def test_roi_head_without_the_whole_window_max_pool_keeps_the_roi_grid():
    import torch.nn as nn
    from pytorchvideo_amd.accelerator.mi355x.blocks import transmute_roi_head
    from pytorchvideo_amd.accelerator.mi355x.session import Session
    from pytorchvideo_amd.models import create_res_roi_pooling_head
    mk = lambda **kw: create_res_roi_pooling_head(in_features=32, out_features=5, resolution=(4, 4), spatial_scale=0.25,
                                                  pool_kernel_size=(2, 1, 1), **kw).eval()
    for kw, want_labels, hw in ((dict(pool_spatial=None), ["det.pool", "det.roi_align", "det.proj", "det.mean"], 16),
                                (dict(pool_spatial=nn.AvgPool2d), ["det.pool", "det.roi_align", "det.pool_spatial", "det.proj", "det.mean"], 1)):
        blk = transmute_roi_head(mk(**kw))
        sess = Session(dtype=torch.float32)
        blk.convert((2, 32, 2, 8, 8), session=sess, num_boxes=3)
        assert [o[3].split("|")[0] for o in sess.ops] == want_labels
        roi = [o[2] for o in sess.ops if o[3] == "det.roi_align"][0]
        assert roi["pool_max"] == 0 and roi["dtype"] == L.PV_F32
        proj = [o[2] for o in sess.ops if o[3].startswith("det.proj")][0]
        assert proj["B"] == 3 and proj["Ho"] * proj["Wo"] == hw
    # an unknown roi layer is declined (reference convention: the transmuter returns None)
    head = mk()
    head.roi_layer = nn.Identity()
    assert transmute_roi_head(head) is None
    # the deploy form is specialised to the box count, and to T == 1 after the head pool
    blk = transmute_roi_head(mk())
    with pytest.raises(RuntimeError):
        blk.convert((2, 32, 2, 8, 8), session=Session(dtype=torch.float32))
    with pytest.raises(Exception, match="Temporal dimension should be 1"):
        transmute_roi_head(mk()).convert((2, 32, 4, 8, 8), session=Session(dtype=torch.float32), num_boxes=3)


def test_roi_align_rejects_invalid_descriptors_without_a_gpu(pv_lib):
    d = L.RoiAlignDesc()
    assert pv_lib.pv_roi_align(C.byref(d), None) == L.PV_ERR_INVALID
    d.x = d.boxes = d.y = 4096
    d.B, d.H, d.W, d.C, d.R, d.ph, d.pw, d.ldx, d.ldy, d.x_bs = 1, 4, 4, 16, 2, 7, 7, 16, 12, 256
    d.spatial_scale, d.dtype = 0.0625, L.PV_BF16
    assert pv_lib.pv_roi_align(C.byref(d), None) == L.PV_ERR_INVALID      # ldy not a multiple of 8
    d.ldy, d.spatial_scale = 16, 0.0
    assert pv_lib.pv_roi_align(C.byref(d), None) == L.PV_ERR_INVALID      # spatial_scale must be positive


# ----------------------------------------------------------------------------- GPU parity
def _run_roi_align(x_nchw, boxes, out, scale, sr, aligned, pool_max, dtype, ld_pad=0):
    """x [B,C,H,W] fp32 CPU -> channels-last device buffer -> pv_roi_align -> [R,C,ph,pw] (or [R,C]) fp32 CPU."""
    from gpu_util import call
    B, Cc, H, W = x_nchw.shape
    ld = (Cc + 7) // 8 * 8 + ld_pad
    xd = torch.zeros((B, H, W, ld), dtype=dtype, device="cuda")
    xd[..., :Cc] = x_nchw.permute(0, 2, 3, 1).to(dtype).cuda()
    ph, pw = (out, out) if isinstance(out, int) else out
    R = boxes.shape[0]
    yd = torch.full((R, 1 if pool_max else ph * pw, ld), 7.0, dtype=dtype, device="cuda")
    bd = boxes.float().cuda().contiguous()
    d = L.RoiAlignDesc()
    d.x, d.boxes, d.y, d.x_bs = xd.data_ptr(), bd.data_ptr(), yd.data_ptr(), H * W * ld
    d.ldx, d.ldy, d.B, d.H, d.W, d.C, d.R, d.ph, d.pw = ld, ld, B, H, W, Cc, R, ph, pw
    d.sampling_ratio, d.aligned, d.pool_max, d.spatial_scale = sr, int(aligned), int(pool_max), scale
    d.dtype = L.PV_BF16 if dtype == torch.bfloat16 else L.PV_F32
    call("pv_roi_align", d)
    pad8 = (Cc + 7) // 8 * 8
    assert torch.all(yd[..., Cc:pad8] == 0)                     # padding channels are written as zeros
    if ld_pad:
        assert torch.all(yd[..., pad8:] == 7.0)                 # and nothing beyond the padded row
    y = yd[..., :Cc].float().cpu()
    return y[:, 0, :] if pool_max else y.reshape(R, ph, pw, Cc).permute(0, 3, 1, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cc,out,scale,sr,aligned", [
    (2304, (7, 7), 1 / 16.0, 0, False),   # SlowFast detection head geometry: 5 channel slabs per box
    (20, (7, 7), 1 / 16.0, 2, False),     # channel count that is not a multiple of 8
    (64, (3, 5), 0.5, 0, False),          # large boxes: many samples per bin, mostly clamped to the border
    (40, (2, 2), 1 / 16.0, 0, True),      # aligned, fewer bins than waves
    (8, 1, 0.25, 3, True),
])
def test_roi_align_kernel_matches_oracle(dtype, Cc, out, scale, sr, aligned):
    x = seeded_input((2, Cc, 9, 11), 7)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    want = OF.roi_align(x, BOXES, out, scale, sr, aligned)
    tol = 1e-5 if dtype == torch.float32 else 6e-3               # bf16: one rounding of the stored result (2^-8)
    got = _run_roi_align(x, BOXES, out, scale, sr, aligned, False, dtype, ld_pad=8)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    # fused with the whole-window max pool of the detection head (MaxPool2d(resolution, stride=1))
    got = _run_roi_align(x, BOXES, out, scale, sr, aligned, True, dtype)
    assert (got - want.amax(dim=(2, 3))).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    # a box that names a clip outside the batch pools zeros instead of reading out of bounds
    bad = BOXES.clone()
    bad[0, 0], bad[1, 0] = 2, -1
    got = _run_roi_align(x, bad, out, scale, sr, aligned, False, dtype)
    assert got[:2].abs().max().item() == 0.0
    assert (got[2:] - want[2:]).abs().max().item() <= tol * max(1.0, want.abs().max().item())


def _deploy_detection(m, x, boxes, dtype):
    from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
    transmute_model(m, "mi355x")
    xd = [t.cuda().to(dtype) for t in x] if isinstance(x, list) else x.cuda().to(dtype)
    return convert_to_deployable_form(m, (xd, boxes), dtype=dtype), xd


# Scores are sigmoid outputs in (0, 1): fp32 kernels 1e-3 absolute; bf16 2.5e-2 -- the backbone's bf16 logit error
# (<= 1e-2 of the logit range, the bound every other model test uses) times the sigmoid's slope, on logits of +-3.
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 2.5e-2)])
@pytest.mark.parametrize("name", ["resnet_det_r50_small", "slowfast_det_r50_small"])
def test_detection_model_matches_oracle(name, dtype, tol):
    g, m, x = _golden(name)
    sd = m.state_dict()
    xo = x
    if dtype == torch.bfloat16:
        sd = quantize_like_kernels(sd)
        xo = [t.bfloat16().float() for t in x] if isinstance(x, list) else x.bfloat16().float()
    want = _oracle_scores(g, sd, xo)
    dm, xd = _deploy_detection(m, x, g["boxes"], dtype)
    assert type(dm.detection_head).__name__ == "Mi355xRoIHeadBlock" and dm.detection_head.convert_flag
    got = dm(list(xd) if isinstance(xd, list) else xd, g["boxes"])
    assert tuple(got.shape) == tuple(want.shape) and got.dtype == torch.float32
    err = (got.cpu() - want).abs().max().item()
    assert err <= tol, "score error %.3e" % err
    if dtype == torch.float32:
        assert (got.cpu() - g["logits"]).abs().max().item() <= tol            # and the reference fixture itself
    # the boxes are data: new values go through the same graph, no re-conversion
    moved = g["boxes"].clone()
    moved[:, 1:] = moved[:, 1:] * 0.5 + 3.0
    want2 = _oracle_scores(dict(g, boxes=moved), sd, xo)
    got2 = dm(list(xd) if isinstance(xd, list) else xd, moved.cuda())
    assert (got2.cpu() - want2).abs().max().item() <= tol
    assert (want2 - want).abs().max().item() > 10 * tol or dtype == torch.bfloat16   # the second call is a different answer
    with pytest.raises(RuntimeError):
        dm(list(xd) if isinstance(xd, list) else xd, moved[:2])                      # specialised to the box count
    # the backbone on its own still answers with the feature map (model.model is part of the public tree)
    feats = dm.model(list(xd) if isinstance(xd, list) else xd)
    assert tuple(feats.shape) == tuple(g["blocks"][0]["shape"])


@pytest.mark.gpu
def test_detection_head_converted_on_its_own_takes_torch_features():
    """Per-block deploy (no whole-model fusion): the RoI head block is fed a torch feature tensor."""
    from pytorchvideo_amd.accelerator.mi355x.blocks import transmute_roi_head
    from pytorchvideo_amd.models import create_res_roi_pooling_head
    import torch.nn as nn
    torch.manual_seed(0)
    head = create_res_roi_pooling_head(in_features=40, out_features=6, resolution=(4, 4), spatial_scale=0.25,
                                       pool_kernel_size=(2, 1, 1), pool_spatial=nn.AvgPool2d, activation=nn.Softmax,
                                       output_with_global_average=True).eval()
    x = seeded_input((2, 40, 2, 8, 8), 11)
    boxes = torch.tensor([[0, 1.0, 2.0, 20.0, 30.0], [1, 0.0, 0.0, 31.0, 31.0], [1, 8.0, 4.0, 12.0, 28.0]])
    with torch.no_grad():
        want = head(x, boxes)                                    # host original form (== oracle by the CPU tests)
    blk = transmute_roi_head(head)
    blk.convert(tuple(x.shape), dtype=torch.float32, num_boxes=3)
    got = blk(x.cuda(), boxes.cuda())
    assert tuple(got.shape) == tuple(want.shape)
    assert (got.cpu() - want).abs().max().item() <= 1e-4


@pytest.mark.gpu
def test_device_packer_feeds_a_detection_model_from_the_decoded_uint8_clip():
    """SURVEY 8f-1 + 8f-4 together: uint8 frames -> subsample per pathway, /255, normalise, bf16, channels-last in
    the ingest kernel; boxes uploaded; one graph replay."""
    from pytorchvideo_amd import transforms as TR
    g, m, x = _golden("slowfast_det_r50_small")
    mean, std = (0.45, 0.45, 0.45), (0.225, 0.225, 0.225)
    clip = torch.randint(0, 256, (1, 3, 16, 96, 96), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    norm = TR.Normalize(mean, std)
    slow, fast = [norm(TR.div_255(t.float())) for t in TR.uniform_temporal_subsample_repeated(clip, (4, 1), 2)]
    dm, xd = _deploy_detection(m, [slow, fast], g["boxes"], torch.bfloat16)
    want = dm(list(xd), g["boxes"]).clone()
    got = TR.DevicePacker(dm, mean, std, div255=True, frame_ratios=(4, 1))(clip.cuda(), g["boxes"])
    assert tuple(got.shape) == tuple(want.shape)
    assert (got - want).abs().max().item() <= 2.5e-2           # the two packings agree to one bf16 rounding of the input
    with pytest.raises(RuntimeError):
        TR.DevicePacker(dm, mean, std, div255=True, frame_ratios=(4, 1))(clip.cuda())
