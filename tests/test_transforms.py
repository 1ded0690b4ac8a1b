"""Host mirrors of the reference's pre-path transforms (pytorchvideo/transforms/functional.py:19-41,
134-160; transforms/transforms.py:177-195,414-430) and the fused device path (DevicePacker)."""
import pytest
import torch

from pytorchvideo_amd import transforms as TR


def test_uniform_temporal_subsample_is_linspace_index_select():
    x = torch.arange(2 * 3 * 32 * 2 * 2, dtype=torch.float32).reshape(2, 3, 32, 2, 2)
    for n in (1, 4, 8, 32, 40):   # 40 > T: nearest-neighbour repeats
        got = TR.uniform_temporal_subsample(x, n, temporal_dim=2)
        idx = torch.clamp(torch.linspace(0, 31, n), 0, 31).long()
        assert torch.equal(got, x[:, :, idx])
    # the SlowFast packing of the reference tutorial: slow = T/4 frames, fast = all of them
    slow, fast = TR.uniform_temporal_subsample_repeated(x, (4, 1), temporal_dim=2)
    assert slow.shape[2] == 8 and torch.equal(fast, x)
    assert torch.equal(slow, x[:, :, torch.linspace(0, 31, 8).long()])
    # default temporal_dim=-3 is the T of a (C,T,H,W) clip
    assert torch.equal(TR.uniform_temporal_subsample(x[0], 8), x[0][:, torch.linspace(0, 31, 8).long()])


def test_div255_and_normalize_match_their_definitions():
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (3, 4, 5, 6), generator=g, dtype=torch.uint8)
    mean, std = (0.45, 0.45, 0.45), (0.225, 0.225, 0.225)
    y = TR.Normalize(mean, std)(TR.Div255()(u8.float()))
    want = (u8.float() / 255.0 - torch.tensor(mean).view(3, 1, 1, 1)) / torch.tensor(std).view(3, 1, 1, 1)
    assert torch.allclose(y, want, rtol=0, atol=1e-6)
    yb = TR.Normalize(mean, std)(TR.div_255(u8.float()[None]))   # batched clips too
    assert torch.allclose(yb[0], want, rtol=0, atol=1e-6)


def test_host_transforms_equal_the_reference_functions():
    """uniform_temporal_subsample(_repeated) and div_255 against outputs of the real reference's
    pytorchvideo/transforms/functional.py:19-41,134-160 (tests/golden/transforms.pt, made by make_transforms_golden.py)."""
    import os
    import sys
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold_dir)
    from make_transforms_golden import CASES, REPEATED, clip
    g = torch.load(os.path.join(gold_dir, "transforms.pt"), weights_only=False)
    for i, (shape, n, dim) in enumerate(CASES):
        assert torch.equal(TR.uniform_temporal_subsample(clip(shape, i), n, dim), g["subsample"][i])
    for i, (shape, ratios, dim) in enumerate(REPEATED):
        got = TR.uniform_temporal_subsample_repeated(clip(shape, 50 + i), ratios, dim)
        assert len(got) == len(g["repeated"][i]) and all(torch.equal(a, b) for a, b in zip(got, g["repeated"][i]))
    assert torch.equal(TR.div_255(clip((3, 4, 5, 6), 99)), g["div_255"])


def test_device_packer_rejects_models_that_were_not_converted_whole():
    with pytest.raises(RuntimeError):
        TR.DevicePacker(torch.nn.Identity())


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(9, 11), (8, 12)], ids=["odd_frame", "frame_of_8n_voxels"])   # scalar / 8-voxel-per-thread kernel
@pytest.mark.parametrize("c4", [True, False])
def test_ingest_fuses_frame_selection_scaling_and_normalisation(c4, hw):
    """pv_ingest_ncdhw with t_index / ch_scale / ch_shift on uint8 frames against the host transforms."""
    from pytorchvideo_amd import _lib as L
    from gpu_util import call
    B, T, (H, W), Tout = 2, 16, hw, 4
    g = torch.Generator().manual_seed(5)
    clip = torch.randint(0, 256, (B, 3, T, H, W), generator=g, dtype=torch.uint8).cuda()
    mean, std = (0.45, 0.40, 0.50), (0.225, 0.25, 0.2)
    want = TR.Normalize(mean, std)(TR.div_255(TR.uniform_temporal_subsample(clip.float(), Tout, 2)))
    idx = TR.temporal_indices(T, Tout).to(torch.int32).cuda()
    scale = (1.0 / (255.0 * torch.tensor(std, dtype=torch.float64))).float().cuda()
    shift = (-torch.tensor(mean, dtype=torch.float64) / torch.tensor(std, dtype=torch.float64)).float().cuda()
    cp = 4 if c4 else 8
    dst_dtype = torch.bfloat16 if c4 else torch.float32
    dst = torch.full((B, Tout, H, W, cp), 7.0, dtype=dst_dtype, device="cuda")
    d = L.LayoutDesc()
    d.src, d.dst = clip.data_ptr(), dst.data_ptr()
    d.B, d.C, d.T, d.H, d.W, d.c_p, d.ld, d.bs = B, 3, Tout, H, W, cp, cp, Tout * H * W * cp
    d.src_dtype, d.dst_dtype = L.PV_U8, (L.PV_BF16 if c4 else L.PV_F32)
    d.t_index, d.src_T, d.ch_scale, d.ch_shift = idx.data_ptr(), T, scale.data_ptr(), shift.data_ptr()
    call("pv_ingest_ncdhw", d)
    got = dst[..., :3].permute(0, 4, 1, 2, 3).float()
    tol = 2 ** -8 * want.abs().max().item() if c4 else 1e-5   # one bf16 rounding, or fp32 re-association
    assert (got - want).abs().max().item() <= tol
    assert torch.all(dst[..., 3:] == 0)


@pytest.mark.gpu
def test_device_packer_equals_host_packing_on_slowfast_and_x3d():
    from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
    from pytorchvideo_amd.models.slowfast import create_slowfast
    from pytorchvideo_amd.models.x3d import create_x3d
    from pytorchvideo_amd.utils import randomize_norm_stats
    from gpu_util import rel_err
    mean, std = (0.45, 0.45, 0.45), (0.225, 0.225, 0.225)
    g = torch.Generator().manual_seed(11)
    clip = torch.randint(0, 256, (2, 3, 16, 64, 64), generator=g, dtype=torch.uint8)
    norm = TR.Normalize(mean, std)

    torch.manual_seed(0)
    sf = randomize_norm_stats(create_slowfast(model_depth=18, slowfast_channel_reduction_ratio=(8,), model_num_class=10,
                                              head_pool_kernel_sizes=((4, 2, 2), (16, 2, 2))), 0).eval()
    slow, fast = [norm(TR.div_255(t.float())) for t in TR.uniform_temporal_subsample_repeated(clip, (4, 1), 2)]
    transmute_model(sf, "mi355x")
    dep = convert_to_deployable_form(sf, [slow.cuda().bfloat16(), fast.cuda().bfloat16()], dtype=torch.bfloat16)
    want = dep([slow.cuda().bfloat16(), fast.cuda().bfloat16()]).clone()
    got = TR.DevicePacker(dep, mean, std, div255=True, frame_ratios=(4, 1))(clip.cuda())
    assert rel_err(got, want) <= 1e-2     # inputs agree to one bf16 rounding

    torch.manual_seed(0)
    x3 = randomize_norm_stats(create_x3d(input_clip_length=16, input_crop_size=64, model_num_class=10), 0).eval()
    x = norm(TR.div_255(clip.float())).cuda().bfloat16()
    transmute_model(x3, "mi355x")
    dep3 = convert_to_deployable_form(x3, x, dtype=torch.bfloat16)
    want3 = dep3(x).clone()
    got3 = TR.DevicePacker(dep3, mean, std, div255=True)(clip.cuda())
    assert rel_err(got3, want3) <= 1e-2
    # the split-batch deploy form (streams=2) takes the packer too: one packer per sub-batch, one joint-graph launch
    dep3s = convert_to_deployable_form(x3, x, dtype=torch.bfloat16, streams=2)
    assert torch.equal(TR.DevicePacker(dep3s, mean, std, div255=True)(clip.cuda()), got3)
    deps = convert_to_deployable_form(sf, [slow.cuda().bfloat16(), fast.cuda().bfloat16()], dtype=torch.bfloat16, streams=2)
    assert torch.equal(TR.DevicePacker(deps, mean, std, div255=True, frame_ratios=(4, 1))(clip.cuda()), got)


def _ensemble_fixture():
    """tests/golden/ensemble.pt: outputs of the reference's OWN method bodies (_test_step_with_data_ensembling,
    _ensemble_at_video_level, on_test_epoch_end -- pytorchvideo_trainer/module/video_classification.py:244-311), lifted
    out of the reference source with `ast` and executed unchanged by tests/golden/make_ensemble_golden.py."""
    import os
    return torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ensemble.pt"), weights_only=False)


@pytest.mark.parametrize("method", ["sum", "max"])
def test_ensembling_restatement_is_pinned_to_the_reference_fixture(method):
    """The dict loop of video_classification.py:290-311 restated on the host (the multi-rank comparator of
    tests/test_distributed_gloo.py) reproduces the reference's own outputs bit for bit."""
    fx = _ensemble_fixture()
    ref = fx["methods"][method]
    preds, cnts = {}, {}
    for logits, ids, _labels in fx["batches"]:
        p = torch.softmax(logits, dim=-1)
        for i, v in enumerate(ids):
            if v not in preds:
                preds[v], cnts[v] = torch.zeros(p.shape[1]), 0
            preds[v] = preds[v] + p[i] if method == "sum" else torch.max(preds[v], p[i])
            cnts[v] += 1
    assert list(preds.keys()) == ref["video_order"] and [cnts[v] for v in ref["video_order"]] == ref["counts"]
    assert torch.equal(torch.stack([preds[v] / cnts[v] for v in ref["video_order"]]), ref["video_preds"])
    assert set(fx["source"]) == {"_test_step_with_data_ensembling", "_ensemble_at_video_level", "on_test_epoch_end"}


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["sum", "max"])
def test_video_level_ensembling_matches_the_reference_fixture(method):
    """pv_ensemble_scores / VideoEnsembler against what the reference's own loop produced on the same logits."""
    from pytorchvideo_amd.ensemble import VideoEnsembler
    fx = _ensemble_fixture()
    ref = fx["methods"][method]
    V, Cc = max(ref["video_order"]) + 1, ref["video_preds"].shape[1]
    e = VideoEnsembler(V, Cc, method=method)
    for logits, ids, _labels in fx["batches"]:
        e.update(logits.cuda(), ids)
    assert e.counts.cpu()[ref["video_order"]].tolist() == ref["counts"]
    got = e.merge().result().cpu()[ref["video_order"]]
    # softmax in fp32 on both sides; the 30-clip video sums 30 terms in index order on the device as in the loop
    assert (got - ref["video_preds"]).abs().max().item() <= 1e-6
    with pytest.raises(RuntimeError):
        e.update(torch.zeros(2, Cc + 1, device="cuda"), [0, 1])
