"""Host-side logic that needs no GPU: factory API surface (shapes, errors, state_dict keys),
the transmuter/convert plumbing, BN folding, weight packing and the arena planner."""
import pytest
import torch
import torch.nn as nn

from pytorchvideo_amd.accelerator import (EFFICIENT_BLOCK_TRANSMUTER_REGISTRY, EfficientBlockBase,
                                          transmute_model)
from pytorchvideo_amd import _lib as L
from pytorchvideo_amd.accelerator.mi355x import emit as E
from pytorchvideo_amd.accelerator.mi355x.blocks import Mi355xBlock
from pytorchvideo_amd.accelerator.mi355x.session import Session, _Arena
from pytorchvideo_amd.models import create_x3d


def test_create_x3d_config1_shape_and_errors():
    # reference tests/test_models_x3d.py:62-80,130-135
    m = create_x3d(model_num_class=400, input_clip_length=4, input_crop_size=160).eval()
    with torch.no_grad():
        assert m(torch.rand(2, 3, 4, 160, 160)).shape == (2, 400)
        with pytest.raises(RuntimeError):
            m(torch.rand(2, 4, 4, 160, 160))  # wrong channel count
    with pytest.raises(AssertionError):
        create_x3d(input_clip_length=4, input_crop_size=16)


def test_transmute_keeps_state_dict_and_original_form():
    torch.manual_seed(0)
    m = create_x3d(input_clip_length=4, input_crop_size=64).eval()
    keys = list(m.state_dict().keys())
    x = torch.rand(1, 3, 4, 64, 64)
    with torch.no_grad():
        want = m(x)
    assert "mi355x" in EFFICIENT_BLOCK_TRANSMUTER_REGISTRY
    transmute_model(m, "mi355x")
    assert all(isinstance(b, Mi355xBlock) and isinstance(b, EfficientBlockBase) for b in m.blocks)
    assert list(m.state_dict().keys()) == keys
    with torch.no_grad():
        assert torch.equal(m(x), want)  # original form = identical maths
    # strict load of a reference-keyed checkpoint still works
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, strict=True)


def test_transmuter_declines_what_it_does_not_know():
    # unknown activation / norm -> left in place (reference convention: return None)
    m = create_x3d(input_clip_length=4, input_crop_size=64, activation=nn.Tanh)
    transmute_model(m, "mi355x")
    assert not isinstance(m.blocks[0], Mi355xBlock)
    assert not isinstance(m.blocks[1], Mi355xBlock)
    with pytest.raises(AssertionError):
        transmute_model(m, "no_such_device")


def test_plan_build_without_gpu_counts_ops():
    m = create_x3d(input_clip_length=4, input_crop_size=160).eval()
    transmute_model(m, "mi355x")
    sess, cur = Session(dtype=torch.bfloat16), None
    for i, b in enumerate(m.blocks):
        b.convert((2, 3, 4, 160, 160) if i == 0 else None, session=sess, input_ref=cur)
        cur = b._out_ref
    labels = [o[3].split("|")[0] for o in sess.ops]
    # conv_a is evaluated inside conv_b's kernel (fused pointwise producer) while the block input is narrow
    fused = labels.count("conv_ab") + labels.count("conv_ab.dw+se")
    # round 6 (csrc/pv_block.hip): the blocks WITHOUT squeeze-excitation of res2 / res3 / res4 are ONE launch each (1 + 2 + 5); of
    # res4's blocks WITH it, the five stride-1 ones run conv_a + conv_b + squeeze sums in one launch
    whole, ab_se = labels.count("block.fused"), labels.count("conv_ab.fused+se")
    assert (whole, ab_se) == (8, 5)
    assert fused == 6 and labels.count("conv_a") == 26 - fused - whole - ab_se and labels.count("conv_c") == 26 - whole
    assert labels.count("se_gate") == 15  # SE in every other block: 2+3+6+4
    assert (cur.B, cur.C, cur.f32) == (2, 400, True)
    with pytest.raises(AssertionError):
        m.blocks[0].convert((2, 3, 4, 160, 160), session=sess)  # no double convert
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            sess.finalize()  # no GPU -> loud failure, never a CPU fallback


@pytest.mark.parametrize("width,label", [(4, "dwconv"), (16, "conv_b")])
def test_grouped_conv_b_of_csn_is_planned_not_declined(width, label):
    """create_csn(stage_conv_b_width_per_group=w) (reference models/csn.py:34,169): 2/4/8 channels per group ride in
    the depthwise kernel (pv_dwconv3d_desc.gw), other widths become block-diagonal dense convs -- no block stays torch."""
    from pytorchvideo_amd.models import create_csn
    m = create_csn(model_depth=50, model_num_class=5, head_pool_kernel_size=(1, 1, 1), stage_conv_b_width_per_group=width).eval()
    transmute_model(m, "mi355x")
    assert all(isinstance(b, Mi355xBlock) for b in m.blocks)
    sess, cur = Session(dtype=torch.bfloat16), None
    for i, b in enumerate(m.blocks):
        b.convert((1, 3, 4, 32, 32) if i == 0 else None, session=sess, input_ref=cur)
        cur = b._out_ref
    convs_b = [o for o in sess.ops if o[3].split("|")[0] == "conv_b"]
    assert len(convs_b) == 16                                      # 3 + 4 + 6 + 3 bottlenecks
    if width == 4:
        assert all(o[0] == L.OP_DWCONV3D and o[2]["gw"] == 4 for o in convs_b)
    else:
        assert all(o[0] == L.OP_CONV3D for o in convs_b)


def test_batchnorm_mvit_blocks_are_transmuted_and_planned():
    """create_multiscale_vision_transformers(norm="batchnorm") (reference models/vision_transformers.py:336-339): BatchNorm1d
    block norms and BatchNorm3d + GELU before the pooling convs become pv_affine_rows launches -- no block stays torch."""
    from pytorchvideo_amd.accelerator.mi355x import emit_mvit as EM
    from pytorchvideo_amd.models import create_multiscale_vision_transformers
    m = create_multiscale_vision_transformers(
        spatial_size=64, temporal_size=4, depth=2, head_num_classes=5, norm="batchnorm", embed_dim_mul=[[1, 2.0]],
        atten_head_mul=[[1, 2.0]], pool_q_stride_size=[[1, 1, 2, 2]], pool_kv_stride_adaptive=[1, 4, 4],
        pool_kvq_kernel=[3, 3, 3]).eval()
    transmute_model(m, "mi355x")
    assert all(type(b).__name__ == "Mi355xMViTBlock" for b in m.blocks)
    sess = Session(dtype=torch.bfloat16)
    x = sess.alloc_act(1, 1, 1, 2 * 16 * 16 + 1, m.blocks[1].dim, f32=True)
    x.thw, x.has_cls = (2, 16, 16), True
    EM.emit_multiscale_block(sess, m.blocks[1], x)
    labels = [o[3].split("|")[0] for o in sess.ops]
    assert labels.count("norm1") == 1 and labels.count("norm2") == 1 and not any("layernorm" in l for l in labels)
    assert all(o[0] == L.OP_AFFINE_ROWS for o in sess.ops if o[3] in ("norm1", "norm2"))
    bn = [o for o in sess.ops if o[3].endswith(".bn_gelu")]
    assert len(bn) == 3 and all(o[2]["act"] == L.ACT_GELU and o[2]["n_prefix"] == 1 for o in bn)    # q, k and v
    assert not any(o[0] == L.OP_LAYERNORM for o in sess.ops)


def test_block_diagonal_expansion_of_a_grouped_conv_is_exact():
    conv = nn.Conv3d(12, 18, (1, 3, 3), padding=(0, 1, 1), groups=3, bias=False)
    x = torch.randn(1, 12, 2, 5, 5)
    cg_in, cg_out = 4, 6
    wd = torch.zeros(18, 12, 1, 3, 3)
    for g in range(3):
        wd[g * cg_out:(g + 1) * cg_out, g * cg_in:(g + 1) * cg_in] = conv.weight.detach()[g * cg_out:(g + 1) * cg_out]
    with torch.no_grad():
        assert torch.allclose(torch.nn.functional.conv3d(x, wd, padding=(0, 1, 1)), conv(x), atol=1e-6)


def test_fold_norm_matches_batchnorm_eval():
    bn = nn.BatchNorm3d(7).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.5, 0.5), bn.running_var.uniform_(0.5, 1.5)
    bias = torch.randn(7)
    scale, shift = E.fold_norm(bn, 7, bias)
    x = torch.randn(2, 7, 3, 4, 5)
    with torch.no_grad():
        want = bn(x + bias.view(1, 7, 1, 1, 1))
    got = x * scale.view(1, 7, 1, 1, 1) + shift.view(1, 7, 1, 1, 1)
    assert torch.allclose(got, want, atol=1e-5)
    with pytest.raises(RuntimeError):
        E.fold_norm(bn, 8)


def test_arena_reuses_and_coalesces():
    a = _Arena()
    o1, o2, o3 = a.alloc(1000), a.alloc(5000), a.alloc(300)
    assert len({o1, o2, o3}) == 3 and all(o % 256 == 0 for o in (o1, o2, o3))
    peak = a.peak
    a.release(o2)
    assert a.alloc(4000) == o2            # first fit into the hole
    a.release(o1)
    a.release(o3)
    assert a.peak == peak
    big = a.alloc(2 * peak)               # must not overlap live blocks
    assert big >= o2 + 4096 or big + 2 * peak <= o2


def test_block_wrapper_resolves_class_level_helpers_of_the_adopted_module():
    """The original-form forward of an adopted module may use static / class methods, properties and constants of
    its class: the wrapper resolves them on the class it adopted."""

    class Helper(nn.Module):
        GAIN = 3.0

        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 4)

        @staticmethod
        def twice(x):
            return 2 * x

        @classmethod
        def gain(cls):
            return cls.GAIN

        @property
        def width(self):
            return self.lin.out_features

        def forward(self, x):
            return self.twice(self.lin(x)) * self.gain() + self.width

    h = Helper().eval()
    xin = torch.randn(2, 4)
    with torch.no_grad():
        assert torch.equal(Mi355xBlock(h)(xin), h(xin))
    from pytorchvideo_amd.models import create_multiscale_vision_transformers as create
    m = create(spatial_size=32, temporal_size=4, depth=2, patch_embed_dim=32, num_heads=1, head_num_classes=5,
               pool_q_stride_size=[[1, 1, 2, 2]], pool_kv_stride_adaptive=[1, 2, 2], pool_kvq_kernel=[3, 3, 3],
               embed_dim_mul=[[1, 2.0]], atten_head_mul=[[1, 2.0]]).eval()
    x = torch.randn(1, 3, 4, 32, 32)
    with torch.no_grad():
        want = m(x)
        transmute_model(m, "mi355x")
        assert type(m.blocks[0]).__name__ == "Mi355xMViTBlock"
        assert torch.equal(m(x), want)      # original form of the transmuted model: same function


def _plan_labels_and_fields(model, sizes, mvit_input=None):
    """Build the launch plan on the host (no GPU needed until finalize) and return [(label, fields)]."""
    from pytorchvideo_amd.accelerator.mi355x import conversion as CV
    transmute_model(model, "mi355x")
    sess = Session(dtype=torch.bfloat16)
    if mvit_input is not None:
        assert CV._is_fusable_mvit(model, mvit_input)
        assert CV._try_fuse_mvit(model, sess, torch.bfloat16, mvit_input)
    else:
        cur = None
        for i, b in enumerate(model.blocks):
            b.convert(sizes if i == 0 else None, session=sess, input_ref=cur)
            cur = b._out_ref
    return [(o[3], o[2]) for o in sess.ops]


def test_slowfast_plan_uses_the_first_layer_layout_and_zero_copy_fusion():
    from pytorchvideo_amd.models import create_slowfast
    m = create_slowfast(model_depth=18, model_num_class=7, head_pool_kernel_sizes=((2, 2, 2), (8, 2, 2))).eval()
    ops = _plan_labels_and_fields(m, [(2, 3, 2, 64, 64), (2, 3, 8, 64, 64)])
    stems = [f for l, f in ops if l.startswith("stem.conv")]
    assert len(stems) == 2 and all(f["cin"] == 4 and f["ldx"] == 4 for f in stems)        # RGB in the 4-channel layout
    fast = [f for f in stems if f["cout"] == 8][0]
    assert fast["c4_wpair"] == 2                                                           # two outputs per MFMA column
    laterals = [f for l, f in ops if l.startswith("lateral_fuse")]
    assert len(laterals) == 4
    # the lateral conv writes into the slow buffer's channel slice: its row stride is wider than its channel count
    assert all(f["ldy"] > f["cout"] for f in laterals)
    assert not any(l.startswith("cat") for l, _ in ops)


def test_mvit_plan_fuses_kv_pooling_and_position_tables():
    from pytorchvideo_amd.models import create_multiscale_vision_transformers as create
    from pytorchvideo_amd.models.hub import mvit_video_base_config
    cfg = dict(mvit_video_base_config, temporal_size=4, spatial_size=64, head_num_classes=5)
    m = create(**cfg).eval()
    x = torch.zeros(8, 3, 4, 64, 64, dtype=torch.bfloat16)
    ops = _plan_labels_and_fields(m, None, mvit_input=x)
    labels = [l.split("|")[0] for l, _ in ops]
    patch = [f for l, f in ops if l.startswith("patch_embed")][0]
    assert patch["pos_spatial"] is not None and patch["pos_temporal"] is not None and patch["y_f32"] == 1
    pos = [f for l, f in ops if l == "pos_encoding"][0]
    assert pos["cls_only"] == 1
    assert labels.count("attn.qkv") == 16 and labels.count("attn.core") == 16            # one fused q|k|v GEMM per block
    assert labels.count("attn.pool_k") + labels.count("attn.pool_v") <= 2 * 16
    # every stream-producing GEMM keeps the residual stream in fp32
    assert all(f["y_f32"] == 1 for l, f in ops if l.split("|")[0] in ("attn.proj", "mlp.fc2"))
    # small token grids: q, k and v of a block are pooled (conv + cls + LayerNorm) by ONE launch
    assert labels.count("attn.pool_qkv") >= 14 and not any(l.endswith(".norm") and "pool" in l for l in labels)


def test_an_op_that_touches_a_released_buffer_is_rejected_at_emission():
    """Ops run in emission order, so every arena pointer of an op must lie in a live allocation when the op is
    emitted; the emitters release a buffer only after its last consumer."""
    from pytorchvideo_amd import _lib as L
    sess = Session(dtype=torch.bfloat16)
    a, b = sess.alloc_act(1, 1, 4, 4, 16), sess.alloc_act(1, 1, 4, 4, 16)
    f = dict(a=a.ptr, b=b.ptr, y=b.ptr, rows=16, C=16, lda=16, ldb=16, ldy=16, act=L.ACT_NONE, dtype=L.PV_BF16)
    sess.add_op(L.OP_ADD_ACT, f, label="ok")
    sess.release(a)
    with pytest.raises(RuntimeError, match="not inside a live allocation"):
        sess.add_op(L.OP_ADD_ACT, f, label="stale")
    f2 = dict(f, a=b.channel_slice(8, 8).ptr)       # a channel slice of a live buffer is fine
    sess.add_op(L.OP_ADD_ACT, f2, label="slice")


def test_slowfast_with_a_declined_head_converts_block_by_block():
    """A multi-pathway model with one block the transmuter declines (unknown head activation) is converted block by
    block: the size hook has to record the LIST input of MultiPathWayWithFuse / PoolConcatPathway (the reference's
    hook records tensors only, model_conversion.py:26-28) and the batch is patched per pathway."""
    from pytorchvideo_amd.accelerator.mi355x import conversion as CV
    from pytorchvideo_amd.accelerator.mi355x.blocks import Mi355xMultiPathBlock
    from pytorchvideo_amd.models import create_slowfast
    m = create_slowfast(model_depth=18, model_num_class=7, head_pool_kernel_sizes=((2, 2, 2), (8, 2, 2)),
                        head_activation=nn.Tanh).eval()
    transmute_model(m, "mi355x")
    assert isinstance(m.blocks[0], Mi355xMultiPathBlock) and not isinstance(m.blocks[-1], Mi355xBlock)   # head declined
    x = [torch.randn(3, 3, 2, 64, 64), torch.randn(3, 3, 8, 64, 64)]
    lut = CV._record_input_sizes(m, CV._one_clip(x))
    assert lut[".blocks.0"] == [(1, 3, 2, 64, 64), (1, 3, 8, 64, 64)]
    sess = Session(dtype=torch.bfloat16)
    CV._convert_children(m, lut, 3, "", sess, torch.bfloat16, {})
    assert all(b.convert_flag for b in m.blocks[:-1])
    assert [r.B for r in m.blocks[0]._in_ref] == [3, 3] and len(sess.ops) > 30


def test_fused_mlp_rows16_weight_image_replayed_with_the_mfma_lane_roles():
    """The LDS image pv_mlp_rows streams (pack_mlp_weights, include/pv_mi355x.h pv_mlp_desc: block j carries W1 / b1 of hidden
    block j and W2 of hidden block j - 1, zeros at both ends, two blocks of padding) replayed on the host with exactly the roles csrc/pv_mlp.hip::mlp_rows16_kernel gives the lanes of v_mfma_f32_16x16x32_bf16
    (A[m = l&15][k = 8 (l>>4) + j], B[k][n = l&15], D[m = 4 (l>>4) + r][n]): phase A per 16-unit half, the activation's
    registers as the phase-B operand through the permuted K order, phase B one block behind, output channel of (ob, m) --
    equal to fc2(act(fc1(x))) on the same bf16-rounded weights."""
    import torch
    from pytorchvideo_amd.accelerator.mi355x.emit_mvit import pack_mlp_weights
    torch.manual_seed(0)
    H, Cin, Cout, R = 96, 64, 96, 16
    w1, b1, w2 = torch.randn(H, Cin) * 0.2, torch.randn(H), torch.randn(Cout, H) * 0.2
    img = pack_mlp_weights(w1, b1, w2)
    KS2, NOB16, NH = Cin // 32, Cout // 16, H // 32
    stage = 2 * KS2 * 1024 + NOB16 * 1024 + 256
    assert img.numel() == (NH + 3) * stage and not img[(NH + 1) * stage:].any()
    x = torch.randn(R, Cin).bfloat16().float()
    act = torch.relu
    Y = torch.zeros(NOB16, 16, R)                     # [ob][m][n]
    Hprev = torch.zeros(2, 16, R)                     # D_0 / D_1 of the previous block after the activation: [uh][m' = 4 g + r][n]
    for j in range(NH + 1):
        blk = img[j * stage:(j + 1) * stage]
        a = blk[:2 * KS2 * 1024].view(torch.int16).view(torch.bfloat16).float().reshape(2 * KS2, 4, 16, 8)     # [f][g][m][j8]
        b = blk[2 * KS2 * 1024:2 * KS2 * 1024 + NOB16 * 1024].view(torch.int16).view(torch.bfloat16).float().reshape(NOB16, 4, 16, 8)
        c = blk[2 * KS2 * 1024 + NOB16 * 1024:].view(torch.float32)
        assert not c[32:].any()
        D = torch.zeros(2, 16, R)
        for uh in range(2):
            D[uh] = c[16 * uh:16 * uh + 16][:, None].expand(16, R).clone()           # b1 as the C operand
            for ks in range(KS2):
                A = a[2 * ks + uh].permute(1, 0, 2).reshape(16, 32)                  # A[m][k = 8 g + j8]
                B = x[:, 32 * ks:32 * ks + 32].t()                                   # B[k][n] = x[n][32 ks + k]
                D[uh] += A @ B
        # phase B of block j - 1: B[k = 8 g + j8][n] = j8 < 4 ? H_0[4 g + j8][n] : H_1[4 g + j8 - 4][n]
        Bk = torch.zeros(32, R)
        for g in range(4):
            for j8 in range(8):
                Bk[8 * g + j8] = Hprev[0, 4 * g + j8] if j8 < 4 else Hprev[1, 4 * g + j8 - 4]
        for ob in range(NOB16):
            A = b[ob].permute(1, 0, 2).reshape(16, 32)
            Y[ob] += A @ Bk
        Hprev = act(D).bfloat16().float()
    out = torch.zeros(R, Cout)
    for ob in range(NOB16):
        for m in range(16):
            out[:, 32 * (ob >> 1) + 8 * (m >> 2) + 4 * (ob & 1) + (m & 3)] = Y[ob, m]
    bf = lambda t: t.bfloat16().float()
    want = torch.nn.functional.linear(bf(act(torch.nn.functional.linear(x, bf(w1), b1))), bf(w2))
    assert (out - want).abs().max().item() <= 1e-4 * want.abs().max().item()


def test_ln_linear_weight_image_follows_the_documented_layout():
    import torch
    from pytorchvideo_amd.accelerator.mi355x.emit_mvit import _chi, pack_ln_linear_weights
    torch.manual_seed(1)
    N, Cin = 96, 64
    w, b = torch.randn(N, Cin), torch.randn(N)
    img = pack_ln_linear_weights(w, b)
    KS, NB = Cin // 16, N // 32
    stage = KS * 1024 + 256
    assert img.numel() == (NB + 2) * stage and not img[NB * stage:].any()
    for nb in range(NB):
        blk = img[nb * stage:(nb + 1) * stage]
        a = blk[:KS * 1024].view(torch.int16).view(torch.bfloat16).reshape(KS, 2, 32, 8)
        c = blk[KS * 1024:].view(torch.float32)
        for ks in range(KS):
            for hi in range(2):
                for rho in (0, 6, 19, 31):
                    for j in range(8):
                        assert a[ks, hi, rho, j] == w[32 * nb + _chi(rho), 32 * (ks >> 1) + 16 * hi + 8 * (ks & 1) + j].to(torch.bfloat16)
        for hi in range(2):
            for r in range(16):
                assert c[hi * 16 + r] == b[32 * nb + 16 * hi + r]


def test_grouped_temporal_conv_declines_the_stem_fusion_and_the_fused_producer():
    """check_conv3d accepts channel-wise grouped convs (2 / 4 / 8 channels per group) as well as depthwise ones; the two
    fusions that need a TRUE depthwise conv must decline them instead of reshaping a [cout, gw, k, 1, 1] weight."""
    import torch.nn as nn
    from pytorchvideo_amd.accelerator.mi355x import emit as E
    sess = Session(dtype=torch.bfloat16)
    x = sess.alloc_input(1, 4, 16, 16, 3)
    first = nn.Conv3d(3, 24, (1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1), bias=False)
    dw = nn.Conv3d(24, 24, (5, 1, 1), padding=(2, 0, 0), groups=24, bias=False)
    grouped = nn.Conv3d(24, 24, (5, 1, 1), padding=(2, 0, 0), groups=6, bias=False)       # 4 channels per group
    assert E.group_width(dw) == 1 and E.group_width(grouped) == 4
    assert E.can_fuse_temporal_dw(sess, first, grouped, None, L.ACT_NONE, x, L.ACT_RELU) is False
    y = sess.alloc_act(1, 4, 8, 8, 24)
    conv_b = nn.Conv3d(48, 48, 3, padding=1, groups=12, bias=False)                      # 4 channels per group
    conv_a = nn.Conv3d(24, 48, 1, bias=False)
    with pytest.raises(E.Unsupported):
        E.emit_dwconv(sess, conv_b, y, producer=(conv_a, None, L.ACT_RELU))


def test_bench_weight_fill_is_the_parity_tests_instance_bit_for_bit():
    """bench.py times `pytorchvideo_amd.utils.synthetic_trained_like_weights`; tests/test_gpu_full_geometry.py asserts the
    north star on `oracle.weights.trained_like_fill`.  The two draw the same key-addressed values and calibrate the same
    way: identical state_dicts (X3D incl. SE blocks, SlowFast incl. the lateral fusions, MViT without flagged norms)."""
    import torch
    from oracle.weights import seeded_input, trained_like_fill
    from pytorchvideo_amd.models import create_multiscale_vision_transformers, create_slowfast, create_x3d
    from pytorchvideo_amd.utils import synthetic_trained_like_weights
    cases = [
        (lambda: create_x3d(input_clip_length=4, input_crop_size=64), seeded_input((2, 3, 4, 64, 64), 3)),
        (lambda: create_slowfast(model_depth=18, slowfast_fusion_conv_stride=(4, 1, 1), head_pool_kernel_sizes=((2, 2, 2), (8, 2, 2))),
         [seeded_input((2, 3, 2, 64, 64), 4), seeded_input((2, 3, 8, 64, 64), 5)]),
        (lambda: create_multiscale_vision_transformers(spatial_size=32, temporal_size=4, depth=2, head_num_classes=101), None),
    ]
    for make, x in cases:
        torch.manual_seed(0)
        a = make()
        torch.manual_seed(0)
        b = make()
        if x is None:
            x = seeded_input((1, 3, 4, 32, 32), 6)
        trained_like_fill(a, x, 0)
        synthetic_trained_like_weights(b, x, 0)
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        assert all(torch.equal(sa[k], sb[k]) for k in sa), [k for k in sa if not torch.equal(sa[k], sb[k])][:3]
        finals = [m for m in a.modules() if getattr(m, "block_final_bn", False)]
        assert all(0.05 <= float(m.weight.min()) and float(m.weight.max()) <= 0.2 for m in finals)


def test_the_oracles_storage_emulation_routes_pools_by_the_emitters_threshold():
    from oracle import functional as OF
    from pytorchvideo_amd.accelerator.mi355x import tuning
    assert OF.POOL_STREAM_MIN_ELEMS_DEFAULT == tuning.OPTIONS["pool_stream_min_elems"]
    with OF.storage_emulation(pool_stream_min_elems=123):
        assert OF._POOL_STREAM_MIN_ELEMS == 123
    assert OF._POOL_STREAM_MIN_ELEMS == OF.POOL_STREAM_MIN_ELEMS_DEFAULT


def test_slowfast_plan_fuses_conv_b_and_conv_c_where_conv_b_is_narrow(monkeypatch):
    """Round 6 (`pv_conv3d` pw2_*, emit.can_fuse_conv_bc): 13 of SlowFast-R50's bottlenecks -- the fast pathway's 12 blocks without
    a projection shortcut and block 1 of the slow pathway's res2 -- run conv_b -> conv_c as one launch; first blocks (shortcut folded
    into conv_c), slow res2's last block (writes into the lateral concat buffer) and the wide slow stages keep two launches.
    Host half of the conversion only (emit_only): runs without a GPU."""
    from pytorchvideo_amd.accelerator import convert_to_deployable_form
    from pytorchvideo_amd.accelerator.mi355x import tuning
    from pytorchvideo_amd.models import create_slowfast

    def n_ops(flag):
        monkeypatch.setitem(tuning.OPTIONS, "fuse_bc", flag)
        m = create_slowfast().eval()
        transmute_model(m, "mi355x")
        x = [torch.zeros(1, dtype=torch.bfloat16).expand(2, 3, 8, 256, 256), torch.zeros(1, dtype=torch.bfloat16).expand(2, 3, 32, 256, 256)]
        return convert_to_deployable_form(m, x, dtype=torch.bfloat16, emit_only=True)["ops"]

    assert tuning.OPTIONS["fuse_bc"] is True
    assert n_ops(False) - n_ops(True) == 13


def test_conv_b_conv_c_fusion_decisions_of_the_emitter():
    """emit.can_fuse_conv_bc on single bottlenecks (host only; the library's pv_conv3d_pw2_supported decides the geometry)."""
    from pytorchvideo_amd.models.resnet import create_bottleneck_block

    def block(inner=64, out=256, k=(1, 3, 3), groups=1, stride=(1, 1, 1)):
        return create_bottleneck_block(dim_in=out, dim_inner=inner, dim_out=out, conv_a_kernel_size=(1, 1, 1), conv_a_stride=(1, 1, 1),
                                       conv_a_padding=(0, 0, 0), conv_b_kernel_size=k, conv_b_stride=stride,
                                       conv_b_padding=tuple(kk // 2 for kk in k), conv_b_num_groups=groups).eval()

    def decide(bb, dtype=torch.bfloat16, inner=64, out=256, res_ld=None, with_out=None, shortcut=None, stride=(1, 1, 1)):
        sess = Session(dtype=dtype)
        a = sess.alloc_act(2, 4, 16, 16, inner)
        To, Ho, Wo = 4 // stride[0], 16 // stride[1], 16 // stride[2]
        r = sess.alloc_act(2, To, Ho, Wo, out, ld=res_ld)
        o = None if with_out is None else sess.alloc_act(2, To, Ho, Wo, with_out[0], ld=with_out[1]).channel_slice(0, out)
        return E.can_fuse_conv_bc(sess, bb, a, r, o, shortcut)

    assert decide(block()) is True                                                  # SlowFast / ResNet res2: 64 -> 64 -> 256 + identity
    assert decide(block(8, 32), inner=8, out=32) is True                            # fast pathway
    assert decide(block(stride=(1, 2, 2)), stride=(1, 2, 2)) is True                # strided conv_b
    assert decide(block(), dtype=torch.float32) is False                            # fp32 parity mode: two launches
    assert decide(block(128, 512), inner=128, out=512) is False                     # 128 inner channels: an MFMA-bound GEMM
    assert decide(block(64, 256, k=(3, 3, 3))) is False                             # K = 1728: not a streaming problem
    assert decide(block(64, 256, k=(1, 1, 1))) is False                             # a pointwise pair
    assert decide(block(64, 256, groups=64)) is False                               # depthwise / grouped conv_b (CSN)
    assert decide(block(), with_out=(320, 320)) is False                            # output = a slice of the lateral concat buffer:
    #                                                                                 residual and output strides differ
    assert decide(block(), shortcut=("conv_s", "norm_s", "x2")) is False            # projection shortcut folded into conv_c
