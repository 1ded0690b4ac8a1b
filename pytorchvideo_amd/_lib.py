"""ctypes binding of include/pv_mi355x.h (the C ABI of the gfx950 kernels).

The structures below mirror the header field-for-field; `pv_plan_add` double-checks the
descriptor size on the C side so a drifted binding fails loudly instead of corrupting
memory.  There is deliberately no CPU fallback here: if the shared library is missing,
`lib()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PV_MI355X_LIB: a debug variant of the SAME library (csrc/build.py VARIANTS: the AddressSanitizer build of the host shim)
# for the sanitizer job -- not a way to select another implementation: the ABI version and every symbol are checked below.
LIB_PATH = os.environ.get("PV_MI355X_LIB") or os.path.join(_HERE, "_lib", "libpv_mi355x.so")

PV_OK, PV_ERR_UNSUPPORTED, PV_ERR_INVALID, PV_ERR_HIP = 0, -1, -2, -3
PV_F32, PV_BF16, PV_U8 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SWISH, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3, 4
POOL_MAX, POOL_AVG = 0, 1
(OP_CONV3D, OP_DWCONV3D, OP_SE_GATE, OP_POOL3D, OP_LAYERNORM, OP_SOFTMAX_ROWS, OP_MEAN_ROWS,
 OP_POSENC, OP_ATTENTION, OP_ADD_ACT, OP_INGEST, OP_EGRESS, OP_TOKEN_POOL, OP_ROI_ALIGN, OP_LATERAL,
 OP_AFFINE_ROWS, OP_MLP_ROWS, OP_LN_LINEAR, OP_BOTTLENECK) = range(1, 20)

_p, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


def _ints(*names):
    return [(n, _i32) for n in names]


Conv3dDesc = _struct("Conv3dDesc", [
    ("x", _p), ("w", _p), ("y", _p), ("scale", _p), ("shift", _p), ("residual", _p), ("a_gate", _p),
    ("x_bs", _i64), ("y_bs", _i64), ("r_bs", _i64)]
    + _ints("ldx", "ldy", "ldr", "B", "Ti", "Hi", "Wi", "cin", "To", "Ho", "Wo", "cout",
            "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw", "act", "a_act", "dtype", "y_f32", "r_f32")
    + [("dwt_w", _p)] + _ints("dwt_k", "c4_wpair") + [("pos_spatial", _p), ("pos_temporal", _p)]
    + _ints("dil_t", "dil_h", "dil_w")
    + [("x2", _p), ("x2_scale", _p), ("x2_bs", _i64)] + _ints("x2_ld", "x2_cin", "x2_Hi", "x2_Wi", "x2_st", "x2_sh", "x2_sw")
    + [("pw2_w", _p), ("pw2_scale", _p), ("pw2_shift", _p)] + _ints("pw2_cout", "pw2_act"))

DwConv3dDesc = _struct("DwConv3dDesc", [
    ("x", _p), ("w", _p), ("y", _p), ("scale", _p), ("shift", _p), ("psum", _p),
    ("x_bs", _i64), ("y_bs", _i64)]
    + _ints("ldx", "ldy", "B", "Ti", "Hi", "Wi", "C", "To", "Ho", "Wo",
            "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw", "w_mod", "act", "dtype", "n_prefix")
    + [("pw_w", _p), ("pw_scale", _p), ("pw_shift", _p)] + _ints("pw_cin", "pw_act", "gw"))

LateralDesc = _struct("LateralDesc", [
    ("x", _p), ("w", _p), ("y", _p), ("scale", _p), ("shift", _p), ("x_bs", _i64), ("y_bs", _i64)]
    + _ints("ldx", "ldy", "B", "Ti", "H", "W", "cin", "To", "cout", "kt", "st", "pt", "act", "dtype"))

EnsembleDesc = _struct("EnsembleDesc", [
    ("logits", _p), ("video_index", _p), ("accum", _p), ("counts", _p)] + _ints("N", "C", "ld", "V", "mode"))

SeGateDesc = _struct("SeGateDesc", [
    ("psum", _p), ("gate", _p), ("w1", _p), ("b1", _p), ("w2", _p), ("b2", _p)]
    + _ints("B", "C", "c_p", "cr", "nblk") + [("inv_count", _f32)])

Pool3dDesc = _struct("Pool3dDesc", [
    ("x", _p), ("y", _p), ("x_bs", _i64), ("y_bs", _i64)]
    + _ints("ldx", "ldy", "B", "Ti", "Hi", "Wi", "C", "To", "Ho", "Wo",
            "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw", "mode", "n_prefix", "dtype"))

LayoutDesc = _struct("LayoutDesc", [
    ("src", _p), ("dst", _p)] + _ints("B", "C", "T", "H", "W", "c_p", "ld")
    + [("bs", _i64)] + _ints("src_dtype", "dst_dtype")
    + [("t_index", _p)] + _ints("src_T") + [("ch_scale", _p), ("ch_shift", _p)])

RowsDesc = _struct("RowsDesc", [
    ("x", _p), ("y", _p), ("gamma", _p), ("beta", _p), ("rows", _i64)]
    + _ints("C", "ldx", "ldy", "rows_per_batch") + [("eps", _f32), ("dtype", _i32), ("x_f32", _i32), ("g_period", _i32),
                                                    ("act", _i32), ("n_prefix", _i32)])

PosencDesc = _struct("PosencDesc", [
    ("x", _p), ("cls_token", _p), ("pos_spatial", _p), ("pos_temporal", _p), ("pos_class", _p)]
    + _ints("B", "T", "HW", "C", "ld", "dtype", "cls_only"))

AttentionDesc = _struct("AttentionDesc", [
    ("q", _p), ("k", _p), ("v", _p), ("o", _p),
    ("q_bs", _i64), ("k_bs", _i64), ("v_bs", _i64), ("o_bs", _i64)]
    + _ints("ldq", "ldk", "ldv", "ldo", "B", "heads", "head_dim", "Nq", "Nk")
    + [("scale", _f32)] + _ints("residual_q", "dtype"))

AddDesc = _struct("AddDesc", [
    ("a", _p), ("b", _p), ("y", _p), ("rows", _i64)]
    + _ints("C", "lda", "ldb", "ldy", "act", "dtype"))

TokenPoolDesc = _struct("TokenPoolDesc", [
    ("x", _p * 3), ("y", _p * 3), ("w", _p * 3), ("gamma", _p * 3), ("beta", _p * 3),
    ("x_bs", _i64 * 3), ("y_bs", _i64 * 3), ("ldx", _i32 * 3), ("ldy", _i32 * 3),
    ("st", _i32 * 3), ("sh", _i32 * 3), ("sw", _i32 * 3), ("To", _i32 * 3), ("Ho", _i32 * 3), ("Wo", _i32 * 3)]
    + _ints("n", "B", "Ti", "Hi", "Wi", "heads", "head_dim", "kt", "kh", "kw", "n_prefix")
    + [("eps", _f32), ("dtype", _i32)])

RoiAlignDesc = _struct("RoiAlignDesc", [
    ("x", _p), ("boxes", _p), ("y", _p), ("x_bs", _i64)]
    + _ints("ldx", "ldy", "B", "H", "W", "C", "R", "ph", "pw", "sampling_ratio", "aligned", "pool_max")
    + [("spatial_scale", _f32), ("dtype", _i32)])

MlpDesc = _struct("MlpDesc", [
    ("x", _p), ("w12", _p), ("y", _p), ("b2", _p), ("residual", _p), ("ln_gamma", _p), ("ln_beta", _p), ("M", _i64)]
    + _ints("C", "H", "Cout", "ldx", "ldr", "ldy", "act", "dtype") + [("ln_eps", _f32)]
    + [("yn", _p), ("nn_gamma", _p), ("nn_beta", _p)] + _ints("ldyn") + [("nn_eps", _f32)])

LnLinearDesc = _struct("LnLinearDesc", [
    ("x", _p), ("wb", _p), ("y", _p), ("ln_gamma", _p), ("ln_beta", _p), ("M", _i64)]
    + _ints("C", "N", "ldx", "ldy", "act", "dtype") + [("ln_eps", _f32)])

BottleneckDesc = _struct("BottleneckDesc", [
    ("x", _p), ("y", _p), ("residual", _p), ("wa", _p), ("wb", _p), ("wc", _p),
    ("sa", _p), ("ha", _p), ("sb", _p), ("hb", _p), ("sc", _p), ("hc", _p), ("x_bs", _i64), ("y_bs", _i64), ("r_bs", _i64)]
    + _ints("ldx", "ldy", "ldr", "B", "T", "H", "W", "cin", "C", "cout", "act_a", "act_b", "act_out", "dtype", "mode")
    + [("psum", _p)])
BLOCK_FULL, BLOCK_AB = 0, 1

GatherSrc = _struct("GatherSrc", [("ptr", _p), ("row_bytes", C.c_size_t), ("row_pitch", C.c_size_t), ("rows", _i64)])

DESC_FOR_OP = {
    OP_CONV3D: Conv3dDesc, OP_DWCONV3D: DwConv3dDesc, OP_SE_GATE: SeGateDesc, OP_POOL3D: Pool3dDesc,
    OP_LAYERNORM: RowsDesc, OP_SOFTMAX_ROWS: RowsDesc, OP_MEAN_ROWS: RowsDesc, OP_POSENC: PosencDesc,
    OP_ATTENTION: AttentionDesc, OP_ADD_ACT: AddDesc, OP_INGEST: LayoutDesc, OP_EGRESS: LayoutDesc,
    OP_TOKEN_POOL: TokenPoolDesc, OP_ROI_ALIGN: RoiAlignDesc, OP_LATERAL: LateralDesc, OP_AFFINE_ROWS: RowsDesc,
    OP_MLP_ROWS: MlpDesc, OP_LN_LINEAR: LnLinearDesc, OP_BOTTLENECK: BottleneckDesc,
}

# every symbol the header declares: (name, restype, argtypes)
_SYMBOLS = [
    ("pv_version", C.c_int, []),
    ("pv_last_error", C.c_char_p, []),
    ("pv_device_count", C.c_int, []),
    ("pv_conv3d", C.c_int, [C.POINTER(Conv3dDesc), _p]),
    ("pv_conv3d_dwt_supported", C.c_int, [C.POINTER(Conv3dDesc)]),
    ("pv_conv3d_x2_supported", C.c_int, [C.POINTER(Conv3dDesc)]),
    ("pv_conv3d_pw2_supported", C.c_int, [C.POINTER(Conv3dDesc)]),
    ("pv_dwconv3d", C.c_int, [C.POINTER(DwConv3dDesc), _p]),
    ("pv_dwconv3d_psum_blocks", C.c_int, [C.POINTER(DwConv3dDesc)]),
    ("pv_dwconv3d_pw_supported", C.c_int, [C.POINTER(DwConv3dDesc)]),
    ("pv_se_gate", C.c_int, [C.POINTER(SeGateDesc), _p]),
    ("pv_ensemble_scores", C.c_int, [C.POINTER(EnsembleDesc), _p]),
    ("pv_pool3d", C.c_int, [C.POINTER(Pool3dDesc), _p]),
    ("pv_ingest_ncdhw", C.c_int, [C.POINTER(LayoutDesc), _p]),
    ("pv_egress_ncdhw", C.c_int, [C.POINTER(LayoutDesc), _p]),
    ("pv_layernorm", C.c_int, [C.POINTER(RowsDesc), _p]),
    ("pv_affine_rows", C.c_int, [C.POINTER(RowsDesc), _p]),
    ("pv_softmax_rows", C.c_int, [C.POINTER(RowsDesc), _p]),
    ("pv_mean_rows", C.c_int, [C.POINTER(RowsDesc), _p]),
    ("pv_add_posenc", C.c_int, [C.POINTER(PosencDesc), _p]),
    ("pv_attention", C.c_int, [C.POINTER(AttentionDesc), _p]),
    ("pv_add_act", C.c_int, [C.POINTER(AddDesc), _p]),
    ("pv_token_pool", C.c_int, [C.POINTER(TokenPoolDesc), _p]),
    ("pv_roi_align", C.c_int, [C.POINTER(RoiAlignDesc), _p]),
    ("pv_lateral_fuse", C.c_int, [C.POINTER(LateralDesc), _p]),
    ("pv_mlp_rows", C.c_int, [C.POINTER(MlpDesc), _p]),
    ("pv_mlp_rows_supported", C.c_int, [C.POINTER(MlpDesc)]),
    ("pv_ln_linear_rows", C.c_int, [C.POINTER(LnLinearDesc), _p]),
    ("pv_ln_linear_rows_supported", C.c_int, [C.POINTER(LnLinearDesc)]),
    ("pv_bottleneck", C.c_int, [C.POINTER(BottleneckDesc), _p]),
    ("pv_bottleneck_supported", C.c_int, [C.POINTER(BottleneckDesc)]),
    ("pv_bottleneck_psum_blocks", C.c_int, [C.POINTER(BottleneckDesc)]),
    ("pv_tune_set", C.c_int, [C.c_char_p, C.c_int]),
    ("pv_tune_clear", C.c_int, []),
    ("pv_plan_create", _p, []),
    ("pv_plan_destroy", None, [_p]),
    ("pv_plan_add", C.c_int, [_p, C.c_int, _p, C.c_size_t]),
    ("pv_plan_size", C.c_int, [_p]),
    ("pv_plan_launch", C.c_int, [_p, _p]),
    ("pv_plan_launch_range", C.c_int, [_p, C.c_int, C.c_int, _p]),
    ("pv_plan_graph_build", C.c_int, [_p, _p]),
    ("pv_plan_graph_launch", C.c_int, [_p, _p]),
    ("pv_joint_create", _p, []),
    ("pv_joint_destroy", None, [_p]),
    ("pv_joint_build", C.c_int, [_p, C.POINTER(_p), C.c_int]),
    ("pv_joint_launch", C.c_int, [_p, _p]),
    ("pv_joint_branches", C.c_int, [_p]),
    ("pv_plan_profile", C.c_int, [_p, _p, C.c_int, C.POINTER(C.c_float)]),
    ("pv_plan_op_kernel", C.c_char_p, [_p, C.c_int]),
    ("pv_comm_probe", C.c_int, [C.c_char_p]),
    ("pv_comm_unique_id", C.c_int, [_p, C.c_char_p]),
    ("pv_comm_create", C.c_int, [C.POINTER(_p), _p, C.c_int, C.c_int, C.c_char_p]),
    ("pv_comm_destroy", None, [_p]),
    ("pv_comm_rank", C.c_int, [_p]),
    ("pv_comm_world", C.c_int, [_p]),
    ("pv_comm_library", C.c_char_p, [_p]),
    ("pv_comm_all_gather", C.c_int, [_p, _p, _p, C.c_size_t, _p]),
    ("pv_forward_gather", C.c_int, [_p, _p, _p, C.POINTER(GatherSrc), C.c_int, _p, _p, _p]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SYMBOLS]
ABI_VERSION = 32

_lib = None


class PvError(RuntimeError):
    """Raised for any non-OK status.  Shape/descriptor errors are RuntimeError like the
    reference's (torch raises RuntimeError for a channel mismatch, tests/test_models_x3d.py:64-67)."""


def lib():
    """Load (once) and return the C-ABI library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PvError(
                "libpv_mi355x.so is not built (%s). Run `python -m pytorchvideo_amd.csrc.build`; "
                "the MI355X deploy form has no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64.so.7; import it FIRST so that our library binds to the
        # HIP runtime torch already initialised (two runtimes in one process cannot share
        # device pointers or streams).
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, res, args in _SYMBOLS:
            fn = getattr(h, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if h.pv_version() != ABI_VERSION:
            raise PvError("ABI mismatch: library %d, binding %d" % (h.pv_version(), ABI_VERSION))
        _lib = h
    return _lib


def tune(**knobs):
    """Development knobs of the library (kernel-routing A/Bs; see include/pv_mi355x.h): tune(gemm8=0)."""
    for k, v in knobs.items():
        check(lib().pv_tune_set(k.encode(), int(v)), "pv_tune_set(%s)" % k)


def check(status, what=""):
    if status >= 0:
        return status
    names = {PV_ERR_UNSUPPORTED: "unsupported", PV_ERR_INVALID: "invalid descriptor", PV_ERR_HIP: "HIP error"}
    msg = names.get(status, "status %d" % status)
    if status == PV_ERR_HIP:
        msg += ": " + (lib().pv_last_error() or b"").decode()
    raise PvError("%s: %s" % (what or "pv call", msg))
