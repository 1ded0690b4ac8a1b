/*
 * pv_mi355x.h -- C ABI of the MI355X (gfx950) forward path for PyTorchVideo-style models.
 *
 * The reference (facebookresearch/pytorchvideo) has no native code and therefore no FFI
 * of its own; every FLOP on its hot path is an ATen op called from Python.  The entry
 * points below are what a reference-side binding (ctypes) binds for
 * the deploy form of an `EfficientBlockBase` (pytorchvideo/accelerator/efficient_blocks/
 * efficient_block_base.py:8-35) registered under the target device "mi355x" in
 * EFFICIENT_BLOCK_TRANSMUTER_REGISTRY (pytorchvideo/accelerator/deployment/common/
 * model_transmuter.py:16).  Each function cites the reference op graph it replaces.
 *
 * Conventions
 *   - plain pointers (device memory) + sizes in POD descriptors; no torch types.
 *   - activations are channels-last: voxel (b,t,h,w) of a tensor lives at
 *         ptr + b*bs + ((t*H + h)*W + w)*ld          (element units)
 *     and carries `round_up(C,8)` channels, the padding channels being exactly 0.
 *     Token tensors (B,N,C) are the same thing with T=H=1, W=N.
 *   - dtype is PV_F32 or PV_BF16 for activations and packed weights; all accumulation,
 *     BN/bias epilogues, softmax, LayerNorm statistics are fp32.
 *   - every call is asynchronous on `stream`, never allocates, never synchronises and is
 *     safe under hipGraph capture.  Return value: PV_OK or a negative pv_status.
 */
#ifndef PV_MI355X_H_
#define PV_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pv_stream_t; /* hipStream_t */

enum pv_status {
  PV_OK = 0,
  PV_ERR_UNSUPPORTED = -1, /* shape/option not implemented by the kernels            */
  PV_ERR_INVALID = -2,     /* inconsistent descriptor (reference raises RuntimeError) */
  PV_ERR_HIP = -3          /* a HIP runtime call failed; see pv_last_error()          */
};

enum pv_dtype { PV_F32 = 0, PV_BF16 = 1, PV_U8 = 2 /* source of pv_ingest_ncdhw only: decoded video frames */ };

enum pv_act {
  PV_ACT_NONE = 0,
  PV_ACT_RELU = 1,
  PV_ACT_SWISH = 2,  /* x*sigmoid(x): pytorchvideo/layers/swish.py:7-34 */
  PV_ACT_GELU = 3,   /* exact erf form: nn.GELU default, layers/attention.py:74 */
  PV_ACT_SIGMOID = 4
};

enum pv_pool_mode { PV_POOL_MAX = 0, PV_POOL_AVG = 1 };

/* ---- library ---------------------------------------------------------------------- */
int pv_version(void);             /* ABI version, bumped on any descriptor change */
const char* pv_last_error(void);  /* text of the last PV_ERR_HIP on this thread   */
int pv_device_count(void);        /* hipGetDeviceCount; 0 when no GPU is visible  */

/* ---- dense convolution / linear as implicit GEMM on MFMA ---------------------------
 * Replaces nn.Conv3d(groups=1, dilation=1) [+ BatchNorm3d eval | + bias] [+ residual]
 * [+ activation] and nn.Linear [+ bias][+ GELU][+ residual]:
 *   models/x3d.py:169-171,210-212,297-305,480-494 (1x1x1 convs), models/resnet.py:98-132
 *   (conv_a T x1x1, conv_b 1x3x3, conv_c), models/stem.py:80-107 (stems),
 *   models/slowfast.py:672-694 (lateral 7x1x1 stride 4), models/stem.py:295-338 (patch
 *   embed), layers/attention.py:102-114,425-451,541 (Mlp / qkv / proj Linear),
 *   models/head.py:376-382 (head Linear).
 * y = act( (sum_k x'[..]*w[..]) * scale[c] + shift[c] + residual ),
 * x' = a_act(x * a_gate[b][k]) when a_gate/a_act are given (squeeze-excitation scale and
 * Swish of models/x3d.py:190-207 folded into the consumer conv_c).
 * Weights are packed [cout][kt*kh*kw][cin] (cin = padded channel count, tap-major K).
 * First-layer special case (RGB input): cin == 4 && ldx == 4 && dtype == PV_BF16 -- the input carries
 * 4 channels per voxel (8 bytes) and the weights are packed [cout][kt][kh][round_up(kw,2)][4] with
 * zeros in the padding (models/stem.py:80-107,295-338, models/x3d.py:66-88).
 */
typedef struct pv_conv3d_desc {
  const void* x;         /* input activations                                   */
  const void* w;         /* packed weights, dtype `dtype`                        */
  void* y;               /* output                                              */
  const float* scale;    /* [cout] or NULL (=1)                                 */
  const float* shift;    /* [cout] or NULL (=0)                                 */
  const void* residual;  /* same geometry as y, dtype `dtype`, or NULL           */
  const float* a_gate;   /* [B][cin] fp32 multiplier on x, or NULL               */
  int64_t x_bs, y_bs, r_bs; /* batch strides, elements                          */
  int32_t ldx, ldy, ldr;    /* voxel strides, elements                          */
  int32_t B, Ti, Hi, Wi, cin; /* cin: channels read per voxel (multiple of 8)   */
  int32_t To, Ho, Wo, cout;   /* cout: true output channels                     */
  int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
  int32_t act;           /* pv_act applied last                                  */
  int32_t a_act;         /* pv_act applied to x (after a_gate) on load           */
  int32_t dtype;         /* pv_dtype of x, w, residual                           */
  int32_t y_f32;         /* 1: y is fp32 regardless of dtype (logits, residual stream) */
  int32_t r_f32;         /* 1: residual is fp32 regardless of dtype              */
  /* Optional fused depthwise temporal conv -- the X3D stem (models/x3d.py:66-88: Conv2plus1d with
   * norm=None, activation=None between conv_xy and the depthwise 5x1x1 conv_t): when dwt_w is set
   *   y = act(scale * dwt(conv(x)) + shift),  dwt = depthwise conv over T, dwt_k taps, stride 1,
   *   padding dwt_k/2, taps [dwt_k][round_up(cout,8)] fp32.
   * Only for the first-layer layout, where pv_conv3d_dwt_supported(d) is 1. */
  const float* dwt_w;
  int32_t dwt_k;
  /* First-layer layout with at most 8 output channels (SlowFast's fast stem, models/slowfast.py:55-60):
   * c4_wpair = 2 makes two W-adjacent outputs one MFMA column, so the 16 filter rows of the matrix
   * instruction are all used and the overlapping halves of the two windows are loaded once.  Weights
   * are then packed [2][round_up(cout,8)][kt][kh][round_up(kw+sw,2)][4]: row (j, co) holds co's filter
   * shifted right by j*sw voxels, zeros elsewhere.  0 / 1 = one output per column (layout above). */
  int32_t c4_wpair;
  /* First-layer layout with an fp32 output (MViT's patch embedding, models/stem.py:289-338): the position
   * tables of SpatioTemporalClsPositionalEncoding.forward (layers/positional_encoding.py:112-136) are added
   * in the epilogue: y[to][ho][wo][c] += pos_spatial[ho*Wo+wo][c] + pos_temporal[to][c]  (separable), or
   * += pos_spatial[(to*Ho+ho)*Wo+wo][c] when pos_temporal is NULL (full table, cls row skipped by the caller).
   * Tables are fp32 [rows][cout].  The cls row itself is written by pv_add_posenc(cls_only). */
  const float* pos_spatial;
  const float* pos_temporal;
  /* Dilation of the taps (create_resnet's stage_conv_b_dilation, models/resnet.py:601-1003: the
   * detection backbone of models/hub/resnet.py:72-88 dilates res5); 0 means 1.  Dense convs only,
   * not the first-layer layout; kernel extent (k-1)*dilation+1 enters the output-size check. */
  int32_t dil_t, dil_h, dil_w;
  /* Optional second operand concatenated along K -- the projection shortcut of a residual block
   * (models/resnet.py:428-438 built, :1179-1189 added) folded into conv_c:
   *   y = act(scale * (W1 . x) + x2_scale * (W2 . x2s) + shift),  x2s[b][to][ho][wo] = x2[b][to*x2_st][ho*x2_sh][wo*x2_sw]
   * (the shortcut's strided 1x1x1 conv).  Weights [cout][round_up(cin,32) + round_up(x2_cin,32)], both parts
   * zero padded.  The two products are accumulated separately and joined in the epilogue, so the folded BatchNorm
   * scales stay fp32 exactly as in the unfused pair (shift = the two shifts added).  a_gate / a_act apply to the
   * first operand only.  Pointwise convs where pv_conv3d_x2_supported(d) is 1. */
  const void* x2;
  const float* x2_scale;     /* [cout] or NULL (=1) */
  int64_t x2_bs;
  int32_t x2_ld, x2_cin, x2_Hi, x2_Wi, x2_st, x2_sh, x2_sw;
  /* Optional pointwise conv BEHIND this conv in the same launch (round 6) -- conv_b -> conv_c of a ResNet / SlowFast
   * bottleneck (BottleneckBlock.forward, models/resnet.py:1345-1365; block built by create_bottleneck_block,
   * models/resnet.py:98-132): when pw2_w is set
   *   m = bf16(act(scale * conv(x) + shift))                       (conv_b + norm_b + act_b: `cout` channels, never stored)
   *   y = pw2_act(pw2_scale * (W2 . m) + pw2_shift + residual)     (conv_c + norm_c + the block's residual join + final act)
   * y, residual, ldy, ldr, y_bs, r_bs describe the pw2_cout-channel output; `act` is the inner activation.  W2 is
   * [pw2_cout][round_up(cout,8)] in `dtype`.  The rounding of m is the one the unfused pair has at the tensor it stores.
   * bf16 only, narrow first convs (the tap-streaming kernel's range) where pv_conv3d_pw2_supported(d) is 1. */
  const void* pw2_w;
  const float* pw2_scale;    /* [pw2_cout] or NULL (=1) */
  const float* pw2_shift;    /* [pw2_cout] or NULL (=0) */
  int32_t pw2_cout, pw2_act;
} pv_conv3d_desc;
int pv_conv3d(const pv_conv3d_desc* d, pv_stream_t stream);
/* 1 if this geometry (pointers are ignored) can run with the fused temporal conv, else 0 */
int pv_conv3d_dwt_supported(const pv_conv3d_desc* d);
/* 1 if this geometry (pointers are ignored; x2_cin > 0) can run with the second K operand, else 0 */
int pv_conv3d_x2_supported(const pv_conv3d_desc* d);
/* 1 if this geometry (pointers are ignored; pw2_cout > 0) can run with the pointwise conv behind it, else 0 */
int pv_conv3d_pw2_supported(const pv_conv3d_desc* d);

/* ---- development knobs ---------------------------------------------------------------------
 * Kernel-routing choices that were settled by A/B measurements on the MI355X stay overridable for the tools
 * that re-measure them (tools/bench_gemm.py, tools/gpu_*.sh): pv_tune_set("gemm8", 0) etc.  Process-wide, read
 * when an op is launched eagerly or recorded into a graph; nothing is ever read from the environment. */
int pv_tune_set(const char* key, int value);
int pv_tune_clear(void);

/* ---- SlowFast lateral connection (fast -> slow fusion) -----------------------------------
 * Replaces FuseFastToSlow.forward (models/slowfast.py:720-729) as built by
 * FastToSlowFusionBuilder.create_module (models/slowfast.py:661-694), called from
 * MultiPathWayWithFuse.forward (models/net.py:107-122):
 *   fuse = act(BatchNorm3d(Conv3d(cin -> cout, kernel (kt,1,1), stride (st,1,1), padding (pt,0,0), bias=False)(x_fast)))
 *   x_slow_fuse = torch.cat([x_slow, fuse], dim=1)
 * x is the fast pathway (B, Ti, H, W, cin) channels-last; y points at channel C_slow of the slow pathway's
 * (wider) buffer -- voxel stride ldy = that buffer's row width -- so the concatenation is the store itself.
 * Weights packed [cout][kt][cin] (cin padded to 8) in `dtype`; scale / shift = the folded BatchNorm.
 * bf16: the dedicated streaming kernel (csrc/pv_lateral.hip); fp32 parity mode: the same arithmetic through
 * pv_conv3d's (kt,1,1) path. */
typedef struct pv_lateral_desc {
  const void* x; const void* w; void* y;
  const float* scale; const float* shift;   /* [cout] or NULL */
  int64_t x_bs, y_bs;                       /* batch strides, elements */
  int32_t ldx, ldy;                         /* voxel strides, elements */
  int32_t B, Ti, H, W, cin;                 /* cin: channels read per fast voxel (multiple of 8) */
  int32_t To, cout;                         /* slow frames, true output channels */
  int32_t kt, st, pt;                       /* temporal kernel / stride (alpha) / padding */
  int32_t act, dtype;
} pv_lateral_desc;
int pv_lateral_fuse(const pv_lateral_desc* d, pv_stream_t stream);

/* ---- depthwise convolution ---------------------------------------------------------
 * Replaces nn.Conv3d(groups=C) [+ BatchNorm3d eval][+ activation]:
 *   models/x3d.py:66-88 (stem 5x1x1), models/x3d.py:180-189 (3x3x3), models/csn.py:169,
 *   layers/attention.py:363-404 (MViT pool_q/k/v; weights shared over heads -> w_mod).
 * Weights packed [taps][round_up(w_mod or C, 8)] fp32.  If psum != NULL the kernel also
 * writes per-block partial sums of the (pre-activation) output over T,H,W:
 * psum[b][blk][c], blk < pv_dwconv3d_psum_blocks(d) -- the squeeze half of
 * fvcore SqueezeExcitation (mean over T,H,W) without atomics (deterministic).
 */
typedef struct pv_dwconv3d_desc {
  const void* x; const float* w; void* y;
  const float* scale; const float* shift; /* [C] or NULL */
  float* psum;                            /* or NULL      */
  int64_t x_bs, y_bs;
  int32_t ldx, ldy;
  int32_t B, Ti, Hi, Wi, C;  /* C: true channels; round_up(C,8) are read/written */
  int32_t To, Ho, Wo;
  int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
  int32_t w_mod;             /* 0, or weights indexed by (c % w_mod)             */
  int32_t act;
  int32_t dtype;
  int32_t n_prefix;          /* rows before the grid in each batch item (the cls    */
                             /* token of layers/attention.py:185-186) copied verbatim */
  /* Optional fused pointwise producer -- conv_a + norm_a + act_a of an X3D / ir-CSN bottleneck
   * (models/x3d.py:169-189, models/resnet.py:1345-1352): when pw_w is set, x is the 1x1x1 conv's
   * INPUT (pw_cin channels, row stride ldx) and the depthwise conv consumes
   * pw_act(pw_scale * (pw_w . x) + pw_shift) rounded to bf16, evaluated on the fly per halo tile --
   * the expanded tensor never exists in memory.  Only where pv_dwconv3d_pw_supported(d) is 1. */
  const void* pw_w;          /* [round_up(C,32)][round_up(pw_cin,32)] bf16, zero padded, or NULL */
  const float* pw_scale;     /* [C] or NULL */
  const float* pw_shift;     /* [C] or NULL */
  int32_t pw_cin, pw_act;
  /* Channel-wise GROUPED convolution (create_csn(stage_conv_b_width_per_group = gw), models/csn.py:34,169 ->
   * nn.Conv3d(C, C, groups = C / gw)): gw = 2, 4 or 8 input channels per output channel, all inside the output
   * channel's own 8-channel chunk.  0 / 1 = depthwise.  Weights [taps][gw][round_up(C,8)] fp32: w[t][j][c]
   * multiplies input channel (c / gw) * gw + j.  No w_mod, no fused producer, C % gw == 0. */
  int32_t gw;
} pv_dwconv3d_desc;
int pv_dwconv3d(const pv_dwconv3d_desc* d, pv_stream_t stream);
int pv_dwconv3d_psum_blocks(const pv_dwconv3d_desc* d);
/* 1 if this geometry (pointers are ignored) can run with the fused pointwise producer, else 0 */
int pv_dwconv3d_pw_supported(const pv_dwconv3d_desc* d);

/* ---- MViT attention pooling (fused) ----------------------------------------------------
 * _AttentionPool.forward with pool_mode="conv" (layers/attention.py:162-212, pools built at
 * :363-404): for up to three token tensors at once (q, k, v of one MultiScaleAttention) the
 * depthwise Conv3d over the (T,H,W) token grid (weights [taps][head_dim] fp32, shared by all
 * heads, padding = kernel/2, no bias), the cls-token pass-through (n_prefix rows) and the
 * LayerNorm(head_dim) applied to every (token, head) afterwards -- one launch instead of six.
 * Tensor i reads x[i] (B, n_prefix + Ti*Hi*Wi, heads*head_dim) with row stride ldx[i] and writes
 * y[i] (B, n_prefix + To[i]*Ho[i]*Wo[i], heads*head_dim) with row stride ldy[i].
 */
typedef struct pv_token_pool_desc {
  const void* x[3];
  void* y[3];
  const float* w[3];      /* [kt*kh*kw][head_dim]                    */
  const float* gamma[3];  /* LayerNorm weight [head_dim] or NULL     */
  const float* beta[3];   /* LayerNorm bias   [head_dim] or NULL     */
  int64_t x_bs[3], y_bs[3];
  int32_t ldx[3], ldy[3];
  int32_t st[3], sh[3], sw[3];
  int32_t To[3], Ho[3], Wo[3];
  int32_t n;              /* tensors in this launch (1..3)           */
  int32_t B, Ti, Hi, Wi, heads, head_dim;
  int32_t kt, kh, kw;
  int32_t n_prefix;
  float eps;
  int32_t dtype;
} pv_token_pool_desc;
int pv_token_pool(const pv_token_pool_desc* d, pv_stream_t stream);

/* ---- squeeze-excitation gate -------------------------------------------------------
 * fvcore.nn.squeeze_excitation.SqueezeExcitation as used at models/x3d.py:190-198:
 * gate[b][c] = sigmoid(W2 . relu(W1 . mean_{T,H,W}(x) + b1) + b2); the multiply is done
 * by the consumer (pv_conv3d a_gate).  w1 [cr][C], w2 [C][cr] fp32 row-major.
 */
typedef struct pv_se_gate_desc {
  const float* psum;   /* [B][nblk][c_p]           */
  float* gate;         /* [B][c_p]; padding = 0    */
  const float* w1; const float* b1; const float* w2; const float* b2;
  int32_t B, C, c_p, cr, nblk;
  float inv_count;     /* 1/(T*H*W)                */
} pv_se_gate_desc;
int pv_se_gate(const pv_se_gate_desc* d, pv_stream_t stream);

/* ---- pooling -----------------------------------------------------------------------
 * nn.MaxPool3d / nn.AvgPool3d (floor mode, zero/-inf padding, count_include_pad):
 *   models/stem.py:98-104 (stem max pool), models/x3d.py:791-806 and
 *   models/slowfast.py:608-620 (head average pools), layers/attention.py:677-679,720-727
 *   (MViT skip-path max pool on tokens; n_prefix = 1 copies the cls token through).
 */
typedef struct pv_pool3d_desc {
  const void* x; void* y;
  int64_t x_bs, y_bs;
  int32_t ldx, ldy;
  int32_t B, Ti, Hi, Wi, C, To, Ho, Wo;
  int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
  int32_t mode;      /* pv_pool_mode */
  int32_t n_prefix;  /* rows before the grid in each batch item copied verbatim */
  int32_t dtype;
} pv_pool3d_desc;
int pv_pool3d(const pv_pool3d_desc* d, pv_stream_t stream);

/* ---- layout ingest / egress --------------------------------------------------------
 * The reference API is NCDHW (models/net.py:41-44).  Ingest converts a contiguous
 * [B,C,src_T,H,W] tensor (fp32, bf16 or uint8) to NDHWC `dtype` with C padded to c_p (zeros).
 * The steps that sit immediately before the path in the reference's data pipeline ride along
 * (SURVEY 8f-1): frame selection -- destination frame t reads source frame t_index[t], the
 * index_select of uniform_temporal_subsample (transforms/functional.py:19-41), so SlowFast's slow
 * pathway (uniform_temporal_subsample_repeated, :134-160) is ingested straight from the fast clip --
 * and a per-channel affine y = x*ch_scale[c] + ch_shift[c] in fp32, which is Div255 + Normalize
 * (transforms/transforms.py:177-195,414-430) with scale = 1/(255 std), shift = -mean/std.
 * Egress is the inverse layout change only (channels [0,C)), used by block-level forwards and tests.
 */
typedef struct pv_layout_desc {
  const void* src; void* dst;
  int32_t B, C, T, H, W;    /* logical NCDHW extent of the NCDHW side           */
  int32_t c_p, ld;          /* NDHWC side: padded channels written (multiple of 8, or 4 with ld == 4:
                               the 4-channel first-layer layout), voxel stride */
  int64_t bs;               /* NDHWC side batch stride                           */
  int32_t src_dtype, dst_dtype;
  /* ingest only (ignored by egress): */
  const int32_t* t_index;   /* [T] source frame of every destination frame, or NULL (identity)   */
  int32_t src_T;            /* frames in the source when t_index is given (else = T)              */
  const float* ch_scale;    /* [C] or NULL                                                        */
  const float* ch_shift;    /* [C] or NULL                                                        */
} pv_layout_desc;
int pv_ingest_ncdhw(const pv_layout_desc* d, pv_stream_t stream);
int pv_egress_ncdhw(const pv_layout_desc* d, pv_stream_t stream);

/* ---- row ops on (rows, C) matrices -------------------------------------------------
 * pv_layernorm: nn.LayerNorm(eps) over C (models/vision_transformers.py:333-335,
 *   layers/attention.py:199-205).
 * pv_softmax_rows: nn.Softmax(dim=channel) head activation (models/head.py:384).
 * pv_mean_rows: AdaptiveAvgPool3d(1)+view (models/head.py:386-390): y[b][c] = mean over
 *   `rows_per_batch` rows, fp32 out.
 * pv_add_posenc: SpatioTemporalClsPositionalEncoding.forward
 *   (layers/positional_encoding.py:112-136): writes the cls row and adds the separable
 *   spatial + temporal (+ class) embedding in place.
 */
typedef struct pv_rows_desc {
  const void* x; void* y;
  const float* gamma; const float* beta;  /* layernorm only */
  int64_t rows; int32_t C, ldx, ldy;
  int32_t rows_per_batch;                 /* mean_rows only */
  float eps;
  int32_t dtype;                          /* of x; y same except mean_rows (fp32) */
  int32_t x_f32;                          /* layernorm: 1 = x is fp32 while y is `dtype` */
  int32_t g_period;                       /* layernorm: 0, or gamma/beta are [g_period][C] tables and row r */
                                          /* uses table row r % g_period (per-head norms of several tensors  */
                                          /* packed side by side, layers/attention.py:202-205); C <= 256    */
  int32_t act;                            /* affine_rows only: PV_ACT_* applied after the affine map                */
  int32_t n_prefix;                       /* affine_rows only: the first n_prefix rows of every rows_per_batch block */
                                          /* (the cls token) pass through untouched                                 */
} pv_rows_desc;
int pv_layernorm(const pv_rows_desc* d, pv_stream_t stream);
/* BatchNorm in eval mode on token rows -- the norm="batchnorm" MViT (models/vision_transformers.py:336-339):
 * y[r][c] = act(x[r][c] * gamma[c] + beta[c]) with gamma / beta the folded running statistics.
 *   nn.BatchNorm1d block norms (layers/attention.py:738-753): x fp32 stream (x_f32) -> bf16 GEMM operand;
 *   nn.BatchNorm3d(head_dim) + GELU BEFORE the pooling conv (layers/attention.py:186-190): in place on the q / k / v
 *   columns of the qkv GEMM output (ldx = ldy = row stride, C = heads * head_dim with g_period = 0 and per-channel
 *   tables repeated per head by the caller), cls rows skipped (rows_per_batch, n_prefix).
 * gamma == NULL: scale 1; beta == NULL: shift 0 (a plain fp32 -> bf16 copy of rows, used for the cls rows of a head
 * without a final norm). */
int pv_affine_rows(const pv_rows_desc* d, pv_stream_t stream);
int pv_softmax_rows(const pv_rows_desc* d, pv_stream_t stream);
int pv_mean_rows(const pv_rows_desc* d, pv_stream_t stream);

typedef struct pv_posenc_desc {
  void* x;                    /* [B][1+T*HW][ld] tokens, row 0 = cls (written here) */
  const float* cls_token;     /* [C] or NULL (no cls: rows start at the grid)       */
  const float* pos_spatial;   /* [HW][C]  (separable) or full [T*HW(+1)][C]          */
  const float* pos_temporal;  /* [T][C]  or NULL when pos_spatial is the full table  */
  const float* pos_class;     /* [C] or NULL                                        */
  int32_t B, T, HW, C, ld;
  int32_t dtype;
  int32_t cls_only;           /* 1: write the cls row only (the tables were added by the producing conv) */
} pv_posenc_desc;
int pv_add_posenc(const pv_posenc_desc* d, pv_stream_t stream);

/* ---- fused pooled attention --------------------------------------------------------
 * softmax((q*scale) k^T) v [+ q] of layers/attention.py:531-539 without materialising
 * the scores.  q/k/v/o are token tensors whose channel dim is heads*head_dim
 * (head h at channel offset h*head_dim); fp32 online softmax, MFMA QK^T and PV.
 */
typedef struct pv_attention_desc {
  const void* q; const void* k; const void* v; void* o;
  int64_t q_bs, k_bs, v_bs, o_bs;
  int32_t ldq, ldk, ldv, ldo;
  int32_t B, heads, head_dim, Nq, Nk;
  float scale;
  int32_t residual_q;   /* 1: o += q (residual_pool, layers/attention.py:536-537) */
  int32_t dtype;
} pv_attention_desc;
int pv_attention(const pv_attention_desc* d, pv_stream_t stream);

/* ---- video-level ensembling (the step right after the path) ---------------
 * pytorchvideo_trainer/module/video_classification.py:244-311: preds = softmax(logits) of every clip
 * (30 views per video in the model zoo's test protocol) are summed -- or max-ed -- into its video's
 * score row and the clip is counted: accum[video_index[i]][:] (op)= softmax(logits[i][:]),
 * counts[video_index[i]] += 1, clips taken in index order (deterministic; no atomics).
 * The final division by the count is left to the caller (after the cross-rank reduction).
 */
typedef struct pv_ensemble_desc {
  const float* logits;          /* [N][ld] fp32                                  */
  const int32_t* video_index;   /* [N], each in [0, V)                           */
  float* accum;                 /* [V][C] fp32, updated in place                 */
  int32_t* counts;              /* [V], updated in place                         */
  int32_t N, C, ld, V;
  int32_t mode;                 /* 0 = sum, 1 = max                              */
} pv_ensemble_desc;
int pv_ensemble_scores(const pv_ensemble_desc* d, pv_stream_t stream);

/* ---- elementwise -------------------------------------------------------------------
 * y = act(a + b) over (rows, C): residual joins that cannot ride in a conv epilogue.
 */
typedef struct pv_add_desc {
  const void* a; const void* b; void* y;
  int64_t rows; int32_t C, lda, ldb, ldy;
  int32_t act, dtype;
} pv_add_desc;
int pv_add_act(const pv_add_desc* d, pv_stream_t stream);

/* ---- RoIAlign (+ the whole-window max pool that follows it in the detection head) ---------
 * ResNetRoIHead.forward, models/head.py:437-482: x.squeeze(T) -> roi_layer(x, bboxes) -> pool_spatial.
 * roi_layer is torchvision.ops.RoIAlign (a third-party op, default of create_res_roi_pooling_head,
 * models/head.py:212): for box n = (batch index, x1, y1, x2, y2) and output bin (ph, pw) the mean of
 * grid_h x grid_w bilinear samples of the feature map, grid = sampling_ratio, or ceil(roi size / bins)
 * when sampling_ratio <= 0; `aligned` = 0 is torchvision's default (no half-pixel shift, roi size >= 1).
 * pool_max = 1 additionally takes the maximum over all ph x pw bins (MaxPool2d(resolution, stride=1),
 * models/head.py:317) and writes one row per box: the [R, C, ph, pw] tensor is never stored.
 * Boxes are read on the device at launch time (their VALUES may change between graph replays, their
 * count may not).  A box whose batch index is outside [0, B) produces zeros.
 */
typedef struct pv_roi_align_desc {
  const void* x;        /* [B][H][W][ldx] features (temporal dim already pooled to 1) */
  const float* boxes;   /* [R][5] fp32: batch index, x1, y1, x2, y2 in input-image pixels */
  void* y;              /* pool_max ? [R][ldy] : [R][ph][pw][ldy]                     */
  int64_t x_bs;         /* elements between batch items of x                          */
  int32_t ldx, ldy;
  int32_t B, H, W, C, R;
  int32_t ph, pw;       /* output resolution (head_spatial_resolution)               */
  int32_t sampling_ratio, aligned, pool_max;
  float spatial_scale;
  int32_t dtype;        /* storage type of x and y                                    */
} pv_roi_align_desc;
int pv_roi_align(const pv_roi_align_desc* d, pv_stream_t stream);

/* ---- X3D bottleneck block, fully fused (round 6) ------------------------------------------------------------
 * Replaces a whole residual block without squeeze-excitation -- pytorchvideo/models/x3d.py:169-212
 * (create_x3d_bottleneck_block: conv_a 1x1x1 + norm_a + ReLU, conv_b depthwise 3x3x3 + norm_b + Swish, conv_c 1x1x1 + norm_c),
 * models/resnet.py:1345-1365 (BottleneckBlock.forward) and :1179-1189 (ResBlock.forward: + shortcut, activation) -- in ONE launch
 * that never writes the expanded tensor (csrc/pv_block.hip):
 *     y = act_out( r + sc * (Wc . act_b( sb * dw3x3x3( act_a( sa * (Wa . x) + ha ) ) + hb )) + hc )
 * x (B, T, H, W, cin), r = residual (B, T, H, W, cout) or NULL, y (B, T, H, W, cout): bf16 channels-last, all strides 1,
 * the depthwise conv zero-pads the EXPANDED tensor by 1 on every side.  Host-packed operands (Cp = round_up(C, 32),
 * cin_p = round_up(cin, 32)):
 *   wa  [Cp/16][cin_p/32][64 lanes][8] bf16   Wa[16 mt + (l&15)][32 ks + 8 (l>>4) + j]   (zeros beyond C / cin)
 *   wb  [27][Cp] fp32                         depthwise taps, t = (kt*3 + kh)*3 + kw
 *   wc  [cout/16][Cp/32][64 lanes][8] bf16    Wc[16 mt + (l&15)][32 ks + 8 (l>>4) + j]   (zeros beyond C)
 *   sa, ha, sb, hb [Cp] fp32; sc, hc [cout] fp32: the folded BatchNorms (zeros in the padding channels)
 * (pytorchvideo_amd/accelerator/mi355x/emit.py::emit_fused_bottleneck packs them.)  pv_bottleneck_supported(d) == 1 for the
 * geometries the kernel is instantiated for: W <= 14, X3D res4 (cin 96, C 216, cout 96). */
typedef struct pv_bottleneck_desc {
  const void* x; void* y; const void* residual;
  const void* wa; const float* wb; const void* wc;
  const float* sa; const float* ha; const float* sb; const float* hb; const float* sc; const float* hc;
  int64_t x_bs, y_bs, r_bs;          /* batch strides, elements */
  int32_t ldx, ldy, ldr;             /* voxel strides, elements */
  int32_t B, T, H, W;
  int32_t cin, C, cout;              /* true channel counts */
  int32_t act_a, act_b, act_out;     /* pv_act after norm_a / norm_b / the residual join */
  int32_t dtype;                     /* PV_BF16 */
  /* mode PV_BLOCK_AB -- blocks WITH squeeze-excitation (x3d.py:190-207: norm_b = Sequential(BN, SE)): the launch stops behind
   * conv_b.  y receives  sb * dw3x3x3(act_a(...)) + hb  (no activation: conv_c applies gate and Swish while loading its operand)
   * as bf16 (B, T, H, W, C) with voxel stride ldy, and psum[b][blk][round_up(C, 8)] (fp32, blk < pv_bottleneck_psum_blocks(d))
   * the sums of those values over the block's voxels -- the squeeze of fvcore's SqueezeExcitation, one writer per entry, no
   * atomics; pv_se_gate turns them into the gate.  wc, sc, hc, residual, cout, act_b, act_out are ignored. */
  int32_t mode;
  float* psum;
} pv_bottleneck_desc;
#define PV_BLOCK_FULL 0
#define PV_BLOCK_AB 1
int pv_bottleneck(const pv_bottleneck_desc* d, pv_stream_t stream);
int pv_bottleneck_supported(const pv_bottleneck_desc* d);
int pv_bottleneck_psum_blocks(const pv_bottleneck_desc* d);

/* ---- fused MLP of a MultiScaleBlock on token rows -------------------------------------------------------
 * Replaces  norm2 -> Mlp.fc1 -> GELU -> Mlp.fc2 -> + residual  (pytorchvideo/layers/attention.py:102-114 Mlp.forward,
 * :750-757 the block's second half) in ONE launch whose hidden tensor never leaves the chip (csrc/pv_mlp.hip):
 *     y[m][:] = R[m][:] + b2 + W2 . act(W1 . xn[m][:] + b1)
 *   ln_gamma != NULL:  x is the fp32 token stream [M][ldx]; xn = LayerNorm(x; ln_gamma, ln_beta, ln_eps) is computed
 *                      in the kernel and R = x (the row is read once); needs C == Cout, residual == NULL.
 *   ln_gamma == NULL:  x is a bf16 operand tensor [M][ldx] (the LayerNorm output); R = residual (fp32 [M][ldr]) or 0.
 * y is fp32 [M][ldy].  dtype must be PV_BF16 (weights bf16, fp32 accumulation / bias / activation / LayerNorm).
 * `w12` is the host-packed LDS image the kernel streams: H/32 + 1 blocks of  C/16*1024 + Cout/32*2048 + 256  bytes, block j
 * holding W1 of hidden block j and W2 of hidden block j - 1 (the kernel multiplies phase B one block behind phase A so that the
 * activation issues in the shadow of MFMAs; W2 of block -1, W1 and b1 of block H/32 are zeros), FOLLOWED BY TWO MORE BLOCKS OF
 * PADDING (any finite values; the kernel prefetches two blocks ahead without a branch).  Block j, in fragments of
 * v_mfma_f32_16x16x32_bf16 (round 6: 16 token rows per wave, lane l = 16 g + m):
 *   [f = 2 ks + uh < C/16][l < 64][j8 < 8]  bf16  W1[32 j + 16 uh + (l&15)][32 ks + 8 (l>>4) + j8]
 *   [ob < Cout/16][l < 64][j8 < 8]          bf16  W2[32 (ob>>1) + 8 ((l&15)>>2) + 4 (ob&1) + (l&3)][32 (j-1) + u(l>>4, j8)],
 *        u(g, j8) = j8 < 4 ? 4 g + j8 : 16 + 4 g + j8 - 4
 *   [u < 32] fp32  b1[32 j + u]  (zeros when the layer has no bias), then 128 bytes of padding
 * (pytorchvideo_amd/accelerator/mi355x/emit_mvit.py::pack_mlp_weights builds it).  b2 is [Cout] fp32 or NULL.
 * pv_mlp_rows_supported(d) == 1 for the (C, Cout) pairs the kernel is instantiated for (MViT-B: 96/192, 192/192,
 * 192/384, 384/384; H any multiple of 32). */
typedef struct pv_mlp_desc {
  const void* x;
  const void* w12;
  void* y;
  const float* b2;
  const float* residual;
  const float* ln_gamma;
  const float* ln_beta;
  int64_t M;             /* token rows */
  int32_t C, H, Cout;    /* in / hidden / out widths */
  int32_t ldx, ldr, ldy; /* row strides, elements */
  int32_t act;           /* pv_act between the two Linears */
  int32_t dtype;         /* PV_BF16 */
  float ln_eps;
  /* Optional (round 4): yn != NULL -> the kernel ALSO writes yn[m][:] = LayerNorm(y[m][:]; nn_gamma, nn_beta, nn_eps) as
   * bf16 [M][ldyn] from the rows it still holds in registers: norm1 of the NEXT MultiScaleBlock (layers/attention.py:729-737),
   * i.e. the operand of that block's q|k|v projection, without a LayerNorm launch reading the stream back. */
  void* yn;
  const float* nn_gamma;
  const float* nn_beta;
  int32_t ldyn;
  float nn_eps;
} pv_mlp_desc;
int pv_mlp_rows(const pv_mlp_desc* d, pv_stream_t stream);
int pv_mlp_rows_supported(const pv_mlp_desc* d);

/* ---- LayerNorm + Linear on token rows ----------------------------------------------------------------------
 * Replaces  norm1 -> the q | k | v Linear(s)  of MultiScaleBlock / MultiScaleAttention (pytorchvideo/layers/attention.py:
 * 729-737 norm1, :425-451 _qkv_proj; the three Linears concatenated along the output dim) in ONE launch:
 *     y[m][:] = act( W . LayerNorm(x[m][:]; ln_gamma, ln_beta, ln_eps) + b )
 * x is the fp32 token stream [M][ldx], y is bf16 [M][ldy]; the bf16 operand tensor between LayerNorm and the GEMM is
 * never written.  `wb` is the host-packed per-output-block LDS image, N/32 blocks of  C/16*1024 + 256  bytes followed
 * by two more blocks of padding (prefetched, never used):
 *   [ks < C/16][hi < 2][rho < 32][j < 8]  bf16  W[32 nb + chi(rho)][32 (ks>>1) + 16 hi + 8 (ks&1) + j]   (chi as above)
 *   [hi < 2][r < 16]  fp32  b[32 nb + 16 hi + r]  (zeros without bias), then 128 bytes of padding
 * (pytorchvideo_amd/accelerator/mi355x/emit_mvit.py::pack_ln_linear_weights).  C in {96, 192, 384, 768}, N % 32 == 0. */
typedef struct pv_ln_linear_desc {
  const void* x;
  const void* wb;
  void* y;
  const float* ln_gamma;
  const float* ln_beta;
  int64_t M;
  int32_t C, N;
  int32_t ldx, ldy;
  int32_t act;
  int32_t dtype;         /* PV_BF16 */
  float ln_eps;
} pv_ln_linear_desc;
int pv_ln_linear_rows(const pv_ln_linear_desc* d, pv_stream_t stream);
int pv_ln_linear_rows_supported(const pv_ln_linear_desc* d);

/* ---- execution plan ----------------------------------------------------------------
 * A deploy-form model is specialised to one input size (reference contract:
 * accelerator/deployment/mobile_cpu/utils/model_conversion.py:100-103), so its forward
 * is a fixed launch list.  The plan records descriptors once; pv_plan_launch replays them
 * from C++ (no Python per layer), optionally through a captured hipGraph.
 */
enum pv_op_kind {
  PV_OP_CONV3D = 1, PV_OP_DWCONV3D = 2, PV_OP_SE_GATE = 3, PV_OP_POOL3D = 4,
  PV_OP_LAYERNORM = 5, PV_OP_SOFTMAX_ROWS = 6, PV_OP_MEAN_ROWS = 7, PV_OP_POSENC = 8,
  PV_OP_ATTENTION = 9, PV_OP_ADD_ACT = 10, PV_OP_INGEST = 11, PV_OP_EGRESS = 12, PV_OP_TOKEN_POOL = 13,
  PV_OP_ROI_ALIGN = 14, PV_OP_LATERAL = 15, PV_OP_AFFINE_ROWS = 16, PV_OP_MLP_ROWS = 17, PV_OP_LN_LINEAR = 18,
  PV_OP_BOTTLENECK = 19
};
typedef struct pv_plan pv_plan;
pv_plan* pv_plan_create(void);
void pv_plan_destroy(pv_plan* p);
int pv_plan_add(pv_plan* p, int op_kind, const void* desc, size_t desc_bytes); /* returns op index */
int pv_plan_size(const pv_plan* p);
int pv_plan_launch(pv_plan* p, pv_stream_t stream);                 /* eager replay   */
int pv_plan_launch_range(pv_plan* p, int first, int last, pv_stream_t stream);
int pv_plan_graph_build(pv_plan* p, pv_stream_t stream);           /* capture+instantiate */
int pv_plan_graph_launch(pv_plan* p, pv_stream_t stream);
/* `n` independent plans (sub-batches of one forward: pytorchvideo_amd.accelerator.mi355x.conversion.SplitBatchDeployed)
 * captured as `n` PARALLEL branches of ONE graph: the runtime runs the branches side by side, so the tail of one
 * sub-batch's kernel overlaps the other's work.  The joint graph is its OWN object (round 3): it does not live in,
 * and is not disturbed by, the graph slot of any member plan (pv_plan_graph_build on a member leaves it intact);
 * rebuild it after pv_plan_add on a member.  Replaces the Python loop over sub-batches; the reference has no
 * counterpart (its forward is one ATen call sequence per batch, models/net.py:41-44). */
typedef struct pv_joint pv_joint;
pv_joint* pv_joint_create(void);
void pv_joint_destroy(pv_joint* j);
int pv_joint_build(pv_joint* j, pv_plan* const* plans, int n);     /* capture + instantiate, 1 <= n <= 16 */
int pv_joint_launch(pv_joint* j, pv_stream_t stream);
int pv_joint_branches(const pv_joint* j);                          /* 0 until built */
/* per-op device time in ms: every op timed in situ between its own pair of HIP events on `stream`, behind a
 * queued un-instrumented replay (the host never paces the measurement); minimum over `iters` passes, minus the
 * null interval of an empty event pair */
int pv_plan_profile(pv_plan* p, pv_stream_t stream, int iters, float* ms_per_op);
/* symbol of the kernel op `i` was routed to ("gemm_glds_kernel", "pw_stream_kernel", ...; the last one when an op launches
 * several), known after pv_plan_profile ran; "" before.  The string lives as long as the plan. */
const char* pv_plan_op_kernel(const pv_plan* p, int i);

/* ---- head collective ----------------------------------------------------------------------
 * The one exchange of the batch-sharded forward: every rank contributes its [B_local, classes] logits and obtains
 * the [B_global, classes] rows (the reference's forward has no collective, models/net.py:41-44; BASELINE.json
 * prescribes this one on the classification head, models/head.py:376-382).  The library binds RCCL itself
 * (dlopen of librccl: ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString)
 * and ENQUEUES the all-gather from C on the stream the forward was launched on, directly behind the graph
 * launch -- no Python, no torch.distributed call per step.  Bootstrap: rank 0 makes the 128-byte id
 * (pv_comm_unique_id), the host side hands it to every rank (any out-of-band channel; pytorchvideo_amd.parallel
 * uses the process group's store), every rank calls pv_comm_create.
 *   lib_paths: ':'-separated candidates for librccl, tried in order (a copy already mapped into the process --
 *   torch's -- is preferred so that one RCCL instance serves both); NULL = "librccl.so:librccl.so.1".
 * pv_comm_all_gather(send, recv, bytes_per_rank): recv[r*bytes .. (r+1)*bytes) = rank r's send; byte-typed (ncclUint8),
 * asynchronous on `stream`, safe behind pv_plan_graph_launch / pv_joint_launch on the same stream.
 * pv_forward_gather: ONE C call per step of the batch-sharded forward (what a step of `bench.py --gpus N` is):
 *   graph launch (plan `p`, or joint graph `j` -- exactly one non-NULL) -> the rank's logits rows collected from up to
 *   16 strided sources (the result buffers of the sub-batch plans) into `staging` -> all-gather into `recv`
 *   ([world][bytes_per_rank], i.e. the [B_global, classes] rows), all enqueued on `stream`.
 *   c == NULL or world 1: the rows are collected straight into `recv` (no RCCL call). */
typedef struct pv_comm pv_comm;
typedef struct pv_gather_src {
  const void* ptr;     /* first row                         */
  size_t row_bytes;    /* bytes copied per row              */
  size_t row_pitch;    /* bytes between rows (>= row_bytes)  */
  int64_t rows;
} pv_gather_src;
int pv_comm_probe(const char* lib_paths);          /* PV_OK when a librccl with the five entry points can be bound */
int pv_comm_unique_id(void* id128, const char* lib_paths);
int pv_comm_create(pv_comm** out, const void* id128, int rank, int world, const char* lib_paths);
void pv_comm_destroy(pv_comm* c);
int pv_comm_rank(const pv_comm* c);
int pv_comm_world(const pv_comm* c);
const char* pv_comm_library(const pv_comm* c);     /* path of the RCCL copy the communicator bound */
int pv_comm_all_gather(pv_comm* c, const void* send, void* recv, size_t bytes_per_rank, pv_stream_t stream);
int pv_forward_gather(pv_plan* p, pv_joint* j, pv_comm* c, const pv_gather_src* srcs, int n_srcs, void* staging,
                      void* recv, pv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PV_MI355X_H_ */
