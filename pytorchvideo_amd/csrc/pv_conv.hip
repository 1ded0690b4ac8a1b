// Dense 3-D convolution / linear layer as an NDHWC implicit GEMM on MFMA (gfx950).
//
//   out[voxel m][channel n] = sum_k  X'[m][k] * W[n][k],   k = tap*cin + c
//
// Mapping to the matrix core: the *weights* are the MFMA A operand (rows = output
// channels) and the *activations* the B operand (cols = voxels), so that a lane's four
// accumulator registers are four consecutive channels of one voxel.  The LDS rows of the
// weight tile are permuted (pairs of 16-channel tiles interleaved in groups of 4) so that
// each lane ends up owning 8 contiguous output channels of its voxel: the epilogue
// (folded-BN scale/shift, residual add, activation) runs in registers and leaves as one
// 16-byte NDHWC store per lane, no LDS transpose.
//
// Staging: global -> registers -> LDS, double-buffered, one barrier per K step; the
// activation tile is gathered on the fly (taps, strides, zero padding) and can be scaled by
// a per-(batch,channel) gate and passed through an activation on its way into LDS
// (squeeze-excitation * Swish of X3D folded into conv_c's load).
#include "pv_common.h"

#include <stdlib.h>
int pv_pwconv_stream_try(const pv_conv3d_desc& d, hipStream_t s);      // pv_pwconv.hip
int pv_gemm_glds_try(const pv_conv3d_desc& d, bool pw, hipStream_t s);  // pv_gemm.hip
int pv_head_rows_try(const pv_conv3d_desc& d, hipStream_t s);              // pv_headgemm.hip
int pv_gemm8_try(const pv_conv3d_desc& d, bool pw, hipStream_t s);      // pv_gemm8.hip
int pv_gemm9_try(const pv_conv3d_desc& d, bool pw, hipStream_t s);      // pv_gemm9.hip
int pv_tapstream_try(const pv_conv3d_desc& d, hipStream_t s, bool dry = false);   // pv_lateral.hip
int pv_stem_c4(const pv_conv3d_desc& d, hipStream_t s);                 // pv_stem.hip
int pv_stem_dwt_supported(const pv_conv3d_desc& d);                     // pv_stem.hip
int pv_pwconv_x2_supported(const pv_conv3d_desc& d);                    // pv_pwconv.hip

namespace {

constexpr int kThreads = 256;
constexpr int kBK = 32;
constexpr int kMaxTaps = 512;

template <typename T> struct LdsLd { static constexpr int v = kBK + 8; };
template <> struct LdsLd<float> { static constexpr int v = kBK + 4; };

template <typename T, int BM, int BN, int WM_, int WN_, bool PW>
__global__ __launch_bounds__(kThreads) void conv_igemm_kernel(const pv_conv3d_desc d) {
  constexpr int LD = LdsLd<T>::v;
  constexpr int TM = BM / WM_ / 16;  // voxel tiles per wave
  constexpr int TN = BN / WN_ / 16;  // channel tiles per wave
  static_assert(WM_ * WN_ == 4, "4 waves");
  static_assert(TN % 2 == 0, "channel tiles are stored in pairs");
  constexpr int KC = kBK / 8;                          // chunks per tile row
  constexpr int XCH = BM * KC / kThreads;              // activation chunks per thread
  constexpr int WCH = (BN * KC + kThreads - 1) / kThreads;
  static_assert(BM * KC % kThreads == 0, "");

  __shared__ __attribute__((aligned(16))) T smem[2 * (BM + BN) * LD];
  __shared__ int s_tap[PW ? 1 : kMaxTaps];
  // the squeeze-excitation gates of the (at most two) clips a tile's rows belong to: read once per workgroup instead of once per
  // staged chunk -- the per-chunk global loads sat on the critical path of every K step (X3D res5's gated conv_c: 39 -> ... us)
  constexpr int kGateMaxC = 512;
  __shared__ __attribute__((aligned(16))) float s_gate[PW ? 2 * kGateMaxC : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM_;
  const int wn = wave / WM_;
  const int l15 = lane & 15;
  const int q = lane >> 4;

  const int S_out = d.To * d.Ho * d.Wo;
  const long M = (long)d.B * S_out;
  const int taps = d.kt * d.kh * d.kw;
  const int K = taps * d.cin;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const int n_tiles_n = (cout_p8 + BN - 1) / BN;
  const int tile_n = blockIdx.x % n_tiles_n;
  const long tile_m = blockIdx.x / n_tiles_n;
  const long m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const T* __restrict__ X = static_cast<const T*>(d.x);
  const T* __restrict__ Wt = static_cast<const T*>(d.w);

  if constexpr (!PW) {
    for (int t = tid; t < taps; t += kThreads) {
      const int dt = t / (d.kh * d.kw);
      const int r = t - dt * d.kh * d.kw;
      const int dh = r / d.kw;
      const int dw = r - dh * d.kw;
      // offsets in voxels: the dilation is folded into the table, the gather loop does not know about it
      s_tap[t] = (dt * (d.dil_t > 1 ? d.dil_t : 1)) | ((dh * (d.dil_h > 1 ? d.dil_h : 1)) << 8) |
                 ((dw * (d.dil_w > 1 ? d.dil_w : 1)) << 16);
    }
    __syncthreads();
  }

  bool gate_lds = false;
  int gate_b0 = 0;
  if constexpr (PW) {
    if (d.a_gate != nullptr) {
      gate_b0 = (int)(m0 / S_out);
      const long m_last = (m0 + BM < M ? m0 + BM : M) - 1;
      gate_lds = d.cin <= kGateMaxC && (d.cin & 7) == 0 && (int)(m_last / S_out) - gate_b0 <= 1;      // (workgroup-uniform)
      if (gate_lds) {
        for (int i = tid; i < 2 * d.cin; i += kThreads) {
          const int bb = gate_b0 + (i >= d.cin ? 1 : 0);
          s_gate[i] = bb < d.B ? d.a_gate[(long)bb * d.cin + (i >= d.cin ? i - d.cin : i)] : 0.f;
        }
        __syncthreads();
      }
    }
  }

  // ---- per-thread staging geometry (rows are fixed across K steps) ----
  const int kc = tid % KC;
  long x_off[XCH];   // PW: element offset of the row; general: batch offset
  int x_t[XCH], x_h[XCH], x_w[XCH];
  int x_b[XCH];
  bool x_ok[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const int r = tid / KC + i * (kThreads / KC);
    const long m = m0 + r;
    x_ok[i] = m < M;
    const long mm = x_ok[i] ? m : 0;
    const int b = (int)(mm / S_out);
    const int sp = (int)(mm - (long)b * S_out);
    x_b[i] = b;
    if constexpr (PW) {
      x_off[i] = (long)b * d.x_bs + (long)sp * d.ldx;
      x_t[i] = x_h[i] = x_w[i] = 0;
    } else {
      const int to = sp / (d.Ho * d.Wo);
      const int r2 = sp - to * d.Ho * d.Wo;
      const int ho = r2 / d.Wo;
      const int wo = r2 - ho * d.Wo;
      x_off[i] = (long)b * d.x_bs;
      x_t[i] = to * d.st - d.pt;
      x_h[i] = ho * d.sh - d.ph;
      x_w[i] = wo * d.sw - d.pw;
    }
  }
  // weight rows: LDS row r holds output channel n0 + perm(r)
  int w_row[WCH];
  long w_off[WCH];
  bool w_ok[WCH];
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int id = tid + i * kThreads;
    const int r = id / KC;
    w_row[i] = r;
    const int wv = r / (TN * 16);
    const int rr = r - wv * (TN * 16);
    const int tn = rr >> 4;
    const int ii = rr & 15;
    const int c = n0 + wv * (TN * 16) + (tn >> 1) * 32 + (ii >> 2) * 8 + (tn & 1) * 4 + (ii & 3);
    w_ok[i] = (id < BN * KC) && (c < d.cout);
    w_off[i] = (long)c * K;
  }

  // Two K tiles in flight in registers, in two FIXED sets used alternately (round 6; no copies between them: a copy waits for the
  // load).  A step's global loads used to be hidden only behind ONE step of MFMAs (~0.1 us of a ~2 us latency), so small-M layers
  // -- X3D res5's gated conv_c: 392 tiles, 14 steps -- ran at one memory latency per step.
  Chunk8<T> xrA[XCH], wrA[WCH], xrB[XCH], wrB[WCH];

  auto load_global = [&](Chunk8<T> (&xr)[XCH], Chunk8<T> (&wr)[WCH], int ks) {
    const int k0 = ks * kBK + kc * 8;
    if constexpr (PW) {
      const bool kok = k0 < K;
#pragma unroll
      for (int i = 0; i < XCH; ++i) {
        if (x_ok[i] && kok) xr[i].load(X + x_off[i] + k0);
        else xr[i].zero();
      }
    } else {
      const int tap = k0 / d.cin;
      const int c = k0 - tap * d.cin;
      const bool kok = tap < taps;
      const int tp = kok ? s_tap[tap] : 0;
      const int dt = tp & 255, dh = (tp >> 8) & 255, dw = tp >> 16;
#pragma unroll
      for (int i = 0; i < XCH; ++i) {
        const int ti = x_t[i] + dt, hi = x_h[i] + dh, wi = x_w[i] + dw;
        const bool ok = x_ok[i] && kok && (unsigned)ti < (unsigned)d.Ti &&
                        (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
        if (ok) xr[i].load(X + x_off[i] + ((long)(ti * d.Hi + hi) * d.Wi + wi) * d.ldx + c);
        else xr[i].zero();
      }
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      if (w_ok[i] && k0 < K) wr[i].load(Wt + w_off[i] + k0);
      else wr[i].zero();
    }
  };

  auto store_lds = [&](Chunk8<T> (&xr)[XCH], Chunk8<T> (&wr)[WCH], int buf, int ks) {
    T* xs = smem + buf * (BM + BN) * LD;
    T* ws = xs + BM * LD;
    const int k0 = ks * kBK + kc * 8;
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int r = tid / KC + i * (kThreads / KC);
      if (d.a_gate != nullptr || d.a_act != PV_ACT_NONE) {
        // pointwise only (checked on the host): k0 is the input channel
        float f[8];
        xr[i].to_f32(f);
        if (d.a_gate != nullptr && k0 < K) {
          float4 g0, g1;
          if (PW && gate_lds) {
            const float* g = s_gate + (x_ok[i] ? x_b[i] - gate_b0 : 0) * d.cin + k0;      // (rows past M carry zeros and clip 0)
            g0 = *reinterpret_cast<const float4*>(g);
            g1 = *reinterpret_cast<const float4*>(g + 4);
          } else {
            const float* g = d.a_gate + (long)x_b[i] * d.cin + k0;
            g0 = *reinterpret_cast<const float4*>(g);
            g1 = *reinterpret_cast<const float4*>(g + 4);
          }
          f[0] *= g0.x; f[1] *= g0.y; f[2] *= g0.z; f[3] *= g0.w;
          f[4] *= g1.x; f[5] *= g1.y; f[6] *= g1.z; f[7] *= g1.w;
        }
        pv_apply_act_n<sizeof(T) == 2>(f, d.a_act);
        xr[i].from_f32(f);
      }
      xr[i].store(xs + r * LD + kc * 8);
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int id = tid + i * kThreads;
      if (id < BN * KC) wr[i].store(ws + w_row[i] * LD + kc * 8);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // which channel-tile pairs of this wave hold real channels (wave-uniform)
  const int wave_c0 = n0 + wn * (TN * 16);

  const int nk = (K + kBK - 1) / kBK;
  auto compute = [&](int buf) {
    const T* xs = smem + buf * (BM + BN) * LD;
    const T* ws = xs + BM * LD;
    if constexpr (sizeof(T) == 2) {
      bf16x8 xf[TM], wf[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t)
        xf[t] = *reinterpret_cast<const bf16x8*>(xs + (wm * TM * 16 + t * 16 + l15) * LD + q * 8);
#pragma unroll
      for (int t = 0; t < TN; ++t)
        wf[t] = *reinterpret_cast<const bf16x8*>(ws + (wn * TN * 16 + t * 16 + l15) * LD + q * 8);
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        if (wave_c0 + (a >> 1) * 32 < cout_p8) {
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < kBK / 4; ++kk) {
        float xf[TM], wf[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) xf[t] = (float)xs[(wm * TM * 16 + t * 16 + l15) * LD + kk * 4 + q];
#pragma unroll
        for (int t = 0; t < TN; ++t) wf[t] = (float)ws[(wn * TN * 16 + t * 16 + l15) * LD + kk * 4 + q];
#pragma unroll
        for (int a = 0; a < TN; ++a) {
          if (wave_c0 + (a >> 1) * 32 < cout_p8) {
#pragma unroll
            for (int b = 0; b < TM; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[a], xf[b], acc[a][b], 0, 0, 0);
          }
        }
      }
    }
  };
  load_global(xrA, wrA, 0);
  store_lds(xrA, wrA, 0, 0);
  if (nk > 1) load_global(xrA, wrA, 1);
  if (nk > 2) load_global(xrB, wrB, 2);
  __syncthreads();
  for (int ks = 0; ks < nk; ks += 2) {
    // even step: tile ks + 1 waits in set A, tile ks + 2 is on its way into set B
    compute(0);
    if (ks + 1 < nk) store_lds(xrA, wrA, 1, ks + 1);
    if (ks + 3 < nk) load_global(xrA, wrA, ks + 3);
    __syncthreads();
    if (ks + 1 >= nk) break;
    // odd step: tile ks + 2 waits in set B, tile ks + 3 is on its way into set A
    compute(1);
    if (ks + 2 < nk) store_lds(xrB, wrB, 0, ks + 2);
    if (ks + 4 < nk) load_global(xrB, wrB, ks + 4);
    __syncthreads();
  }

  // ---- epilogue: lane owns channels c0..c0+7 of voxel m for every (pair p, tile t) ----
  const T* __restrict__ R = static_cast<const T*>(d.residual);
#pragma unroll
  for (int p = 0; p < TN / 2; ++p) {
    const int c0 = wave_c0 + p * 32 + q * 8;
    if (c0 >= cout_p8) continue;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = (c0 + j) < d.cout;
      sc[j] = ok ? (d.scale ? d.scale[c0 + j] : 1.0f) : 0.0f;
      sh[j] = ok ? (d.shift ? d.shift[c0 + j] : 0.0f) : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const long m = m0 + wm * (TM * 16) + t * 16 + l15;
      if (m >= M) continue;
      const int b = (int)(m / S_out);
      const long sp = m - (long)b * S_out;
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[2 * p][t][j] * sc[j] + sh[j];
        v[4 + j] = acc[2 * p + 1][t][j] * sc[4 + j] + sh[4 + j];
      }
      if (R != nullptr) {
        float rf[8];
        if (d.r_f32) {
          Chunk8<float> rc;
          rc.load(static_cast<const float*>(d.residual) + (long)b * d.r_bs + sp * d.ldr + c0);
          rc.to_f32(rf);
        } else {
          Chunk8<T> rc;
          rc.load(R + (long)b * d.r_bs + sp * d.ldr + c0);
          rc.to_f32(rf);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rf[j];
      }
      pv_apply_act_n<sizeof(T) == 2>(v, d.act);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (c0 + j >= d.cout) v[j] = 0.0f;
      const long yo = (long)b * d.y_bs + sp * d.ldy + c0;
      if (d.y_f32) {
        Chunk8<float> oc;
        oc.from_f32(v);
        oc.store(static_cast<float*>(d.y) + yo);
      } else {
        Chunk8<T> oc;
        oc.from_f32(v);
        oc.store(static_cast<T*>(d.y) + yo);
      }
    }
  }
}

template <typename T, int BM, int BN, int WM_, int WN_>
int launch_cfg(const pv_conv3d_desc& d, bool pw, hipStream_t s) {
  const long M = (long)d.B * d.To * d.Ho * d.Wo;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const long tiles = pv_ceil_div(M, BM) * pv_ceil_div(cout_p8, BN);
  if (tiles <= 0 || tiles > 0x7fffffffL) return PV_ERR_INVALID;
  dim3 grid((unsigned)tiles), block(kThreads);
  if (pw) PV_LAUNCH((conv_igemm_kernel<T, BM, BN, WM_, WN_, true>), grid, block, 0, s, d);
  else PV_LAUNCH((conv_igemm_kernel<T, BM, BN, WM_, WN_, false>), grid, block, 0, s, d);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

template <typename T> int launch_conv(const pv_conv3d_desc& d, bool pw, hipStream_t s) {
  const int cout_p8 = pv_round_up(d.cout, 8);
  if (cout_p8 <= 32) return launch_cfg<T, 256, 32, 4, 1>(d, pw, s);
  if (cout_p8 <= 64) return launch_cfg<T, 128, 64, 4, 1>(d, pw, s);
  return launch_cfg<T, 128, 128, 2, 2>(d, pw, s);
}

}  // namespace

extern "C" int pv_conv3d_dwt_supported(const pv_conv3d_desc* d) {
  if (!d || d->B <= 0 || d->cout <= 0 || d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
  return pv_stem_dwt_supported(*d);
}

extern "C" int pv_conv3d_pw2_supported(const pv_conv3d_desc* d) {
  if (!d || d->B <= 0 || d->cin <= 0 || d->cout <= 0 || d->pw2_cout <= 0 || d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
  if (d->kt < 1 || d->kh < 1 || d->kw < 1 || d->kt * d->kh * d->kw == 1) return 0;   // a pointwise pair is the x2 / streaming kernels' work
  return pv_tapstream_try(*d, nullptr, true) == PV_OK;
}

extern "C" int pv_conv3d_x2_supported(const pv_conv3d_desc* d) {
  if (!d || d->B <= 0 || d->cout <= 0 || d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
  if (d->kt * d->kh * d->kw != 1 || d->st != 1 || d->sh != 1 || d->sw != 1 || d->pt || d->ph || d->pw) return 0;
  return pv_pwconv_x2_supported(*d);
}

extern "C" int pv_conv3d(const pv_conv3d_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const pv_conv3d_desc& d = *dp;
  if (!d.x || !d.w || !d.y) return PV_ERR_INVALID;
  if (d.B <= 0 || d.cin <= 0 || d.cout <= 0 || d.To <= 0 || d.Ho <= 0 || d.Wo <= 0) return PV_ERR_INVALID;
  const bool c4 = d.cin == 4 && d.ldx == 4 && d.dtype == PV_BF16;   // 4-channel-padded first layer (pv_stem.hip)
  if (!c4 && (d.cin % 8 || d.ldx % 8 || d.x_bs % 8)) return PV_ERR_INVALID;
  if (d.ldy % 8 || d.y_bs % 8 || (c4 && d.x_bs % 4)) return PV_ERR_INVALID;
  if (d.residual && (d.ldr % 8 || d.r_bs % 8)) return PV_ERR_INVALID;
  if (d.kt < 1 || d.kh < 1 || d.kw < 1 || d.st < 1 || d.sh < 1 || d.sw < 1) return PV_ERR_INVALID;
  const int taps = d.kt * d.kh * d.kw;
  if (taps > kMaxTaps || d.kt > 255 || d.kh > 255 || d.kw > 255) return PV_ERR_UNSUPPORTED;
  if (d.dil_t < 0 || d.dil_h < 0 || d.dil_w < 0) return PV_ERR_INVALID;
  const int et = (d.kt - 1) * (d.dil_t > 1 ? d.dil_t : 1) + 1, eh = (d.kh - 1) * (d.dil_h > 1 ? d.dil_h : 1) + 1,
            ew = (d.kw - 1) * (d.dil_w > 1 ? d.dil_w : 1) + 1;   // kernel extents
  if (et > 255 || eh > 255 || ew > 255) return PV_ERR_UNSUPPORTED;
  const bool dilated = et != d.kt || eh != d.kh || ew != d.kw;
  if (dilated && c4) return PV_ERR_UNSUPPORTED;
  // output geometry must be what nn.Conv3d would produce (RuntimeError in the reference)
  if ((d.Ti + 2 * d.pt - et) / d.st + 1 != d.To || (d.Hi + 2 * d.ph - eh) / d.sh + 1 != d.Ho ||
      (d.Wi + 2 * d.pw - ew) / d.sw + 1 != d.Wo)
    return PV_ERR_INVALID;
  const bool pw = taps == 1 && d.st == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.ph == 0 && d.pw == 0;
  if ((d.a_gate || d.a_act != PV_ACT_NONE) && !pw) return PV_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c4) return pv_stem_c4(d, s);
  if (d.dwt_w || d.pos_spatial || d.pos_temporal) return PV_ERR_UNSUPPORTED;   // first-layer layout only
  if (d.pw2_w) {   // a pointwise conv behind this one: the tap-streaming kernel only, no fallback
    if (pw || d.x2 || d.pw2_cout <= 0 || d.dtype != PV_BF16) return PV_ERR_UNSUPPORTED;
    return pv_tapstream_try(d, s);
  }
  if (d.x2) {   // second K operand: streaming pointwise kernel only, no fallback
    if (!pw || d.dtype != PV_BF16 || !pv_pwconv_x2_supported(d)) return PV_ERR_UNSUPPORTED;
    return pv_pwconv_stream_try(d, s);
  }
  if (d.dtype == PV_BF16) {
    // pv_tune "conv_route" (experiments): 1 = prefer the streaming kernel, 2 = prefer the LDS-DMA GEMM, 3 = generic only
    const int route = pv_tune("conv_route", 0);
    const int cout_p8 = pv_round_up(d.cout, 8);
    // Pointwise layers with a short reduction (X3D widths, SlowFast's fast pathway, MViT's first block)
    // are streaming problems: weights resident in LDS, activations straight into MFMA operands.
    // From ~192 input channels on the LDS-DMA GEMM wins on every measured shape, narrow outputs included
    // (profiles/r1: route sweeps on MViT-B and SlowFast-R50).  pv_tune "conv_small_cin" overrides the threshold.
    const int small_cin = pv_tune("conv_small_cin", 128);
    const bool small = d.cin <= small_cin;
    if (route != 3) {
      if (pw && route == 0 && pv_tune("head_rows", 1)) {     // a handful of rows (classification heads): K-parallel kernel
        const int r = pv_head_rows_try(d, s);
        if (r != PV_ERR_UNSUPPORTED) return r;
      }
      if (pw && (route == 1 || (route == 0 && small))) {
        const int r = pv_pwconv_stream_try(d, s);
        if (r != PV_ERR_UNSUPPORTED) return r;
      }
      // the 128-channel LDS-DMA tile wastes the matrix core on narrow outputs, and per-chunk tap
      // decoding dominates when a tap is only 8 channels wide (RGB stems): those stay on the
      // generic register-staged kernel
      // a 64-channel output fills half of the GEMM's 128-channel tile; with a short reduction (K <= 640) the
      // tap-streaming kernel (whole filter in LDS, operands straight into MFMA registers) takes those first
      if (!pw && cout_p8 <= 64 && (long)taps * d.cin <= 640 && pv_tune("tapstream_first", 1)) {
        const int r = pv_tapstream_try(d, s);
        if (r != PV_ERR_UNSUPPORTED) return r;
      }
      const bool gemm_ok = route == 2 || pw || (cout_p8 >= 64 && d.cin >= 64);
      int r = gemm_ok ? pv_gemm9_try(d, pw, s) : PV_ERR_UNSUPPORTED;     // 256 x 256 tiles, eight-phase loop: layers that fill the chip with them
      if (r != PV_ERR_UNSUPPORTED) return r;
      r = gemm_ok ? pv_gemm8_try(d, pw, s) : PV_ERR_UNSUPPORTED;         // large tiles, 4-stage ring: big layers
      if (r != PV_ERR_UNSUPPORTED) return r;
      r = gemm_ok ? pv_gemm_glds_try(d, pw, s) : PV_ERR_UNSUPPORTED;
      if (r != PV_ERR_UNSUPPORTED) return r;
      if (pw) {
        r = pv_pwconv_stream_try(d, s);
        if (r != PV_ERR_UNSUPPORTED) return r;
      }
    }
  }
  if (d.dtype == PV_BF16 && !pw) {   // narrow dense convs (SlowFast's fast pathway, ...): the tap-streaming kernel
    const int r = pv_tapstream_try(d, s);
    if (r != PV_ERR_UNSUPPORTED) return r;
  }
  if (d.dtype == PV_BF16) return launch_conv<bf16_t>(d, pw, s);
  if (d.dtype == PV_F32) return launch_conv<float>(d, pw, s);
  return PV_ERR_UNSUPPORTED;
}
