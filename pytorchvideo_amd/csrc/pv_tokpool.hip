// Fused MViT attention pooling: depthwise conv over the token grid + cls pass-through + LayerNorm
// over head_dim, for q / k / v of one MultiScaleAttention in ONE launch
// (reference: _AttentionPool.forward, pytorchvideo/layers/attention.py:162-212).
//
// The pooled tensors are small (785 .. 12545 tokens) and the unfused chain (conv, cls copy, norm, per
// tensor) is six latency-bound launches per block.  Here a 16-lane group owns one (output token,
// head): lane l carries channels 8l..8l+7 of the head (head_dim <= 128), gathers its window straight
// from the (L2-resident) token tensor, and the LayerNorm statistics are two 16-lane butterfly
// reductions -- no intermediate tensor is written.  blockIdx.z selects the tensor (q, k or v).
#include "pv_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxTaps = 64;

template <typename T, bool K333>
__global__ __launch_bounds__(kThreads) void token_pool_kernel(const pv_token_pool_desc d) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];   // [taps][hd_p] filter of this tensor
  const int z = blockIdx.z;
  const int hd = d.head_dim, hd_p = pv_round_up(hd, 8);
  const int taps = d.kt * d.kh * d.kw;
  // head_dim % 8 == 0 (checked on the host): the filter is a dense [taps][hd] fp32 block -> 16-byte copies
  for (int i = threadIdx.x; i < taps * hd / 4; i += kThreads)
    reinterpret_cast<f32x4*>(s_w)[i] = reinterpret_cast<const f32x4*>(d.w[z])[i];
  __syncthreads();

  const int lane16 = threadIdx.x & 15;
  const int grp = threadIdx.x >> 4;
  const int To = d.To[z], Ho = d.Ho[z], Wo = d.Wo[z];
  const int n_out = d.n_prefix + To * Ho * Wo;
  const int b = blockIdx.y;
  // grid-stride over (token, head) pairs: the filter is staged once per resident workgroup
  for (long unit = (long)blockIdx.x * (kThreads / 16) + grp; unit - grp < (long)n_out * d.heads;
       unit += (long)gridDim.x * (kThreads / 16)) {
  const bool unit_ok = unit < (long)n_out * d.heads;
  const int tok = unit_ok ? (int)(unit / d.heads) : 0;
  const int h = unit_ok ? (int)(unit - (long)tok * d.heads) : 0;
  const bool act = unit_ok && lane16 * 8 < hd;
  const T* __restrict__ X = static_cast<const T*>(d.x[z]) + (long)b * d.x_bs[z] + h * hd + lane16 * 8;

  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (act) {
    if (tok < d.n_prefix) {   // cls token: passed through the pool, normalised below
      Chunk8<T> c;
      c.load(X + (long)tok * d.ldx[z]);
      c.to_f32(acc);
    } else {
      const int g = tok - d.n_prefix;
      const int to = g / (Ho * Wo);
      const int r2 = g - to * Ho * Wo;
      const int ho = r2 / Wo, wo = r2 - ho * Wo;
      const int t0 = to * d.st[z] - d.kt / 2, h0 = ho * d.sh[z] - d.kh / 2, w0 = wo * d.sw[z] - d.kw / 2;
      if (K333) {
        // 3x3x3 window: nine independent 16-byte loads in flight per temporal tap (out-of-grid taps
        // read the centre voxel and are multiplied by zero weights -- no divergent branches)
        const long centre = (long)d.n_prefix + (long)((to * d.st[z]) * d.Hi + ho * d.sh[z]) * d.Wi + wo * d.sw[z];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
          const int ti = t0 + dt;
          const bool tok_t = (unsigned)ti < (unsigned)d.Ti;
          Chunk8<T> c[9];
          bool okv[9];
#pragma unroll
          for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
              const int hi = h0 + dh, wi = w0 + dw;
              const bool ok = tok_t && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
              okv[dh * 3 + dw] = ok;
              const long vox = ok ? (long)d.n_prefix + (long)(ti * d.Hi + hi) * d.Wi + wi : centre;
              c[dh * 3 + dw].load(X + vox * d.ldx[z]);
            }
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            float f[8];
            c[i].to_f32(f);
            const float* wp = s_w + (dt * 9 + i) * hd_p + lane16 * 8;
            const f32x4 w0v = *reinterpret_cast<const f32x4*>(wp), w1v = *reinterpret_cast<const f32x4*>(wp + 4);
            const float m = okv[i] ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[j] += f[j] * (w0v[j] * m);
              acc[4 + j] += f[4 + j] * (w1v[j] * m);
            }
          }
        }
      } else {
      for (int dt = 0; dt < d.kt; ++dt) {
        const int ti = t0 + dt;
        if ((unsigned)ti >= (unsigned)d.Ti) continue;
        for (int dh = 0; dh < d.kh; ++dh) {
          const int hi = h0 + dh;
          if ((unsigned)hi >= (unsigned)d.Hi) continue;
          for (int dw = 0; dw < d.kw; ++dw) {
            const int wi = w0 + dw;
            if ((unsigned)wi >= (unsigned)d.Wi) continue;
            Chunk8<T> c;
            c.load(X + ((long)d.n_prefix + (long)(ti * d.Hi + hi) * d.Wi + wi) * d.ldx[z]);
            float f[8];
            c.to_f32(f);
            const float* wp = s_w + ((dt * d.kh + dh) * d.kw + dw) * hd_p + lane16 * 8;
            const f32x4 w0v = *reinterpret_cast<const f32x4*>(wp), w1v = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[j] += f[j] * w0v[j];
              acc[4 + j] += f[4 + j] * w1v[j];
            }
          }
        }
      }
      }
    }
  }
  // LayerNorm over the head's channels: 16-lane butterflies (inactive lanes contribute zeros)
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += (lane16 * 8 + j < hd) ? acc[j] : 0.f;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)hd;
  float v = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float dlt = (lane16 * 8 + j < hd) ? acc[j] - mean : 0.f;
    v += dlt * dlt;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const float rstd = rsqrtf(v / (float)hd + d.eps);
  if (act) {
    const bool has_norm = d.gamma[z] != nullptr || d.beta[z] != nullptr;
    float o8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane16 * 8 + j;
      float val = acc[j];
      if (has_norm) val = (acc[j] - mean) * rstd * (d.gamma[z] ? d.gamma[z][c < hd ? c : 0] : 1.f) +
                          (d.beta[z] ? d.beta[z][c < hd ? c : 0] : 0.f);
      o8[j] = c < hd ? val : 0.f;
    }
    Chunk8<T> oc;
    oc.from_f32(o8);
    oc.store(static_cast<T*>(d.y[z]) + (long)b * d.y_bs[z] + (long)tok * d.ldy[z] + h * hd + lane16 * 8);
  }
  }
}

}  // namespace

extern "C" int pv_token_pool(const pv_token_pool_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const pv_token_pool_desc& d = *dp;
  if (d.n < 1 || d.n > 3 || d.B <= 0 || d.heads <= 0 || d.head_dim <= 0 || d.Ti <= 0 || d.Hi <= 0 || d.Wi <= 0)
    return PV_ERR_INVALID;
  if (d.kt < 1 || d.kh < 1 || d.kw < 1 || d.n_prefix < 0) return PV_ERR_INVALID;
  if (d.head_dim % 8 || d.head_dim > 128 || d.kt * d.kh * d.kw > kMaxTaps || d.B > 65535) return PV_ERR_UNSUPPORTED;
  long max_units = 0;
  for (int i = 0; i < d.n; ++i) {
    if (!d.x[i] || !d.y[i] || !d.w[i]) return PV_ERR_INVALID;
    if (d.st[i] < 1 || d.sh[i] < 1 || d.sw[i] < 1) return PV_ERR_INVALID;
    if (d.ldx[i] % 8 || d.ldy[i] % 8 || d.x_bs[i] % 8 || d.y_bs[i] % 8) return PV_ERR_INVALID;
    // nn.Conv3d output size with padding = kernel // 2 (RuntimeError in the reference otherwise)
    if ((d.Ti + 2 * (d.kt / 2) - d.kt) / d.st[i] + 1 != d.To[i] || (d.Hi + 2 * (d.kh / 2) - d.kh) / d.sh[i] + 1 != d.Ho[i] ||
        (d.Wi + 2 * (d.kw / 2) - d.kw) / d.sw[i] + 1 != d.Wo[i])
      return PV_ERR_INVALID;
    const long units = ((long)d.n_prefix + (long)d.To[i] * d.Ho[i] * d.Wo[i]) * d.heads;
    if (units > max_units) max_units = units;
  }
  long gx = pv_ceil_div(max_units, kThreads / 16);
  const long resident = pv_ceil_div(256 * 8, (long)d.B * d.n);   // ~8 workgroups per CU over the whole grid
  if (gx > resident) gx = resident;
  dim3 grid((unsigned)gx, (unsigned)d.B, (unsigned)d.n), block(kThreads);
  const size_t lds = sizeof(float) * (size_t)d.kt * d.kh * d.kw * pv_round_up(d.head_dim, 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool k333 = d.kt == 3 && d.kh == 3 && d.kw == 3;
  if (d.dtype == PV_BF16 && k333) PV_LAUNCH((token_pool_kernel<bf16_t, true>), grid, block, lds, s, d);
  else if (d.dtype == PV_BF16) PV_LAUNCH((token_pool_kernel<bf16_t, false>), grid, block, lds, s, d);
  else if (d.dtype == PV_F32 && k333) PV_LAUNCH((token_pool_kernel<float, true>), grid, block, lds, s, d);
  else if (d.dtype == PV_F32) PV_LAUNCH((token_pool_kernel<float, false>), grid, block, lds, s, d);
  else return PV_ERR_UNSUPPORTED;
  PV_LAUNCH_CHECK();
  return PV_OK;
}
