// Depthwise 3-D convolution on NDHWC (channels-last) activations: an HBM-bound stencil.
//
// A thread owns one 8-channel chunk (16 B of bf16) of NW consecutive output columns, so
// every global access is a 16-byte vector and each loaded input column is reused for up
// to KW taps x NW outputs in registers.  Lanes run fastest over the channel chunks of a
// voxel, i.e. a wave reads whole contiguous NDHWC voxel rows.  Filter taps live in LDS as
// fp32.  The folded-BN affine, the activation and (optionally) the per-block partial sums
// for the squeeze-excitation mean are fused into the epilogue; partial sums are written
// per block (no atomics) so the result is run-to-run deterministic.
#include "pv_common.h"

namespace {

constexpr int kThreads = 256;

// KW = 0 selects the fully runtime (one output per thread) variant.
template <typename T, int KW, int SW, int NW>
__global__ __launch_bounds__(kThreads) void dwconv_kernel(const pv_dwconv3d_desc d, int gpb,
                                                          int wgroups, int units_per_batch) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int c_p = pv_round_up(d.C, 8);
  const int CG = c_p / 8;
  const int w_p = d.w_mod > 0 ? pv_round_up(d.w_mod, 8) : c_p;
  const int taps = d.kt * d.kh * d.kw;
  float* s_w = s_mem;                 // [taps][w_p]
  float* s_red = s_mem + taps * w_p;  // [gpb][c_p]   (psum only)

  const int tid = threadIdx.x;
  for (int i = tid; i < taps * w_p; i += kThreads) s_w[i] = d.w[i];
  __syncthreads();

  const int cg = tid % CG;
  const int g = tid / CG;
  const int b = blockIdx.y;
  const long u = (long)blockIdx.x * gpb + g;
  const bool active = (g < gpb) && (u < units_per_batch);

  constexpr int NWc = (KW == 0) ? 1 : NW;
  float acc[NWc][8];
#pragma unroll
  for (int n = 0; n < NWc; ++n)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[n][j] = 0.f;

  int to = 0, ho = 0, wo0 = 0;
  const int c0 = cg * 8;
  const int wc0 = d.w_mod > 0 ? (c0 % d.w_mod) : c0;
  if (active) {
    const int wg = (int)(u % wgroups);
    const long r = u / wgroups;
    ho = (int)(r % d.Ho);
    to = (int)(r / d.Ho);
    wo0 = wg * NWc;
    const T* __restrict__ X = static_cast<const T*>(d.x) + (long)b * d.x_bs + (long)d.n_prefix * d.ldx + c0;
    const int t0 = to * d.st - d.pt, h0 = ho * d.sh - d.ph;
    if constexpr (KW == 0) {
      const int w0 = wo0 * d.sw - d.pw;
      for (int dt = 0; dt < d.kt; ++dt) {
        const int ti = t0 + dt;
        if ((unsigned)ti >= (unsigned)d.Ti) continue;
        for (int dh = 0; dh < d.kh; ++dh) {
          const int hi = h0 + dh;
          if ((unsigned)hi >= (unsigned)d.Hi) continue;
          const T* row = X + (long)(ti * d.Hi + hi) * d.Wi * d.ldx;
          for (int dw = 0; dw < d.kw; ++dw) {
            const int wi = w0 + dw;
            if ((unsigned)wi >= (unsigned)d.Wi) continue;
            Chunk8<T> c;
            c.load(row + (long)wi * d.ldx);
            float f[8];
            c.to_f32(f);
            const float* wp = s_w + ((dt * d.kh + dh) * d.kw + dw) * w_p + wc0;
            const float4 w0v = *reinterpret_cast<const float4*>(wp);
            const float4 w1v = *reinterpret_cast<const float4*>(wp + 4);
            acc[0][0] += f[0] * w0v.x; acc[0][1] += f[1] * w0v.y;
            acc[0][2] += f[2] * w0v.z; acc[0][3] += f[3] * w0v.w;
            acc[0][4] += f[4] * w1v.x; acc[0][5] += f[5] * w1v.y;
            acc[0][6] += f[6] * w1v.z; acc[0][7] += f[7] * w1v.w;
          }
        }
      }
    } else {
      constexpr int IW = (NW - 1) * SW + KW;  // input columns feeding NW outputs
      const int w0 = wo0 * SW - d.pw;
      for (int dt = 0; dt < d.kt; ++dt) {
        const int ti = t0 + dt;
        if ((unsigned)ti >= (unsigned)d.Ti) continue;
        for (int dh = 0; dh < d.kh; ++dh) {
          const int hi = h0 + dh;
          if ((unsigned)hi >= (unsigned)d.Hi) continue;
          const T* row = X + (long)(ti * d.Hi + hi) * d.Wi * d.ldx;
          Chunk8<T> in[IW];
#pragma unroll
          for (int i = 0; i < IW; ++i) {
            const int wi = w0 + i;
            if ((unsigned)wi < (unsigned)d.Wi) in[i].load(row + (long)wi * d.ldx);
            else in[i].zero();
          }
          const float* wp = s_w + ((dt * d.kh + dh) * KW) * w_p + wc0;
#pragma unroll
          for (int dw = 0; dw < KW; ++dw) {
            const float4 w0v = *reinterpret_cast<const float4*>(wp + dw * w_p);
            const float4 w1v = *reinterpret_cast<const float4*>(wp + dw * w_p + 4);
#pragma unroll
            for (int n = 0; n < NW; ++n) {
              float f[8];
              in[n * SW + dw].to_f32(f);
              acc[n][0] += f[0] * w0v.x; acc[n][1] += f[1] * w0v.y;
              acc[n][2] += f[2] * w0v.z; acc[n][3] += f[3] * w0v.w;
              acc[n][4] += f[4] * w1v.x; acc[n][5] += f[5] * w1v.y;
              acc[n][6] += f[6] * w1v.z; acc[n][7] += f[7] * w1v.w;
            }
          }
        }
      }
    }
  }

  // ---- epilogue ----
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = (c0 + j) < d.C;
    sc[j] = ok ? (d.scale ? d.scale[c0 + j] : 1.f) : 0.f;
    sh[j] = ok ? (d.shift ? d.shift[c0 + j] : 0.f) : 0.f;
  }
  float ps[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ps[j] = 0.f;
  if (active) {
    T* __restrict__ Y = static_cast<T*>(d.y) + (long)b * d.y_bs + (long)d.n_prefix * d.ldy + c0;
#pragma unroll
    for (int n = 0; n < NWc; ++n) {
      const int wo = wo0 + n;
      if (wo >= d.Wo) continue;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = acc[n][j] * sc[j] + sh[j];
        ps[j] += v[j];
        v[j] = pv_apply_act(v[j], d.act);
        if (c0 + j >= d.C) v[j] = 0.f;
      }
      Chunk8<T> o;
      o.from_f32(v);
      o.store(Y + ((long)(to * d.Ho + ho) * d.Wo + wo) * d.ldy);
    }
  }
  if (d.psum != nullptr) {
    if (g < gpb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s_red[g * c_p + c0 + j] = ps[j];
    }
    __syncthreads();
    for (int c = tid; c < c_p; c += kThreads) {
      float s = 0.f;
      for (int gg = 0; gg < gpb; ++gg) s += s_red[gg * c_p + c];
      d.psum[((long)b * gridDim.x + blockIdx.x) * c_p + c] = s;
    }
  }
}

// copy the n_prefix leading rows (cls token) of every batch item verbatim
template <typename T>
__global__ void dw_prefix_kernel(const pv_dwconv3d_desc d) {
  const int CG = pv_round_up(d.C, 8) / 8;
  const int total = d.B * d.n_prefix * CG;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int cg = id % CG;
  const int r = (id / CG) % d.n_prefix;
  const int b = id / (CG * d.n_prefix);
  Chunk8<T> c;
  c.load(static_cast<const T*>(d.x) + (long)b * d.x_bs + (long)r * d.ldx + cg * 8);
  c.store(static_cast<T*>(d.y) + (long)b * d.y_bs + (long)r * d.ldy + cg * 8);
}

struct DwGeom {
  int nw, gpb, wgroups, units_per_batch, nblk;
  size_t lds;
};

int variant_nw(const pv_dwconv3d_desc& d) {
  if (d.kw == 3 && (d.sw == 1 || d.sw == 2)) return 4;
  if (d.kw == 1 && d.sw == 1) return 4;
  return 1;
}

bool geom(const pv_dwconv3d_desc& d, DwGeom* g) {
  const int c_p = pv_round_up(d.C, 8);
  const int CG = c_p / 8;
  if (CG > kThreads) return false;
  g->nw = variant_nw(d);
  g->gpb = kThreads / CG;
  g->wgroups = (d.Wo + g->nw - 1) / g->nw;
  const long upb = (long)d.To * d.Ho * g->wgroups;
  if (upb > 0x7fffffffL) return false;
  g->units_per_batch = (int)upb;
  g->nblk = (int)pv_ceil_div(upb, g->gpb);
  const int w_p = d.w_mod > 0 ? pv_round_up(d.w_mod, 8) : c_p;
  g->lds = sizeof(float) * ((size_t)d.kt * d.kh * d.kw * w_p + (d.psum ? (size_t)g->gpb * c_p : 0));
  return true;
}

template <typename T, int KW, int SW, int NW>
int launch_variant(const pv_dwconv3d_desc& d, const DwGeom& g, hipStream_t s) {
  auto kern = dwconv_kernel<T, KW, SW, NW>;
  if (g.lds > 64 * 1024) {
    if (g.lds > 160 * 1024) return PV_ERR_UNSUPPORTED;
    PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
  }
  dim3 grid(g.nblk, d.B), block(kThreads);
  hipLaunchKernelGGL(kern, grid, block, g.lds, s, d, g.gpb, g.wgroups, g.units_per_batch);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

template <typename T> int launch_dw(const pv_dwconv3d_desc& d, const DwGeom& g, hipStream_t s) {
  if (d.n_prefix > 0) {
    const int total = d.B * d.n_prefix * (pv_round_up(d.C, 8) / 8);
    hipLaunchKernelGGL(dw_prefix_kernel<T>, dim3((unsigned)pv_ceil_div(total, kThreads)), dim3(kThreads), 0, s, d);
    PV_LAUNCH_CHECK();
  }
  if (d.kw == 3 && d.sw == 1) return launch_variant<T, 3, 1, 4>(d, g, s);
  if (d.kw == 3 && d.sw == 2) return launch_variant<T, 3, 2, 4>(d, g, s);
  if (d.kw == 1 && d.sw == 1) return launch_variant<T, 1, 1, 4>(d, g, s);
  return launch_variant<T, 0, 1, 1>(d, g, s);
}

int validate(const pv_dwconv3d_desc& d) {
  if (!d.x || !d.w || !d.y) return PV_ERR_INVALID;
  if (d.B <= 0 || d.C <= 0 || d.To <= 0 || d.Ho <= 0 || d.Wo <= 0) return PV_ERR_INVALID;
  if (d.ldx % 8 || d.ldy % 8 || d.x_bs % 8 || d.y_bs % 8) return PV_ERR_INVALID;
  if (d.kt < 1 || d.kh < 1 || d.kw < 1 || d.st < 1 || d.sh < 1 || d.sw < 1) return PV_ERR_INVALID;
  if (d.w_mod < 0 || (d.w_mod > 0 && d.w_mod % 8)) return PV_ERR_UNSUPPORTED;
  if ((d.Ti + 2 * d.pt - d.kt) / d.st + 1 != d.To || (d.Hi + 2 * d.ph - d.kh) / d.sh + 1 != d.Ho ||
      (d.Wi + 2 * d.pw - d.kw) / d.sw + 1 != d.Wo)
    return PV_ERR_INVALID;
  if (d.B > 65535) return PV_ERR_UNSUPPORTED;
  if (d.n_prefix < 0 || (d.n_prefix > 0 && d.psum)) return PV_ERR_INVALID;
  return PV_OK;
}

}  // namespace

extern "C" int pv_dwconv3d_psum_blocks(const pv_dwconv3d_desc* d) {
  if (!d) return PV_ERR_INVALID;
  DwGeom g;
  if (!geom(*d, &g)) return PV_ERR_UNSUPPORTED;
  return g.nblk;
}

extern "C" int pv_dwconv3d(const pv_dwconv3d_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const pv_dwconv3d_desc& d = *dp;
  const int v = validate(d);
  if (v != PV_OK) return v;
  DwGeom g;
  if (!geom(d, &g)) return PV_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d.dtype == PV_BF16) return launch_dw<bf16_t>(d, g, s);
  if (d.dtype == PV_F32) return launch_dw<float>(d, g, s);
  return PV_ERR_UNSUPPORTED;
}
