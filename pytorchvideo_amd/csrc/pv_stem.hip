// First-layer ("stem") convolutions on a 4-channel-padded input: SlowFast's (1,7,7) / (5,7,7)
// stems (models/slowfast.py:55-60 -> models/stem.py:80-107), X3D's 1x3x3 stem conv
// (models/x3d.py:66-88) and MViT's (3,7,7) patch embedding (models/stem.py:295-338).
//
// With Cin = 3 an 8-channel-padded implicit GEMM multiplies 5/8 zeros and decodes a tap per 16-byte
// chunk.  Here the input is NDHWC with the channel dim padded to 4 (8 bytes per voxel), so two
// W-adjacent voxels are one 16-byte chunk and a whole (dt,dh) row of the window -- kw voxels -- is one
// contiguous run of memory: K is enumerated as (dt, dh, voxel pair, voxel-in-pair, channel) and the
// MFMA B operand (lane (n = lane&15, q = lane>>4) holds k = 8q..8q+7 of voxel column n) is exactly
// two 8-byte loads at x[ti+dt][hi+dh][wi0 + 2*pair + {0,1}][0..3], straight from global memory into
// operand registers (no LDS round trip; the overlapping windows of neighbouring outputs hit in L1/L2).
// Image-border taps use buffer addressing: an out-of-range offset reads as zero.
// Weights are packed [cout][kt][kh][kw rounded up to even][4] (zeros in the padding) and stay in LDS
// for the life of the workgroup; the epilogue (folded BN or bias, activation) is in registers.
#include "pv_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPairs = 1024;   // K <= 8192

template <int NT, int TM>
__global__ __launch_bounds__(kThreads) void stem_c4_kernel(const pv_conv3d_desc d, int ksteps, int ngroups) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int Kp = ksteps * 32;
  const int WLD = Kp + 8;   // 16 B x odd: conflict-free ds_read_b128
  bf16_t* w_s = reinterpret_cast<bf16_t*>(smem_raw);
  int* tab_s = reinterpret_cast<int*>(smem_raw + (size_t)NT * 16 * WLD * 2);   // [ksteps*4] pair -> (dt, dh, dw)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n16 = lane & 15, q = lane >> 4;
  const int KWP = (d.kw + 1) & ~1;
  const int PPR = KWP / 2;                 // voxel pairs per (dt,dh) row
  const int K = d.kt * d.kh * KWP * 4;     // packed K (multiple of 8)
  const int npairs = ksteps * 4;
  const long S_out = (long)d.To * d.Ho * d.Wo;
  const long M = (long)d.B * S_out;
  const int n0 = blockIdx.y * NT * 16;

  {
    const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
    const int cpr = Kp / 8;
    for (int id = tid; id < NT * 16 * cpr; id += kThreads) {
      const int r = id / cpr, kc = id - r * cpr;
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (n0 + r < d.cout && kc * 8 < K) v = *reinterpret_cast<const bf16x8*>(Wt + (long)(n0 + r) * K + kc * 8);
      *reinterpret_cast<bf16x8*>(w_s + r * WLD + kc * 8) = v;
    }
    for (int pi = tid; pi < npairs; pi += kThreads) {
      const int row = pi / PPR, pv = pi - row * PPR;
      const int dt = row / d.kh, dh = row - dt * d.kh;
      tab_s[pi] = row < d.kt * d.kh ? (dt | (dh << 8) | ((2 * pv) << 16)) : -1;
    }
  }
  __syncthreads();

  // buffer descriptor over the whole input (31-bit byte offsets, checked on the host)
  constexpr unsigned kOOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(d.x), 0, (int)((unsigned)d.B * (unsigned)d.x_bs * 2u), 0x00020000);
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long m_base = ((long)g * 4 + wave) * (TM * 16);
    if (m_base >= M) continue;
    int vb[TM], vt[TM], vh[TM], vw[TM];
    long vy[TM];
    bool vok[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const long m = m_base + t * 16 + n16;
      vok[t] = m < M;
      const long mm = vok[t] ? m : 0;
      const long b = mm / S_out;
      const long sp = mm - b * S_out;
      const int to = (int)(sp / (d.Ho * d.Wo));
      const int r2 = (int)(sp - (long)to * d.Ho * d.Wo);
      const int ho = r2 / d.Wo;
      vb[t] = (int)b;
      vt[t] = to * d.st - d.pt;
      vh[t] = ho * d.sh - d.ph;
      vw[t] = (r2 - ho * d.Wo) * d.sw - d.pw;
      vy[t] = b * d.y_bs + sp * d.ldy;
    }
    auto load_step = [&](u32x2 (&dst)[TM][2], int ks) {
      const int tp = tab_s[ks * 4 + q];
      const int dt = tp & 255, dh = (tp >> 8) & 255, dw = tp >> 16;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int ti = vt[t] + dt, hi = vh[t] + dh, wi = vw[t] + dw;
        const bool rok = vok[t] && tp >= 0 && (unsigned)ti < (unsigned)d.Ti && (unsigned)hi < (unsigned)d.Hi;
        const unsigned base = (unsigned)vb[t] * (unsigned)d.x_bs + ((unsigned)(ti * d.Hi + hi) * d.Wi + wi) * 4u;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bool ok = rok && (unsigned)(wi + e) < (unsigned)d.Wi && dw + e < d.kw;
          dst[t][e] = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(ok ? (base + 4u * e) * 2u : kOOB), 0, 0);
        }
      }
    };

    f32x4 acc[NT][TM];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int t = 0; t < TM; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x2 xa[TM][2], xb[TM][2];
    auto mma = [&](const u32x2 (&src)[TM][2], int ks) {
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w_s + (a * 16 + n16) * WLD + ks * 32 + q * 8);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 u = {src[t][0][0], src[t][0][1], src[t][1][0], src[t][1][1]};
          acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, u), acc[a][t], 0, 0, 0);
        }
      }
    };
    load_step(xa, 0);
    for (int ks = 0; ks < ksteps; ks += 2) {
      if (ks + 1 < ksteps) load_step(xb, ks + 1);
      mma(xa, ks);
      if (ks + 1 < ksteps) {
        if (ks + 2 < ksteps) load_step(xa, ks + 2);
        mma(xb, ks + 1);
      }
    }

    // ---- epilogue: lane holds channels n0 + a*16 + q*4 .. +3 of voxel n16 ----
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int c0 = n0 + a * 16 + q * 4;
      if (c0 >= pv_round_up(d.cout, 8)) continue;   // padding channels up to the 8-multiple are written as zeros
      float sc[4], sh[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = c0 + j < d.cout;
        sc[j] = ok ? (d.scale ? d.scale[c0 + j] : 1.f) : 0.f;
        sh[j] = ok ? (d.shift ? d.shift[c0 + j] : 0.f) : 0.f;
      }
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (!vok[t]) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = pv_apply_act(acc[a][t][j] * sc[j] + sh[j], d.act);
          if (c0 + j >= d.cout) v[j] = 0.f;
        }
        if (d.y_f32) {
          *reinterpret_cast<f32x4*>(static_cast<float*>(d.y) + vy[t] + c0) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(d.y) + vy[t] + c0) = o;
        }
      }
    }
  }
}

template <int NT, int TM> int launch_stem(const pv_conv3d_desc& d, int ksteps, hipStream_t s) {
  const long M = (long)d.B * d.To * d.Ho * d.Wo;
  const long ngroups = pv_ceil_div(M, 4 * TM * 16);
  const int nsplit = (int)pv_ceil_div(pv_round_up(d.cout, 8), NT * 16);
  if (ngroups > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  const size_t lds = (size_t)NT * 16 * (ksteps * 32 + 8) * 2 + (size_t)ksteps * 4 * 4;
  if (lds > 160 * 1024) return PV_ERR_UNSUPPORTED;
  auto kern = stem_c4_kernel<NT, TM>;
  if (lds > 64 * 1024)
    PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long gx = ngroups < 4096 ? ngroups : 4096;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)nsplit), dim3(kThreads), lds, s, d, ksteps, (int)ngroups);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

}  // namespace

// pv_conv3d with cin == 4 (see include/pv_mi355x.h): bf16 input with 4 channels per voxel, weights
// packed [cout][kt][kh][round_up(kw,2)][4].
int pv_stem_c4(const pv_conv3d_desc& d, hipStream_t s) {
  if (d.dtype != PV_BF16 || d.cin != 4 || d.ldx != 4) return PV_ERR_UNSUPPORTED;
  if (d.residual || d.a_gate || d.a_act != PV_ACT_NONE) return PV_ERR_UNSUPPORTED;
  if ((long)d.B * d.x_bs > 0x3fffffffL) return PV_ERR_UNSUPPORTED;   // 31-bit byte offsets
  if (d.ldy % 4 || d.y_bs % 4) return PV_ERR_INVALID;
  const int KWP = (d.kw + 1) & ~1;
  const int K = d.kt * d.kh * KWP * 4;
  const int ksteps = (K + 31) / 32;
  if (ksteps * 4 > kMaxPairs || d.kt > 255 || d.kh > 255 || d.kw > 255) return PV_ERR_UNSUPPORTED;
  const int cout_p8 = pv_round_up(d.cout, 8);
  if (cout_p8 <= 16) return launch_stem<1, 4>(d, ksteps, s);
  if (cout_p8 <= 32) return launch_stem<2, 4>(d, ksteps, s);
  if (cout_p8 <= 64) return launch_stem<4, 2>(d, ksteps, s);
  return launch_stem<6, 1>(d, ksteps, s);   // e.g. 96 patch-embedding channels; wider outputs split over blockIdx.y
}
