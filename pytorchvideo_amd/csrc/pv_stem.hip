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
#include <stdlib.h>
#include "pv_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPairs = 1024;   // K <= 8192

// JP = 2 (c4_wpair): an MFMA column is a PAIR of W-adjacent outputs whose filters -- the same taps shifted
// by the stride -- occupy rows [0,8) and [8,16) of one filter tile; for an 8-channel layer (SlowFast's
// fast stem) that doubles the useful rows per instruction and halves the operand loads per output.
template <int NT, int TM, int JP = 1>
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
  const int kw_eff = d.kw + (JP - 1) * d.sw;   // columns spanned by the JP windows of an MFMA column
  const int KWP = (kw_eff + 1) & ~1;
  const int PPR = KWP / 2;                 // voxel pairs per (dt,dh) row
  const int K = d.kt * d.kh * KWP * 4;     // packed K (multiple of 8)
  const int npairs = ksteps * 4;
  const int Wo2 = (d.Wo + JP - 1) / JP;    // MFMA columns per output row
  const long S_out = (long)d.To * d.Ho * Wo2;
  const long M = (long)d.B * S_out;
  const int n0 = blockIdx.y * NT * 16;
  const int w_rows = JP == 1 ? d.cout : JP * pv_round_up(d.cout, 8);   // filter rows the host packed

  {
    const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
    const int cpr = Kp / 8;
    for (int id = tid; id < NT * 16 * cpr; id += kThreads) {
      const int r = id / cpr, kc = id - r * cpr;
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (n0 + r < w_rows && kc * 8 < K) v = *reinterpret_cast<const bf16x8*>(Wt + (long)(n0 + r) * K + kc * 8);
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
    int vb[TM], vt[TM], vh[TM], vw[TM], vwo[TM], vpt[TM], vps[TM];
    long vy[TM];
    bool vok[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const long m = m_base + t * 16 + n16;
      vok[t] = m < M;
      const long mm = vok[t] ? m : 0;
      const long b = mm / S_out;
      const long sp = mm - b * S_out;
      const int to = (int)(sp / (d.Ho * Wo2));
      const int r2 = (int)(sp - (long)to * d.Ho * Wo2);
      const int ho = r2 / Wo2;
      const int wo = (r2 - ho * Wo2) * JP;     // first output of the column
      vb[t] = (int)b;
      vt[t] = to * d.st - d.pt;
      vh[t] = ho * d.sh - d.ph;
      vw[t] = wo * d.sw - d.pw;
      vwo[t] = wo;
      vpt[t] = to;
      vps[t] = d.pos_temporal ? ho * d.Wo + wo : (to * d.Ho + ho) * d.Wo + wo;   // row of the spatial / full table
      vy[t] = b * d.y_bs + (((long)to * d.Ho + ho) * d.Wo + wo) * d.ldy;
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
          const bool ok = rok && (unsigned)(wi + e) < (unsigned)d.Wi && dw + e < kw_eff;
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

    if constexpr (JP == 2) {
      // ---- epilogue, paired columns: rows 4q..4q+3 of the tile = channels 4(q&1)..+3 of output j = q>>1 ----
      const int co0 = 4 * (q & 1), j = q >> 1;
      float sc[4], sh[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const bool ok = co0 + jj < d.cout;
        sc[jj] = ok ? (d.scale ? d.scale[co0 + jj] : 1.f) : 0.f;
        sh[jj] = ok ? (d.shift ? d.shift[co0 + jj] : 0.f) : 0.f;
      }
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (!vok[t] || vwo[t] + j >= d.Wo) continue;
        float v[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) v[jj] = acc[0][t][jj] * sc[jj] + sh[jj];
        pv_apply_act_n<true>(v, d.act);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          if (co0 + jj >= d.cout) v[jj] = 0.f;
        const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(d.y) + vy[t] + (long)j * d.ldy + co0) = o;
      }
      continue;
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
        for (int j = 0; j < 4; ++j) v[j] = acc[a][t][j] * sc[j] + sh[j];
        pv_apply_act_n<true>(v, d.act);
        if (d.pos_spatial != nullptr) {   // position tables of the token stream (fp32 [rows][cout]); wave-uniform branch
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c0 + j < d.cout) {
              v[j] += d.pos_spatial[(long)vps[t] * d.cout + c0 + j];
              if (d.pos_temporal != nullptr) v[j] += d.pos_temporal[(long)vpt[t] * d.cout + c0 + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j >= d.cout) v[j] = 0.f;
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

template <int NT, int TM, int JP = 1> int launch_stem(const pv_conv3d_desc& d, int ksteps, hipStream_t s) {
  const long M = (long)d.B * d.To * d.Ho * ((d.Wo + JP - 1) / JP);
  const long ngroups = pv_ceil_div(M, 4 * TM * 16);
  const int nsplit = JP == 1 ? (int)pv_ceil_div(pv_round_up(d.cout, 8), NT * 16) : 1;
  if (ngroups > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  const size_t lds = (size_t)NT * 16 * (ksteps * 32 + 8) * 2 + (size_t)ksteps * 4 * 4;
  if (lds > 160 * 1024) return PV_ERR_UNSUPPORTED;
  auto kern = stem_c4_kernel<NT, TM, JP>;
  if (lds > 64 * 1024)
    PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long gx = ngroups < 4096 ? ngroups : 4096;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)nsplit), dim3(kThreads), lds, s, d, ksteps, (int)ngroups);
  PV_LAUNCH_CHECK();
  return PV_OK;
}


// ---------------------------------------------------------------------------------------
// X3D stem in one pass: conv_xy (1 x kh x kw from the 4-channel input, MFMA) followed, with nothing in
// between (models/x3d.py:66-88: Conv2plus1d(norm=None, activation=None)), by the depthwise temporal
// conv_t (DK x 1 x 1, stride 1, padding DK/2) + folded BN + activation.  A lane owns its voxels for
// the whole clip: it walks the T axis, evaluates one frame of conv_xy per step and scatters it into a
// register ring of DK partial output frames (frame t feeds outputs t-DK/2 .. t+DK/2), so the
// 24-channel intermediate -- 2x the bytes of the RGB input -- never exists in memory and the temporal
// conv needs no halo, no LDS and no second launch.  The ring index is a compile-time constant (the T
// loop is unrolled DK-fold), so "rotating" the ring costs no moves.
template <int V> struct IntC { static constexpr int value = V; };

template <int NT, int TM, int KS, int DK, int ACT>
__global__ __launch_bounds__(kThreads) void stem_c4_dwt_kernel(const pv_conv3d_desc d, int groups_per_clip) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int Kp = KS * 32;
  constexpr int WLD = Kp + 8;
  constexpr int HALF = DK / 2;
  bf16_t* w_s = reinterpret_cast<bf16_t*>(smem_raw);
  int* tab_s = reinterpret_cast<int*>(smem_raw + (size_t)NT * 16 * WLD * 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n16 = lane & 15, q = lane >> 4;
  const int KWP = (d.kw + 1) & ~1;
  const int PPR = KWP / 2;
  const int K = d.kh * KWP * 4;
  const int S_sp = d.Ho * d.Wo;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const int b = blockIdx.x / groups_per_clip, g = blockIdx.x - b * groups_per_clip;

  {
    const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
    constexpr int cpr = Kp / 8;
    for (int id = tid; id < NT * 16 * cpr; id += kThreads) {
      const int r = id / cpr, kc = id - r * cpr;
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (r < d.cout && kc * 8 < K) v = *reinterpret_cast<const bf16x8*>(Wt + (long)r * K + kc * 8);
      *reinterpret_cast<bf16x8*>(w_s + r * WLD + kc * 8) = v;
    }
    for (int pi = tid; pi < KS * 4; pi += kThreads) {
      const int dh = pi / PPR, pv = pi - dh * PPR;
      tab_s[pi] = dh < d.kh ? ((dh << 8) | ((2 * pv) << 16)) : -1;
    }
  }
  __syncthreads();

  constexpr unsigned kOOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(d.x), 0, (int)((unsigned)d.B * (unsigned)d.x_bs * 2u), 0x00020000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      d.y, 0, (int)((unsigned)d.B * (unsigned)d.y_bs * 2u), 0x00020000);
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

  // ---- this lane's voxels (fixed for the whole clip) and their load / store offsets inside a frame ----
  unsigned xo[TM][KS][2];   // byte offset of the two 8-byte halves of each k-step's fragment, or kOOB
  unsigned yo[TM];          // byte offset of (voxel, channel 4q) inside an output frame, or kOOB
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int sp = ((g * 4 + wave) * TM + t) * 16 + n16;
    const bool vok = sp < S_sp;
    const int ho = sp / d.Wo, wo = sp - ho * d.Wo;
    const int h0 = ho * d.sh - d.ph, w0 = wo * d.sw - d.pw;
    yo[t] = vok ? (unsigned)(sp * d.ldy + q * 4) * 2u : kOOB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int tp = tab_s[ks * 4 + q];
      const int dh = (tp >> 8) & 255, dw = tp >> 16;
      const int hi = h0 + dh, wi = w0 + dw;
      const bool rok = vok && tp >= 0 && (unsigned)hi < (unsigned)d.Hi;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool ok = rok && (unsigned)(wi + e) < (unsigned)d.Wi && dw + e < d.kw;
        xo[t][ks][e] = ok ? (unsigned)((hi * d.Wi + wi + e) * 4) * 2u : kOOB;
      }
    }
  }
  const unsigned x_frame = (unsigned)(d.Hi * d.Wi * 4) * 2u, y_frame = (unsigned)(S_sp * d.ldy) * 2u;
  const unsigned x_clip = (unsigned)b * (unsigned)d.x_bs * 2u, y_clip = (unsigned)b * (unsigned)d.y_bs * 2u;

  // ---- temporal taps, folded BN: channels a*16 + 4q .. +3 ----
  f32x4 wt[DK][NT], sc[NT], sh[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = a * 16 + q * 4 + j;
      const bool ok = c < d.cout;
      sc[a][j] = ok ? (d.scale ? d.scale[c] : 1.f) : 0.f;
      sh[a][j] = ok ? (d.shift ? d.shift[c] : 0.f) : 0.f;
#pragma unroll
      for (int k = 0; k < DK; ++k) wt[k][a][j] = ok ? d.dwt_w[k * cout_p8 + c] : 0.f;
    }

  u32x2 xr[TM][KS][2];
  auto load_frame = [&](int t) {
    const unsigned fb = x_clip + (unsigned)t * x_frame;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 2; ++e)
          xr[i][ks][e] = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(xo[i][ks][e] + fb), 0, 0);   // kOOB + fb stays out of range
  };

  f32x4 ring[DK][NT][TM];
#pragma unroll
  for (int k = 0; k < DK; ++k)
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int i = 0; i < TM; ++i) ring[k][a][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto step = [&](auto Rc, int t) {
    constexpr int R = decltype(Rc)::value;   // t mod DK
    if (t < d.Ti) {
      f32x4 h[NT][TM];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int i = 0; i < TM; ++i) h[a][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w_s + (a * 16 + n16) * WLD + ks * 32 + q * 8);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const u32x4 u = {xr[i][ks][0][0], xr[i][ks][0][1], xr[i][ks][1][0], xr[i][ks][1][1]};
            h[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, u), h[a][i], 0, 0, 0);
          }
        }
      if (t + 1 < d.Ti) load_frame(t + 1);   // next frame's loads fly under this frame's FMAs and stores
      // frame t feeds output t + HALF - k with tap k
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        const int slot = (R + HALF - k + DK) % DK;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int i = 0; i < TM; ++i) ring[slot][a][i] += wt[k][a] * h[a][i];
      }
    }
    // output t - HALF is complete after frame t; its ring slot is recycled for output t + HALF + 1
    // (also for the "outputs" before the first frame, which only exist as garbage in the ring)
    const int to = t - HALF;
    constexpr int slot = (R - HALF + DK) % DK;
    if (to >= 0 && to < d.To) {
      const unsigned fb = y_clip + (unsigned)to * y_frame;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const bool cok = a * 16 + q * 4 < cout_p8;   // padding channels up to the 8-multiple are written as zeros
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x4 v = ring[slot][a][i] * sc[a] + sh[a];
          if (ACT == PV_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          const unsigned off = cok ? yo[i] + a * 32u + fb : kOOB;   // kOOB + anything stays out of range
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, (int)off, 0, 0);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int i = 0; i < TM; ++i) ring[slot][a][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  load_frame(0);
  for (int t0 = 0; t0 < d.Ti + HALF; t0 += DK) {
    step(IntC<0>{}, t0);
    step(IntC<1>{}, t0 + 1);
    step(IntC<2>{}, t0 + 2);
    if (DK > 3) {
      step(IntC<3 % DK>{}, t0 + 3);
      step(IntC<4 % DK>{}, t0 + 4);
    }
  }
}

template <int NT, int TM, int KS, int DK> int launch_stem_dwt(const pv_conv3d_desc& d, hipStream_t s) {
  const long S_sp = (long)d.Ho * d.Wo;
  const long gpc = pv_ceil_div(S_sp, 4 * TM * 16);
  if (gpc * d.B > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  const size_t lds = (size_t)NT * 16 * (KS * 32 + 8) * 2 + (size_t)KS * 4 * 4;
  dim3 grid((unsigned)(gpc * d.B)), block(kThreads);
  if (d.act == PV_ACT_RELU) hipLaunchKernelGGL((stem_c4_dwt_kernel<NT, TM, KS, DK, PV_ACT_RELU>), grid, block, lds, s, d, (int)gpc);
  else hipLaunchKernelGGL((stem_c4_dwt_kernel<NT, TM, KS, DK, PV_ACT_NONE>), grid, block, lds, s, d, (int)gpc);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

}  // namespace

// geometry test for the fused temporal conv: first-layer 1 x kh x kw conv whose K fits two MFMA steps
// (3x3 taps), at most 32 output channels, 3 or 5 temporal taps
int pv_stem_dwt_supported(const pv_conv3d_desc& d) {
  if (d.dtype != PV_BF16 || d.cin != 4 || d.ldx != 4 || d.y_f32) return 0;
  if (d.kt != 1 || d.st != 1 || d.pt != 0 || d.To != d.Ti) return 0;
  if (d.dwt_k != 3 && d.dwt_k != 5) return 0;
  if (d.act != PV_ACT_NONE && d.act != PV_ACT_RELU) return 0;
  if (d.cout > 32 || d.kh * ((d.kw + 1) & ~1) * 4 > 64 || d.kh > 255 || d.kw > 255) return 0;
  return 1;
}

// pv_conv3d with cin == 4 (see include/pv_mi355x.h): bf16 input with 4 channels per voxel, weights
// packed [cout][kt][kh][round_up(kw,2)][4].
int pv_stem_c4(const pv_conv3d_desc& d, hipStream_t s) {
  if (d.dtype != PV_BF16 || d.cin != 4 || d.ldx != 4) return PV_ERR_UNSUPPORTED;
  if (d.residual || d.a_gate || d.a_act != PV_ACT_NONE) return PV_ERR_UNSUPPORTED;
  if ((long)d.B * d.x_bs > 0x3fffffffL) return PV_ERR_UNSUPPORTED;   // 31-bit byte offsets
  if (d.ldy % 4 || d.y_bs % 4) return PV_ERR_INVALID;
  const int jp = d.c4_wpair == 2 ? 2 : 1;
  if (d.c4_wpair < 0 || d.c4_wpair > 2 || (jp == 2 && (d.cout > 8 || d.y_f32 || d.dwt_w))) return PV_ERR_UNSUPPORTED;
  if ((d.pos_spatial || d.pos_temporal) && (!d.pos_spatial || !d.y_f32 || jp == 2 || d.dwt_w)) return PV_ERR_UNSUPPORTED;
  const int KWP = (d.kw + (jp - 1) * d.sw + 1) & ~1;
  const int K = d.kt * d.kh * KWP * 4;
  const int ksteps = (K + 31) / 32;
  if (d.dwt_w) {   // fused depthwise temporal conv (X3D stem)
    if (!pv_stem_dwt_supported(d)) return PV_ERR_UNSUPPORTED;
    if ((long)d.B * d.y_bs > 0x3fffffffL) return PV_ERR_UNSUPPORTED;
    return d.dwt_k == 5 ? launch_stem_dwt<2, 2, 2, 5>(d, s) : launch_stem_dwt<2, 2, 2, 3>(d, s);
  }
  if (ksteps * 4 > kMaxPairs || d.kt > 255 || d.kh > 255 || d.kw > 255) return PV_ERR_UNSUPPORTED;
  const int cout_p8 = pv_round_up(d.cout, 8);
  if (jp == 2) return launch_stem<1, 4, 2>(d, ksteps, s);
  if (cout_p8 <= 16) return launch_stem<1, 4>(d, ksteps, s);
  if (cout_p8 <= 32) return launch_stem<2, 4>(d, ksteps, s);
  const int mid = pv_tune("stem_mid", 4);
  if (cout_p8 <= 64) return mid == 2 ? launch_stem<2, 4>(d, ksteps, s) : launch_stem<4, 2>(d, ksteps, s);
  // wider outputs (MViT's 96 patch-embedding channels) split over blockIdx.y: 48 filter rows per workgroup keep
  // the LDS-resident filter slab (K up to 672) small enough for two workgroups per CU
  const int wide = pv_tune("stem_wide", 2);
  if (wide == 6) return launch_stem<6, 1>(d, ksteps, s);
  if (wide == 2) return launch_stem<2, 2>(d, ksteps, s);
  return launch_stem<3, 2>(d, ksteps, s);
}
