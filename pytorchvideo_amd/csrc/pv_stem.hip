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
  PV_LAUNCH(kern, dim3((unsigned)gx, (unsigned)nsplit), dim3(kThreads), lds, s, d, ksteps, (int)ngroups);
  pv_note_kernel("stem_c4_kernel");   // (launched through a function pointer: PV_LAUNCH saw only the variable)
  PV_LAUNCH_CHECK();
  return PV_OK;
}


// ---------------------------------------------------------------------------------------
// 7 x 7 / stride-2 stems with the input tile in LDS: SlowFast's slow (1,7,7) and fast (5,7,7) stems
// (models/slowfast.py:209-229 -> models/stem.py:80-107).
//
// The kernel above gathers every MFMA B operand straight from global memory: 2.2 KB of window per output voxel of
// the (5,7,7) stem go through L1 / the texture addresser (PMC: 554 MB of HBM traffic against 369 MB algorithmic,
// 18.6 k VALU instructions per wave, most of them address arithmetic) -- 0.79 ms for 99 GFLOP.  Here a workgroup
// owns an 8 x 32 output tile of one clip and walks the T axis:
//   * each input frame's halo tile (21 x 70 voxels x 8 bytes) is staged into LDS ONCE and stays in a ring of kt + 1
//     frames: the 5 temporal taps re-read LDS, not memory; the next frame's loads are in flight during the MFMAs;
//   * with kw = 7 padded to 8, one (dt, dh) row of the window is exactly one K step of 32: lane (n, q) reads ONE
//     aligned 16-byte chunk (voxel pair q of output column n) with a compile-time LDS offset -- no tap table, no
//     bounds tests (the halo is zero-filled at staging time);
//   * the filter is held in LDS as ready-made A fragments (lane-linear 1 KB each: conflict-free ds_read_b128);
//   * <= 8 output channels (fast stem): the 16 MFMA rows are 8 channels x TWO vertically adjacent outputs -- input
//     row r of a row pair feeds output row h with dh = r and output row h+1 with dh = r - 2 -- so no half of the
//     instruction is idle; 64 channels (slow stem): 4 channel tiles share every B fragment.
template <int RP, int NT, int TM, int KT>
__global__ __launch_bounds__(kThreads) void stem7_kernel(const pv_conv3d_desc d, int tiles_h, int tiles_w, int wpitch) {
  constexpr int TH = 8, TW = TM * 16;
  constexpr int IH = (TH - 1) * 2 + 7;       // 21 input rows
  constexpr int IW = (TW - 1) * 2 + 8;       // 70 input columns (even: 16-byte chunks = voxel pairs)
  constexpr int NVOX = IH * IW;
  constexpr int FRAME = NVOX * 4;            // bf16 elements per staged frame
  constexpr int NLD = (NVOX + kThreads - 1) / kThreads;
  constexpr int JR = RP == 2 ? 9 : 7;        // A fragments per temporal tap (row offsets of a row pair / dh)
  constexpr int NA = RP == 2 ? 1 : NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // KT > 0 (row-pair form only): the temporal extent is a compile-time constant and every wave keeps all KT * 9 A
  // fragments in registers (180 VGPRs for the (5,7,7) stem) -- the LDS then holds nothing but the frame ring
  // (70 KB: two workgroups per CU, one's barrier waits hidden behind the other's MFMAs) and A costs no LDS bandwidth.
  constexpr bool kAReg = RP == 2 && KT > 0;
  bf16_t* wf_s = reinterpret_cast<bf16_t*>(smem_raw);               // [kt][JR][NA][64 lanes][8]
  const int nfrag = kAReg ? 0 : d.kt * JR * NA;
  bf16_t* ring = wf_s + (size_t)nfrag * 512;                         // [kt + 1][IH][IW][4]
  const int nslot = d.kt + 1;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int tw = blockIdx.x % tiles_w;
  const int th = (blockIdx.x / tiles_w) % tiles_h;
  const int b = blockIdx.x / (tiles_w * tiles_h);
  const int ho0 = th * TH, wo0 = tw * TW;
  const int hi0 = ho0 * 2 - 3, wi0 = wo0 * 2 - 3;
  const int cout_p8 = pv_round_up(d.cout, 8);

  // ---- filter -> A fragments ----
  {
    const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
    const long Kh = (long)d.kt * 7 * wpitch;     // elements per filter row as the host packed it
    for (int id = tid; id < nfrag * 64; id += kThreads) {
      const int f = id >> 6, l = id & 63;
      const int r = l & 15, qq = l >> 4;
      const int a = f % NA, j = (f / NA) % JR, dt = f / (NA * JR);
      int ch, dh;
      if (RP == 2) { ch = r & 7; dh = j - 2 * (r >> 3); }
      else { ch = a * 16 + r; dh = j; }
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ch < d.cout && dh >= 0 && dh < 7)
        v = *reinterpret_cast<const bf16x8*>(Wt + (long)ch * Kh + (long)(dt * 7 + dh) * wpitch + qq * 8);
      *reinterpret_cast<bf16x8*>(wf_s + (size_t)id * 8) = v;
    }
  }

  // ---- staging geometry: thread -> voxels of the halo tile (fixed for the whole clip) ----
  constexpr unsigned kOOB = 0x80000000u;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const bf16_t* X = static_cast<const bf16_t*>(d.x) + (long)b * d.x_bs;
  const unsigned frame_bytes = (unsigned)(d.Hi * d.Wi) * 8u;
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)(frame_bytes * (unsigned)d.Ti), 0x00020000);
  unsigned g_off[NLD];
#pragma unroll
  for (int n = 0; n < NLD; ++n) {
    const int i = tid + n * kThreads;
    const int ir = i / IW, ic = i - ir * IW;
    const int hi = hi0 + ir, wi = wi0 + ic;
    const bool ok = i < NVOX && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
    g_off[n] = ok ? (unsigned)(hi * d.Wi + wi) * 8u : kOOB;
  }
  u32x2 st[NLD];
  auto load_frame = [&](int ti) {
    const bool live = (unsigned)ti < (unsigned)d.Ti;   // frames outside the clip are the conv's temporal zero padding
#pragma unroll
    for (int n = 0; n < NLD; ++n)
      st[n] = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)((live && g_off[n] != kOOB) ? g_off[n] + (unsigned)ti * frame_bytes : kOOB), 0, 0);
  };
  auto slot_of = [&](int ti) { return (ti + 8 * nslot) % nslot; };
  auto store_frame = [&](int ti) {
    bf16_t* dst = ring + (size_t)slot_of(ti) * FRAME;
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      const int i = tid + n * kThreads;
      if (i < NVOX) *reinterpret_cast<u32x2*>(dst + (size_t)i * 4) = st[n];
    }
  };

  // ---- epilogue constants ----
  float sc[RP == 2 ? 1 : NT][4], sh[RP == 2 ? 1 : NT][4];
#pragma unroll
  for (int a = 0; a < (RP == 2 ? 1 : NT); ++a)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int c = RP == 2 ? 4 * (q & 1) + jj : a * 16 + 4 * q + jj;
      const bool ok = c < d.cout;
      sc[a][jj] = ok ? (d.scale ? d.scale[c] : 1.f) : 0.f;
      sh[a][jj] = ok ? (d.shift ? d.shift[c] : 0.f) : 0.f;
    }
  bf16_t* Y = static_cast<bf16_t*>(d.y) + (long)b * d.y_bs;
  const unsigned y_frame_bytes = (unsigned)(d.Ho * d.Wo * d.ldy) * 2u;
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)(y_frame_bytes * (unsigned)d.To), 0x00020000);

  // prologue: the first kt frames (all of them in flight at once when kt is a compile-time constant)
  if constexpr (KT > 0) {
    u32x2 pf[KT][NLD];
#pragma unroll
    for (int dt = 0; dt < KT; ++dt) {
      load_frame(dt - d.pt);
#pragma unroll
      for (int n = 0; n < NLD; ++n) pf[dt][n] = st[n];
    }
#pragma unroll
    for (int dt = 0; dt < KT; ++dt) {
#pragma unroll
      for (int n = 0; n < NLD; ++n) st[n] = pf[dt][n];
      store_frame(dt - d.pt);
    }
  } else {
    for (int dt = 0; dt < d.kt; ++dt) {
      load_frame(dt - d.pt);
      store_frame(dt - d.pt);
    }
  }
  bf16x8 areg[kAReg ? KT * JR : 1];
  if constexpr (kAReg) {
    const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
    const long Kh = (long)KT * 7 * wpitch;
    const int ch = n16 & 7;
#pragma unroll
    for (int f = 0; f < KT * JR; ++f) {
      const int j = f % JR, dt = f / JR;
      const int dh = j - 2 * (n16 >> 3);
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ch < d.cout && dh >= 0 && dh < 7)
        v = *reinterpret_cast<const bf16x8*>(Wt + (long)ch * Kh + (long)(dt * 7 + dh) * wpitch + q * 8);
      areg[f] = v;
    }
  }
  __syncthreads();

  for (int to = 0; to < d.To; ++to) {
    const int ti_new = to + 1 - d.pt + d.kt - 1;      // the frame output to+1 needs on top of this one's
    load_frame(to + 1 < d.To ? ti_new : -1);
    if constexpr (RP == 2) {
      f32x4 acc[TM];
#pragma unroll
      for (int t = 0; t < TM; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < (KT > 0 ? KT : d.kt); ++dt) {
        const bf16_t* fb = ring + (size_t)slot_of(to - d.pt + dt) * FRAME + ((4 * wave) * IW + 2 * n16 + 2 * q) * 4;
        const bf16_t* wp = wf_s + (size_t)(dt * JR) * 512 + lane * 8;
#pragma unroll
        for (int j = 0; j < JR; ++j) {
          bf16x8 af;
          if constexpr (kAReg) af = areg[dt * JR + j];
          else af = *reinterpret_cast<const bf16x8*>(wp + j * 512);
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            const bf16x8 bfv = *reinterpret_cast<const bf16x8*>(fb + (j * IW + 32 * t) * 4);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv, acc[t], 0, 0, 0);
          }
        }
      }
      store_frame(ti_new);
      // rows 4q..4q+3 of the tile = channels 4(q&1)..+3 of output row 2*wave + (q >> 1)
      const int ho = ho0 + 2 * wave + (q >> 1), co0 = 4 * (q & 1);
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int wo = wo0 + t * 16 + n16;
        float v[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) v[jj] = acc[t][jj] * sc[0][jj] + sh[0][jj];
        pv_apply_act_n<true>(v, d.act);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          if (co0 + jj >= d.cout) v[jj] = 0.f;
        const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        const bool ok = ho < d.Ho && wo < d.Wo && co0 < cout_p8;
        const unsigned off = ok ? (unsigned)to * y_frame_bytes + (unsigned)((ho * d.Wo + wo) * d.ldy + co0) * 2u : kOOB;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, (int)off, 0, 0);
      }
    } else {
      f32x4 acc[NT][2][TM];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int t = 0; t < TM; ++t) acc[a][rr][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int dt = 0; dt < d.kt; ++dt) {
        const bf16_t* fb = ring + (size_t)slot_of(to - d.pt + dt) * FRAME + (2 * n16 + 2 * q) * 4;
        const bf16_t* wp = wf_s + (size_t)(dt * JR * NT) * 512 + lane * 8;
#pragma unroll
        for (int dh = 0; dh < 7; ++dh) {
          bf16x8 bfv[2][TM];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int t = 0; t < TM; ++t)
              bfv[rr][t] = *reinterpret_cast<const bf16x8*>(fb + ((2 * (wave + 4 * rr) + dh) * IW + 32 * t) * 4);
#pragma unroll
          for (int a = 0; a < NT; ++a) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(wp + (dh * NT + a) * 512);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
              for (int t = 0; t < TM; ++t)
                acc[a][rr][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv[rr][t], acc[a][rr][t], 0, 0, 0);
          }
        }
      }
      store_frame(ti_new);
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int c0 = a * 16 + 4 * q;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int ho = ho0 + wave + 4 * rr;
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            const int wo = wo0 + t * 16 + n16;
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = acc[a][rr][t][jj] * sc[a][jj] + sh[a][jj];
            pv_apply_act_n<true>(v, d.act);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              if (c0 + jj >= d.cout) v[jj] = 0.f;
            const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            const bool ok = ho < d.Ho && wo < d.Wo && c0 < cout_p8;
            const unsigned off = ok ? (unsigned)to * y_frame_bytes + (unsigned)((ho * d.Wo + wo) * d.ldy + c0) * 2u : kOOB;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ry, (int)off, 0, 0);
          }
        }
      }
    }
    __syncthreads();   // frame ti_new is visible; nobody reads the slot it replaced any more
  }
}

// ---------------------------------------------------------------------------------------
// MViT patch embedding: (kt,7,7) conv, stride (st,4,4), 3 -> 96 channels, fp32 tokens + position tables
// (models/stem.py:289-338, models/vision_transformers.py:357-368).
//
// 96 x 672 filter values are 126 KB as MFMA fragments: with them in LDS no input tile fits beside them twice, and the
// generic kernel above re-gathers every window through L1 (PMC: 736 MB per launch against 257 MB algorithmic).  Here the
// FILTER never enters LDS: wave w owns output channels 16w..16w+15 and streams its kt * 7 A fragments (one per (dt, dh)
// window row: kw = 7 padded to 8 voxels x 4 channels = one K = 32 step) from L2 into registers, one temporal tap ahead; LDS holds only the input halo tile of the kt frames (8 x 16 outputs: 35 x 68 voxels x 8 B per frame),
// staged once and read by every wave -- the B operand of output column n, voxel pair q is ONE aligned 16-byte
// ds_read at a compile-time offset (the halo is zero-filled at staging time: no tap table, no bounds tests).
// 57 KB of LDS and ~170 VGPRs: two workgroups per CU, one loading while the other multiplies.
template <int KT, int NW>
__global__ __launch_bounds__(NW * 64, 3) void stem_pe_kernel(const pv_conv3d_desc d, int tiles_h, int tiles_w, int wpitch) {
  constexpr int TH = 8, TW = 16, S = 4;
  constexpr int IH = (TH - 1) * S + 7;       // 35 input rows
  constexpr int IW = (TW - 1) * S + 8;       // 68 input columns = 34 voxel pairs
  constexpr int HP = IW / 4;                 // 17 even (or odd) pairs per row
  constexpr int NVOX = IH * IW;
  constexpr int NTHR = NW * 64;
  constexpr int NLD = (NVOX + NTHR - 1) / NTHR;   // 8-byte loads per thread per frame
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // [KT][IH][even pairs 0..16 | odd pairs 0..16][2 voxels][4]: output column n reads pair 2n + q -- with the pairs of
  // one parity stored contiguously the 16 lanes of a ds_read_b128 phase touch 16 consecutive 16-byte chunks (the
  // natural layout would put lanes n and n + 8 on the same banks: columns are 32 bytes apart)
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  int bid = blockIdx.x;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h; bid /= tiles_h;
  const int to = bid % d.To;
  const int b = bid / d.To;
  const int ho0 = th * TH, wo0 = tw * TW;
  const int hi0 = ho0 * S - 3, wi0 = wo0 * S - 3, ti0 = to * d.st - d.pt;

  constexpr unsigned kOOB = 0x80000000u;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const bf16_t* X = static_cast<const bf16_t*>(d.x) + (long)b * d.x_bs;
  const unsigned frame_bytes = (unsigned)(d.Hi * d.Wi) * 8u;
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(X), 0, (int)(frame_bytes * (unsigned)d.Ti), 0x00020000);
  // per-thread staging geometry of one frame (the same for every frame): global byte offset, LDS element index
  unsigned g_off[NLD];
  int l_idx[NLD];
#pragma unroll
  for (int n = 0; n < NLD; ++n) {
    const int i = tid + n * NTHR;
    const int ir = i / IW, ic = i - ir * IW;
    const int hi = hi0 + ir, wi = wi0 + ic;
    const bool ok = i < NVOX && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
    g_off[n] = ok ? (unsigned)(hi * d.Wi + wi) * 8u : kOOB;
    const int pr = ic >> 1;
    l_idx[n] = i < NVOX ? ((ir * (IW / 2) + (pr & 1) * HP + (pr >> 1)) * 2 + (ic & 1)) * 4 : -1;
  }
  u32x2 st[2][NLD];   // two frames in flight
  auto load_frame = [&](int fr) {
    const int ti = ti0 + fr;
    const bool live = (unsigned)ti < (unsigned)d.Ti;
#pragma unroll
    for (int n = 0; n < NLD; ++n)
      st[fr & 1][n] = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)((live && g_off[n] != kOOB) ? g_off[n] + (unsigned)ti * frame_bytes : kOOB), 0, 0);
  };
  auto store_frame = [&](int fr) {
#pragma unroll
    for (int n = 0; n < NLD; ++n)
      if (l_idx[n] >= 0) *reinterpret_cast<u32x2*>(tile + (size_t)fr * NVOX * 4 + l_idx[n]) = st[fr & 1][n];
  };

  load_frame(0);
  if (KT > 1) load_frame(1);
  // ---- this wave's filter rows as A fragments, streamed from L2 one temporal tap ahead of their use: every fragment
  //      feeds exactly 8 MFMAs of this workgroup, so nothing is gained by holding all kt * 7 of them (84 VGPRs) ----
  const int ch = wave * 16 + n16;
  const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w) + (long)(ch < d.cout ? ch : 0) * KT * 7 * wpitch + q * 8;
  bf16x8 areg[2][7];
  auto load_a = [&](int dt, bf16x8 (&a)[7]) {
#pragma unroll
    for (int dh = 0; dh < 7; ++dh) {
      a[dh] = *reinterpret_cast<const bf16x8*>(Wt + (long)(dt * 7 + dh) * wpitch);
      if (ch >= d.cout) a[dh] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  load_a(0, areg[0]);
#pragma unroll
  for (int fr = 0; fr < KT; ++fr) {
    store_frame(fr);
    if (fr + 2 < KT) load_frame(fr + 2);
  }
  __syncthreads();

  // ---- 8 output rows x 16 columns x this wave's 16 channels ----
  f32x4 acc[TH];
#pragma unroll
  for (int r = 0; r < TH; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* bbase = tile + ((q & 1) * HP + n16 + (q >> 1)) * 8;   // pair 2 n16 + q: parity q & 1, index n16 + q / 2
#pragma unroll
  for (int dt = 0; dt < KT; ++dt) {
    if (dt + 1 < KT) load_a(dt + 1, areg[(dt + 1) & 1]);
#pragma unroll
    for (int dh = 0; dh < 7; ++dh) {
      const bf16x8 af = areg[dt & 1][dh];
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const bf16x8 bfv = *reinterpret_cast<const bf16x8*>(bbase + ((size_t)dt * NVOX + (S * r + dh) * IW) * 4);
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv, acc[r], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keeps tap dt+2's fragment loads from being hoisted up here (registers)
  }

  // ---- epilogue: lane holds channels 16*wave + 4q .. +3 of column n16 for every row ----
  const int c0 = wave * 16 + 4 * q;
  const int wo = wo0 + n16;
  if (c0 >= pv_round_up(d.cout, 8) || wo >= d.Wo) return;
  float sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = c0 + j < d.cout;
    sc[j] = ok ? (d.scale ? d.scale[c0 + j] : 1.f) : 0.f;
    sh[j] = ok ? (d.shift ? d.shift[c0 + j] : 0.f) : 0.f;
  }
  // every position-table value of the tile is requested BEFORE the first store: the compiler cannot prove that y does
  // not alias the tables and would otherwise wait for each row's loads behind the previous row's store (8 round trips)
  float pos[TH][4];
#pragma unroll
  for (int r = 0; r < TH; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) pos[r][j] = 0.f;
  if (d.pos_spatial != nullptr) {
    float pt[4] = {0.f, 0.f, 0.f, 0.f};
    if (d.pos_temporal != nullptr) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < d.cout) pt[j] = d.pos_temporal[(long)to * d.cout + c0 + j];
    }
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      const int ho = ho0 + r < d.Ho ? ho0 + r : d.Ho - 1;
      const long ps = d.pos_temporal ? (long)ho * d.Wo + wo : ((long)to * d.Ho + ho) * d.Wo + wo;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < d.cout) pos[r][j] = d.pos_spatial[ps * d.cout + c0 + j] + pt[j];
    }
  }
#pragma unroll
  for (int r = 0; r < TH; ++r) {
    const int ho = ho0 + r;
    if (ho >= d.Ho) break;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = acc[r][j] * sc[j] + sh[j];
    pv_apply_act_n<true>(v, d.act);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += pos[r][j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j >= d.cout) v[j] = 0.f;
    const long yo = (long)b * d.y_bs + (((long)to * d.Ho + ho) * d.Wo + wo) * d.ldy + c0;
    if (d.y_f32) {
      *reinterpret_cast<f32x4*>(static_cast<float*>(d.y) + yo) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(d.y) + yo) = o;
    }
  }
}

// number of waves (16-channel tiles) if the patch-embedding kernel takes this conv, else 0
int stem_pe_waves(const pv_conv3d_desc& d) {
  if (d.dtype != PV_BF16 || d.cin != 4 || d.ldx != 4 || d.dwt_w || d.c4_wpair == 2) return 0;
  if (d.kh != 7 || d.kw != 7 || d.sh != 4 || d.sw != 4 || d.ph != 3 || d.pw != 3) return 0;
  if ((d.kt != 1 && d.kt != 3) || d.pt != d.kt / 2 || d.st < 1) return 0;
  if (d.dil_t > 1 || d.dil_h > 1 || d.dil_w > 1) return 0;
  if ((long)d.Ti * d.Hi * d.Wi * 8 > 0x7fffffffL) return 0;
  const int nw = pv_round_up(d.cout, 16) / 16;
  return (nw == 4 || nw == 6 || nw == 8) ? nw : 0;
}

template <int KT, int NW> int launch_stem_pe(const pv_conv3d_desc& d, int wpitch, hipStream_t s) {
  constexpr int TH = 8, TW = 16;
  const int tiles_h = (d.Ho + TH - 1) / TH, tiles_w = (d.Wo + TW - 1) / TW;
  const size_t lds = (size_t)KT * ((TH - 1) * 4 + 7) * ((TW - 1) * 4 + 8) * 8;
  const long blocks = (long)d.B * d.To * tiles_h * tiles_w;
  if (blocks > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  PV_LAUNCH((stem_pe_kernel<KT, NW>), dim3((unsigned)blocks), dim3(NW * 64), lds, s, d, tiles_h, tiles_w, wpitch);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

// geometry of the LDS-staged 7 x 7 stem kernel: 0 = not this kernel's
int stem7_variant(const pv_conv3d_desc& d) {
  if (d.dtype != PV_BF16 || d.cin != 4 || d.ldx != 4 || d.y_f32 || d.pos_spatial || d.pos_temporal || d.dwt_w) return 0;
  if (d.kh != 7 || d.kw != 7 || d.sh != 2 || d.sw != 2 || d.ph != 3 || d.pw != 3 || d.st != 1) return 0;
  if (d.kt < 1 || d.kt > 7 || d.pt != d.kt / 2 || d.To != d.Ti) return 0;
  if (d.dil_t > 1 || d.dil_h > 1 || d.dil_w > 1) return 0;
  if ((long)d.Ti * d.Hi * d.Wi * 8 > 0x7fffffffL || (long)d.To * d.Ho * d.Wo * d.ldy * 2 > 0x7fffffffL) return 0;
  if (d.cout <= 8) return 2;                                   // row pairs
  if (d.cout % 16 == 0 && d.cout <= 64) return 1;              // channel tiles
  return 0;
}

template <int RP, int NT, int TM, int KT = 0> int launch_stem7(const pv_conv3d_desc& d, int wpitch, hipStream_t s) {
  constexpr int TH = 8, TW = TM * 16;
  constexpr int FRAME_B = ((TH - 1) * 2 + 7) * ((TW - 1) * 2 + 8) * 8;
  const int tiles_h = (d.Ho + TH - 1) / TH, tiles_w = (d.Wo + TW - 1) / TW;
  const int nfrag = (RP == 2 && KT > 0) ? 0 : d.kt * (RP == 2 ? 9 : 7 * NT);
  const size_t lds = (size_t)nfrag * 1024 + (size_t)(d.kt + 1) * FRAME_B;
  if (lds > 160 * 1024) return PV_ERR_UNSUPPORTED;
  auto kern = stem7_kernel<RP, NT, TM, KT>;
  if (lds > 64 * 1024)
    PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long blocks = (long)d.B * tiles_h * tiles_w;
  if (blocks > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  PV_LAUNCH(kern, dim3((unsigned)blocks), dim3(kThreads), lds, s, d, tiles_h, tiles_w, wpitch);
  pv_note_kernel("stem7_kernel");   // (launched through a function pointer: PV_LAUNCH saw only the variable)
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
  if (d.act == PV_ACT_RELU) PV_LAUNCH((stem_c4_dwt_kernel<NT, TM, KS, DK, PV_ACT_RELU>), grid, block, lds, s, d, (int)gpc);
  else PV_LAUNCH((stem_c4_dwt_kernel<NT, TM, KS, DK, PV_ACT_NONE>), grid, block, lds, s, d, (int)gpc);
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
  if (const int nw = pv_tune("stem_pe", 1) ? stem_pe_waves(d) : 0) {   // 7 x 7 / stride 4: filter in registers, tile in LDS
#define PV_PE(KT_)                                                              \
    (nw == 6 ? launch_stem_pe<KT_, 6>(d, KWP * 4, s) : nw == 4 ? launch_stem_pe<KT_, 4>(d, KWP * 4, s) \
             : launch_stem_pe<KT_, 8>(d, KWP * 4, s))
    const int r = d.kt == 3 ? PV_PE(3) : PV_PE(1);
#undef PV_PE
    if (r != PV_ERR_UNSUPPORTED) return r;
  }
  if (const int v7 = pv_tune("stem7", 1) ? stem7_variant(d) : 0) {   // 7 x 7 / stride 2: input tiles through LDS
    int r;
    if (v7 == 2 && d.kt == 5) r = launch_stem7<2, 1, 2, 5>(d, KWP * 4, s);   // SlowFast's fast stem
    else if (v7 == 2) r = launch_stem7<2, 1, 2>(d, KWP * 4, s);
    else if (d.cout == 64) r = launch_stem7<1, 4, 2>(d, KWP * 4, s);
    else if (d.cout == 48) r = launch_stem7<1, 3, 2>(d, KWP * 4, s);
    else if (d.cout == 32) r = launch_stem7<1, 2, 2>(d, KWP * 4, s);
    else r = launch_stem7<1, 1, 2>(d, KWP * 4, s);
    if (r != PV_ERR_UNSUPPORTED) return r;
  }
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
