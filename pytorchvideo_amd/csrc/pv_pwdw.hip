// Fused pointwise -> depthwise 3x3x3 convolution: conv_a + norm_a + act_a + conv_b + norm_b
// (+ act_b, + squeeze-excitation partial sums) of an X3D / ir-CSN bottleneck
// (models/x3d.py:169-207, models/resnet.py:1345-1365) in ONE pass over the block input.
//
// Unfused, conv_a writes the expanded tensor (2.25x the block width in X3D) to HBM and conv_b reads
// it straight back: for X3D-M that round trip is the largest byte mover of the whole network.  Here
// the plane-streaming depthwise kernel (pv_dwconv.hip) gets a different producer for its LDS planes:
// instead of copying a halo tile of the expanded tensor from global memory, every wave computes its
// share of that tile with MFMA from the (narrow) block input:
//     h[voxel][32 channels of this workgroup's slab] = act(scale * (Wa[32 x Cin] . x[voxel][Cin]) + shift)
// evaluated as Wa . X^T (A operand = filter rows, B operand = 16 voxels x 32 input channels loaded
// straight from global memory into operand registers), so that a lane ends up with 4 consecutive
// channels of one voxel = one 8-byte LDS write in the depthwise kernel's plane layout.  Voxels of the
// halo that fall outside the image are written as ZERO (conv_b zero-pads the *expanded* tensor, which
// is not the image of a zero input because of the folded-BN shift).  The value is rounded to bf16
// exactly where the unfused path stores it, so both paths give bit-identical results.
// A workgroup still owns one 32-channel slab: the slab's filter rows are the only part of conv_a it
// evaluates, so nothing but the halo (1.7x at stride 1) is recomputed; slabs of one tile are
// consecutive workgroups of one XCD and share the input tile through that XCD's L2.
#include "pv_common.h"

int pv_plane_variant(const pv_dwconv3d_desc& d);          // pv_dwconv.hip
int pv_plane_tiles(const pv_dwconv3d_desc& d, int nw);    // pv_dwconv.hip

namespace {

constexpr int kPlaneThreads = 256;  // 4 waves = 4 output rows
constexpr int kPR = kPlaneThreads / 64;

template <int S, int NW, int KS, int ACT>
__global__ __launch_bounds__(kPlaneThreads) void pwdw_plane_kernel(const pv_dwconv3d_desc d, int ntiles, int ngroups) {
  constexpr int TW = 4 * NW;               // tile width (outputs)
  constexpr int IH = (kPR - 1) * S + 3;
  constexpr int IW = (TW - 1) * S + 3;
  constexpr bool kDeep = S == 1;           // small planes: keep two of them in flight
  constexpr int IWP = (IW + 1) & ~1;       // LDS row pitch (even: columns are pair-swapped)
  constexpr int NVOX = IH * IW;
  constexpr int NMT = (NVOX + 15) / 16;    // 16-voxel MFMA column tiles per plane
  constexpr int MT = (NMT + kPR - 1) / kPR;  // ... per wave
  constexpr int NC = (NW - 1) * S + 3;     // input columns feeding a lane's NW outputs
  __shared__ __attribute__((aligned(16))) bf16_t s_in[2][IH * IWP + 1][32];   // +1: dump row for the ragged last column tile
  __shared__ float s_ps[kPR][32];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int row = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = output row of the tile (scalar)
  const int cp = lane & 15;                // channel pair inside the 32-channel slab (depthwise part)
  const int sx = lane >> 4;                // column group (depthwise part)
  const int n16 = lane & 15, q = lane >> 4;  // MFMA roles: voxel column / k-group (pointwise part)
  const int c_p = pv_round_up(d.C, 8);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = slot % ngroups;
  const long tb = (long)(slot / ngroups) * 8 + xcd;     // (clip, tile) index
  if (tb >= (long)ntiles * d.B) return;
  const int tile_id = (int)(tb % ntiles);
  const int cbase = grp * 32;
  const int ch = cbase + cp * 2;
  const bool ch_ok = ch < c_p;
  const int tiles_w = (d.Wo + TW - 1) / TW;
  const int th = tile_id / tiles_w, tw = tile_id - th * tiles_w;
  const int b = (int)(tb / ntiles);
  const int ho0 = th * kPR, wo0 = tw * TW;
  const int ho = ho0 + row, wo_first = wo0 + sx * NW;
  const int hi0 = ho0 * S - 1, wi0 = wo0 * S - 1;

  constexpr unsigned kOOB = 0x80000000u;
  const bf16_t* X = static_cast<const bf16_t*>(d.x) + (long)b * d.x_bs;
  bf16_t* Y = static_cast<bf16_t*>(d.y) + (long)b * d.y_bs;
  const unsigned x_plane_bytes = (unsigned)(d.Hi * d.Wi * d.ldx) * 2u;
  const unsigned y_plane_bytes = (unsigned)(d.Ho * d.Wo * d.ldy) * 2u;
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(x_plane_bytes * (unsigned)d.Ti), 0x00020000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, (int)(y_plane_bytes * (unsigned)d.To), 0x00020000);

  // ---- depthwise taps and folded BN of conv_b: registers for the whole kernel ----
  float2 wt[27];
#pragma unroll
  for (int t = 0; t < 27; ++t)
    wt[t] = ch_ok ? *reinterpret_cast<const float2*>(d.w + (long)t * c_p + ch) : float2{0.f, 0.f};
  float2 sc = {0.f, 0.f}, sh = {0.f, 0.f};
  if (ch_ok) {
    sc.x = ch < d.C ? (d.scale ? d.scale[ch] : 1.f) : 0.f;
    sc.y = ch + 1 < d.C ? (d.scale ? d.scale[ch + 1] : 1.f) : 0.f;
    sh.x = ch < d.C ? (d.shift ? d.shift[ch] : 0.f) : 0.f;
    sh.y = ch + 1 < d.C ? (d.shift ? d.shift[ch + 1] : 0.f) : 0.f;
  }

  // ---- pointwise filter fragments (A operand, k = 8q..8q+7 of step ks) and the folded BN of conv_a.
  //      Row r of MFMA tile nt is slab channel 8*(r>>2) + 4*nt + (r&3): the accumulator rows 4q..4q+3 of the
  //      two tiles are then the 8 CONSECUTIVE channels 8q..8q+7 of a voxel = one 16-byte LDS write ----
  bf16x8 wa[2][KS];
  {
    const bf16_t* __restrict__ Wa = static_cast<const bf16_t*>(d.pw_w);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wa[nt][ks] = *reinterpret_cast<const bf16x8*>(Wa + (long)(cbase + 8 * (n16 >> 2) + 4 * nt + (n16 & 3)) * (KS * 32) + ks * 32 + q * 8);
  }
  float2 sa[2][2], ha[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cbase + q * 8 + nt * 4 + r;
      const bool ok = c < d.C;
      const float s_ = ok ? (d.pw_scale ? d.pw_scale[c] : 1.f) : 0.f;
      const float h_ = ok ? (d.pw_shift ? d.pw_shift[c] : 0.f) : 0.f;
      if (r & 1) { sa[nt][r >> 1].y = s_; ha[nt][r >> 1].y = h_; }
      else { sa[nt][r >> 1].x = s_; ha[nt][r >> 1].x = h_; }
    }
  // ReLU on the packed pair: a negative bf16 is a negative int16, so max(.,0) per 16-bit half clears it
  const bool a_relu = d.pw_act == PV_ACT_RELU;

  // ---- producer geometry: this wave's column tiles of the halo plane ----
  auto colperm = [](int c) { return c ^ ((c / (NW * S)) & 1); };
  unsigned x_off[MT];     // byte offset of (voxel, channel 8q) inside an input plane, or kOOB
  int h_lds[MT];          // LDS element index of (voxel, channel 8q); lanes past the tile write the dump row
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int v = (row + i * kPR) * 16 + n16;
    const int ih = v / IW, iw = v - ih * IW;
    const int hi = hi0 + ih, wi = wi0 + iw;
    const bool in_tile = v < NVOX;
    const bool ok = in_tile && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi;
    x_off[i] = ok ? (unsigned)((hi * d.Wi + wi) * d.ldx + q * 8) * 2u : kOOB;
    h_lds[i] = (in_tile ? ih * IWP + colperm(iw) : IH * IWP) * 32 + q * 8;
  }
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 xa[MT][KS], xb[MT][KS];
  auto load_plane = [&](u32x4 (&reg)[MT][KS], int p) {
    if (p >= d.Ti) return;   // wave-uniform
    const unsigned pb = (unsigned)p * x_plane_bytes;
#pragma unroll
    for (int i = 0; i < MT; ++i)   // column tiles past the plane read nothing (kOOB) and write the dump row
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        reg[i][ks] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(x_off[i] + pb + ks * 64u), 0, 0);
  };
  auto produce_plane = [&](const u32x4 (&reg)[MT][KS], int p) {
    if (p >= d.Ti) return;
    bf16_t* dst = &s_in[p & 1][0][0];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, reg[i][ks]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][ks], xf, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][ks], xf, acc[1], 0, 0, 0);
      }
      // halo voxels outside the image are conv_b's zero padding: mask the packed result
      const unsigned inside = x_off[i] != kOOB ? 0xffffffffu : 0u;
      u32x4 o;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef short s16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float v0 = acc[nt][2 * h] * sa[nt][h].x + ha[nt][h].x;
          const float v1 = acc[nt][2 * h + 1] * sa[nt][h].y + ha[nt][h].y;
          const bf16x2_t pk = {(bf16_t)v0, (bf16_t)v1};
          s16x2 pi = __builtin_bit_cast(s16x2, pk);
          if (a_relu) pi = __builtin_elementwise_max(pi, s16x2{0, 0});
          o[nt * 2 + h] = __builtin_bit_cast(unsigned, pi) & inside;
        }
      }
      *reinterpret_cast<u32x4*>(dst + h_lds[i]) = o;
    }
  };

  float2 acc[3][NW];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int n = 0; n < NW; ++n) acc[a][n] = float2{0.f, 0.f};
  float2 ps = {0.f, 0.f};

  unsigned y_off[NW];
  float y_mask[NW];   // 1 for outputs that exist: the squeeze-excitation sums skip the ragged edge
#pragma unroll
  for (int n = 0; n < NW; ++n) {
    const int wo = wo_first + n;
    const bool ok = ch_ok && ho < d.Ho && wo < d.Wo;
    y_off[n] = ok ? (unsigned)((ho * d.Wo + wo) * d.ldy + ch) * 2u : kOOB;
    y_mask[n] = ok ? 1.f : 0.f;
  }
  const bool has_psum = d.psum != nullptr;
  auto finalize = [&](float2 (&a)[NW], int t) {
    const unsigned tbytes = (unsigned)t * y_plane_bytes;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      float v0 = a[n].x * sc.x + sh.x, v1 = a[n].y * sc.y + sh.y;
      if (has_psum) { ps.x += v0 * y_mask[n]; ps.y += v1 * y_mask[n]; }
      if (ACT == PV_ACT_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      else if (ACT == PV_ACT_SWISH) { v0 *= pv_sigmoid(v0); v1 *= pv_sigmoid(v1); }
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const bf16x2_t o = {(bf16_t)v0, (bf16_t)v1};
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), ry, (int)(y_off[n] + tbytes), 0, 0);
    }
  };

  auto plane = [&](int p) {
    const bf16_t* base = &s_in[p & 1][0][0] + cp * 2;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      float2 x[NC];
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const uint32_t u = *reinterpret_cast<const uint32_t*>(base + ((row * S + dh) * IWP + colperm(sx * NW * S + i)) * 32);
        x[i].x = __uint_as_float(u << 16);
        x[i].y = __uint_as_float(u & 0xffff0000u);
      }
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const float2 w0 = wt[(0 * 3 + dh) * 3 + dw];   // kt = 0 feeds output p+1
        const float2 w1 = wt[(1 * 3 + dh) * 3 + dw];   // kt = 1: output p
        const float2 w2 = wt[(2 * 3 + dh) * 3 + dw];   // kt = 2: output p-1
#pragma unroll
        for (int n = 0; n < NW; ++n) {
          const float2 xv = x[n * S + dw];
          acc[2][n].x += xv.x * w0.x; acc[2][n].y += xv.y * w0.y;
          acc[1][n].x += xv.x * w1.x; acc[1][n].y += xv.y * w1.y;
          acc[0][n].x += xv.x * w2.x; acc[0][n].y += xv.y * w2.y;
        }
      }
    }
    if (p >= 1) finalize(acc[0], p - 1);
    if (p == d.Ti - 1) finalize(acc[1], p);
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      acc[0][n] = acc[1][n];
      acc[1][n] = acc[2][n];
      acc[2][n] = float2{0.f, 0.f};
    }
  };

  load_plane(xa, 0);
  if (kDeep) load_plane(xb, 1);
  produce_plane(xa, 0);
  __syncthreads();
  if (kDeep) {
    for (int p = 0; p < d.Ti; p += 2) {
      load_plane(xa, p + 2);       // xb holds plane p+1
      plane(p);
      produce_plane(xb, p + 1);
      __syncthreads();
      if (p + 1 < d.Ti) {
        load_plane(xb, p + 3);     // xa holds plane p+2
        plane(p + 1);
        produce_plane(xa, p + 2);
        __syncthreads();
      }
    }
  } else {
    for (int p = 0; p < d.Ti; ++p) {
      load_plane(xa, p + 1);
      plane(p);
      produce_plane(xa, p + 1);
      __syncthreads();
    }
  }

  if (d.psum != nullptr) {
    ps.x += __shfl_xor(ps.x, 16, 64); ps.y += __shfl_xor(ps.y, 16, 64);
    ps.x += __shfl_xor(ps.x, 32, 64); ps.y += __shfl_xor(ps.y, 32, 64);
    if (lane < 16) { s_ps[row][cp * 2] = ps.x; s_ps[row][cp * 2 + 1] = ps.y; }
    __syncthreads();
    if (tid < 32 && cbase + tid < c_p) {
      float a = 0.f;
#pragma unroll
      for (int rr = 0; rr < kPR; ++rr) a += s_ps[rr][tid];
      float* dst = d.psum + ((long)b * ntiles + tile_id) * c_p + cbase + tid;
      *dst = a;
    }
  }
}

template <int S, int NW, int KS> int launch_pwdw(const pv_dwconv3d_desc& d, hipStream_t s) {
  const int c_p = pv_round_up(d.C, 8);
  const int ntiles = pv_plane_tiles(d, NW), ngroups = (c_p + 31) / 32;
  const long blocks = pv_ceil_div((long)ntiles * d.B, 8) * 8 * ngroups;
  if (blocks > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  dim3 grid((unsigned)blocks), block(kPlaneThreads);
  if (d.act == PV_ACT_NONE) PV_LAUNCH((pwdw_plane_kernel<S, NW, KS, PV_ACT_NONE>), grid, block, 0, s, d, ntiles, ngroups);
  else if (d.act == PV_ACT_RELU) PV_LAUNCH((pwdw_plane_kernel<S, NW, KS, PV_ACT_RELU>), grid, block, 0, s, d, ntiles, ngroups);
  else PV_LAUNCH((pwdw_plane_kernel<S, NW, KS, PV_ACT_SWISH>), grid, block, 0, s, d, ntiles, ngroups);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

template <int KS> int launch_ks(const pv_dwconv3d_desc& d, int nw, hipStream_t s) {
  if (d.sw == 1) return nw == 4 ? launch_pwdw<1, 4, KS>(d, s) : launch_pwdw<1, 2, KS>(d, s);
  return nw == 4 ? launch_pwdw<2, 4, KS>(d, s) : launch_pwdw<2, 2, KS>(d, s);
}

int ksteps_of(const pv_dwconv3d_desc& d) {
  const int ks = (d.pw_cin + 31) / 32;
  return (ks == 1 || ks == 2 || ks == 3 || ks == 4 || ks == 6) ? ks : 0;
}

}  // namespace

// geometry-only test (no pointers needed): can pv_dwconv3d take this descriptor with a fused
// pointwise producer?
int pv_pwdw_supported(const pv_dwconv3d_desc& d) {
  if (d.dtype != PV_BF16 || d.w_mod != 0 || d.n_prefix != 0) return 0;
  if (d.pw_cin <= 0 || d.ldx < pv_round_up(d.pw_cin, 8)) return 0;
  if (!ksteps_of(d)) return 0;
  if (d.pw_act != PV_ACT_NONE && d.pw_act != PV_ACT_RELU) return 0;
  return pv_plane_variant(d) != 0;
}

int pv_pwdw_launch(const pv_dwconv3d_desc& d, hipStream_t s) {
  if (!pv_pwdw_supported(d)) return PV_ERR_UNSUPPORTED;
  const int nw = pv_plane_variant(d);
  switch (ksteps_of(d)) {
    case 1: return launch_ks<1>(d, nw, s);
    case 2: return launch_ks<2>(d, nw, s);
    case 3: return launch_ks<3>(d, nw, s);
    case 4: return launch_ks<4>(d, nw, s);
    case 6: return launch_ks<6>(d, nw, s);
  }
  return PV_ERR_UNSUPPORTED;
}
