// Depthwise 3-D convolution on NDHWC (channels-last) activations: an HBM-bound stencil.
//
// A thread owns one 8-channel chunk (16 B of bf16) of NW consecutive output columns, so
// every global access is a 16-byte vector and each loaded input column is reused for up
// to KW taps x NW outputs in registers.  Lanes run fastest over the channel chunks of a
// voxel, i.e. a wave reads whole contiguous NDHWC voxel rows.  Filter taps live in LDS as
// fp32.  The folded-BN affine, the activation and (optionally) the per-block partial sums
// for the squeeze-excitation mean are fused into the epilogue; partial sums are written
// per block (no atomics) so the result is run-to-run deterministic.
#include <type_traits>
#include "pv_common.h"

namespace {

constexpr int kThreads = 256;

// one tap of a channel-wise grouped conv: output channel c of the chunk sums GW inputs of its own group
template <int GW> __device__ __forceinline__ void grouped_taps(float (&acc)[8], const float (&f)[8], const float* wp, int w_p) {
#pragma unroll
  for (int j = 0; j < GW; ++j) {
    const float4 w0v = *reinterpret_cast<const float4*>(wp + j * w_p);
    const float4 w1v = *reinterpret_cast<const float4*>(wp + j * w_p + 4);
    const float w[8] = {w0v.x, w0v.y, w0v.z, w0v.w, w1v.x, w1v.y, w1v.z, w1v.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] += f[(c / GW) * GW + j] * w[c];
  }
}

// KW = 0 selects the fully runtime (one output per thread) variant.
template <typename T, int KW, int SW, int NW>
__global__ __launch_bounds__(kThreads) void dwconv_kernel(const pv_dwconv3d_desc d, int gpb,
                                                          int wgroups, int units_per_batch, int w_global) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int c_p = pv_round_up(d.C, 8);
  const int CG = c_p / 8;
  const int w_p = d.w_mod > 0 ? pv_round_up(d.w_mod, 8) : c_p;
  const int taps = d.kt * d.kh * d.kw;
  const int gw = d.gw > 1 ? d.gw : 1;  // input channels per output channel (grouped conv: KW == 0 variant only)
  // filter taps in LDS, [taps][gw][w_p] -- unless they would not fit (grouped conv over hundreds of channels): then
  // they are read where they lie (L1 / L2 hits: every workgroup reads the same few hundred KB)
  const float* s_w = w_global ? d.w : s_mem;
  float* s_red = s_mem + (w_global ? 0 : taps * gw * w_p);  // [gpb][c_p]   (psum only)

  const int tid = threadIdx.x;
  if (!w_global) {
    for (int i = tid; i < taps * gw * w_p; i += kThreads) s_mem[i] = d.w[i];
    __syncthreads();
  }

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
            const float* wp = s_w + (((dt * d.kh + dh) * d.kw + dw) * gw) * w_p + wc0;
            if (gw == 1) {
              const float4 w0v = *reinterpret_cast<const float4*>(wp);
              const float4 w1v = *reinterpret_cast<const float4*>(wp + 4);
              acc[0][0] += f[0] * w0v.x; acc[0][1] += f[1] * w0v.y;
              acc[0][2] += f[2] * w0v.z; acc[0][3] += f[3] * w0v.w;
              acc[0][4] += f[4] * w1v.x; acc[0][5] += f[5] * w1v.y;
              acc[0][6] += f[6] * w1v.z; acc[0][7] += f[7] * w1v.w;
            } else if (gw == 2) {
              grouped_taps<2>(acc[0], f, wp, w_p);
            } else if (gw == 4) {
              grouped_taps<4>(acc[0], f, wp, w_p);
            } else {
              grouped_taps<8>(acc[0], f, wp, w_p);
            }
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
      }
      pv_apply_act_n<sizeof(T) == 2>(v, d.act);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (c0 + j >= d.C) v[j] = 0.f;
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
      float* dst = d.psum + ((long)b * gridDim.x + blockIdx.x) * c_p + c;
      *dst = s;
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


// ---------------------------------------------------------------------------------------
// 3x3x3 depthwise conv, temporal stride 1, spatial stride 1 or 2 (X3D conv_b): plane streaming.
//
// The generic kernel above re-reads every input voxel ~13x through the vector cache and is
// bound by the texture-address path, not by HBM.  Here a workgroup (4 waves) owns a
// (4 x 4*NW) output tile of one clip and 32 channels and walks the T axis:
//   * every input plane (tile + halo, 32 channels = 64 B per voxel) is staged through LDS
//     exactly once, double-buffered, one barrier per plane;
//   * a lane owns ONE CHANNEL PAIR (lane&15) of NW consecutive output columns of one row
//     (wave = row, lane>>4 = column group), so its 27 x 2 filter taps are loop-invariant and
//     stay in 54 registers for the whole kernel -- no per-tap weight traffic at all;
//   * each input dword (2 bf16 channels) is read from LDS once per plane, converted to fp32
//     once and scattered with packed FMAs into THREE rolling accumulator sets (outputs t-1, t,
//     t+1): LDS traffic and conversions are 3x lower than in a gather formulation;
//   * output plane t-1 is finished after input plane t: folded BN, activation, store (16 lanes
//     cover 64 contiguous bytes of a voxel), and -- for squeeze-excitation -- lane-private
//     partial sums reduced once per workgroup (no atomics: bitwise reproducible).
constexpr int kPlaneThreads = 256;  // 4 waves = 4 output rows
constexpr int kPR = kPlaneThreads / 64;

// (A four-waves-per-SIMD build of the narrow tile -- <= 128 registers, no second plane in flight, so that res4's 1792
// workgroups are 2 rounds instead of 3 -- measured -1.8 % on X3D-M in the two-branch form in round 4 and is gone.)
template <int S, int NW, int ACT>
__global__ __launch_bounds__(kPlaneThreads, 1) void dw3_plane_kernel(const pv_dwconv3d_desc d, int ntiles, int ngroups) {
  constexpr int TW = 4 * NW;               // tile width (outputs)
  constexpr int IH = (kPR - 1) * S + 3;
  constexpr int IW = (TW - 1) * S + 3;
  constexpr bool kDeep = S == 1;   // small planes: keep two of them in flight
  constexpr int IWP = (IW + 1) & ~1;       // LDS row pitch (even: columns are pair-swapped)
  constexpr int NVOX = IH * IW;
  constexpr int NITEM = NVOX * 4;          // 16-byte items per plane
  constexpr int NST = (NITEM + kPlaneThreads - 1) / kPlaneThreads;
  constexpr int NC = (NW - 1) * S + 3;     // input columns feeding a lane's NW outputs
  __shared__ __attribute__((aligned(16))) bf16_t s_in[2][IH * IWP][32];
  __shared__ float s_ps[kPR][32];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int row = tid >> 6;                // wave = output row of the tile
  const int cp = lane & 15;                // channel pair inside the 32-channel slab
  const int sx = lane >> 4;                // column group
  const int c_p = pv_round_up(d.C, 8);
  // 1-D grid decoded so that the channel groups of one tile (which share every 128-byte line of
  // the input when C is not a multiple of 64 channels) are consecutive workgroups of the SAME XCD:
  // the second group's reads hit in that XCD's L2 instead of going back to the fabric.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = slot % ngroups;
  const long tb = (long)(slot / ngroups) * 8 + xcd;     // (clip, tile) index
  if (tb >= (long)ntiles * d.B) return;                 // grid is padded to a multiple of 8 tiles
  const int tile_id = (int)(tb % ntiles);
  const int cbase = grp * 32;
  const int ch = cbase + cp * 2;           // first of this lane's two channels
  const bool ch_ok = ch < c_p;
  const int tiles_w = (d.Wo + TW - 1) / TW;
  const int th = tile_id / tiles_w, tw = tile_id - th * tiles_w;
  const int b = (int)(tb / ntiles);
  const int ho0 = th * kPR, wo0 = tw * TW;
  const int ho = ho0 + row, wo_first = wo0 + sx * NW;
  const int hi0 = ho0 * S - 1, wi0 = wo0 * S - 1;

  // Buffer descriptors over this clip's input / output: out-of-range offsets (halo outside the
  // image, padded channels, masked outputs) read as 0 / are dropped by the hardware, so the plane
  // loop has no exec-masked branches and the compiler keeps counted vmcnt waits.
  constexpr unsigned kOOB = 0x80000000u;
  // token tensors (MViT pooling): the n_prefix leading rows (cls token) are skipped here and
  // copied by dw_prefix_kernel
  const bf16_t* X = static_cast<const bf16_t*>(d.x) + (long)b * d.x_bs + (long)d.n_prefix * d.ldx;
  bf16_t* Y = static_cast<bf16_t*>(d.y) + (long)b * d.y_bs + (long)d.n_prefix * d.ldy;
  const unsigned x_plane_bytes = (unsigned)(d.Hi * d.Wi * d.ldx) * 2u;
  const unsigned y_plane_bytes = (unsigned)(d.Ho * d.Wo * d.ldy) * 2u;
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(x_plane_bytes * (unsigned)d.Ti), 0x00020000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, (int)(y_plane_bytes * (unsigned)d.To), 0x00020000);

  // ---- this lane's 27 x 2 filter taps, folded-BN scale/shift: registers for the whole kernel ----
  // (w_mod > 0: one filter shared by every head, channel c uses filter column c % w_mod)
  float2 wt[27];
  const int w_p = d.w_mod > 0 ? pv_round_up(d.w_mod, 8) : c_p;
  const int wch = d.w_mod > 0 ? ch % d.w_mod : ch;
#pragma unroll
  for (int t = 0; t < 27; ++t)
    wt[t] = ch_ok ? *reinterpret_cast<const float2*>(d.w + (long)t * w_p + wch) : float2{0.f, 0.f};
  float2 sc = {0.f, 0.f}, sh = {0.f, 0.f};
  if (ch_ok) {
    sc.x = ch < d.C ? (d.scale ? d.scale[ch] : 1.f) : 0.f;
    sc.y = ch + 1 < d.C ? (d.scale ? d.scale[ch + 1] : 1.f) : 0.f;
    sh.x = ch < d.C ? (d.shift ? d.shift[ch] : 0.f) : 0.f;
    sh.y = ch + 1 < d.C ? (d.shift ? d.shift[ch + 1] : 0.f) : 0.f;
  }

  // ---- staging geometry (fixed across planes) ----
  // LDS column permutation: the two 16-lane halves of a 32-lane bank group read columns that are
  // NW*S apart; with 64-byte voxels they would share banks, so odd groups of NW*S columns are
  // stored pair-swapped (col ^ 1) -> conflict-free ds_read_b32
  auto colperm = [](int c) { return c ^ ((c / (NW * S)) & 1); };
  unsigned st_off[NST];   // byte offset inside a plane, or kOOB (reads as zero)
  int st_lds[NST];        // LDS element index, or -1 (no such item)
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int id = tid + i * kPlaneThreads;
    const int v = id >> 2, c = id & 3;
    const int ih = v / IW, iw = v - ih * IW;
    const int hi = hi0 + ih, wi = wi0 + iw;
    const bool ok = id < NITEM && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi &&
                    (cbase + c * 8) < c_p;
    st_off[i] = ok ? (unsigned)((hi * d.Wi + wi) * d.ldx + cbase + c * 8) * 2u : kOOB;
    st_lds[i] = id < NITEM ? (ih * IWP + colperm(iw)) * 32 + c * 8 : -1;
  }
  // two register sets: planes p+1 and p+2 are in flight while plane p is computed (with few workgroups
  // per CU the bytes in flight, not the ALUs, set the streaming rate)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 st_a[NST], st_b[NST];
  auto load_plane = [&](u32x4 (&reg)[NST], int p) {
    if (p >= d.Ti) return;   // wave-uniform
    const unsigned pb = (unsigned)p * x_plane_bytes;
#pragma unroll
    for (int i = 0; i < NST; ++i) reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(st_off[i] + pb), 0, 0);
  };
  auto store_plane = [&](const u32x4 (&reg)[NST], int p) {
    if (p >= d.Ti) return;
#pragma unroll
    for (int i = 0; i < NST; ++i)
      if ((i + 1) * kPlaneThreads <= NITEM || st_lds[i] >= 0)
        *reinterpret_cast<u32x4*>(&s_in[p & 1][0][0] + st_lds[i]) = reg[i];
  };

  float2 acc[3][NW];   // acc[0] = output p-1 (finished by plane p), acc[1] = output p, acc[2] = output p+1
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int n = 0; n < NW; ++n) acc[a][n] = float2{0.f, 0.f};
  float2 ps = {0.f, 0.f};

  // per-output store offsets inside an output plane (kOOB: masked)
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
    const unsigned tb = (unsigned)t * y_plane_bytes;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      float v0 = a[n].x * sc.x + sh.x, v1 = a[n].y * sc.y + sh.y;
      if (has_psum) { ps.x += v0 * y_mask[n]; ps.y += v1 * y_mask[n]; }
      if (ACT == PV_ACT_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      else if (ACT == PV_ACT_SWISH) { v0 *= pv_sigmoid(v0); v1 *= pv_sigmoid(v1); }
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const bf16x2_t o = {(bf16_t)v0, (bf16_t)v1};
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), ry, (int)(y_off[n] + tb), 0, 0);
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

  load_plane(st_a, 0);
  if (kDeep) load_plane(st_b, 1);
  store_plane(st_a, 0);
  __syncthreads();
  if (kDeep) {
    for (int p = 0; p < d.Ti; p += 2) {
      load_plane(st_a, p + 2);     // st_b holds plane p+1
      plane(p);
      store_plane(st_b, p + 1);
      __syncthreads();
      if (p + 1 < d.Ti) {
        load_plane(st_b, p + 3);   // st_a holds plane p+2
        plane(p + 1);
        store_plane(st_a, p + 2);
        __syncthreads();
      }
    }
  } else {
    for (int p = 0; p < d.Ti; ++p) {
      load_plane(st_a, p + 1);
      plane(p);
      store_plane(st_a, p + 1);
      __syncthreads();
    }
  }

  if (d.psum != nullptr) {
    // lanes cp, cp+16, cp+32, cp+48 of a wave hold the same channel pair; then across the rows of the tile
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

// which layers the plane-streaming kernel takes, and with how many outputs per lane
int plane_variant(const pv_dwconv3d_desc& d) {
  if (d.dtype != PV_BF16 || (d.n_prefix != 0 && d.psum) || d.gw > 1) return 0;
  if (d.kt != 3 || d.kh != 3 || d.kw != 3 || d.st != 1 || d.pt != 1 || d.ph != 1 || d.pw != 1) return 0;
  if (d.sh != d.sw || (d.sw != 1 && d.sw != 2)) return 0;
  if (d.act != PV_ACT_NONE && d.act != PV_ACT_RELU && d.act != PV_ACT_SWISH) return 0;
  if ((long)d.Ti * d.Hi * d.Wi * d.ldx > 0x3fffffffL || (long)d.To * d.Ho * d.Wo * d.ldy > 0x3fffffffL)
    return 0;   // 31-bit byte offsets inside a clip (buffer addressing)
  // 4 outputs per lane (16-wide tiles, 186+ VGPRs: 2 waves per SIMD) or 2 (8-wide tiles, 134 VGPRs: 3 waves per SIMD).
  // The plain kernel lives on the number of waves with loads in flight: narrow tiles win on every grid measured (X3D-M 14^2, MViT 56^2 pool_q)
  // (X3D-M res4 conv_b: -7 %); the fused conv_a producer (pw_cin > 0) recomputes its halo, so it keeps the wide tile.
  const bool fused = d.pw_w != nullptr || d.pw_cin > 0;
  return d.Wo >= (fused ? 12 : pv_tune("dw_wide_min_wo", 64)) ? 4 : 2;
}

int plane_tiles(const pv_dwconv3d_desc& d, int nw) { return ((d.Ho + kPR - 1) / kPR) * ((d.Wo + 4 * nw - 1) / (4 * nw)); }

template <int S, int NW> int launch_plane(const pv_dwconv3d_desc& d, hipStream_t s) {
  const int c_p = pv_round_up(d.C, 8);
  const int ntiles = plane_tiles(d, NW), ngroups = (c_p + 31) / 32;
  const long blocks = pv_ceil_div((long)ntiles * d.B, 8) * 8 * ngroups;
  if (blocks > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  dim3 grid((unsigned)blocks), block(kPlaneThreads);
  if (d.n_prefix > 0) {
    const int total = d.B * d.n_prefix * (c_p / 8);
    PV_LAUNCH(dw_prefix_kernel<bf16_t>, dim3((unsigned)pv_ceil_div(total, kThreads)), dim3(kThreads), 0, s, d);
    PV_LAUNCH_CHECK();
  }
  if (d.act == PV_ACT_NONE) PV_LAUNCH((dw3_plane_kernel<S, NW, PV_ACT_NONE>), grid, block, 0, s, d, ntiles, ngroups);
  else if (d.act == PV_ACT_RELU) PV_LAUNCH((dw3_plane_kernel<S, NW, PV_ACT_RELU>), grid, block, 0, s, d, ntiles, ngroups);
  else PV_LAUNCH((dw3_plane_kernel<S, NW, PV_ACT_SWISH>), grid, block, 0, s, d, ntiles, ngroups);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

struct DwGeom {
  int nw, gpb, wgroups, units_per_batch, nblk, w_global;
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
  g->nw = d.gw > 1 ? 1 : variant_nw(d);
  g->gpb = kThreads / CG;
  g->wgroups = (d.Wo + g->nw - 1) / g->nw;
  const long upb = (long)d.To * d.Ho * g->wgroups;
  if (upb > 0x7fffffffL) return false;
  g->units_per_batch = (int)upb;
  g->nblk = (int)pv_ceil_div(upb, g->gpb);
  const int w_p = d.w_mod > 0 ? pv_round_up(d.w_mod, 8) : c_p;
  const size_t w_bytes = sizeof(float) * (size_t)d.kt * d.kh * d.kw * (d.gw > 1 ? d.gw : 1) * w_p;
  const size_t red_bytes = sizeof(float) * (d.psum ? (size_t)g->gpb * c_p : 0);
  g->w_global = w_bytes + red_bytes > 96 * 1024;
  g->lds = (g->w_global ? 0 : w_bytes) + red_bytes;
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
  PV_LAUNCH(kern, grid, block, g.lds, s, d, g.gpb, g.wgroups, g.units_per_batch, g.w_global);
  pv_note_kernel("dwconv_kernel");   // (launched through a function pointer: PV_LAUNCH saw only the variable)
  PV_LAUNCH_CHECK();
  return PV_OK;
}

template <typename T> int launch_dw(const pv_dwconv3d_desc& d, const DwGeom& g, hipStream_t s) {
  if (d.n_prefix > 0) {
    const int total = d.B * d.n_prefix * (pv_round_up(d.C, 8) / 8);
    PV_LAUNCH(dw_prefix_kernel<T>, dim3((unsigned)pv_ceil_div(total, kThreads)), dim3(kThreads), 0, s, d);
    PV_LAUNCH_CHECK();
  }
  if (d.gw > 1) return launch_variant<T, 0, 1, 1>(d, g, s);   // channel-wise grouped: the runtime variant
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
  if (d.gw < 0 || (d.gw > 1 && d.gw != 2 && d.gw != 4 && d.gw != 8)) return PV_ERR_UNSUPPORTED;
  if (d.gw > 1 && (d.C % d.gw || d.w_mod > 0 || d.pw_w != nullptr)) return PV_ERR_INVALID;
  return PV_OK;
}

}  // namespace

int pv_plane_variant(const pv_dwconv3d_desc& d) { return plane_variant(d); }
int pv_plane_tiles(const pv_dwconv3d_desc& d, int nw) { return plane_tiles(d, nw); }
int pv_pwdw_supported(const pv_dwconv3d_desc& d);           // pv_pwdw.hip
int pv_pwdw_launch(const pv_dwconv3d_desc& d, hipStream_t s);  // pv_pwdw.hip

extern "C" int pv_dwconv3d_pw_supported(const pv_dwconv3d_desc* d) {
  if (!d || d->B <= 0 || d->C <= 0 || d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
  return pv_pwdw_supported(*d);
}

extern "C" int pv_dwconv3d_psum_blocks(const pv_dwconv3d_desc* d) {
  if (!d) return PV_ERR_INVALID;
  if (d->B <= 0 || d->C <= 0 || d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return PV_ERR_INVALID;
  if (const int nw = plane_variant(*d)) return plane_tiles(*d, nw);  // pointers are not needed for the count
  DwGeom g;
  if (!geom(*d, &g)) return PV_ERR_UNSUPPORTED;
  return g.nblk;
}

extern "C" int pv_dwconv3d(const pv_dwconv3d_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const pv_dwconv3d_desc& d = *dp;
  const int v = validate(d);
  if (v != PV_OK) return v;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d.pw_w) return pv_pwdw_launch(d, s);   // fused conv_a -> conv_b; no unfused fallback inside the library
  if (const int nw = plane_variant(d)) {
    if (d.sw == 1) return nw == 4 ? launch_plane<1, 4>(d, s) : launch_plane<1, 2>(d, s);
    return nw == 4 ? launch_plane<2, 4>(d, s) : launch_plane<2, 2>(d, s);
  }
  DwGeom g;
  if (!geom(d, &g)) return PV_ERR_UNSUPPORTED;
  if (d.dtype == PV_BF16) return launch_dw<bf16_t>(d, g, s);
  if (d.dtype == PV_F32) return launch_dw<float>(d, g, s);
  return PV_ERR_UNSUPPORTED;
}
