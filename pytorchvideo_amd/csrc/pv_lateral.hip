// SlowFast's lateral connection as its own kernel: FuseFastToSlow.forward (models/slowfast.py:720-729) as
// built by FastToSlowFusionBuilder.create_module (models/slowfast.py:661-694):
//
//   fuse        = act(BN(Conv3d(C_f -> 2 C_f, kernel (kt,1,1), stride (alpha,1,1), padding (kt/2,0,0), bias=False)(x_fast)))
//   x_slow_fuse = cat([x_slow, fuse], dim=1)
//
// The conv is a time-strided gather of kt fast frames per slow frame at ONE spatial position: no spatial
// halo, K = kt * C_f, and (8 -> 16 ... 128 -> 256 channels) an arithmetic intensity of 40-300 FLOP/B -- an
// HBM-bound stream, not a tiled GEMM.  So the kernel is a streaming pass with an MFMA in the middle:
//   * the filter slab of a workgroup (<= 128 output channels x kt*C_f, at most ~116 KB) is staged into LDS
//     once; the workgroup then walks 16-voxel tiles of the SLOW grid in a grid-stride loop, no barriers;
//   * the activation operand goes global -> MFMA B registers directly: for K step s lane (n = lane & 15,
//     q = lane >> 4) loads the 16 bytes x_fast[b][alpha*t_s - kt/2 + tap][h][w][c .. c+7] with (tap, c) the
//     position of k = 32 s + 8 q in the tap-major K order -- a per-lane frame offset from a tiny LDS table,
//     frames outside the clip are the conv's zero padding (buffer addressing: out-of-range offset reads 0);
//   * filter rows are permuted in LDS so that a lane's accumulators are 8 CONSECUTIVE output channels of its
//     voxel: BN scale/shift + ReLU in registers, one 16-byte store per lane STRAIGHT INTO THE CHANNEL SLICE
//     [C_slow, C_slow + 2 C_f) of the slow pathway's buffer -- the reference's torch.cat (a full copy of the
//     slow tensor) does not exist;
//   * each fast frame is needed by up to two slow frames (kt = 7, alpha = 4: 7 reads per 4 frames); tiles
//     are ordered frame-major inside a clip, so the second use hits in L2 / Infinity Cache.
//
// The same streaming structure serves every NARROW dense convolution of the path -- SlowFast's fast pathway
// (conv_a (3,1,1) and conv_b (1,3,3) on 8-64 channels), the narrow ends of CSN / R(2+1)D: K = taps x cin of a
// few hundred, <= 128 output channels, 16-64 bytes per voxel.  Those are HBM streams as well (a c8 -> c8 1x3x3 layer
// moves 67 MB for 10 GFLOP), which a 128-wide LDS-tiled GEMM serves at 0.75-1.8 TB/s; here the tap is a per-lane
// (dt, dh, dw) offset with bounds tests instead of a frame offset only (pv_tapstream_try, called by pv_conv3d).
#include "pv_common.h"

namespace {

constexpr int kLatThreads = 512;   // 8 waves share one filter slab

struct TapConv {       // the convolution the kernel evaluates (a lateral connection is kh = kw = sh = sw = 1)
  const void* x; const void* w; void* y;
  const float* scale; const float* shift;
  long x_bs, y_bs;
  int ldx, ldy;
  int B, Ti, Hi, Wi, cin;
  int To, Ho, Wo, cout;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int act;
  // a pointwise conv behind this one in the same launch (pv_conv3d_desc.pw2_*: conv_b -> conv_c of a bottleneck): then y,
  // y_bs, ldy and the residual describe ITS cout2-channel output, and this conv's own output exists only as MFMA operands
  const void* w2; const float* scale2; const float* shift2; const void* r;
  long r_bs;
  int ldr, cout2, act2;
};

struct LatTile {
  unsigned xoff;   // byte offset of x[b][t0][h0][w0][0] (may be "negative": wraps, fixed by the tap offset)
  int t0, h0, w0;  // input coordinates of tap (0,0,0): stride * output coordinate - padding
  unsigned yoff;   // byte offset of y[b][to][ho][wo][n0]
  bool ok;
};

// NT: 16-channel MFMA row tiles per workgroup (slab = NT * 16 output channels); TM: voxel tiles per wave tile
// PW2: a pointwise conv behind the tap conv (conv_b -> conv_c of a bottleneck).  The epilogue's registers -- lane (n16, q) holds
// channels 32 p + 8 q .. + 7 of voxel n16 after BatchNorm and the activation, rounded to bf16 -- ARE the B operand of a
// 16 x 16 x 32 MFMA for K step p of the second product, so the inner tensor goes from accumulators to operands without
// leaving the lane: no LDS exchange, no HBM round trip.  W2 sits in LDS behind the first slab (rows permuted the same way, so
// the second epilogue again owns 8 consecutive channels per lane); the second product runs 32 output channels at a time with
// the residual of the next 32 requested before the MFMAs of the current ones.  nsplit is 1 in this mode.
template <int NT, int TM, bool PW2>
__global__ __launch_bounds__(kLatThreads, (PW2 && NT == 2) ? 4 : 1) void tap_stream_kernel(const TapConv d, int ksteps, int ngroups,
                                                                   int nchunks, int nsplit, int xcont) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int Kp = ksteps * 32;
  const int WLD = Kp + 8;   // row stride in elements: 16 B x odd -> conflict-free ds_read_b128
  bf16_t* w_s = reinterpret_cast<bf16_t*>(smem_raw);
  float* sc_s = reinterpret_cast<float*>(smem_raw + (size_t)NT * 16 * WLD * 2);
  float* sh_s = sc_s + NT * 16;
  int2* tap_s = reinterpret_cast<int2*>(sh_s + NT * 16);   // [ksteps][4]: (dt | dh << 8 | dw << 16, byte offset of tap + channel)
  // PW2: [pairs2 * 32][W2LD] weights, then scale / shift of the second conv
  constexpr int W2LD = NT * 16 + 8;
  const int pairs2 = PW2 ? (pv_round_up(d.cout2, 8) + 31) / 32 : 0;
  bf16_t* w2_s = reinterpret_cast<bf16_t*>(smem_raw + (((size_t)NT * 16 * WLD * 2 + (size_t)2 * NT * 16 * 4 + (size_t)ksteps * 4 * 8 + 15) & ~(size_t)15));
  float* sc2_s = reinterpret_cast<float*>(w2_s + (size_t)pairs2 * 32 * W2LD);
  float* sh2_s = sc2_s + pairs2 * 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int cin_p = d.cin;                 // multiple of 8
  const int cout_p8 = pv_round_up(d.cout, 8);
  const int HWo = d.Ho * d.Wo;
  const long M = (long)d.B * d.To * HWo;
  const int taps = d.kt * d.kh * d.kw;
  // the N splits of one voxel chunk are consecutive workgroups of one XCD: the activations are fetched from
  // HBM once and re-read from that XCD's L2 by the other splits.  XCONT (round 6): an XCD takes a CONTIGUOUS run of
  // chunks -- the consumers of an input frame (the kt output frames around it) or of an input row (the kh output rows)
  // are neighbouring chunks, and dealt round-robin they sat on kt / kh different XCDs, each filling its own L2 with
  // the same lines (PMC round 5: the fast pathway's (3,1,1) layers fetched 3.8 x their input)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chunk = xcont ? xcd * ((nchunks + 7) >> 3) + slot / nsplit : (slot / nsplit) * 8 + xcd;
  const int n0 = (slot % nsplit) * (NT * 16);

  // ---- stage the filter slab (LDS row r = channel n0 + perm(r)), BN scale / shift, the tap table ----
  {
    const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
    const int K = taps * cin_p;
    const int cpr = Kp / 8;
    for (int id = tid; id < NT * 16 * cpr; id += kLatThreads) {
      const int r = id / cpr, kc = id - r * cpr;
      const int tn = r >> 4, ii = r & 15;
      const int c = n0 + (tn >> 1) * 32 + (ii >> 2) * 8 + (tn & 1) * 4 + (ii & 3);
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < d.cout && kc * 8 < K) v = *reinterpret_cast<const bf16x8*>(Wt + (long)c * K + kc * 8);
      *reinterpret_cast<bf16x8*>(w_s + r * WLD + kc * 8) = v;
    }
    for (int i = tid; i < NT * 16; i += kLatThreads) {
      const int c = n0 + i;
      const bool ok = c < d.cout;
      sc_s[i] = ok ? (d.scale ? d.scale[c] : 1.f) : 0.f;
      sh_s[i] = ok ? (d.shift ? d.shift[c] : 0.f) : 0.f;
    }
    for (int i = tid; i < ksteps * 4; i += kLatThreads) {
      const int k = (i >> 2) * 32 + (i & 3) * 8;
      const int tap = k / cin_p, c = k - tap * cin_p;
      const int dt = tap / (d.kh * d.kw), r2 = tap - dt * d.kh * d.kw, dh = r2 / d.kw, dw = r2 - dh * d.kw;
      // taps past the last one (the zero-padded tail of the last K step) are marked invalid
      tap_s[i] = tap < taps ? int2{dt | (dh << 8) | (dw << 16),
                                   (int)((((unsigned)dt * (unsigned)d.Hi + (unsigned)dh) * (unsigned)d.Wi + (unsigned)dw) *
                                             (unsigned)d.ldx * 2u + (unsigned)c * 2u)}
                            : int2{-1, 0};
    }
    if constexpr (PW2) {
      const bf16_t* __restrict__ W2 = static_cast<const bf16_t*>(d.w2);
      const int K2 = cout_p8;              // W2 is [cout2][round_up(cout, 8)]
      constexpr int cpr2 = NT * 16 / 8;
      for (int id = tid; id < pairs2 * 32 * cpr2; id += kLatThreads) {
        const int r = id / cpr2, kc = id - r * cpr2;
        const int tn = r >> 4, ii = r & 15;
        const int c = (tn >> 1) * 32 + (ii >> 2) * 8 + (tn & 1) * 4 + (ii & 3);
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c < d.cout2 && kc * 8 < K2) v = *reinterpret_cast<const bf16x8*>(W2 + (long)c * K2 + kc * 8);
        *reinterpret_cast<bf16x8*>(w2_s + r * W2LD + kc * 8) = v;
      }
      for (int i = tid; i < pairs2 * 32; i += kLatThreads) {
        const bool ok = i < d.cout2;
        sc2_s[i] = ok ? (d.scale2 ? d.scale2[i] : 1.f) : 0.f;
        sh2_s[i] = ok ? (d.shift2 ? d.shift2[i] : 0.f) : 0.f;
      }
    }
  }
  __syncthreads();

  constexpr unsigned kOOB = 0x80000000u;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(d.x), 0, (int)((unsigned)d.B * (unsigned)d.x_bs * 2u), 0x00020000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      d.y, 0, (int)((unsigned)d.B * (unsigned)d.y_bs * 2u), 0x00020000);
  const bool has_r = PW2 && d.r != nullptr;
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(has_r ? d.r : d.x), 0, has_r ? (int)((unsigned)d.B * (unsigned)d.y_bs * 2u) : 0, 0x00020000);
  const int live_pairs = min(NT / 2, (cout_p8 - n0 + 31) / 32);   // wave-uniform
  constexpr int NP = NT / 2;
  constexpr int NWV = kLatThreads / 64;

  auto tile_of = [&](int g, LatTile (&t)[TM]) {
    const long m_base = ((long)g * NWV + wave) * (TM * 16);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long m = m_base + i * 16 + n16;
      const bool ok = g < ngroups && m < M;
      const unsigned mm = ok ? (unsigned)m : 0u;
      const unsigned bt = mm / (unsigned)HWo, hw = mm - bt * (unsigned)HWo;   // bt = b * To + to
      const unsigned b = bt / (unsigned)d.To, to = bt - b * (unsigned)d.To;
      const unsigned ho = hw / (unsigned)d.Wo, wo = hw - ho * (unsigned)d.Wo;
      t[i].ok = ok;
      t[i].t0 = (int)to * d.st - d.pt;
      t[i].h0 = (int)ho * d.sh - d.ph;
      t[i].w0 = (int)wo * d.sw - d.pw;
      t[i].xoff = (unsigned)((long)b * d.x_bs + (((long)t[i].t0 * d.Hi + t[i].h0) * d.Wi + t[i].w0) * d.ldx) * 2u;
      t[i].yoff = (unsigned)((long)b * d.y_bs + ((long)to * HWo + hw) * d.ldy + n0) * 2u;
    }
  };
  auto load_x = [&](u32x4 (&dst)[TM], const LatTile (&t)[TM], int ks) {
    const int2 tp = tap_s[ks * 4 + q];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const bool ok = t[i].ok && tp.x >= 0 && (unsigned)(t[i].t0 + (tp.x & 255)) < (unsigned)d.Ti &&
                      (unsigned)(t[i].h0 + ((tp.x >> 8) & 255)) < (unsigned)d.Hi && (unsigned)(t[i].w0 + (tp.x >> 16)) < (unsigned)d.Wi;
      dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok ? t[i].xoff + (unsigned)tp.y : kOOB), 0, 0);
    }
  };

  LatTile cur[TM], nxt[TM];
  u32x4 xf[TM], xn[TM];
  int g = chunk < nchunks ? chunk : ngroups;   // padded workgroups (grid is a multiple of 8 chunks) do nothing
  tile_of(g, cur);
  load_x(xf, cur, 0);
  for (; g < ngroups; g += nchunks) {
    f32x4 acc[NT][TM];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[a][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    tile_of(g + nchunks, nxt);
    for (int ks = 0; ks < ksteps; ++ks) {
      // the next K step's operand (or the next tile's first one) is requested before this step's MFMAs
      if (ks + 1 < ksteps) load_x(xn, cur, ks + 1);
      else load_x(xn, nxt, 0);
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        if ((a >> 1) < live_pairs) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w_s + (a * 16 + n16) * WLD + ks * 32 + q * 8);
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[a][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, xf[i]), acc[a][i], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) xf[i] = xn[i];
    }
    // ---- epilogue: lane (n16, q) owns channels n0 + 32 p + 8 q .. + 7 of its voxel ----
    bf16x8 mid[PW2 ? NP : 1][TM];
    if constexpr (PW2) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i) mid[p][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p >= live_pairs) break;
      const int cl = p * 32 + q * 8;
      const int c0 = n0 + cl;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc_s + cl), s1 = *reinterpret_cast<const f32x4*>(sc_s + cl + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh_s + cl), h1 = *reinterpret_cast<const f32x4*>(sh_s + cl + 4);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[2 * p][i][j] * s0[j] + h0[j];
          v[4 + j] = acc[2 * p + 1][i][j] * s1[j] + h1[j];
        }
        pv_apply_act_n<true>(v, d.act);
        if (c0 + 8 > d.cout) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (c0 + j >= d.cout) v[j] = 0.f;   // the padding up to the 8-multiple is written as zeros
        }
        bf16x8 ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) ob[j] = (bf16_t)v[j];
        if constexpr (PW2) {
          mid[p][i] = ob;      // channels past cout are zeros (zero weights, zero shift, zeroed above)
        } else {
          const bool ok = cur[i].ok && c0 < cout_p8;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ob), ry,
                                                 (int)(ok ? cur[i].yoff + (unsigned)cl * 2u : kOOB), 0, 0);
        }
      }
    }
    if constexpr (PW2) {
      // ---- the second product, 32 output channels per pass; lane (n16, q) owns channels 32 p2 + 8 q .. + 7 of its voxel ----
      const int cout2_p8 = pv_round_up(d.cout2, 8);
      u32x4 rq[TM], rn[TM];
      auto load_r = [&](u32x4 (&dst)[TM], int p2) {
        const int c0 = p2 * 32 + q * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          dst[i] = __builtin_amdgcn_raw_buffer_load_b128(
              rr, (int)(has_r && p2 < pairs2 && cur[i].ok && c0 < cout2_p8 ? cur[i].yoff + (unsigned)c0 * 2u : kOOB), 0, 0);
      };
      // (the residual has the output's strides: one offset serves both.  The narrow variant runs four waves per SIMD and
      // requests a pass's residual at the top of the pass; the wide one, two waves per SIMD, one pass ahead)
      constexpr bool kAhead = NT > 2;
      if (kAhead) load_r(rq, 0);
      for (int p2 = 0; p2 < pairs2; ++p2) {
        if (kAhead) load_r(rn, p2 + 1);
        else load_r(rq, p2);
        f32x4 a2[2][TM];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < TM; ++i) a2[h][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sp = 0; sp < NP; ++sp) {
          if (sp >= live_pairs) break;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w2_s + ((p2 * 2 + h) * 16 + n16) * W2LD + sp * 32 + q * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i) a2[h][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, mid[sp][i], a2[h][i], 0, 0, 0);
          }
        }
        const int cl = p2 * 32 + q * 8;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc2_s + cl), s1 = *reinterpret_cast<const f32x4*>(sc2_s + cl + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh2_s + cl), h1 = *reinterpret_cast<const f32x4*>(sh2_s + cl + 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float v[8];
          const bf16x8 rb = __builtin_bit_cast(bf16x8, rq[i]);   // zeros where there is no residual (out-of-range read)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = a2[0][i][j] * s0[j] + h0[j] + (float)rb[j];
            v[4 + j] = a2[1][i][j] * s1[j] + h1[j] + (float)rb[4 + j];
          }
          pv_apply_act_n<true>(v, d.act2);
          if (cl + 8 > d.cout2) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (cl + j >= d.cout2) v[j] = 0.f;
          }
          bf16x8 ob;
#pragma unroll
          for (int j = 0; j < 8; ++j) ob[j] = (bf16_t)v[j];
          const bool ok = cur[i].ok && cl < cout2_p8;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ob), ry,
                                                 (int)(ok ? cur[i].yoff + (unsigned)cl * 2u : kOOB), 0, 0);
        }
        if (kAhead) {
#pragma unroll
          for (int i = 0; i < TM; ++i) rq[i] = rn[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) cur[i] = nxt[i];
  }
}

template <int NT, int TM, bool PW2 = false>
int launch_lateral(const TapConv& d, int ksteps, size_t lds, int xcont, hipStream_t s) {
  const long M = (long)d.B * d.To * d.Ho * d.Wo;
  const long ngroups = pv_ceil_div(M, (kLatThreads / 64) * TM * 16);
  const int nsplit = (int)pv_ceil_div(pv_round_up(d.cout, 8), NT * 16);
  if (PW2 && nsplit != 1) return PV_ERR_UNSUPPORTED;
  auto kern = tap_stream_kernel<NT, TM, PW2>;
  if (lds > 64 * 1024)
    PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // one resident generation of workgroups; the rest is the grid-stride loop (the slab is staged once per workgroup)
  long per_cu = (160 * 1024) / (long)(lds + 1024);
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
  long nchunks = pv_ceil_div(256 * per_cu, nsplit);
  if (nchunks > ngroups) nchunks = ngroups;
  const long blocks = pv_ceil_div(nchunks, 8) * 8 * nsplit;
  PV_LAUNCH(kern, dim3((unsigned)blocks), dim3(kLatThreads), lds, s, d, ksteps, (int)ngroups, (int)nchunks, nsplit, xcont);
  pv_note_kernel("tap_stream_kernel");   // (launched through a function pointer: PV_LAUNCH saw only the variable)
  PV_LAUNCH_CHECK();
  return PV_OK;
}

// PV_OK: launched; PV_ERR_UNSUPPORTED: not a geometry of the streaming kernel (nothing launched)
// xcont: chunk -> XCD mapping (see the kernel).  Contiguous runs for the convolutions (same-box A/B on SlowFast-R50: the fast
// pathway's res4 conv_a 27 -> 17 us, res3 30 -> 28 us, model +0.6 %); the time-strided lateral connections measured 3-6 % slower
// with it at 64^2 and equal elsewhere, and keep the round-robin deal.
// dry: geometry check only (pv_conv3d_pw2_supported), nothing is launched
int tapstream_launch(const TapConv& d, int k_limit, int xcont, hipStream_t s, bool dry = false) {
  const int taps = d.kt * d.kh * d.kw;
  const long K = (long)taps * d.cin;
  if (K > k_limit || d.kt > 255 || d.kh > 255 || d.kw > 255) return PV_ERR_UNSUPPORTED;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const int ksteps = (int)((K + 31) / 32);
  const bool small_offsets = (long)d.B * d.x_bs * 2 <= 0x7fffffffL && (long)d.B * d.y_bs * 2 <= 0x7fffffffL &&
                             (long)d.B * d.To * d.Ho * d.Wo <= 0x7fffffffL;
  if (!small_offsets) return PV_ERR_UNSUPPORTED;
  int nt = cout_p8 <= 32 ? 2 : (cout_p8 <= 64 ? 4 : 8);
  auto lds_of = [&](int t) { return (size_t)t * 16 * (ksteps * 32 + 8) * 2 + (size_t)2 * t * 16 * 4 + (size_t)ksteps * 4 * 8; };
  if (d.cout2 > 0) {
    // the second conv's slab behind the first: the first conv must produce all of its channels in one workgroup
    if (cout_p8 > 64 || d.cout2 > 1024 || (d.r && (d.r_bs != d.y_bs || d.ldr != d.ldy))) return PV_ERR_UNSUPPORTED;
    const size_t pairs2 = (size_t)(pv_round_up(d.cout2, 8) + 31) / 32;
    const size_t lds = ((lds_of(nt) + 15) & ~(size_t)15) + pairs2 * 32 * (nt * 16 + 8) * 2 + 2 * pairs2 * 32 * 4;
    if (lds > 150 * 1024) return PV_ERR_UNSUPPORTED;
    if (dry) return PV_OK;
    if (nt == 2) return launch_lateral<2, 2, true>(d, ksteps, lds, xcont, s);
    return launch_lateral<4, 2, true>(d, ksteps, lds, xcont, s);
  }
  while (nt > 2 && lds_of(nt) > 120 * 1024) nt >>= 1;
  if (lds_of(nt) > 120 * 1024) return PV_ERR_UNSUPPORTED;
  const size_t lds = lds_of(nt);
  if (nt == 2) return launch_lateral<2, 2>(d, ksteps, lds, xcont, s);
  if (nt == 4) return launch_lateral<4, 2>(d, ksteps, lds, xcont, s);
  return launch_lateral<8, 1>(d, ksteps, lds, xcont, s);
}

}  // namespace

// Narrow dense convolutions of pv_conv3d (called before its generic kernel): bf16, no residual / gate / second operand,
// no dilation, <= 128 output channels, K = taps * cin <= 640.  Returns PV_ERR_UNSUPPORTED for everything else.
// With pw2_w set (a pointwise conv behind it, <= 64 channels in between): the residual belongs to the second conv.
// dry: the second-conv mode's geometry check (pv_conv3d_pw2_supported: pointers are ignored, pw2_cout > 0 selects the mode)
int pv_tapstream_try(const pv_conv3d_desc& c, hipStream_t s, bool dry) {
  const bool pw2 = dry ? c.pw2_cout > 0 : c.pw2_w != nullptr;
  if (c.dtype != PV_BF16 || c.y_f32 || c.a_gate || c.a_act != PV_ACT_NONE || c.x2 || c.dwt_w || c.pos_spatial)
    return PV_ERR_UNSUPPORTED;
  if (c.residual && (!pw2 || c.r_f32)) return PV_ERR_UNSUPPORTED;
  if (pw2 && (c.pw2_cout <= 0 || (!dry && c.ldy < pv_round_up(c.pw2_cout, 8)))) return PV_ERR_UNSUPPORTED;
  if (c.dil_t > 1 || c.dil_h > 1 || c.dil_w > 1 || c.cin % 8 || pv_round_up(c.cout, 8) > 128) return PV_ERR_UNSUPPORTED;
  if (!pw2 && !pv_tune("tapstream", 1)) return PV_ERR_UNSUPPORTED;
  TapConv d;
  d.x = c.x; d.w = c.w; d.y = c.y; d.scale = c.scale; d.shift = c.shift;
  d.x_bs = c.x_bs; d.y_bs = c.y_bs; d.ldx = c.ldx; d.ldy = c.ldy;
  d.B = c.B; d.Ti = c.Ti; d.Hi = c.Hi; d.Wi = c.Wi; d.cin = c.cin;
  d.To = c.To; d.Ho = c.Ho; d.Wo = c.Wo; d.cout = c.cout;
  d.kt = c.kt; d.kh = c.kh; d.kw = c.kw; d.st = c.st; d.sh = c.sh; d.sw = c.sw; d.pt = c.pt; d.ph = c.ph; d.pw = c.pw;
  d.act = c.act;
  d.w2 = pw2 ? c.pw2_w : nullptr; d.scale2 = c.pw2_scale; d.shift2 = c.pw2_shift; d.r = pw2 ? (dry && !c.residual && c.ldr > 0 ? static_cast<const void*>(&c) /* geometry check: a residual is announced by its strides */ : c.residual) : nullptr;
  d.r_bs = c.r_bs; d.ldr = c.ldr; d.cout2 = pw2 ? c.pw2_cout : 0; d.act2 = c.pw2_act;
  return tapstream_launch(d, 640, pv_tune("tap_xcont", 1), s, dry);
}

extern "C" int pv_lateral_fuse(const pv_lateral_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const pv_lateral_desc& l = *dp;
  if (!l.x || !l.w || !l.y) return PV_ERR_INVALID;
  if (l.B <= 0 || l.Ti <= 0 || l.H <= 0 || l.W <= 0 || l.cin <= 0 || l.cout <= 0 || l.To <= 0) return PV_ERR_INVALID;
  if (l.kt < 1 || l.st < 1 || l.pt < 0 || l.cin % 8 || l.ldx % 8 || l.ldy % 8 || l.x_bs % 8 || l.y_bs % 8) return PV_ERR_INVALID;
  if ((l.Ti + 2 * l.pt - l.kt) / l.st + 1 != l.To || l.ldx < l.cin || l.ldy < pv_round_up(l.cout, 8)) return PV_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Where the op stops being a stream: K = kt * cin >= 448 (SlowFast-R50's 64 -> 128 and 128 -> 256 sites, 150-300
  // FLOP/B) is MFMA work whose operands want full 128-byte lines staged through LDS; measured on the R50 sites
  // (16 clips, same box): streaming kernel 26 / 61 / 60 / 59 us, LDS-DMA implicit GEMM 38 / 87 / 53 / 41 us.
  if (l.dtype == PV_BF16) {
    TapConv d;
    d.x = l.x; d.w = l.w; d.y = l.y; d.scale = l.scale; d.shift = l.shift;
    d.x_bs = l.x_bs; d.y_bs = l.y_bs; d.ldx = l.ldx; d.ldy = l.ldy;
    d.B = l.B; d.Ti = l.Ti; d.Hi = l.H; d.Wi = l.W; d.cin = l.cin;
    d.To = l.To; d.Ho = l.H; d.Wo = l.W; d.cout = l.cout;
    d.kt = l.kt; d.kh = 1; d.kw = 1; d.st = l.st; d.sh = 1; d.sw = 1; d.pt = l.pt; d.ph = 0; d.pw = 0;
    d.act = l.act;
    d.w2 = nullptr; d.scale2 = nullptr; d.shift2 = nullptr; d.r = nullptr; d.r_bs = 0; d.ldr = 0; d.cout2 = 0; d.act2 = 0;
    const int r = tapstream_launch(d, 256, pv_tune("lat_xcont", 0), s);
    if (r != PV_ERR_UNSUPPORTED) return r;
  }
  // fp32 parity mode, MFMA-bound widths and geometries outside the streaming kernel's range: the same arithmetic
  // as a (kt,1,1) convolution through the dense-conv entry point -- still the HIP library, never a host path
  pv_conv3d_desc c = {};
  c.x = l.x; c.w = l.w; c.y = l.y; c.scale = l.scale; c.shift = l.shift;
  c.x_bs = l.x_bs; c.y_bs = l.y_bs; c.ldx = l.ldx; c.ldy = l.ldy;
  c.B = l.B; c.Ti = l.Ti; c.Hi = l.H; c.Wi = l.W; c.cin = l.cin;
  c.To = l.To; c.Ho = l.H; c.Wo = l.W; c.cout = l.cout;
  c.kt = l.kt; c.kh = 1; c.kw = 1; c.st = l.st; c.sh = 1; c.sw = 1; c.pt = l.pt;
  c.act = l.act; c.dtype = l.dtype;
  return pv_conv3d(&c, stream);
}
