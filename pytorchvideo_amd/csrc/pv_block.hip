// A whole X3D bottleneck in ONE launch (gfx950), round 6:
//
//   y = act_out( r + sc * (Wc . act_b( sb * dw3x3x3( act_a( sa * (Wa . x) + ha ) ) + hb )) + hc )
//
// i.e. conv_a + norm_a + ReLU -> depthwise conv_b + norm_b + Swish -> conv_c + norm_c -> + shortcut -> ReLU of
// pytorchvideo/models/x3d.py:169-212 (create_x3d_bottleneck_block) / models/resnet.py:1345-1365 (BottleneckBlock.forward)
// and :1179-1189 (ResBlock.forward) for the blocks WITHOUT squeeze-excitation (every second block of a stage): the
// expanded tensor (2.25 x the block width) is never written.  As three launches an X3D-M res4 block moves
// 19 + 43 | 43 + 43 | 43 + 19 + 19 MB and takes 18 + 34 + 26 us on 100 k voxels -- every one of them latency-bound on a
// 14 x 14 map; here the block reads x once (plus the halo rows) and writes y once.
//
// Mapping on CDNA4 (one workgroup = 8 waves = one clip x TWO output rows x the full width, walking T; two waves per SIMD with
// DIFFERENT roles, so that the matrix pipe and the VALU work at the same time -- measured with every wave doing every phase in
// turn, one wave per SIMD: 74 us per block, no faster than the three launches):
//   MATRIX waves (0-3), iteration i:
//   [A] conv_a for the 4 halo rows of input plane i (wave w = halo row w = one 16-voxel MFMA column tile, all CP/16 channel
//       tiles; v_mfma_f32_16x16x32_bf16 with the filter as A operand from LDS and the voxels' input channels as B operand
//       straight from global memory), BN + ReLU, rounded to bf16 exactly where the unfused path stores it, written into the LDS
//       plane E[i & 1][voxel][channel]; voxels outside the image are written as ZERO (conv_b zero-pads the EXPANDED tensor);
//   [C] conv_c of output plane i-3 from MID[(i-3) & 1] (B operand) and the filter streamed from L2 into registers one
//       iteration ahead (A operand; both filters + two E planes + two MID planes do not fit 160 KB): 12 tiles of 16 channels
//       x 16 voxels, 3 per wave; BN + residual + ReLU in the epilogue, 8 bytes per lane to global memory.
//   STENCIL waves (4-7), iteration i:
//   [B] the 3x3x3 stencil of input plane i-1 on the VALU: thread = (channel pair, output row), a full row of 14 outputs per
//       thread, the 27 x 2 taps in registers for the whole kernel, three rolling accumulator sets over T (plane p feeds outputs
//       p-1, p, p+1), so every plane is produced once and read once; BN + Swish of the finished output plane i-2 -> bf16 ->
//       LDS MID[(i-2) & 1][voxel][channel].
//   One barrier per iteration.  Halo rows are recomputed by the neighbouring workgroup (4 rows of conv_a per 2 rows of output).
#include "pv_common.h"

namespace {

constexpr int kTH = 2;            // output rows per workgroup
constexpr int kHR = kTH + 2;      // halo rows of the expanded tensor per plane
constexpr int kWP = 16;           // voxel slots per plane row in LDS (image columns 0 .. 15)
constexpr int kNW = 14;           // outputs per stencil thread = the widest supported map

template <int CP, int KSA, int COUT> struct BlkGeom {
  static constexpr int NP = CP / 2;                 // channel pairs
  static constexpr int WPR = (NP + 63) / 64;        // stencil waves per output row
  static constexpr int SWAVES = kTH * WPR;          // stencil waves
  static constexpr int MWAVES = kHR * kWP / 16;     // matrix waves: one 16-voxel producer tile (= one halo row) each
  static constexpr int MTA = CP / 16;               // conv_a channel tiles
  static constexpr int KSC = CP / 32;               // conv_c K steps
  static constexpr int MTC = COUT / 16;             // conv_c channel tiles
  static constexpr int TPW = MTC * kTH / MWAVES;    // conv_c tiles per matrix wave
  static constexpr int E_STRIDE = CP * 2 + 8;       // bytes per voxel of E (+8: the producer's 8-byte stores of 16 voxels hit distinct banks)
  static constexpr int MID_STRIDE = CP * 2 + 16;    // bytes per voxel of MID (+16: conflict-free ds_read_b128 of 16 voxels)
  static constexpr int WA_BYTES = MTA * KSA * 1024;
  static constexpr int E_BYTES = kHR * kWP * E_STRIDE;
  static constexpr int MID_BYTES = kTH * kWP * MID_STRIDE;
  static constexpr int PAR_BYTES = (2 * CP + 2 * COUT) * 4;     // sa | ha | sc | hc
  static constexpr int OUT_STRIDE = COUT * 2 + 8;   // bytes per voxel of the output staging planes (+8: the epilogue's 8-byte stores hit distinct banks)
  static constexpr int OUT_BYTES = kTH * kWP * OUT_STRIDE;
  static constexpr int TOTAL = WA_BYTES + 2 * E_BYTES + 2 * MID_BYTES + PAR_BYTES + 2 * OUT_BYTES;
  static_assert(MWAVES == 4 && SWAVES == 4, "four matrix waves + four stencil waves");
  static_assert(MTC * kTH % MWAVES == 0, "conv_c tiles split evenly over the matrix waves");
  static_assert(TOTAL <= 160 * 1024, "LDS");
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((vector_size(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2_t pk = {(bf16_t)a, (bf16_t)b};
  return __builtin_bit_cast(unsigned, pk);
}

// MODE 0: the whole block.  MODE 1 (blocks WITH squeeze-excitation): stop behind conv_b -- the stencil waves store
// sb * dw(...) + hb (no activation: it is applied with the gate by conv_c's operand load) as bf16 to y (B, T, H, W, C) and add
// the fp32 values up for the squeeze: psum[b][2 * tile + row][c] = the sum over this workgroup's T x 1 x W output voxels of
// channel c (deterministic: one writer per entry); the matrix waves run conv_a only.
// ABL (development variant of the library, timing only, WRONG results): 1 no stencil FMAs, 2 no conv_a MFMAs, 3 no conv_c, 4 no
// barriers, 5 no filter-fragment / residual loads, 6 no stencil activation
template <int CP, int KSA, int COUT, int ACT_B, int MODE, int ABL = 0>
__global__ __launch_bounds__(512, 2) void bottleneck_block_kernel(const pv_bottleneck_desc d, int tiles_h) {
  using G = BlkGeom<CP, KSA, COUT>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::TOTAL];
  unsigned char* const s_wa = smem;
  unsigned char* const s_e = s_wa + G::WA_BYTES;            // two planes
  unsigned char* const s_mid = s_e + 2 * G::E_BYTES;        // two planes
  float* const s_sa = reinterpret_cast<float*>(s_mid + 2 * G::MID_BYTES);
  float* const s_ha = s_sa + CP;
  float* const s_sc = s_ha + CP;
  float* const s_hc = s_sc + COUT;
  unsigned char* const s_out = reinterpret_cast<unsigned char*>(s_hc + COUT);      // two planes of finished output rows

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / tiles_h;
  const int h0 = (blockIdx.x - b * tiles_h) * kTH;
  const int T = d.T, H = d.H, W = d.W;

  // ---- conv_a's filter and the folded BatchNorms -> LDS (once per workgroup; L2-resident after the first workgroups) ----
  {
    const u32x4* wa = static_cast<const u32x4*>(d.wa);
    u32x4* la = reinterpret_cast<u32x4*>(s_wa);
    for (int i = tid; i < G::WA_BYTES / 16; i += 512) la[i] = wa[i];
    for (int i = tid; i < CP; i += 512) { s_sa[i] = d.sa[i]; s_ha[i] = d.ha[i]; }
    if constexpr (MODE == 0)
      for (int i = tid; i < COUT; i += 512) { s_sc[i] = d.sc[i]; s_hc[i] = d.hc[i]; }
  }
  __syncthreads();

  if (wave < G::MWAVES) {
    // =========================================== MATRIX waves ===========================================
    constexpr unsigned kOOB = 0x80000000u;
    // producer role: halo row `wave`, voxel column n16, k-group q
    const bf16_t* X = static_cast<const bf16_t*>(d.x) + (long)b * d.x_bs;
    const unsigned plane_b = (unsigned)(H * W * d.ldx) * 2u;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(plane_b * (unsigned)T), 0x00020000);
    const int hh = h0 - 1 + wave;
    const bool vox_ok = (unsigned)hh < (unsigned)H && n16 < W;
    const unsigned x_off = vox_ok ? (unsigned)((hh * W + n16) * d.ldx + q * 8) * 2u : kOOB;
    const unsigned e_mask = vox_ok ? 0xffffffffu : 0u;
    const int e_dst = (wave * kWP + n16) * G::E_STRIDE + q * 8;      // + plane buffer + mt * 32
    // conv_c role: output row `cnt`, channel tiles cmt0 .. cmt0 + TPW - 1
    const int cnt = wave % kTH, cmt0 = (wave / kTH) * G::TPW;
    const int oh = h0 + cnt;
    const bool o_ok = oh < H && n16 < W;
    const bf16_t* R = static_cast<const bf16_t*>(d.residual) + (long)b * d.r_bs;
    bf16_t* Yp = static_cast<bf16_t*>(d.y) + (long)b * d.y_bs;
    const unsigned rplane_b = (unsigned)(H * W * d.ldr) * 2u, yplane_b = (unsigned)(H * W * d.ldy) * 2u;
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, (int)(rplane_b * (unsigned)T), 0x00020000);
    __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)Yp, 0, (int)(yplane_b * (unsigned)T), 0x00020000);
    const unsigned r_off = o_ok ? (unsigned)((oh * W + n16) * d.ldr + cmt0 * 16 + q * 4) * 2u : kOOB;
    const unsigned y_off = o_ok ? (unsigned)((oh * W + n16) * d.ldy + cmt0 * 16 + q * 4) * 2u : kOOB;
    const bool has_res = d.residual != nullptr;
    const bool out_relu = d.act_out == PV_ACT_RELU;
    const bool a_relu = d.act_a == PV_ACT_RELU;
    // conv_c's filter fragments of this wave: the same 21 KB every plane, from L2 (it stays hot: every workgroup reads it)
    const u32x4* wcg = static_cast<const u32x4*>(d.wc) + (long)cmt0 * G::KSC * 64 + lane;

    // conv_c's filter fragments of this wave stay in REGISTERS for the whole kernel (84 of them: the matrix waves hold no
    // stencil state), loaded once from L2
    u32x4 wcf[G::TPW][G::KSC];
    if constexpr (MODE == 0 && ABL != 3) {
#pragma unroll
      for (int j = 0; j < G::TPW; ++j)
#pragma unroll
        for (int ks = 0; ks < G::KSC; ++ks) wcf[j][ks] = wcg[(j * G::KSC + ks) * 64];
    }
    u32x4 xf[KSA], xn[KSA];
    u32x2 res[G::TPW], resn[G::TPW];
    auto load_x = [&](u32x4 (&dst)[KSA], int p) {      // (planes past the clip read zeros through the range check: p * plane_b >= the descriptor's size)
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks)
        dst[ks] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(x_off + (unsigned)p * plane_b + (unsigned)ks * 64u), 0, 0);
    };
    auto load_res = [&](u32x2 (&dst)[G::TPW], int t) { // (t < 0: garbage from another plane, never used; t >= T: zeros)
#pragma unroll
      for (int j = 0; j < G::TPW; ++j)
        dst[j] = (has_res && MODE == 0) ? __builtin_amdgcn_raw_buffer_load_b64(rr, (int)(r_off + (unsigned)(t < 0 ? 0 : t) * rplane_b + (unsigned)j * 32u), 0, 0) : u32x2{0u, 0u};
    };
    load_x(xn, 0);
    load_res(resn, -3);

    // The matrix waves only LOAD from global memory (finished rows go through LDS to the stencil waves, which only STORE):
    // vmcnt counts stores too and hipcc waits conservatively where paths join, so a wave that did both waited for its stores
    // whenever it needed a load -- measured: the block took 53 us whatever was removed from its arithmetic (ablations,
    // profiles/r6/bench_block_ablations_call10.txt).  Everything requested in iteration i is consumed in iteration i + 1.
    for (int i = 0; i <= T + 3; ++i) {
      const int tc = i - 3;                       // output plane of this iteration's conv_c
      const bool do_c = MODE == 0 && tc >= 0 && tc < T && ABL != 3;
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks) { asm volatile("" : "+v"(xn[ks])); xf[ks] = xn[ks]; }
#pragma unroll
      for (int j = 0; j < G::TPW; ++j) { asm volatile("" : "+v"(resn[j])); res[j] = resn[j]; }
      load_x(xn, i + 1);
      load_res(resn, tc + 1);
      if (i < T) {
        // ---- [A] conv_a + BN + ReLU of plane i's halo rows -> E[i & 1] ----
        unsigned char* const eb = s_e + (i & 1) * G::E_BYTES + e_dst;
        static_assert(G::MTA % 2 == 0, "channel tiles in pairs (two independent MFMA chains)");
#pragma unroll
        for (int mt = 0; mt < G::MTA; mt += 2) {
          f32x4 ca4 = {0.f, 0.f, 0.f, 0.f}, cb4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KSA; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(s_wa + ((mt * KSA + ks) * 64 + lane) * 16);
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(s_wa + (((mt + 1) * KSA + ks) * 64 + lane) * 16);
            if constexpr (ABL != 2) {
              ca4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, xf[ks]), ca4, 0, 0, 0);
              cb4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, __builtin_bit_cast(bf16x8, xf[ks]), cb4, 0, 0, 0);
            } else {
              asm volatile("" :: "v"(af), "v"(bf), "v"(xf[ks]));
            }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 c4 = h ? cb4 : ca4;
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(s_sa + (mt + h) * 16 + q * 4);
            const f32x4 h4 = *reinterpret_cast<const f32x4*>(s_ha + (mt + h) * 16 + q * 4);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = c4[r] * s4[r] + h4[r];
              if (a_relu) v[r] = fmaxf(v[r], 0.f);
            }
            u32x2 o = {pack2(v[0], v[1]) & e_mask, pack2(v[2], v[3]) & e_mask};
            *reinterpret_cast<u32x2*>(eb + (mt + h) * 32) = o;
          }
        }
      }
      if (do_c) {
        // ---- [C] conv_c + BN + residual + ReLU of output plane tc from MID[tc & 1] -> the staging plane s_out[tc & 1] ----
        f32x4 c4[G::TPW];
#pragma unroll
        for (int j = 0; j < G::TPW; ++j) c4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* mrow = s_mid + (tc & 1) * G::MID_BYTES + (cnt * kWP + n16) * G::MID_STRIDE + q * 16;
#pragma unroll
        for (int ks = 0; ks < G::KSC; ++ks) {
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(mrow + ks * 64);
#pragma unroll
          for (int j = 0; j < G::TPW; ++j)
            c4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wcf[j][ks]), bfr, c4[j], 0, 0, 0);
        }
        unsigned char* const ob = s_out + (tc & 1) * G::OUT_BYTES + (cnt * kWP + n16) * G::OUT_STRIDE + (cmt0 * 16 + q * 4) * 2;
#pragma unroll
        for (int j = 0; j < G::TPW; ++j) {
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(s_sc + (cmt0 + j) * 16 + q * 4);
          const f32x4 h4 = *reinterpret_cast<const f32x4*>(s_hc + (cmt0 + j) * 16 + q * 4);
          float v[4];
          v[0] = c4[j][0] * s4[0] + h4[0] + __uint_as_float(res[j][0] << 16);
          v[1] = c4[j][1] * s4[1] + h4[1] + __uint_as_float(res[j][0] & 0xffff0000u);
          v[2] = c4[j][2] * s4[2] + h4[2] + __uint_as_float(res[j][1] << 16);
          v[3] = c4[j][3] * s4[3] + h4[3] + __uint_as_float(res[j][1] & 0xffff0000u);
          if (out_relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          *reinterpret_cast<u32x2*>(ob + j * 32) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
        }
      }
      if constexpr (ABL != 4) __syncthreads();
    }
  } else {
    // =========================================== STENCIL waves ===========================================
    const int sw = wave - G::MWAVES;
    const int srow = sw / G::WPR;
    const int pair = (sw - srow * G::WPR) * 64 + lane;
    const bool p_ok = pair < G::NP;
    const int pch = p_ok ? 2 * pair : 0;
    float2 wt[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wt[t] = p_ok ? *reinterpret_cast<const float2*>(d.wb + (long)t * CP + pch) : float2{0.f, 0.f};
    const float2 sb2 = p_ok ? *reinterpret_cast<const float2*>(d.sb + pch) : float2{0.f, 0.f};
    const float2 hb2 = p_ok ? *reinterpret_cast<const float2*>(d.hb + pch) : float2{0.f, 0.f};
    float2 acc[3][kNW];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int n = 0; n < kNW; ++n) acc[a][n] = float2{0.f, 0.f};
    // MODE 1: this thread's output row in global memory and its squeeze sums
    const int soh = h0 + srow;
    const bool st_ok = MODE == 1 && p_ok && 2 * pair < pv_round_up(d.C, 8) && soh < H;
    bf16_t* Mp = static_cast<bf16_t*>(d.y) + (long)b * d.y_bs;
    const unsigned mplane_b = (unsigned)(H * W * d.ldy) * 2u, mvox_b = (unsigned)d.ldy * 2u;
    __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)Mp, 0, (int)(mplane_b * (unsigned)T), 0x00020000);
    const unsigned m_off = (unsigned)((soh * W) * d.ldy + 2 * pair) * 2u;
    float2 ps = {0.f, 0.f};

    // MODE 0: this wave's share of a finished output plane in the staging buffer: 16-byte chunks, (voxel slot, chunk) = item / item % CH
    constexpr int CH = COUT * 2 / 16;             // 16-byte chunks per voxel
    constexpr int ITEMS = kTH * kWP * CH, ROUNDS = (ITEMS + 255) / 256;
    bf16_t* Yo = static_cast<bf16_t*>(d.y) + (long)b * d.y_bs;
    const unsigned yplane_b = (unsigned)(H * W * d.ldy) * 2u;
    __amdgpu_buffer_rsrc_t ryo = __builtin_amdgcn_make_buffer_rsrc((void*)Yo, 0, (int)(yplane_b * (unsigned)T), 0x00020000);
    unsigned o_src[ROUNDS], o_dst[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int item = r * 256 + sw * 64 + lane;
      const int slot = item / CH, chunk = item - slot * CH;
      const int orow = slot / kWP, ocol = slot - orow * kWP;
      const bool ok = MODE == 0 && item < ITEMS && h0 + orow < H && ocol < W;
      o_src[r] = (unsigned)(slot * G::OUT_STRIDE + chunk * 16);
      o_dst[r] = ok ? (unsigned)(((h0 + orow) * W + ocol) * d.ldy) * 2u + (unsigned)chunk * 16u : 0x80000000u;
    }

    for (int i = 0; i <= T + 3; ++i) {
      if (MODE == 0 && i >= 4 && ABL != 3) {
        // ---- the rows the matrix waves finished in the last iteration (output plane i - 4): LDS -> global memory, 16 bytes per lane ----
        const unsigned char* ob = s_out + ((i - 4) & 1) * G::OUT_BYTES;
        const unsigned tb = (unsigned)(i - 4) * yplane_b;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
          const u32x2 lo = *reinterpret_cast<const u32x2*>(ob + o_src[r]);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(ob + o_src[r] + 8);
          const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
          __builtin_amdgcn_raw_buffer_store_b128(v, ryo, (int)(o_dst[r] + tb), 0, 0);
        }
      }
      const int p = i - 1;                        // input plane of this iteration's stencil
      if (p_ok && p >= 0 && p <= T) {
        if (p < T) {
          // ---- [B] plane p into the three rolling accumulator sets ----
          const unsigned char* ebase = s_e + (p & 1) * G::E_BYTES + pair * 4;
#pragma unroll
          for (int dh = 0; dh < 3; ++dh) {
            float2 xv[kNW + 2];
            xv[0] = float2{0.f, 0.f};                                  // image column -1
#pragma unroll
            for (int c = 0; c < kNW + 1; ++c) {                        // image columns 0 .. 14 (column >= W holds zeros)
              const unsigned u = *reinterpret_cast<const unsigned*>(ebase + ((srow + dh) * kWP + c) * G::E_STRIDE);
              xv[c + 1].x = __uint_as_float(u << 16);
              xv[c + 1].y = __uint_as_float(u & 0xffff0000u);
            }
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
              const float2 w0 = wt[(0 * 3 + dh) * 3 + dw];   // kt = 0 feeds output p+1
              const float2 w1 = wt[(1 * 3 + dh) * 3 + dw];   // kt = 1: output p
              const float2 w2 = wt[(2 * 3 + dh) * 3 + dw];   // kt = 2: output p-1
#pragma unroll
              for (int n = 0; n < kNW; ++n) {
                // (hipcc packs each pair into v_pk_fma_f32; single v_fmac_f32 by inline asm measured 56.4 against 52.8 us per block)
                const float2 x2 = xv[n + dw];
                if constexpr (ABL != 1) {
                  acc[2][n].x += x2.x * w0.x; acc[2][n].y += x2.y * w0.y;
                  acc[1][n].x += x2.x * w1.x; acc[1][n].y += x2.y * w1.y;
                  acc[0][n].x += x2.x * w2.x; acc[0][n].y += x2.y * w2.y;
                } else {
                  asm volatile("" :: "v"(x2.x), "v"(x2.y));
                }
              }
            }
          }
        }
        if (p >= 1) {
          if constexpr (MODE == 0) {
            // ---- output plane p-1 is complete: BN + activation -> bf16 -> MID[(p-1) & 1] ----
            unsigned char* mbase = s_mid + ((p - 1) & 1) * G::MID_BYTES + (srow * kWP) * G::MID_STRIDE + pair * 4;
#pragma unroll
            for (int n = 0; n < kNW; ++n) {
              float v0 = acc[0][n].x * sb2.x + hb2.x, v1 = acc[0][n].y * sb2.y + hb2.y;
              if (ACT_B == PV_ACT_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
              else if (ACT_B == PV_ACT_SWISH && ABL != 6) { v0 *= pv_sigmoid(v0); v1 *= pv_sigmoid(v1); }
              *reinterpret_cast<unsigned*>(mbase + n * G::MID_STRIDE) = pack2(v0, v1);
            }
          } else {
            // ---- output plane p-1: BN -> bf16 -> global memory (256 contiguous bytes per voxel and wave), fp32 sums for the squeeze ----
            const unsigned tb = (unsigned)(p - 1) * mplane_b;
#pragma unroll
            for (int n = 0; n < kNW; ++n) {
              const float v0 = acc[0][n].x * sb2.x + hb2.x, v1 = acc[0][n].y * sb2.y + hb2.y;
              const bool ok = st_ok && n < W;
              if (ok) { ps.x += v0; ps.y += v1; }
              __builtin_amdgcn_raw_buffer_store_b32(pack2(v0, v1), rm, (int)((ok ? m_off + (unsigned)n * mvox_b : 0x80000000u) + tb), 0, 0);
            }
          }
        }
#pragma unroll
        for (int n = 0; n < kNW; ++n) {
          acc[0][n] = acc[1][n];
          acc[1][n] = acc[2][n];
          acc[2][n] = float2{0.f, 0.f};
        }
      }
      if constexpr (ABL != 4) __syncthreads();
    }
    if (MODE == 1 && d.psum != nullptr && p_ok && 2 * pair < pv_round_up(d.C, 8)) {
      const int c_p = pv_round_up(d.C, 8);
      float* dst = d.psum + ((long)b * (tiles_h * kTH) + (blockIdx.x - b * tiles_h) * kTH + srow) * c_p + 2 * pair;
      *reinterpret_cast<float2*>(dst) = soh < H ? ps : float2{0.f, 0.f};
    }
  }
}

int check(const pv_bottleneck_desc& d) {
  if (!d.x || !d.y || !d.wa || !d.wb || !d.sa || !d.ha || !d.sb || !d.hb) return PV_ERR_INVALID;
  if (d.mode != PV_BLOCK_FULL && d.mode != PV_BLOCK_AB) return PV_ERR_INVALID;
  if (d.B <= 0 || d.T <= 0 || d.H <= 0 || d.W <= 0 || d.cin <= 0 || d.C <= 0 || d.cout <= 0) return PV_ERR_INVALID;
  if (d.ldx < pv_round_up(d.cin, 8) || d.ldx % 8) return PV_ERR_INVALID;
  if (d.mode == PV_BLOCK_FULL) {
    if (!d.wc || !d.sc || !d.hc) return PV_ERR_INVALID;
    if (d.ldy < pv_round_up(d.cout, 8) || d.ldy % 4) return PV_ERR_INVALID;
    if (d.residual && (d.ldr < pv_round_up(d.cout, 8) || d.ldr % 4)) return PV_ERR_INVALID;
  } else if (d.ldy < pv_round_up(d.C, 8) || d.ldy % 2) {
    return PV_ERR_INVALID;
  }
  return PV_OK;
}

}  // namespace

// geometry-only (pointers are ignored): 1 if pv_bottleneck can run this block
extern "C" int pv_bottleneck_supported(const pv_bottleneck_desc* dp) {
  if (!dp) return 0;
  const pv_bottleneck_desc& d = *dp;
  if (d.dtype != PV_BF16) return 0;
  if (d.B <= 0 || d.T <= 0 || d.H <= 0 || d.W <= 0 || d.W > kNW) return 0;
  if (d.act_a != PV_ACT_RELU && d.act_a != PV_ACT_NONE) return 0;
  if (d.mode != PV_BLOCK_FULL && d.mode != PV_BLOCK_AB) return 0;
  if (d.mode == PV_BLOCK_FULL) {
    if (d.act_b != PV_ACT_SWISH && d.act_b != PV_ACT_RELU && d.act_b != PV_ACT_NONE) return 0;
    if (d.act_out != PV_ACT_RELU && d.act_out != PV_ACT_NONE) return 0;
  }
  // 31-bit byte offsets inside a clip
  if ((long)d.T * d.H * d.W * (d.ldx > d.ldy ? d.ldx : d.ldy) * 2 > 0x7fffffffL) return 0;
  if ((long)d.B * pv_ceil_div(d.H, kTH) > 0x7fffffffL) return 0;
  // instantiated: X3D res4 (96 -> 216 -> 96)
  if (pv_round_up(d.C, 32) != 224 || pv_round_up(d.cin, 32) != 96 || d.ldx < 96) return 0;
  return d.mode == PV_BLOCK_AB || d.cout == 96;
}

// squeeze partial-sum blocks per clip of mode PV_BLOCK_AB (psum is [B][blocks][round_up(C, 8)] fp32)
extern "C" int pv_bottleneck_psum_blocks(const pv_bottleneck_desc* d) {
  return d ? (int)pv_ceil_div(d->H, kTH) * kTH : 0;
}

extern "C" int pv_bottleneck(const pv_bottleneck_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const int rc = check(*dp);
  if (rc != PV_OK) return rc;
  if (!pv_bottleneck_supported(dp)) return PV_ERR_UNSUPPORTED;
  const pv_bottleneck_desc& d = *dp;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_h = (int)pv_ceil_div(d.H, kTH);
  const dim3 grid((unsigned)(d.B * tiles_h)), block(512);
#ifdef PV_DEV_ABLATION   // timing builds with WRONG results: development variant of the library only (tools/r6/bench_block.py)
  if (const int abl = pv_tune("block_abl", 0); abl && d.mode == PV_BLOCK_FULL && d.act_b == PV_ACT_SWISH) {
    switch (abl) {
      case 1: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_SWISH, 0, 1>), grid, block, 0, s, d, tiles_h); break;
      case 2: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_SWISH, 0, 2>), grid, block, 0, s, d, tiles_h); break;
      case 3: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_SWISH, 0, 3>), grid, block, 0, s, d, tiles_h); break;
      case 4: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_SWISH, 0, 4>), grid, block, 0, s, d, tiles_h); break;
      case 5: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_SWISH, 0, 5>), grid, block, 0, s, d, tiles_h); break;
      default: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_SWISH, 0, 6>), grid, block, 0, s, d, tiles_h); break;
    }
    PV_LAUNCH_CHECK();
    return PV_OK;
  }
#endif
  if (d.mode == PV_BLOCK_AB) PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_NONE, 1>), grid, block, 0, s, d, tiles_h);
  else if (d.act_b == PV_ACT_SWISH) PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_SWISH, 0>), grid, block, 0, s, d, tiles_h);
  else if (d.act_b == PV_ACT_RELU) PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_RELU, 0>), grid, block, 0, s, d, tiles_h);
  else PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, PV_ACT_NONE, 0>), grid, block, 0, s, d, tiles_h);
  PV_LAUNCH_CHECK();
  return PV_OK;
}
