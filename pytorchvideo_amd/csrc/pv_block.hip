// A whole X3D bottleneck in ONE launch (gfx950), round 6:
//
//   y = act_out( r + sc * (Wc . act_b( sb * dw3x3x3( act_a( sa * (Wa . x) + ha ) ) + hb )) + hc )
//
// i.e. conv_a + norm_a + ReLU -> depthwise conv_b + norm_b + Swish -> conv_c + norm_c -> + shortcut -> ReLU of
// pytorchvideo/models/x3d.py:169-212 (create_x3d_bottleneck_block) / models/resnet.py:1345-1365 (BottleneckBlock.forward)
// and :1179-1189 (ResBlock.forward) for the blocks WITHOUT squeeze-excitation (every second block of a stage): the
// expanded tensor (2.25 x the block width) is never written.  As three launches an X3D-M res4 block moves
// 19 + 43 | 43 + 43 | 43 + 19 + 19 MB and takes 18 + 34 + 26 us on 100 k voxels -- every one of them latency-bound on a
// 14 x 14 map; here the block reads x once (plus the halo) and writes y once.  Mode PV_BLOCK_AB serves the blocks WITH
// squeeze-excitation: conv_a + conv_b + the squeeze sums in one launch (conv_c needs the gate of the whole clip).
//
// Mapping on CDNA4 (one workgroup = 8 waves = one clip x a tile of TWO output rows x 14 output columns, walking T; two waves per
// SIMD with DIFFERENT roles, so that the matrix pipe and the VALU work at the same time -- with every wave doing every phase in
// turn, one wave per SIMD, a block took 74 us, no faster than the three launches):
//   MATRIX waves (0-3), iteration i -- they only LOAD from global memory:
//   [A] conv_a for the 4 x 16 halo voxels of input plane i (wave w = halo row w = one 16-voxel MFMA column tile: image columns
//       w0-1 .. w0+14, all CP/16 channel tiles; v_mfma_f32_16x16x32_bf16 with the filter as A operand from LDS and the voxels'
//       input channels as B operand straight from global memory), BN + ReLU, rounded to bf16 exactly where the unfused path
//       stores it, written into the LDS plane E[i & 1][voxel][channel]; voxels outside the image are written as ZERO (conv_b
//       zero-pads the EXPANDED tensor);
//   [C] conv_c of output plane i-3 from MID[(i-3) & 1] (B operand) and its filter fragments, resident in REGISTERS for the whole
//       kernel (A operand; both filters + two E planes + two MID planes do not fit 160 KB at res4): tiles of 16 channels x 16
//       voxels round-robin over the waves; BN + residual + ReLU -> bf16 -> the LDS staging plane OUT[(i-3) & 1].
//   STENCIL waves (4-7), iteration i -- they only STORE to global memory:
//       OUT[(i-4) & 1] -> y, 16 bytes per lane;
//   [B] the 3x3x3 stencil of input plane i-1 on the VALU: thread = (channel pair | channel, output row, row segment of 14 | 7
//       outputs), the 27 taps in registers for the whole kernel, three rolling accumulator sets over T (plane p feeds outputs
//       p-1, p, p+1), so every plane is produced once and read once; BN + Swish of the finished output plane i-2 -> bf16 ->
//       LDS MID[(i-2) & 1][voxel][channel] (mode PV_BLOCK_AB: BN only -> global memory, + the squeeze sums).
//   One barrier per iteration.  The halo (4 x 16 voxels of conv_a per 2 x 14 outputs) is recomputed by the neighbours.
//   Why loads and stores sit in different waves: vmcnt counts stores too and hipcc waits conservatively where paths join, so a
//   wave that did both waited for its stores whenever it needed a load (the block took 53 us whatever was removed from its
//   arithmetic); and what bounds the kernel now is VALU issue (16 lanes per clock: 4 cycles per wave64 instruction, 8 per
//   v_pk_fma_f32), shared by the stencil wave and the matrix wave's epilogues on each SIMD -- profiles/r6/bench_block_ablations_*.txt.
#include "pv_common.h"

namespace {

constexpr int kTH = 2;            // output rows per workgroup
constexpr int kHR = kTH + 2;      // halo rows of the expanded tensor per plane
constexpr int kWP = 16;           // voxel slots per plane row in LDS: image columns w0-1 .. w0+14
constexpr int kTW = 14;           // output columns per workgroup

// CP: inner channels padded to 32; KSA = cin_p / 32; COUTP: output channels padded to 16;
// PAIR: a stencil thread owns a channel PAIR (packed fp32 math) or ONE channel; SEGS: segments a 14-column output row is cut into
template <int CP, int KSA, int COUTP, bool PAIR, int SEGS> struct BlkGeom {
  static constexpr int NI = PAIR ? CP / 2 : CP;     // stencil items (pairs | channels) per voxel
  static constexpr int WPR = (NI + 63) / 64;        // stencil waves per (row, segment)
  static constexpr int NWS = kTW / SEGS;            // outputs per stencil thread
  static constexpr int SWAVES = kTH * SEGS * WPR;   // stencil waves
  static constexpr int MWAVES = kHR * kWP / 16;     // matrix waves: one 16-voxel producer tile (= one halo row) each
  static constexpr int MTA = CP / 16;               // conv_a channel tiles
  static constexpr int KSC = CP / 32;               // conv_c K steps
  static constexpr int MTC = COUTP / 16;            // conv_c channel tiles
  static constexpr int TPW = (MTC * kTH + MWAVES - 1) / MWAVES;    // conv_c tiles per matrix wave (tile = wave + 4 k: row = tile & 1, channel tile = tile >> 1)
  static constexpr int E_STRIDE = CP * 2 + 8;       // bytes per voxel of E (+8: the producer's 8-byte stores of 16 voxels hit distinct banks)
  static constexpr int MID_STRIDE = CP * 2 + 16;    // bytes per voxel of MID (+16: conflict-free ds_read_b128 of 16 voxels)
  static constexpr int OUT_STRIDE = COUTP * 2 + 8;  // bytes per voxel of the output staging planes
  static constexpr int WA_BYTES = MTA * KSA * 1024;
  static constexpr int E_BYTES = kHR * kWP * E_STRIDE;
  static constexpr int MID_BYTES = kTH * kWP * MID_STRIDE;
  static constexpr int OUT_BYTES = kTH * kWP * OUT_STRIDE;
  static constexpr int PAR_BYTES = (2 * CP + 2 * COUTP) * 4;     // sa | ha | sc | hc
  static constexpr int TOTAL = WA_BYTES + 2 * E_BYTES + 2 * MID_BYTES + PAR_BYTES + 2 * OUT_BYTES;
  static_assert(MWAVES == 4 && SWAVES == 4, "four matrix waves + four stencil waves");
  static_assert(kTW % SEGS == 0 && MTA % 2 == 0 && CP % 32 == 0 && COUTP % 16 == 0, "geometry");
  static_assert(TOTAL <= 160 * 1024, "LDS");
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((vector_size(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2_t pk = {(bf16_t)a, (bf16_t)b};
  return __builtin_bit_cast(unsigned, pk);
}
// ReLU on a packed bf16 pair: a negative bf16 is a negative int16, so max(., 0) per 16-bit half clears it (one v_pk_max_i16
// instead of two v_max_f32 + canonicalisation)
__device__ __forceinline__ unsigned relu2(unsigned u) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, u), s16x2{0, 0}));
}

// a stencil thread's value: a channel pair (float2) or one channel (float)
template <bool PAIR> struct SV;
template <> struct SV<true> {
  typedef float2 T;
  static __device__ __forceinline__ T zero() { return float2{0.f, 0.f}; }
  static __device__ __forceinline__ T load(const unsigned char* p, unsigned) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    return float2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
  }
  static __device__ __forceinline__ void fma(T& a, const T& x, const T& w) { a.x += x.x * w.x; a.y += x.y * w.y; }
  static __device__ __forceinline__ T ldw(const float* p, bool ok) { return ok ? *reinterpret_cast<const float2*>(p) : float2{0.f, 0.f}; }
};
template <> struct SV<false> {
  typedef float T;
  static __device__ __forceinline__ T zero() { return 0.f; }
  // p: the 32-bit word that holds the lane's channel and its neighbour; sel: the v_perm_b32 selector that moves the lane's half into
  // the upper half of an fp32 (0x01000c0c even channel, 0x03020c0c odd): one ds_read_b32 + one v_perm_b32 per value
  static __device__ __forceinline__ T load(const unsigned char* p, unsigned sel) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    return __uint_as_float(__builtin_amdgcn_perm(u, u, sel));
  }
  // ONE v_fmac_f32 per tap, opaque to the SLP vectoriser.  With `a += x * w` hipcc packed the taps of neighbouring outputs into
  // v_pk_fma_f32 and, a segment having 7 outputs, built a separate packed sequence for the odd one (two accumulator sets of output 0
  // in one register pair, the products of the third by v_pk_mul_f32 + v_add_f32).  That build was NOT bit-reproducible: beside the
  // other sub-batch's stem kernel or its stride-2 pwdw_plane_kernel -- and beside no other op of the plan -- output column 0 of a
  // thread's segment differed from run to run (~1e-4 of the voxels, all channels, modest values: one accumulator of that output
  // wrong).  What removed it: waiting for ALL of a row's LDS loads before the first FMA, pinning the loads in order, channel
  // pairs (the res3 / res4 form), or this scalar form; what did not: a second barrier per iteration, ordering the MID stores,
  // 32-bit instead of 16-bit LDS loads.  The instruction-level cause is not established: the most suspicious form of that
  // sequence (v_pk_fma_f32 whose destination pair is also its broadcast source) is exact in isolation, alone and beside an
  // LDS-heavy kernel (tools/micro/pk_fma_inplace.hip).  Evidence: profiles/r6/replay_locate_*.txt; regression test:
  // tests/test_gpu_full_geometry.py::test_x3d_m_fused_blocks_are_bit_reproducible_beside_the_other_sub_batchs_kernels.
  // Same VALU time as the packed form (4 cycles per 64 FMAs against 8 per 128), fewer register moves: 69 instead of 126 VGPRs.
  static __device__ __forceinline__ void fma(T& a, const T& x, const T& w) { asm("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(w)); }
  static __device__ __forceinline__ T ldw(const float* p, bool ok) { return ok ? *p : 0.f; }
};

// MODE 0: the whole block.  MODE 1 (blocks WITH squeeze-excitation): stop behind conv_b -- the stencil waves store
// sb * dw(...) + hb (no activation: it is applied with the gate by conv_c's operand load) as bf16 to y (B, T, H, W, C) and add
// the fp32 values up for the squeeze: psum[b][blk][c], blk = ((tile_h * 2 + row) * tiles_w + tile_w) * SEGS + seg = the sum over
// this thread's T x 1 x NWS output voxels of channel c (deterministic: one writer per entry); the matrix waves run conv_a only.
// (A third mode -- excitation + Swish + conv_c of such a block on these waves, reading the tensor mode 1 wrote -- was measured and
// removed: X3D-M 10.33 k -> 10.00 k clips/s against the gated conv_c of csrc/pv_pwconv.hip / pv_conv.hip, profiles/r6/model_ab_gated_conv_c_call19.txt.)
// ABL (development variant of the library, timing only, WRONG results): 1 no stencil FMAs, 2 no conv_a MFMAs, 3 no conv_c, 4 no
// barriers, 6 no stencil activation
template <int CP, int KSA, int COUTP, bool PAIR, int SEGS, int ACT_B, int MODE, int ABL = 0>
__global__ __launch_bounds__(512, 2) void bottleneck_block_kernel(const pv_bottleneck_desc d, int tiles_h, int tiles_w) {
  using G = BlkGeom<CP, KSA, COUTP, PAIR, SEGS>;
  using V = SV<PAIR>;
  typedef typename V::T vt;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::TOTAL];
  unsigned char* const s_wa = smem;
  unsigned char* const s_e = s_wa + G::WA_BYTES;            // two planes
  unsigned char* const s_mid = s_e + 2 * G::E_BYTES;        // two planes
  float* const s_sa = reinterpret_cast<float*>(s_mid + 2 * G::MID_BYTES);
  float* const s_ha = s_sa + CP;
  float* const s_sc = s_ha + CP;
  float* const s_hc = s_sc + COUTP;
  unsigned char* const s_out = reinterpret_cast<unsigned char*>(s_hc + COUTP);      // two planes of finished output rows

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int tiles = tiles_h * tiles_w;
  const int b = blockIdx.x / tiles;
  const int tile = blockIdx.x - b * tiles;
  const int tile_h = tile / tiles_w, tile_w = tile - tile_h * tiles_w;
  const int h0 = tile_h * kTH, w0 = tile_w * kTW;
  const int T = d.T, H = d.H, W = d.W;
  const int cout_p8 = pv_round_up(d.cout, 8);

  // ---- conv_a's filter and the folded BatchNorms -> LDS (once per workgroup; L2-resident after the first workgroups) ----
  {
    const u32x4* wa = static_cast<const u32x4*>(d.wa);
    u32x4* la = reinterpret_cast<u32x4*>(s_wa);
    for (int i = tid; i < G::WA_BYTES / 16; i += 512) la[i] = wa[i];
    for (int i = tid; i < CP; i += 512) { s_sa[i] = d.sa[i]; s_ha[i] = d.ha[i]; }
    if constexpr (MODE == 0)
      for (int i = tid; i < COUTP; i += 512) { s_sc[i] = i < d.cout ? d.sc[i] : 0.f; s_hc[i] = i < d.cout ? d.hc[i] : 0.f; }
  }
  __syncthreads();

  if (wave < G::MWAVES) {
    // =========================================== MATRIX waves ===========================================
    constexpr unsigned kOOB = 0x80000000u;
    // producer role: halo row `wave`, halo column n16 (image column w0 - 1 + n16), k-group q
    const bf16_t* X = static_cast<const bf16_t*>(d.x) + (long)b * d.x_bs;
    const unsigned plane_b = (unsigned)(H * W * d.ldx) * 2u;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)(plane_b * (unsigned)T), 0x00020000);
    const int hh = h0 - 1 + wave, ww = w0 - 1 + n16;
    const bool vox_ok = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
    const unsigned x_off = vox_ok ? (unsigned)((hh * W + ww) * d.ldx + q * 8) * 2u : kOOB;
    const unsigned e_mask = vox_ok ? 0xffffffffu : 0u;
    const int e_dst = (wave * kWP + n16) * G::E_STRIDE + q * 8;      // + plane buffer + mt * 32
    // conv_c role: tiles wave, wave + 4, ...: output row crow = wave & 1, channel tiles (wave >> 1) + 2 k; voxel slot n16 = output column w0 + n16
    const int crow = wave & 1, cmt0 = wave >> 1;
    const int oh = h0 + crow, ow = w0 + n16;
    const bool o_ok = oh < H && n16 < kTW && ow < W;
    const bf16_t* R = static_cast<const bf16_t*>(d.residual) + (long)b * d.r_bs;
    const unsigned rplane_b = (unsigned)(H * W * d.ldr) * 2u;
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, (int)(rplane_b * (unsigned)T), 0x00020000);
    const unsigned r_off = o_ok ? (unsigned)((oh * W + ow) * d.ldr + cmt0 * 16 + q * 4) * 2u : kOOB;
    const bool has_res = d.residual != nullptr;
    const bool out_relu = d.act_out == PV_ACT_RELU;
    const bool a_relu = d.act_a == PV_ACT_RELU;
    const u32x4* wcg = static_cast<const u32x4*>(d.wc) + (long)cmt0 * G::KSC * 64 + lane;

    // conv_c's filter fragments of this wave stay in REGISTERS for the whole kernel (the matrix waves hold no stencil state)
    u32x4 wcf[G::TPW][G::KSC];
    if constexpr (MODE == 0 && ABL != 3) {
#pragma unroll
      for (int k = 0; k < G::TPW; ++k)
#pragma unroll
        for (int ks = 0; ks < G::KSC; ++ks)
          wcf[k][ks] = (cmt0 + 2 * k < G::MTC) ? wcg[((2 * k) * G::KSC + ks) * 64] : u32x4{0u, 0u, 0u, 0u};
    }
    u32x4 xf[KSA], xn[KSA];
    u32x2 res[G::TPW], resn[G::TPW];
    auto load_x = [&](u32x4 (&dst)[KSA], int p) {      // (planes past the clip read zeros through the range check: p * plane_b >= the descriptor's size)
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks)
        dst[ks] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(x_off + (unsigned)p * plane_b + (unsigned)ks * 64u), 0, 0);
    };
    auto load_res = [&](u32x2 (&dst)[G::TPW], int t) { // (t < 0: another plane's values, never used; t >= T: zeros)
#pragma unroll
      for (int k = 0; k < G::TPW; ++k)
        dst[k] = (has_res && MODE == 0) ? __builtin_amdgcn_raw_buffer_load_b64(rr, (int)(r_off + (unsigned)(t < 0 ? 0 : t) * rplane_b + (unsigned)k * 64u), 0, 0) : u32x2{0u, 0u};
    };
    load_x(xn, 0);
    load_res(resn, -3);

    // Everything requested in iteration i is consumed in iteration i + 1 (a full iteration of flight time).
    for (int i = 0; i <= T + 3; ++i) {
      const int tc = i - 3;                       // output plane of this iteration's conv_c
      const bool do_c = MODE == 0 && tc >= 0 && tc < T && ABL != 3;
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks) { asm volatile("" : "+v"(xn[ks])); xf[ks] = xn[ks]; }
#pragma unroll
      for (int k = 0; k < G::TPW; ++k) { asm volatile("" : "+v"(resn[k])); res[k] = resn[k]; }
      load_x(xn, i + 1);
      load_res(resn, tc + 1);
      if (i < T) {
        // ---- [A] conv_a + BN + ReLU of plane i's halo voxels -> E[i & 1] ----
        unsigned char* const eb = s_e + (i & 1) * G::E_BYTES + e_dst;
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
            unsigned o0 = pack2(c4[0] * s4[0] + h4[0], c4[1] * s4[1] + h4[1]);
            unsigned o1 = pack2(c4[2] * s4[2] + h4[2], c4[3] * s4[3] + h4[3]);
            if (a_relu) { o0 = relu2(o0); o1 = relu2(o1); }
            *reinterpret_cast<u32x2*>(eb + (mt + h) * 32) = u32x2{o0 & e_mask, o1 & e_mask};
          }
        }
      }
      if (do_c) {
        // ---- [C] conv_c + BN + residual + ReLU of output plane tc from MID[tc & 1] -> the staging plane OUT[tc & 1] ----
        f32x4 c4[G::TPW];
#pragma unroll
        for (int k = 0; k < G::TPW; ++k) c4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* mrow = s_mid + (tc & 1) * G::MID_BYTES + (crow * kWP + n16) * G::MID_STRIDE + q * 16;
#pragma unroll
        for (int ks = 0; ks < G::KSC; ++ks) {
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(mrow + ks * 64);
#pragma unroll
          for (int k = 0; k < G::TPW; ++k)
            c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wcf[k][ks]), bfr, c4[k], 0, 0, 0);
        }
        unsigned char* const ob = s_out + (tc & 1) * G::OUT_BYTES + (crow * kWP + n16) * G::OUT_STRIDE + (cmt0 * 16 + q * 4) * 2;
#pragma unroll
        for (int k = 0; k < G::TPW; ++k) {
          if (cmt0 + 2 * k < G::MTC) {        // (wave-uniform)
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(s_sc + (cmt0 + 2 * k) * 16 + q * 4);
            const f32x4 h4 = *reinterpret_cast<const f32x4*>(s_hc + (cmt0 + 2 * k) * 16 + q * 4);
            const float v0 = c4[k][0] * s4[0] + h4[0] + __uint_as_float(res[k][0] << 16);
            const float v1 = c4[k][1] * s4[1] + h4[1] + __uint_as_float(res[k][0] & 0xffff0000u);
            const float v2 = c4[k][2] * s4[2] + h4[2] + __uint_as_float(res[k][1] << 16);
            const float v3 = c4[k][3] * s4[3] + h4[3] + __uint_as_float(res[k][1] & 0xffff0000u);
            unsigned o0 = pack2(v0, v1), o1 = pack2(v2, v3);
            if (out_relu) { o0 = relu2(o0); o1 = relu2(o1); }
            *reinterpret_cast<u32x2*>(ob + k * 64) = u32x2{o0, o1};
          }
        }
      }
      if constexpr (ABL != 4) __syncthreads();
      if constexpr (ABL == 8) __syncthreads();
    }
  } else {
    // =========================================== STENCIL waves ===========================================
    const int sw = wave - G::MWAVES;
    const int wpart = sw % G::WPR, rs = sw / G::WPR;
    const int seg = rs % SEGS, srow = rs / SEGS;
    const int item = wpart * 64 + lane;
    constexpr int IB = PAIR ? 4 : 2;              // bytes of an item in E / MID
    constexpr int IC = PAIR ? 2 : 1;              // channels of an item
    const bool p_ok = item < G::NI;
    const unsigned psel = (item & 1) ? 0x03020c0cu : 0x01000c0cu;     // (one channel per lane: which half of the pair word is the lane's)
    const int pch = p_ok ? IC * item : 0;
    vt wt[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wt[t] = V::ldw(d.wb + (long)t * CP + pch, p_ok);
    const vt sb2 = V::ldw(d.sb + pch, p_ok), hb2 = V::ldw(d.hb + pch, p_ok);
    vt acc[3][G::NWS];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int n = 0; n < G::NWS; ++n) acc[a][n] = V::zero();
    // MODE 1: this thread's output row segment in global memory and its squeeze sums
    const int soh = h0 + srow, sow0 = w0 + seg * G::NWS;
    const bool st_ok = MODE == 1 && p_ok && pch < pv_round_up(d.C, 8) && soh < H;
    bf16_t* Mp = static_cast<bf16_t*>(d.y) + (long)b * d.y_bs;
    const unsigned yplane_b = (unsigned)(H * W * d.ldy) * 2u, yvox_b = (unsigned)d.ldy * 2u;
    __amdgpu_buffer_rsrc_t ryo = __builtin_amdgcn_make_buffer_rsrc((void*)Mp, 0, (int)(yplane_b * (unsigned)T), 0x00020000);
    const unsigned m_off = (unsigned)((soh * W + sow0) * d.ldy + pch) * 2u;
    vt ps = V::zero();
    // MODE 0: this wave's share of a finished output plane in the staging buffer: 16-byte chunks, (voxel slot, chunk) = item / item % CH
    constexpr int CH = COUTP * 2 / 16;            // 16-byte chunks per voxel
    constexpr int ITEMS = kTH * kWP * CH, ROUNDS = (ITEMS + 255) / 256;
    unsigned o_src[ROUNDS], o_dst[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int it = r * 256 + sw * 64 + lane;
      const int slot = it / CH, chunk = it - slot * CH;
      const int orow = slot / kWP, ocol = slot - orow * kWP;
      const bool ok = MODE == 0 && it < ITEMS && h0 + orow < H && ocol < kTW && w0 + ocol < W && chunk * 8 < cout_p8;
      o_src[r] = (unsigned)((it < ITEMS ? slot : 0) * G::OUT_STRIDE + chunk * 16);
      o_dst[r] = ok ? (unsigned)(((h0 + orow) * W + w0 + ocol) * d.ldy) * 2u + (unsigned)chunk * 16u : 0x80000000u;
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
          const unsigned char* ebase = s_e + (p & 1) * G::E_BYTES + (seg * G::NWS) * G::E_STRIDE + (PAIR ? item : item >> 1) * 4;
#pragma unroll
          for (int dh = 0; dh < 3; ++dh) {
            vt xv[G::NWS + 2];                      // halo columns seg * NWS .. seg * NWS + NWS + 1 = image columns w0 + seg * NWS - 1 ..
            if constexpr (ABL == 9) {
#pragma unroll
              for (int c = G::NWS + 1; c >= 0; --c) { xv[c] = V::load(ebase + ((srow + dh) * kWP + c) * G::E_STRIDE, psel); asm volatile("" : "+v"(xv[c])); }
            } else {
#pragma unroll
              for (int c = 0; c < G::NWS + 2; ++c) {
                xv[c] = V::load(ebase + ((srow + dh) * kWP + c) * G::E_STRIDE, psel);
                if constexpr (ABL == 11) asm volatile("" : "+v"(xv[c]));
              }
              if constexpr (ABL == 12) __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0), vmcnt / expcnt untouched
            }
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
              const vt w0v = wt[(0 * 3 + dh) * 3 + dw];   // kt = 0 feeds output p+1
              const vt w1v = wt[(1 * 3 + dh) * 3 + dw];   // kt = 1: output p
              const vt w2v = wt[(2 * 3 + dh) * 3 + dw];   // kt = 2: output p-1
#pragma unroll
              for (int n = 0; n < G::NWS; ++n) {
                // (hipcc packs each pair into v_pk_fma_f32; single v_fmac_f32 by inline asm measured 56.4 against 52.8 us per block)
                if constexpr (ABL != 1) {
                  V::fma(acc[2][n], xv[n + dw], w0v);
                  V::fma(acc[1][n], xv[n + dw], w1v);
                  V::fma(acc[0][n], xv[n + dw], w2v);
                } else {
                  asm volatile("" :: "v"(xv[n + dw]));
                }
              }
            }
          }
        }
        if (p >= 1) {
          if constexpr (MODE == 0) {
            // ---- output plane p-1 is complete: BN + activation -> bf16 -> MID[(p-1) & 1] ----
            unsigned char* mbase = s_mid + ((p - 1) & 1) * G::MID_BYTES + (srow * kWP + seg * G::NWS) * G::MID_STRIDE + item * IB;
#pragma unroll
            for (int n = 0; n < G::NWS; ++n) {
              if constexpr (PAIR) {
                float v0 = acc[0][n].x * sb2.x + hb2.x, v1 = acc[0][n].y * sb2.y + hb2.y;
                if (ACT_B == PV_ACT_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                else if (ACT_B == PV_ACT_SWISH && ABL != 6) { v0 *= pv_sigmoid(v0); v1 *= pv_sigmoid(v1); }
                *reinterpret_cast<unsigned*>(mbase + n * G::MID_STRIDE) = pack2(v0, v1);
              } else {
                float v0 = acc[0][n] * sb2 + hb2;
                if (ACT_B == PV_ACT_RELU) v0 = fmaxf(v0, 0.f);
                else if (ACT_B == PV_ACT_SWISH && ABL != 6) v0 *= pv_sigmoid(v0);
                *reinterpret_cast<bf16_t*>(mbase + n * G::MID_STRIDE) = (bf16_t)v0;
                if constexpr (ABL == 13) __builtin_amdgcn_sched_barrier(0);
              }
            }
          } else {
            // ---- output plane p-1: BN -> bf16 -> global memory, fp32 sums for the squeeze ----
            const unsigned tb = (unsigned)(p - 1) * yplane_b;
#pragma unroll
            for (int n = 0; n < G::NWS; ++n) {
              const bool ok = st_ok && sow0 + n < W;
              const unsigned off = (ok ? m_off + (unsigned)n * yvox_b : 0x80000000u) + tb;
              if constexpr (PAIR) {
                const float v0 = acc[0][n].x * sb2.x + hb2.x, v1 = acc[0][n].y * sb2.y + hb2.y;
                if (ok) { ps.x += v0; ps.y += v1; }
                __builtin_amdgcn_raw_buffer_store_b32(pack2(v0, v1), ryo, (int)off, 0, 0);
              } else {
                const float v0 = acc[0][n] * sb2 + hb2;
                if (ok) ps += v0;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)v0), ryo, (int)off, 0, 0);
              }
            }
          }
        }
#pragma unroll
        for (int n = 0; n < G::NWS; ++n) {
          acc[0][n] = acc[1][n];
          acc[1][n] = acc[2][n];
          acc[2][n] = V::zero();
        }
      }
      if constexpr (ABL != 4) __syncthreads();
      if constexpr (ABL == 8) __syncthreads();
    }
    if (MODE == 1 && d.psum != nullptr && p_ok && pch < pv_round_up(d.C, 8)) {
      const int c_p = pv_round_up(d.C, 8);
      const long blk = ((long)(tile_h * kTH + srow) * tiles_w + tile_w) * SEGS + seg;
      float* dst = d.psum + ((long)b * ((long)tiles_h * kTH * tiles_w * SEGS) + blk) * c_p + pch;
      if constexpr (PAIR) *reinterpret_cast<float2*>(dst) = soh < H ? ps : float2{0.f, 0.f};
      else *dst = soh < H ? ps : 0.f;
    }
  }
}

int check(const pv_bottleneck_desc& d) {
  if (d.mode != PV_BLOCK_FULL && d.mode != PV_BLOCK_AB) return PV_ERR_INVALID;
  if (!d.x || !d.y || !d.wa || !d.wb || !d.sa || !d.ha || !d.sb || !d.hb) return PV_ERR_INVALID;
  if (d.B <= 0 || d.T <= 0 || d.H <= 0 || d.W <= 0 || d.cin <= 0 || d.C <= 0 || d.cout <= 0) return PV_ERR_INVALID;
  if (d.ldx < pv_round_up(d.cin, 8) || d.ldx % 8) return PV_ERR_INVALID;
  if (d.mode == PV_BLOCK_FULL) {
    if (!d.wc || !d.sc || !d.hc) return PV_ERR_INVALID;
    if (d.ldy < pv_round_up(d.cout, 8) || d.ldy % 8) return PV_ERR_INVALID;
    if (d.residual && (d.ldr < pv_round_up(d.cout, 8) || d.ldr % 4)) return PV_ERR_INVALID;
  } else if (d.ldy < pv_round_up(d.C, 8) || d.ldy % 2) {
    return PV_ERR_INVALID;
  }
  return PV_OK;
}

// the instantiated geometries: X3D res2 (24 -> 54 -> 24), res3 (48 -> 108 -> 48), res4 (96 -> 216 -> 96)
int variant_of(const pv_bottleneck_desc& d) {
  const int cp = pv_round_up(d.C, 32), cinp = pv_round_up(d.cin, 32);
  const bool c_out = d.mode != PV_BLOCK_AB;      // does the mode run conv_c?
  if (cp == 224 && cinp == 96 && d.ldx >= 96 && (!c_out || d.cout == 96)) return 4;
  if (cp == 128 && cinp == 64 && (!c_out || d.cout == 48)) return 3;
  if (cp == 64 && cinp == 32 && (!c_out || d.cout == 24)) return 2;
  return 0;
}

template <int CP, int KSA, int COUTP, bool PAIR, int SEGS>
int launch_variant(const pv_bottleneck_desc& d, dim3 grid, int tiles_h, int tiles_w, hipStream_t s) {
  const dim3 block(512);
  if (d.mode == PV_BLOCK_AB) PV_LAUNCH((bottleneck_block_kernel<CP, KSA, COUTP, PAIR, SEGS, PV_ACT_NONE, 1>), grid, block, 0, s, d, tiles_h, tiles_w);
  else if (d.act_b == PV_ACT_SWISH) PV_LAUNCH((bottleneck_block_kernel<CP, KSA, COUTP, PAIR, SEGS, PV_ACT_SWISH, 0>), grid, block, 0, s, d, tiles_h, tiles_w);
  else if (d.act_b == PV_ACT_RELU) PV_LAUNCH((bottleneck_block_kernel<CP, KSA, COUTP, PAIR, SEGS, PV_ACT_RELU, 0>), grid, block, 0, s, d, tiles_h, tiles_w);
  else PV_LAUNCH((bottleneck_block_kernel<CP, KSA, COUTP, PAIR, SEGS, PV_ACT_NONE, 0>), grid, block, 0, s, d, tiles_h, tiles_w);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

}  // namespace

// geometry-only (pointers are ignored): 1 if pv_bottleneck can run this block
extern "C" int pv_bottleneck_supported(const pv_bottleneck_desc* dp) {
  if (!dp) return 0;
  const pv_bottleneck_desc& d = *dp;
  if (d.dtype != PV_BF16) return 0;
  if (d.B <= 0 || d.T <= 0 || d.H <= 0 || d.W <= 0) return 0;
  if (d.mode != PV_BLOCK_FULL && d.mode != PV_BLOCK_AB) return 0;
  if (d.act_a != PV_ACT_RELU && d.act_a != PV_ACT_NONE) return 0;
  if (d.mode == PV_BLOCK_FULL) {
    if (d.act_b != PV_ACT_SWISH && d.act_b != PV_ACT_RELU && d.act_b != PV_ACT_NONE) return 0;
    if (d.act_out != PV_ACT_RELU && d.act_out != PV_ACT_NONE) return 0;
  }
  // 31-bit byte offsets inside a clip
  if ((long)d.T * d.H * d.W * (d.ldx > d.ldy ? d.ldx : d.ldy) * 2 > 0x7fffffffL) return 0;
  if ((long)d.B * pv_ceil_div(d.H, kTH) * pv_ceil_div(d.W, kTW) > 0x7fffffffL) return 0;
  const int v = variant_of(d);
  if (!v) return 0;
  // bit s: stage res<s>.  Whole blocks: res2 + res3 + res4.  conv_a + conv_b + squeeze sums: res4 only -- on the larger maps the
  // plane-streaming kernel with the fused pointwise producer (csrc/pv_pwdw.hip) is faster (B = 32: 74 vs 88 us at res3, 141 vs
  // 190 us at res2; profiles/r6/bench_block_stages_call12.txt against profiles/r5/x3d_m_per_op.txt).
  const int stages = d.mode == PV_BLOCK_FULL ? pv_tune("block_stages", 0x1c) : pv_tune("block_stages_ab", 0x10);
  return (stages >> v) & 1;
}

// squeeze partial-sum blocks per clip of mode PV_BLOCK_AB (psum is [B][blocks][round_up(C, 8)] fp32)
extern "C" int pv_bottleneck_psum_blocks(const pv_bottleneck_desc* dp) {
  if (!dp) return 0;
  const int segs = variant_of(*dp) == 4 ? 1 : 2;
  return (int)(pv_ceil_div(dp->H, kTH) * kTH * pv_ceil_div(dp->W, kTW) * segs);
}

extern "C" int pv_bottleneck(const pv_bottleneck_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const int rc = check(*dp);
  if (rc != PV_OK) return rc;
  if (!pv_bottleneck_supported(dp)) return PV_ERR_UNSUPPORTED;
  const pv_bottleneck_desc& d = *dp;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_h = (int)pv_ceil_div(d.H, kTH), tiles_w = (int)pv_ceil_div(d.W, kTW);
  const dim3 grid((unsigned)(d.B * tiles_h * tiles_w));
#ifdef PV_DEV_ABLATION   // timing builds with WRONG results: development variant of the library only (tools/r6/bench_block.py)
  if (const int abl = pv_tune("block_abl", 0); abl && abl < 8 && variant_of(d) == 4 && d.mode == PV_BLOCK_FULL && d.act_b == PV_ACT_SWISH) {
    const dim3 block(512);
    switch (abl) {
      case 1: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, true, 1, PV_ACT_SWISH, 0, 1>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      case 2: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, true, 1, PV_ACT_SWISH, 0, 2>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      case 3: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, true, 1, PV_ACT_SWISH, 0, 3>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      case 4: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, true, 1, PV_ACT_SWISH, 0, 4>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      default: PV_LAUNCH((bottleneck_block_kernel<224, 3, 96, true, 1, PV_ACT_SWISH, 0, 6>), grid, block, 0, s, d, tiles_h, tiles_w); break;
    }
    PV_LAUNCH_CHECK();
    return PV_OK;
  }
#endif
#ifdef PV_DEV_ABLATION   // diagnostic builds of the res2 instantiation (correct results): 8 two barriers per iteration, 9 the stencil's LDS loads in reverse order, 10 channel pairs
  if (const int abl = pv_tune("block_abl", 0); abl >= 8 && variant_of(d) == 2 && d.mode == PV_BLOCK_FULL && d.act_b == PV_ACT_SWISH) {
    const dim3 block(512);
    switch (abl) {
      case 8: PV_LAUNCH((bottleneck_block_kernel<64, 1, 32, false, 2, PV_ACT_SWISH, 0, 8>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      case 9: PV_LAUNCH((bottleneck_block_kernel<64, 1, 32, false, 2, PV_ACT_SWISH, 0, 9>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      case 11: PV_LAUNCH((bottleneck_block_kernel<64, 1, 32, false, 2, PV_ACT_SWISH, 0, 11>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      case 12: PV_LAUNCH((bottleneck_block_kernel<64, 1, 32, false, 2, PV_ACT_SWISH, 0, 12>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      case 13: PV_LAUNCH((bottleneck_block_kernel<64, 1, 32, false, 2, PV_ACT_SWISH, 0, 13>), grid, block, 0, s, d, tiles_h, tiles_w); break;
      default: PV_LAUNCH((bottleneck_block_kernel<64, 1, 32, true, 2, PV_ACT_SWISH, 0, 0>), grid, block, 0, s, d, tiles_h, tiles_w); break;
    }
    PV_LAUNCH_CHECK();
    return PV_OK;
  }
#endif
  switch (variant_of(d)) {
    case 4: return launch_variant<224, 3, 96, true, 1>(d, grid, tiles_h, tiles_w, s);
    case 3: return launch_variant<128, 2, 48, true, 2>(d, grid, tiles_h, tiles_w, s);
    case 2: return launch_variant<64, 1, 32, false, 2>(d, grid, tiles_h, tiles_w, s);
  }
  return PV_ERR_UNSUPPORTED;
}
