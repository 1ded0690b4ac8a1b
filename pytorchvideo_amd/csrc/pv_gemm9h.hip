// The eight-phase GEMM (pv_gemm9.hip) on a 128 (voxels) x 256 (channels) tile: layers whose 256 x 256 tiles leave half of
// the chip idle (round 5; SlowFast-R50 res4: 32768 voxels x 256 channels = 128 tiles on 256 CUs, res5: 64).
//
//   out[voxel m][channel n] = act( (sum_k X'[m][k] W[n][k]) * scale[n] + shift[n] + residual[m][n] )
//
// Same building blocks as pv_gemm9.hip (whole 128-byte rows staged by LDS-DMA one 16 KB unit at a time, multiply-free
// staging addresses, counted `vmcnt`, the two halves of the workgroup one barrier apart, the DMA stream running on across
// output tiles, DMAs issued among the MFMAs), re-cut for the smaller tile:
//   * 512 threads = 8 waves as 2 (voxels) x 4 (channels); a wave owns 64 voxels x 64 channels = 4 accumulator blocks of
//     v_mfma_f32_32x32x16_bf16 (64 registers: half of the big kernel's -- the fragment sets fit without tricks);
//   * a K step of 64 is THREE staging units: channel halves A0 / A1 (128 rows: for each channel group its first / second 32
//     channels) and ONE voxel unit B (128 rows), multiplied in TWO PHASES of 8 MFMAs: phase 0 = A0 x (v0, v1), phase 1 =
//     A1 x (v0, v1).  A0 and B are dead after phase 0, A1 after phase 1; fragment reads per K tile: 4 + 8 + 4 `ds_read_b128`
//     for 16 MFMAs (the 4 x 2 wave layout of the big kernel would need 20: at 128 x 256 the LDS port, not the L2 -> LDS
//     path, is the first ceiling: 176 KB of LDS traffic per K tile of 1024 matrix-pipe cycles);
//   * two phases per K tile leave too little time between "rows dead" and "rows needed again" on two buffers, so LDS holds
//     THREE K tiles (3 x 48 KB): in phase 0 of K tile t the stream requests A0(t+2) and the first half of B(t+2) into the
//     buffer K tile t-1 has just left, in phase 1 the second half of B(t+2) and A1(t+2): three DMAs per thread and phase,
//     behind MFMA 2, 4 and 6, every unit requested four phases before its first read.  K tiles come in triples (K % 192 == 0:
//     every LDS address is a per-lane base + an immediate); the waits are the counted `vmcnt(6)` / `vmcnt(5)`.
#include <stdlib.h>
#include "pv_common.h"

__device__ __attribute__((aligned(16))) unsigned int pv_zero_pageh[4] = {0u, 0u, 0u, 0u};

namespace {

constexpr int kThreadsH = 512;
constexpr int BMH = 128, BNH = 256;      // tile: voxels x channels
constexpr int UNITH = 128 * 64;          // elements of one staging unit (128 rows x 64 K, 16 KB)
constexpr int kBufBytesH = 3 * UNITH * 2;   // one K tile: [A0 | A1 | B], 48 KB
constexpr int kTabH = 3 * kBufBytesH;       // epilogue tables: [tile parity][scale 256 f32 | shift 256 f32]
constexpr int kGeoH = kTabH + 2 * 2048;     // implicit-GEMM staging geometry: 16 bytes per thread [off j0, off j1, mask j0, mask j1]
constexpr int kLdsHBytes = kGeoH + kThreadsH * 16;   // 156 KB

typedef const __attribute__((address_space(1))) void* gptrh_t;
typedef __attribute__((address_space(3))) void* lptrh_t;
typedef int i32x4h __attribute__((ext_vector_type(4)));

// LDS row -> channel inside a 32-row MFMA tile (accumulator register r of lane-half hi is channel 16*hi + r)
__device__ __forceinline__ int chih(int rho) { return 16 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 3) + (rho & 3); }

template <bool PW, bool YF32, bool TR>
__global__ __launch_bounds__(kThreadsH) void gemm_quad_half_kernel(const pv_conv3d_desc d, int tiles_n, int total_tiles, int tap_rot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemh_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smemh_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int S_out = d.To * d.Ho * d.Wo;
  const long M = (long)d.B * S_out;
  const int K = d.kt * d.kh * d.kw * d.cin;
  const int nk = K >> 6;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const int dil_t = d.dil_t > 1 ? d.dil_t : 1, dil_h = d.dil_h > 1 ? d.dil_h : 1, dil_w = d.dil_w > 1 ? d.dil_w : 1;
  const bf16_t* __restrict__ X = static_cast<const bf16_t*>(d.x);
  const char* __restrict__ Wb = static_cast<const char*>(d.w);
  const unsigned long zaddr = (unsigned long)reinterpret_cast<const bf16_t*>(pv_zero_pageh);

  // TR: the transposed tile, 256 voxels x 128 channels -- the SPLIT operand (two 128-row units, one per phase) is the voxels,
  // the WHOLE one (a single unit, read in phase 0 and kept) the weights; everything else is the same loop
  constexpr int BM = TR ? 2 * BMH : BMH, BN = TR ? BNH / 2 : BNH;
  auto tile_origin = [&](int it, long& m0, int& n0) __attribute__((always_inline)) {   // XCD-aware tile order (bijective for any tile count)
    const int xcd = it & 7, slot = it >> 3;
    const int qn = total_tiles >> 3, rn = total_tiles & 7;
    const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
    m0 = (long)(tile / tiles_n) * BM;
    n0 = (tile % tiles_n) * BN;
  };
  // DMA j of a unit covers unit rows 64 j + rho0, rho0 = 8 wave + lane / 8; K chunk (8 elements) that lands on LDS position
  // lane % 8 of those rows (the swizzle key (row >> 1) & 7 is the same for rows 64 apart)
  const int rho0 = 8 * wave + (lane >> 3);
  const int chunk8 = ((lane & 7) ^ ((rho0 >> 1) & 7)) * 8;
  const unsigned a_pitch = (unsigned)K * 2u, a_last = (unsigned)(d.cout - 1) * a_pitch;
  const unsigned b_pitch = (unsigned)d.ldx * 2u, b_last = (unsigned)(M - 1) * b_pitch;
  int iss_c0 = 0, iss_dt = 0, iss_dh = 0, iss_dw = 0;   // channel offset inside the tap, tap coordinates (wave-uniform)
  unsigned iss_wk = 0;                                  // byte offset of the stream's K tile inside a weight row
  unsigned g_a_row = 0, g_b_row = 0;   // byte offsets of this thread's staging rows (half 0, j = 0) of the stream's output tile
  unsigned gm0 = 0, gm1 = 0, gm2 = 0, gm3 = 0;   // (TR, implicit GEMM) window masks of the four voxel rows [v][j]
  // Row maps (g4 = wave & 3, g2 = wave >> 2; DMA j of a unit covers its rows 64 j + rho0):
  //   split unit h, row 32 g4 + i;  whole unit, row 64 g2 + 32 h + i;
  //   !TR: channel n0 + 64 g4 + 32 h + chi(i) (split)   voxel m0 + 64 g2 + 32 h + i (whole)
  //    TR: voxel m0 + 64 g4 + 32 h + i (split)          channel n0 + 64 g2 + 32 h + chi(i) (whole)
  auto geom_of = [&](int it) __attribute__((always_inline)) {
    long m0;
    int n0;
    tile_origin(it, m0, n0);
    iss_c0 = iss_dh = iss_dw = 0;
    // temporal-tap rotation (see pv_gemm9.hip): the tile of output frame t starts at tap (pt - t) mod kt and wraps
    int rot = 0;
    if constexpr (!PW) {
      if (tap_rot) {
        const unsigned sp = (unsigned)m0 % (unsigned)S_out;
        rot = (d.pt - (int)(sp / (unsigned)(d.Ho * d.Wo))) % d.kt;
        rot = rot < 0 ? rot + d.kt : rot;
      }
    }
    iss_dt = rot;
    iss_wk = (unsigned)(rot * d.kh * d.kw * d.cin) * 2u;
    if constexpr (!TR) g_a_row = (unsigned)(n0 + 64 * (rho0 >> 5) + chih(rho0 & 31)) * a_pitch;   // (h, j): 32 h + 128 j rows further, clamped at use
    else g_a_row = (unsigned)(n0 + 32 * (rho0 >> 5) + chih(rho0 & 31)) * a_pitch;                 // j: 64 rows further
    if constexpr (PW) {
      if constexpr (!TR) g_b_row = (unsigned)((int)m0 + rho0) * b_pitch;                          // j: 64 rows further, clamped at use
      else g_b_row = (unsigned)((int)m0 + 64 * (rho0 >> 5) + (rho0 & 31)) * b_pitch;              // (h, j): 32 h + 128 j rows further
    } else {
      int* geo = reinterpret_cast<int*>(smemh_raw + kGeoH) + tid * 4;
      unsigned msk4[4];
#pragma unroll
      for (int q4 = 0; q4 < (TR ? 4 : 2); ++q4) {
        const int j = q4 & 1, h = q4 >> 1;
        long m = TR ? m0 + 128 * j + 64 * (rho0 >> 5) + 32 * h + (rho0 & 31) : m0 + 64 * j + rho0;
        m = m < M ? m : M - 1;                              // M tail: a clamped row, never stored
        const unsigned b = (unsigned)m / (unsigned)S_out;
        const unsigned sp = (unsigned)m - b * (unsigned)S_out;
        const unsigned to = sp / (unsigned)(d.Ho * d.Wo);
        const unsigned r2 = sp - to * (unsigned)(d.Ho * d.Wo);
        const unsigned ho = r2 / (unsigned)d.Wo;
        const int t0 = (int)to * d.st - d.pt, h0 = (int)ho * d.sh - d.ph, w0 = (int)(r2 - ho * (unsigned)d.Wo) * d.sw - d.pw;
        geo[q4] = (int)((long)b * d.x_bs + ((long)(t0 * d.Hi + h0) * d.Wi + w0) * d.ldx) + chunk8;
        unsigned msk = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (q < d.kt && (unsigned)(t0 + q * dil_t) < (unsigned)d.Ti) msk |= 1u << q;
          if (q < d.kh && (unsigned)(h0 + q * dil_h) < (unsigned)d.Hi) msk |= 1u << (8 + q);
          if (q < d.kw && (unsigned)(w0 + q * dil_w) < (unsigned)d.Wi) msk |= 1u << (16 + q);
        }
        msk4[q4] = msk;
      }
      if constexpr (TR) {       // 16 bytes of LDS per thread hold the four offsets; the masks stay in registers
        gm0 = msk4[0]; gm1 = msk4[1]; gm2 = msk4[2]; gm3 = msk4[3];
      } else {                  // [off j0, off j1, mask j0, mask j1]
        geo[2] = (int)msk4[0]; geo[3] = (int)msk4[1];
      }
    }
  };

  // ---- issue side: the DMA stream, K tile by K tile across output tiles (all of this state is wave-uniform) ----
  int iss_it = blockIdx.x, iss_ku = 0, iss_jt = 0;   // work item, K tile inside it, this workgroup's tile count
  bool iss_live = iss_it < total_tiles;
  geom_of(iss_live ? iss_it : 0);

  // source selection with bit masks, not `?:` (a select between two pointers becomes two exec-masked DMAs)
  auto pick = [&](bool ok, unsigned long p) __attribute__((always_inline)) -> const bf16_t* {
    const unsigned long m = 0ul - (unsigned long)ok;
    return reinterpret_cast<const bf16_t*>((p & m) | (zaddr & ~m));
  };
  auto dma_w = [&](int rows, int unit, int j) __attribute__((always_inline)) {   // one weight row per lane: g_a_row + rows
    unsigned off = g_a_row + (unsigned)rows * a_pitch;
    off = (off < a_last ? off : a_last) + iss_wk + (unsigned)(chunk8 * 2);   // N tail: a clamped row, zeroed in the epilogue
    __builtin_amdgcn_global_load_lds((gptrh_t)pick(iss_live, (unsigned long)(Wb + off)),
                                     (lptrh_t)(smem + unit * UNITH + (j * 8 + wave) * 512), 16, 0, 0);
  };
  auto dma_x = [&](int rows, int goff, unsigned gmask, int unit, int j) __attribute__((always_inline)) {   // one voxel row per lane
    if constexpr (PW) {
      unsigned off = g_b_row + (unsigned)rows * b_pitch;
      off = (off < b_last ? off : b_last) + (unsigned)(iss_ku * 128 + chunk8 * 2);   // M tail: a clamped row, never stored
      __builtin_amdgcn_global_load_lds((gptrh_t)pick(iss_live, (unsigned long)(reinterpret_cast<const char*>(X) + off)),
                                       (lptrh_t)(smem + unit * UNITH + (j * 8 + wave) * 512), 16, 0, 0);
    } else {
      // element offset of the K tile's tap + channel block and the mask bits that must be set for this tap (wave-uniform)
      const int uni = ((iss_dt * dil_t * d.Hi + iss_dh * dil_h) * d.Wi + iss_dw * dil_w) * d.ldx + iss_c0;
      const unsigned sel = (1u << iss_dt) | (1u << (8 + iss_dh)) | (1u << (16 + iss_dw));
      const bool ok = iss_live && (gmask & sel) == sel;
      __builtin_amdgcn_global_load_lds((gptrh_t)pick(ok, (unsigned long)(X + (long)(goff + uni))),
                                       (lptrh_t)(smem + unit * UNITH + (j * 8 + wave) * 512), 16, 0, 0);
    }
  };
  // DMA j of split unit h / of the whole unit of the stream's K tile -> LDS buffer buf (gq: this thread's geometry words)
  auto issue_s = [&](int h, int buf, int j, const i32x4h& gq) __attribute__((always_inline)) {
    if constexpr (!TR) dma_w(32 * h + 128 * j, 3 * buf + h, j);
    else dma_x(32 * h + 128 * j, gq[2 * h + j], h == 0 ? (j == 0 ? gm0 : gm1) : (j == 0 ? gm2 : gm3), 3 * buf + h, j);
  };
  auto issue_w = [&](int buf, int j, const i32x4h& gq) __attribute__((always_inline)) {
    if constexpr (!TR) dma_x(64 * j, gq[j], (unsigned)gq[2 + j], 3 * buf + 2, j);
    else dma_w(64 * j, 3 * buf + 2, j);
  };
  auto load_geo = [&]() __attribute__((always_inline)) -> i32x4h {
    if constexpr (PW) {
      return i32x4h{0, 0, 0, 0};
    } else {
      unsigned t = threadIdx.x;   // (rebuilt behind an empty asm: see pv_gemm9.hip)
      asm volatile("" : "+v"(t));
      return *reinterpret_cast<const i32x4h*>(smemh_raw + kGeoH + t * 16u);
    }
  };
  // folded BatchNorm / bias tables of the stream's tile -> LDS, one 4-byte DMA per thread (wave w < 4: scale[64 w .. 64 w + 63]
  // of the tile's 256 channels, w >= 4: shift).  Extra DMAs inside the counted windows only make the waits stricter.
  auto issue_tables = [&]() __attribute__((always_inline)) {
    const float* tab = wave < 4 ? d.scale : d.shift;
    if (tab != nullptr && iss_live) {
      long m0;
      int n0;
      tile_origin(iss_it, m0, n0);
      int n = n0 + 64 * (wave & 3) + lane;
      n = n < d.cout ? n : d.cout - 1;
      __builtin_amdgcn_global_load_lds((gptrh_t)(tab + n), (lptrh_t)(smemh_raw + kTabH + (iss_jt & 1) * 2048 + wave * 256), 4, 0, 0);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {   // next K tile of the stream
    ++iss_ku;
    iss_wk += 128u;
    if constexpr (!PW) {
      iss_c0 += 64;
      if (iss_c0 == d.cin) {
        iss_c0 = 0;
        if (++iss_dw == d.kw) {
          iss_dw = 0;
          if (++iss_dh == d.kh) {
            iss_dh = 0;
            if (++iss_dt == d.kt) { iss_dt = 0; iss_wk = 0u; }   // (a rotated reduction wraps to tap 0 = weight column 0)
          }
        }
      }
    }
    if (iss_ku == nk) {
      iss_ku = 0;
      iss_wk = 0u;
      iss_it += gridDim.x;
      ++iss_jt;
      iss_live = iss_it < total_tiles;
      if (iss_live) geom_of(iss_it);
      issue_tables();
    }
  };

  // ---- read side: this lane's fragment position inside a unit, as a BYTE offset from the LDS base ----
  // split unit h: row 32 g4 + (lane & 31); whole unit: row 64 g2 + 32 h + (lane & 31) (h = 1: + 4096 bytes, same swizzle key).
  // K slice s reads chunk (2 s + hi) ^ key(row) = (hi ^ key(row)) ^ 2 s: the position of slice 0 in a register, slices 1-3 are
  // that register ^ (s * 32)
  const int g4 = wave & 3, g2 = wave >> 2;
  const int wn = TR ? g2 : g4, wm = TR ? g4 : g2;   // the wave's 64 channels / 64 voxels inside the tile
  unsigned rd_a0, rd_b0;                            // split / whole operand
  {
    const int ra = g4 * 32 + (lane & 31), rb = g2 * 64 + (lane & 31);
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) unsigned char*)smemh_raw);
    rd_a0 = lds0 + (unsigned)(ra * 128 + (((lane >> 5) ^ ((ra >> 1) & 7)) << 4));
    rd_b0 = lds0 + (unsigned)(2 * UNITH * 2 + rb * 128 + (((lane >> 5) ^ ((rb >> 1) & 7)) << 4));
  }
  typedef const __attribute__((address_space(3))) bf16x8* lfragh_t;
#define PVH_RD(BASE, S, BYTES) (*(lfragh_t)(unsigned long)(((BASE) ^ ((S) * 32u)) + (unsigned)(BYTES)))

  // s_waitcnt simm16 on gfx9: vmcnt = bits [3:0] | [15:14]; expcnt [6:4] "no wait"; lgkmcnt [11:8] = 0
  constexpr auto vml = [](int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4); };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned kOOB = 0x80000000u;
  constexpr int kStores = YF32 ? 16 : 8;   // stores per thread and output tile (whatever is masked)
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      d.y, 0, (int)((unsigned)d.B * (unsigned)d.y_bs * (YF32 ? 4u : 2u)), 0x00020000);

  // ---- prologue: the first tile's tables, K tiles 0 and 1 in stream order (A0, B, A1) ----
  issue_tables();
  {
    const i32x4h gq = load_geo();
    issue_s(0, 0, 0, gq); issue_s(0, 0, 1, gq); issue_w(0, 0, gq); issue_w(0, 1, gq); issue_s(1, 0, 0, gq); issue_s(1, 0, 1, gq);
    advance();
    issue_s(0, 1, 0, gq); issue_s(0, 1, 1, gq); issue_w(1, 0, gq); issue_w(1, 1, gq); issue_s(1, 1, 0, gq); issue_s(1, 1, 1, gq);
  }
  __builtin_amdgcn_s_waitcnt(vml(8));   // A0, B of K tile 0: this thread's share
  __builtin_amdgcn_s_barrier();
  const bool half_b = wave >= 4;        // wave-uniform
  if (half_b) __builtin_amdgcn_s_barrier();   // the offset between the two halves
  __builtin_amdgcn_sched_barrier(0);

  bool stores_behind = false;   // the previous tile's stores sit behind the units the first two K tiles' waits cover (wave-uniform)
  int jt = 0;                   // tiles finished by this workgroup (table parity)
  f32x16 acc[2][2];             // [channel half a][voxel half v]

  // one phase:  [reads] wait(N) | B1 | 8 MFMAs with the phase's three DMAs behind MFMA 2, 4 and 6 | B2
#define PVH_WAIT_B1(FIRST, N)                                                                       \
  do {                                                                                              \
    if ((FIRST) && stores_behind) __builtin_amdgcn_s_waitcnt(vml((N) + kStores));                   \
    else __builtin_amdgcn_s_waitcnt(vml(N));                                                        \
    __builtin_amdgcn_s_barrier();                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
  // phase P multiplies split unit P (fragments af) with both halves H of the whole unit (bf): weights are always the A operand
#define PVH_M1(P, S, H)                                                                             \
  do {                                                                                              \
    if constexpr (!TR) acc[P][H] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[S], bf[S][H], acc[P][H], 0, 0, 0);   \
    else acc[H][P] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[S][H], af[S], acc[H][P], 0, 0, 0);                 \
  } while (0)
#define PVH_SLOT(X)                                                                                 \
  do { __builtin_amdgcn_sched_barrier(0); X; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PVH_PHASE(FIRST, N, A, D0, D1, D2)                                                          \
  do {                                                                                              \
    PVH_WAIT_B1(FIRST, N);                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    PVH_M1(A, 0, 0); PVH_M1(A, 0, 1); PVH_SLOT(D0);                                                 \
    PVH_M1(A, 1, 0); PVH_M1(A, 1, 1); PVH_SLOT(D1);                                                 \
    PVH_M1(A, 2, 0); PVH_M1(A, 2, 1); PVH_SLOT(D2);                                                 \
    PVH_M1(A, 3, 0); PVH_M1(A, 3, 1);                                                               \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    __builtin_amdgcn_s_barrier();                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#define PVH_READ_A(Q, A)                                                                                                  \
  do {                                                                                                                    \
    asm volatile("" : "+v"(rd_a0));                                                                                       \
    const unsigned ba = rd_a0 + (unsigned)((Q) * kBufBytesH + (A) * UNITH * 2);                                           \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) af[s] = PVH_RD(ba, s, 0);                                               \
  } while (0)
#define PVH_READ_B(Q)                                                                                                     \
  do {                                                                                                                    \
    asm volatile("" : "+v"(rd_b0));                                                                                       \
    const unsigned bb = rd_b0 + (unsigned)((Q) * kBufBytesH);                                                             \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                         \
      _Pragma("unroll") for (int v = 0; v < 2; ++v) bf[s][v] = PVH_RD(bb, s, v * 4096);                                   \
  } while (0)
  // (order of a K tile's six DMAs: A0 A0 B | B A1 A1.  The voxel rows first -- B B A0 | A0 A1 A1 -- and 4 + 2 -- B B A0 A0 |
  // A1 A1 -- measured the same to 0.3 % on every SlowFast shape, profiles/r5/bench_gemm_half_dma_order_call11.txt: the loop is
  // not waiting for data, it pays for issuing the DMAs and for the LDS port.)
  // One K tile on LDS buffer Q (compile-time); the stream writes buffer (Q + 2) % 3.  A phase's fragments were guaranteed by the
  // PREVIOUS phase's wait + barrier; its own wait (before its DMAs are issued) covers what the next phase reads: phase 0 waits
  // for A1 of this K tile (younger: the next K tile's six DMAs), phase 1 for A0 and B of the next K tile (younger: its A1 and
  // this K tile's phase-0 DMAs: five).  F0 / F1: the previous output tile's stores sit behind the units that wait covers (the
  // first three phases after an epilogue).
#define PVH_KTILE(Q, F0, F1)                                                                                              \
  do {                                                                                                                    \
    constexpr int QW = ((Q) + 2) % 3;                                                                                     \
    bf16x8 af[4], bf[4][2];                                                                                               \
    advance();                       /* the stream moves on to the K tile after next */                                   \
    const i32x4h gq = load_geo();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    PVH_READ_B(Q);                                                                                                        \
    PVH_READ_A(Q, 0);                                                                                                     \
    PVH_PHASE(F0, 6, 0, issue_s(0, QW, 0, gq), issue_s(0, QW, 1, gq), issue_w(QW, 0, gq));                                     \
    PVH_READ_A(Q, 1);                                                                                                     \
    PVH_PHASE(F1, 5, 1, issue_w(QW, 1, gq), issue_s(1, QW, 0, gq), issue_s(1, QW, 1, gq));                                     \
  } while (0)

  const int nk3 = nk / 3;   // K tiles come in triples (K % 192 == 0, host check): every output tile starts on LDS buffer 0
  for (int it = blockIdx.x; it < total_tiles; it += gridDim.x, ++jt) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][v][r] = 0.f;

    PVH_KTILE(0, true, true);
    PVH_KTILE(1, true, false);
    PVH_KTILE(2, false, false);
    for (int k3 = 1; k3 < nk3; ++k3) {
      PVH_KTILE(0, false, false);
      PVH_KTILE(1, false, false);
      PVH_KTILE(2, false, false);
    }

    // both halves run their epilogues side by side (see pv_gemm9.hip): one extra barrier of the first half here, matched by the
    // second half's last B2; one of the second half at the end of its epilogue, matched by the first half's next B1
    if (!half_b) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int l31 = lane & 31, hi = lane >> 5;
    // ---- epilogue: lane owns channels cb..cb+15 of voxel m for every (channel half a, voxel half v) ----
    long m0;
    int n0;
    tile_origin(it, m0, n0);
    long e_b[2], e_sp[2];
    bool e_ok[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const long m = m0 + wm * 64 + v * 32 + l31;
      e_ok[v] = m < M;
      const long mm = e_ok[v] ? m : 0;
      e_b[v] = (long)((unsigned)mm / (unsigned)S_out);
      e_sp[v] = mm - e_b[v] * S_out;
    }
    const float* tabs = reinterpret_cast<const float*>(smemh_raw + kTabH + (jt & 1) * 2048);
    typedef f32x4 res_t[2][2][2];   // [v][h8][half]: 8 channels as 2 x f32x4 (fp32) or 1 x 16 bytes (bf16)
    auto load_res = [&](int a, res_t& res) __attribute__((always_inline)) {
      if (d.residual == nullptr) return;
      const int cb = n0 + wn * 64 + a * 32 + 16 * hi;
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          const bool ok = e_ok[v] && cb + h8 * 8 < cout_p8;
          const long ro = ok ? e_b[v] * d.r_bs + e_sp[v] * d.ldr + cb + h8 * 8 : 0;
          if (d.r_f32) {
            const float* rp = static_cast<const float*>(d.residual) + ro;
            res[v][h8][0] = *reinterpret_cast<const f32x4*>(rp);
            res[v][h8][1] = *reinterpret_cast<const f32x4*>(rp + 4);
          } else {
            res[v][h8][0] = *reinterpret_cast<const f32x4*>(static_cast<const bf16_t*>(d.residual) + ro);
          }
        }
    };
    auto finish = [&](int a, const res_t& res) __attribute__((always_inline)) {
      const int cl = wn * 64 + a * 32 + 16 * hi;   // channel inside the tile
      const int cb = n0 + cl;
      if (d.scale != nullptr) {
        f32x4 sc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sc[q] = *reinterpret_cast<const f32x4*>(tabs + cl + 4 * q);
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] *= sc[r >> 2][r & 3];
      }
      if (d.shift != nullptr) {
        f32x4 sh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sh[q] = *reinterpret_cast<const f32x4*>(tabs + 256 + cl + 4 * q);
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] += sh[r >> 2][r & 3];
      }
      if (d.residual != nullptr) {
        if (d.r_f32) {
#pragma unroll
          for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][v][r] += res[v][r >> 3][(r >> 2) & 1][r & 3];
        } else {
#pragma unroll
          for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][v][r] += (float)__builtin_bit_cast(bf16x8, res[v][r >> 3][0])[r & 7];
        }
      }
      if (d.act == PV_ACT_RELU) {
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] = fmaxf(acc[a][v][r], 0.f);
      } else if (d.act == PV_ACT_GELU) {
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] = pv_gelu_fast(acc[a][v][r]);
      } else if (d.act == PV_ACT_SWISH) {
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] *= pv_sigmoid(acc[a][v][r]);
      } else if (d.act == PV_ACT_SIGMOID) {
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] = pv_sigmoid(acc[a][v][r]);
      }
      if (cb + 16 > d.cout) {   // ragged last channel tile: the padding up to the 8-multiple is written as zeros
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] = cb + r < d.cout ? acc[a][v][r] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    {   // (literal channel-half indices: see pv_gemm9.hip)
      res_t r0;
      load_res(0, r0);
      finish(0, r0);
      load_res(1, r0);
      finish(1, r0);
    }
    // ... then nothing but stores (kStores of them, whatever is masked)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int cb = n0 + wn * 64 + a * 32 + 16 * hi;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const unsigned yo = (unsigned)(e_b[v] * d.y_bs + e_sp[v] * d.ldy + cb);
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          const bool ok = e_ok[v] && cb + h8 * 8 < cout_p8;
          const int r0 = h8 * 8;
          if constexpr (YF32) {
            const unsigned off = ok ? (yo + r0) * 4u : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(
                u32x4{__float_as_uint(acc[a][v][r0 + 0]), __float_as_uint(acc[a][v][r0 + 1]),
                      __float_as_uint(acc[a][v][r0 + 2]), __float_as_uint(acc[a][v][r0 + 3])}, ry, (int)off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(
                u32x4{__float_as_uint(acc[a][v][r0 + 4]), __float_as_uint(acc[a][v][r0 + 5]),
                      __float_as_uint(acc[a][v][r0 + 6]), __float_as_uint(acc[a][v][r0 + 7])}, ry,
                (int)(ok ? off + 16u : kOOB), 0, 0);
          } else {
            bf16x8 ob;
#pragma unroll
            for (int r = 0; r < 8; ++r) ob[r] = (bf16_t)acc[a][v][r0 + r];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ob), ry, (int)(ok ? (yo + r0) * 2u : kOOB), 0, 0);
          }
        }
      }
    }
    stores_behind = true;
    if (half_b) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
#undef PVH_KTILE
#undef PVH_PHASE
#undef PVH_SLOT
#undef PVH_M1
#undef PVH_READ_A
#undef PVH_READ_B
#undef PVH_WAIT_B1
#undef PVH_RD
  if (!half_b) __builtin_amdgcn_s_barrier();   // matches the second half's offset barrier
  __builtin_amdgcn_s_waitcnt(vml(0));          // the stream's trailing (zero-page) DMAs land before the LDS is released
}

template <bool PW, bool YF32, bool TR>
int launch9h(const pv_conv3d_desc& d, int tiles_n, long total, hipStream_t s) {
  const size_t lds = (size_t)kLdsHBytes;
  auto kern = gemm_quad_half_kernel<PW, YF32, TR>;
  PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long resident = 256;   // one workgroup per CU
  dim3 grid((unsigned)(total < resident ? total : resident)), block(kThreadsH);
  const int tap_rot = !PW && d.kt > 1 && d.st == 1 && d.dil_t <= 1 && (d.Ho * d.Wo) % (TR ? 2 * BMH : BMH) == 0 && pv_tune("gemm9_tap_rot", 1);
  PV_LAUNCH(kern, grid, block, lds, s, d, tiles_n, (int)total, tap_rot);
  pv_note_kernel("gemm_quad_half_kernel");
  PV_LAUNCH_CHECK();
  return PV_OK;
}

}  // namespace

// Returns PV_OK when this kernel took the op, PV_ERR_UNSUPPORTED to leave it to the other GEMM kernels.  Called by
// pv_gemm9_try for the layers its 256 x 256 tiles cannot spread over the chip.
int pv_gemm9h_try(const pv_conv3d_desc& d, bool pw, hipStream_t s) {
  const int mode = pv_tune("gemm9h", 1);   // 0 off, 1 heuristic, 2 wherever the kernel can run
  if (mode == 0) return PV_ERR_UNSUPPORTED;
  if (d.dtype != PV_BF16 || d.a_gate != nullptr || d.a_act != PV_ACT_NONE || d.x2 != nullptr) return PV_ERR_UNSUPPORTED;
  if (d.kt > 8 || d.kh > 8 || d.kw > 8) return PV_ERR_UNSUPPORTED;                  // 8-bit window masks per axis
  if (d.dil_t > 1 || d.dil_h > 1 || d.dil_w > 1) return PV_ERR_UNSUPPORTED;         // dilated: the 128 x 128 kernel (see pv_gemm9.hip)
  const long K = (long)d.kt * d.kh * d.kw * d.cin;
  if (d.cin % 64 != 0 || K % 192 != 0 || K < 384) return PV_ERR_UNSUPPORTED;        // a K step inside one tap; K tiles in triples
  const long M = (long)d.B * d.To * d.Ho * d.Wo;
  const int cout_p8 = pv_round_up(d.cout, 8);
  // 31-bit element offsets into x, 32-bit byte offsets into w, 31-bit byte offsets in the store descriptor
  if (M > 0x7fffffffL || (long)d.B * d.x_bs > 0x7fffffffL || ((long)d.cout + 256) * K * 2 > 0xffffffffL ||
      (M + 256) * d.ldx * 2 > 0xffffffffL)
    return PV_ERR_UNSUPPORTED;
  if ((long)d.B * d.y_bs * (d.y_f32 ? 4 : 2) > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  // tile shape: 128 voxels x 256 channels where 256-channel tiles fit the layer, else the transposed 256 x 128 tile (SlowFast
  // res3's 1x3x3 convs: 128 channels) -- pv_tune "gemm9h_tr": -1 by the padding each shape costs, 0 / 1 forced
  const double waste256 = (double)(pv_ceil_div(cout_p8, BNH) * BNH - cout_p8) / (double)cout_p8;
  const double waste128 = (double)(pv_ceil_div(cout_p8, BNH / 2) * (BNH / 2) - cout_p8) / (double)cout_p8;
  int tr = pv_tune("gemm9h_tr", -1);
  if (tr < 0) tr = waste256 > 0.15 && waste128 <= 0.15;
  const long tiles_m = pv_ceil_div(M, tr ? 2 * BMH : BMH);
  const int tiles_n = (int)pv_ceil_div(cout_p8, tr ? BNH / 2 : BNH);
  const long total = tiles_m * tiles_n;
  if (total <= 0 || total >= 0x3fffffffL) return PV_ERR_UNSUPPORTED;
  if (mode == 1) {
    // (the 256 x 256 kernel is the better one wherever ITS tiles fill the chip: pv_gemm9_try asks this kernel only below that
    // or where 256-channel tiles would be mostly padding)
    if ((tr ? waste128 : waste256) > 0.15) return PV_ERR_UNSUPPORTED;
    // measured in the two-branch bench forms (profiles/r6/model_ab_gemm9h_min_tiles_call95.txt): 96 -> 32 is +3 % on SlowFast-R50
    // (res5's layers at 8 clips per branch: 64 / 32 tiles, still ahead of the 128 x 128 kernel on a quarter of the chip),
    // transposed 200 -> 100 +2.5 % on MViT-B, +-0 on SlowFast-R50
    if (total < (tr ? pv_tune("gemm9h_tr_min_tiles", 100) : pv_tune("gemm9h_min_tiles", 32))) return PV_ERR_UNSUPPORTED;
  }
  const bool rows = pw && d.x_bs == (long)d.To * d.Ho * d.Wo * d.ldx;
#define PVH_GO(PWv, YFv)                                                \
  return tr ? launch9h<PWv, YFv, true>(d, tiles_n, total, s) : launch9h<PWv, YFv, false>(d, tiles_n, total, s);
  if (d.y_f32) {
    if (rows) { PVH_GO(true, true) } else { PVH_GO(false, true) }
  } else {
    if (rows) { PVH_GO(true, false) } else { PVH_GO(false, false) }
  }
#undef PVH_GO
}
