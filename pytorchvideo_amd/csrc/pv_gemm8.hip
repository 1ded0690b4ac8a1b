// MFMA-bound dense convolution / linear layer, large-tile variant: NDHWC implicit GEMM on a 256-voxel x
// (128 | 256)-channel workgroup tile, 8 waves, a FOUR-stage LDS ring filled by LDS-DMA with counted waits.
//
//   out[voxel m][channel n] = act( (sum_k X'[m][k] W[n][k]) * scale[n] + shift[n] + residual[m][n] )
//   k = tap*cin + c ;  X'[m][k] = x[voxel m shifted by tap][c]  (zero outside the image)
//
// Why a second GEMM kernel next to pv_gemm.hip (128 x 128 tile, 4 waves, two LDS buffers): ablation builds of
// that kernel on the MI355X (tools/bench_gemm.py --tune=gemm_abl=1|2|3) show its operand pipeline sets the pace on
// LARGE problems -- without the MFMAs 32768 x 4096 x 4096 runs only 15 % faster, without the loads 1.6x: a
// 128 x 128 x 64 step stages 32 KB of L2 -> LDS traffic for 2.1 MFLOP (64 FLOP/B), and the chip sustains ~13-14 TB/s
// on that path.  A 256 x 256 tile doubles the FLOP per staged byte, a 256 x 128 tile gives 85.
// Structure of one workgroup (512 threads = 8 waves as 4 (voxels) x 2 (channels), one workgroup per CU):
//   * persistent: walks output tiles in an XCD-aware order; the stage stream of the LDS ring runs on ACROSS
//     tiles, so the first K steps of the next tile are landing while this tile's epilogue stores drain;
//   * ring of 4 stages x BK 32 (256-channel tiles) or 3 stages x BK 64 (128-channel tiles: full 128-byte lines per
//     row); every lane computes its own SOURCE address (implicit-GEMM gather, K / M / N tails -> a zero page) and
//     carries the LDS swizzle there, the LDS image is lane-linear; `ds_read_b128` fragment reads are conflict-free
//     (SQ_LDS_BANK_CONFLICT = 0 measured); waits are counted (`s_waitcnt vmcnt(N)`, raw `s_barrier`), never 0;
//   * the two halves of the workgroup run the same stream one barrier apart, so one wave of every SIMD issues
//     loads and reads fragments while the other multiplies (see the comment at the main loop);
//   * v_mfma_f32_32x32x16_bf16, weights as the A operand with LDS rows permuted so that a lane's 16 accumulator
//     registers are 16 CONSECUTIVE channels of one voxel: the epilogue (folded BN / bias, fp32-or-bf16 residual,
//     activation) stays in registers and leaves as 16-byte buffer stores (masked ones get an out-of-range offset:
//     every wave issues the same number of stores, which is what makes the counted waits exact).
// Where it is used: see the measurements at pv_gemm8_try() -- it wins on problems that fill several rounds of its
// tiles (large batches); the layers of the BASELINE workloads are too small for 256-voxel tiles and stay on the
// 128 x 128 kernel.
#include <stdlib.h>
#include "pv_common.h"

__device__ __attribute__((aligned(16))) unsigned int pv_zero_page8[4] = {0u, 0u, 0u, 0u};

namespace {

constexpr int kThreads8 = 512;
constexpr int BMV8 = 256;      // voxels per tile
constexpr int kMaxTaps8 = 512;

typedef const __attribute__((address_space(1))) void* gptr8_t;
typedef __attribute__((address_space(3))) void* lptr8_t;

__device__ __forceinline__ int chi8(int rho) {   // LDS row -> channel inside a 32-row MFMA tile group
  return (rho & ~31) + 16 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 3) + (rho & 3);
}

template <int NJW, int NJX>
struct Geom8 {        // per-thread staging geometry of one output tile (element offsets fit 31 bits: host check)
  int w_off[NJW];     // weight row of W item j, or -1
  int x_off[NJX];     // PW: voxel row ; general: clip offset ; -1: no such row
  int x_t[NJX], x_h[NJX], x_w[NJX];
};

// CT: 32-channel tiles per wave (2 -> 128-channel workgroup tile, 4 -> 256); BK8: K per stage (32: 64-byte rows,
// 64: full 128-byte lines per row and DMA piece); NS8: ring stages
template <bool PW, int CT, int BK8, int NS8>
__global__ __launch_bounds__(kThreads8, 2) void gemm8_kernel(const pv_conv3d_desc d, int tiles_n, int total_tiles,
                                                             float inv_cin) {
  constexpr int BN = 64 * CT;
  constexpr int CPR = BK8 / 8;                           // 16-byte chunks per tile row
  constexpr int LCPR = CPR == 4 ? 2 : 3;                 // log2
  constexpr int NJW = BN * CPR / kThreads8;              // 16-byte W items per thread per stage
  constexpr int NJX = BMV8 * CPR / kThreads8;            // ... X items
  constexpr int LPS = NJW + NJX;                         // loads per thread per stage
  constexpr int KSL = BK8 / 16;                          // 16-deep MFMA slices per stage
  constexpr int STAGE_ELEMS = (BN + BMV8) * BK8;         // bf16 elements per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem8_raw);
  int* s_tap = reinterpret_cast<int*>(smem8_raw + (size_t)NS8 * STAGE_ELEMS * 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;

  const int S_out = d.To * d.Ho * d.Wo;
  const long M = (long)d.B * S_out;
  const int taps = d.kt * d.kh * d.kw;
  const int K = taps * d.cin;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const bf16_t* __restrict__ X = static_cast<const bf16_t*>(d.x);
  const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(pv_zero_page8);

  if constexpr (!PW) {
    for (int t = tid; t < taps; t += kThreads8) {
      const int dt = t / (d.kh * d.kw);
      const int r = t - dt * d.kh * d.kw;
      const int dh = r / d.kw;
      s_tap[t] = (dt * (d.dil_t > 1 ? d.dil_t : 1)) | ((dh * (d.dil_h > 1 ? d.dil_h : 1)) << 8) |
                 (((r - dh * d.kw) * (d.dil_w > 1 ? d.dil_w : 1)) << 16);
    }
    __syncthreads();
  }

  // XOR swizzle of the chunk index that makes ds_read_b128 conflict-free (64-byte rows: 4 rows per bank row;
  // 128-byte rows: 2)
  auto swz = [](int row) { return CPR == 4 ? (row >> 2) & 3 : (row >> 1) & 7; };
  // item g = j*512 + tid -> (row g/CPR, LDS chunk position g%CPR); the logical K chunk it carries is tile-independent
  int kch_w[NJW], kch_x[NJX];
#pragma unroll
  for (int j = 0; j < NJW; ++j) { const int g = j * kThreads8 + tid; kch_w[j] = (g & (CPR - 1)) ^ swz(g >> LCPR); }
#pragma unroll
  for (int j = 0; j < NJX; ++j) { const int g = j * kThreads8 + tid; kch_x[j] = (g & (CPR - 1)) ^ swz(g >> LCPR); }

  auto tile_origin = [&](int it, long& m0, int& n0) {   // XCD-aware tile order (bijective for any tile count)
    const int xcd = it & 7, slot = it >> 3;
    const int qn = total_tiles >> 3, rn = total_tiles & 7;
    const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
    m0 = (long)(tile / tiles_n) * BMV8;
    n0 = (tile % tiles_n) * BN;
  };
  auto geom_of = [&](int it, Geom8<NJW, NJX>& gg) {
    long m0;
    int n0;
    tile_origin(it, m0, n0);
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      const int row = (j * kThreads8 + tid) >> LCPR;
      const int n = n0 + chi8(row);
      gg.w_off[j] = n < d.cout ? n * K : -1;
    }
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int row = (j * kThreads8 + tid) >> LCPR;
      const long m = m0 + row;
      gg.x_t[j] = gg.x_h[j] = gg.x_w[j] = 0;
      if (m < M) {
        const unsigned b = (unsigned)m / (unsigned)S_out;
        const unsigned sp = (unsigned)m - b * (unsigned)S_out;
        if constexpr (PW) {
          gg.x_off[j] = (int)((long)b * d.x_bs + (long)sp * d.ldx);
        } else {
          const unsigned to = sp / (unsigned)(d.Ho * d.Wo);
          const unsigned r2 = sp - to * (unsigned)(d.Ho * d.Wo);
          const unsigned ho = r2 / (unsigned)d.Wo;
          gg.x_off[j] = (int)((long)b * d.x_bs);
          gg.x_t[j] = (int)to * d.st - d.pt;
          gg.x_h[j] = (int)ho * d.sh - d.ph;
          gg.x_w[j] = (int)(r2 - ho * (unsigned)d.Wo) * d.sw - d.pw;
        }
      } else {
        gg.x_off[j] = -1;
      }
    }
  };

  // source selection with bit masks, not `?:` (a select between two pointers becomes two exec-masked DMAs)
  const unsigned long zaddr = (unsigned long)zero;
  auto pick = [&](bool ok, const bf16_t* p) -> const bf16_t* {
    const unsigned long m = 0ul - (unsigned long)ok;
    return reinterpret_cast<const bf16_t*>(((unsigned long)p & m) | (zaddr & ~m));
  };
  auto stage = [&](int slot, const Geom8<NJW, NJX>& gg, int ks, bool live) {
    bf16_t* wb = smem + slot * STAGE_ELEMS;
    bf16_t* xb = wb + BN * BK8;
    const int k0 = ks * BK8;
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      const int k = k0 + kch_w[j] * 8;
      const bool ok = live && k < K && gg.w_off[j] >= 0;
      __builtin_amdgcn_global_load_lds((gptr8_t)pick(ok, Wt + (gg.w_off[j] >= 0 ? gg.w_off[j] : 0) + k),
                                       (lptr8_t)(wb + (j * kThreads8 + wave * 64) * 8), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int k = k0 + kch_x[j] * 8;
      bool ok = live && k < K && gg.x_off[j] >= 0;
      long xo = gg.x_off[j] >= 0 ? gg.x_off[j] : 0;
      if constexpr (PW) {
        xo += k;
      } else {
        const int tap = (int)(((float)k + 0.5f) * inv_cin);
        const int tp = s_tap[tap < taps ? tap : 0];
        const int ti = gg.x_t[j] + (tp & 255), hh = gg.x_h[j] + ((tp >> 8) & 255), ww = gg.x_w[j] + (tp >> 16);
        ok = ok && (unsigned)ti < (unsigned)d.Ti && (unsigned)hh < (unsigned)d.Hi && (unsigned)ww < (unsigned)d.Wi;
        xo += ((long)(ti * d.Hi + hh) * d.Wi + ww) * d.ldx + (k - tap * d.cin);
      }
      __builtin_amdgcn_global_load_lds((gptr8_t)pick(ok, X + xo), (lptr8_t)(xb + (j * kThreads8 + wave * 64) * 8), 16, 0, 0);
    }
  };

  // read-side rows of this lane's fragments (fixed for the whole kernel)
  int a_row[CT], b_row[2];
#pragma unroll
  for (int t = 0; t < CT; ++t) a_row[t] = wn * 32 * CT + t * 32 + l31;
#pragma unroll
  for (int t = 0; t < 2; ++t) b_row[t] = wm * 64 + t * 32 + l31;
  // s_waitcnt simm16 on gfx9: vmcnt = bits [3:0] | [15:14]; expcnt [6:4] and lgkmcnt [11:8] left at "no wait"
  constexpr auto vm = [](int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned kOOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      d.y, 0, (int)((unsigned)d.B * (unsigned)d.y_bs * (d.y_f32 ? 4u : 2u)), 0x00020000);
  constexpr int E_BF16 = CT * 2 * 2, E_F32 = CT * 2 * 4;   // stores per wave per tile

  const int nk = (K + BK8 - 1) / BK8;
  // ---- issue side: runs NS8 - 1 stages ahead of the compute side, across tile boundaries ----
  Geom8<NJW, NJX> iss;
  int iss_it = blockIdx.x, iss_ks = 0, gi = 0;
  bool iss_live = iss_it < total_tiles;
  geom_of(iss_live ? iss_it : 0, iss);
  auto issue = [&]() {
    stage(gi, iss, iss_ks, iss_live);   // exhausted stream: zero-page loads keep the counts uniform
    gi = gi + 1 == NS8 ? 0 : gi + 1;
    if (++iss_ks == nk) {
      iss_ks = 0;
      iss_it += gridDim.x;
      iss_live = iss_it < total_tiles;
      if (iss_live) geom_of(iss_it, iss);
    }
  };
#pragma unroll
  for (int p = 0; p < NS8 - 1; ++p) issue();

  // The two halves of the workgroup (waves 0-3 / 4-7: one wave of each per SIMD) run the same instruction
  // stream ONE BARRIER APART: a stage is  B1 | L: issue the DMA of stage g+3, read the fragments of stage g |
  // B2 | C: 16 (8) MFMAs , and while one half multiplies (C) the other half issues loads and reads LDS (L), so
  // the SIMD's matrix pipe is fed by one wave while the other does its address arithmetic.
  //   RAW: a wave waits for ITS share of stage g+1 at the end of L(g), i.e. before B2(g); whichever half reads
  //        stage g+1 first does so after a barrier both halves reached after that wait.
  //   WAR: slot (g+3) % 4 was last read in L(g-1); those reads are waited for (lgkmcnt) before B2(g-1), which
  //        both halves pass before either issues the DMA of L(g).
  const bool half_b = wave >= 4;   // wave-uniform (scalar)
  int gslot = 0;         // ring slot of the stage being multiplied
  int pend = 0;          // stages for which the previous tile's stores may still sit in the queue (wave-uniform)
  __builtin_amdgcn_s_waitcnt(vm((NS8 - 2) * LPS));   // this wave's share of stage 0
  if (half_b) __builtin_amdgcn_s_barrier();  // the offset
  for (int it = blockIdx.x; it < total_tiles; it += gridDim.x) {
    long m0;
    int n0;
    tile_origin(it, m0, n0);
    f32x16 acc[CT][2];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][v][r] = 0.f;

    for (int ks = 0; ks < nk; ++ks) {
      __builtin_amdgcn_s_barrier();   // B1
      // ---- L ----
      const bf16_t* wb = smem + gslot * STAGE_ELEMS;
      const bf16_t* xb = wb + BN * BK8;
      bf16x8 af[KSL][CT], bfr[KSL][2];
#pragma unroll
      for (int s = 0; s < KSL; ++s) {
        const int c = 2 * s + hi;
#pragma unroll
        for (int t = 0; t < CT; ++t)
          af[s][t] = *reinterpret_cast<const bf16x8*>(wb + a_row[t] * BK8 + ((c ^ swz(a_row[t])) << 3));
#pragma unroll
        for (int t = 0; t < 2; ++t)
          bfr[s][t] = *reinterpret_cast<const bf16x8*>(xb + b_row[t] * BK8 + ((c ^ swz(b_row[t])) << 3));
      }
      issue();
      // this wave's share of stage gc+1 has landed when at most the two youngest stages (and, after an epilogue,
      // that tile's stores, issued between them) are outstanding; the fragment reads are complete as well
      if (pend > 0) {
        if (d.y_f32) __builtin_amdgcn_s_waitcnt(vm((NS8 - 2) * LPS + E_F32) & ~(15 << 8));
        else __builtin_amdgcn_s_waitcnt(vm((NS8 - 2) * LPS + E_BF16) & ~(15 << 8));
        --pend;
      } else {
        __builtin_amdgcn_s_waitcnt(vm((NS8 - 2) * LPS) & ~(15 << 8));   // ... and lgkmcnt(0)
      }
      __builtin_amdgcn_s_barrier();   // B2
      // ---- C ----
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < KSL; ++s)
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
          for (int v = 0; v < 2; ++v)
            acc[a][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][a], bfr[s][v], acc[a][v], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      gslot = gslot + 1 == NS8 ? 0 : gslot + 1;
    }

    // ---- epilogue: lane owns channels cb..cb+15 of voxel m for every (channel tile a, voxel tile v) ----
    if (pend > 0) {   // K shorter than the ring: the previous tile's stores are still counted -- drain once
      __builtin_amdgcn_s_waitcnt(vm(0));
      pend = 0;
    }
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
#pragma unroll
    for (int a = 0; a < CT; ++a) {
      const int cb = n0 + wn * 32 * CT + a * 32 + 16 * hi;
      f32x4 res[2][2][2];   // [v][h8][half]
      if (d.residual != nullptr) {
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
      }
      if (d.scale != nullptr) {
        float sc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = cb + r < d.cout ? d.scale[cb + r] : 0.f;
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] *= sc[r];
      }
      if (d.shift != nullptr) {
        float sh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sh[r] = cb + r < d.cout ? d.shift[cb + r] : 0.f;
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] += sh[r];
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
      // one channel tile's residual rows and scale / shift tables live at a time (the scheduler would otherwise
      // hoist all CT tiles' loads to the top of the epilogue: 128+ registers on top of the accumulators)
      __builtin_amdgcn_sched_barrier(0);
    }
    // ... then nothing but stores (E_BF16 / E_F32 of them, whatever is masked)
#pragma unroll
    for (int a = 0; a < CT; ++a) {
      const int cb = n0 + wn * 32 * CT + a * 32 + 16 * hi;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const unsigned yo = (unsigned)(e_b[v] * d.y_bs + e_sp[v] * d.ldy + cb);
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          const bool ok = e_ok[v] && cb + h8 * 8 < cout_p8;
          const int r0 = h8 * 8;
          if (d.y_f32) {
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
    pend = NS8 - 2;
  }
  if (!half_b) __builtin_amdgcn_s_barrier();   // matches the second half's offset barrier
  __builtin_amdgcn_s_waitcnt(vm(0));   // the stream's trailing (zero-page) DMAs land before the LDS is released
}

template <bool PW, int CT, int BK, int NS>
int launch8(const pv_conv3d_desc& d, int tiles_n, long total, hipStream_t s) {
  constexpr int BN = 64 * CT;
  const size_t lds = (size_t)NS * (BN + BMV8) * BK * 2 + (PW ? 16 : (size_t)kMaxTaps8 * 4);
  auto kern = gemm8_kernel<PW, CT, BK, NS>;
  PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long resident = 256;   // one workgroup per CU
  dim3 grid((unsigned)(total < resident ? total : resident)), block(kThreads8);
  PV_LAUNCH(kern, grid, block, lds, s, d, tiles_n, (int)total, 1.0f / (float)d.cin);
  pv_note_kernel("gemm8_kernel");   // (launched through a function pointer: PV_LAUNCH saw only the variable)
  PV_LAUNCH_CHECK();
  return PV_OK;
}

}  // namespace

// Returns PV_OK when this kernel took the op, PV_ERR_UNSUPPORTED to leave it to the 128 x 128 kernel.
int pv_gemm8_try(const pv_conv3d_desc& d, bool pw, hipStream_t s) {
  const int mode = pv_tune("gemm8", 1);   // 0 off, 1 heuristic, 2 BN=128, 4 BN=256
  if (mode == 0) return PV_ERR_UNSUPPORTED;
  if (d.dtype != PV_BF16 || d.a_gate != nullptr || d.a_act != PV_ACT_NONE || d.x2 != nullptr) return PV_ERR_UNSUPPORTED;
  const int taps = d.kt * d.kh * d.kw;
  if (taps > kMaxTaps8 || d.kt > 255 || d.kh > 255 || d.kw > 255) return PV_ERR_UNSUPPORTED;
  const long M = (long)d.B * d.To * d.Ho * d.Wo;
  const long K = (long)taps * d.cin;
  const int cout_p8 = pv_round_up(d.cout, 8);
  // 31-bit element offsets in the staging geometry, 31-bit byte offsets in the store descriptor
  if (M > 0x7fffffffL || (long)d.B * d.x_bs > 0x7fffffffL || (long)d.cout * K > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  if ((long)d.B * d.y_bs * (d.y_f32 ? 4 : 2) > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  if (K < 3 * 64) return PV_ERR_UNSUPPORTED;   // at least three stages per tile
  const long tiles_m = pv_ceil_div(M, BMV8);
  int ct = 0;
  if (mode == 2 || mode == 4) {
    ct = mode;
  } else {
    // Measured (tools/bench_gemm.py, same box, 128 x 128 kernel -> this one): 32768 x 4096 x 4096 777 -> 986 TF/s
    // with 256-channel tiles (256 x 128 tiles: 802), M 100k x K 384 x N 1152 516 -> 564 -- but every layer of the
    // BASELINE workloads (M 6k-32k, N 256-3072: 100-600 tiles of 256 x 256) is 10-35 % SLOWER here: a second,
    // partial round of one-per-CU workgroups costs more than the halved L2 -> LDS traffic saves.  So the large
    // tiles take only what fills >= 4 rounds of them; everything else stays on the 128 x 128 kernel.
    const long t256 = tiles_m * pv_ceil_div(cout_p8, 256), t128 = tiles_m * pv_ceil_div(cout_p8, 128);
    const double waste256 = (double)(pv_ceil_div(cout_p8, 256) * 256 - cout_p8) / (double)cout_p8;
    const double waste128 = (double)(pv_ceil_div(cout_p8, 128) * 128 - cout_p8) / (double)cout_p8;
    if (K >= 1024 && waste256 <= 0.15 && t256 >= 1024) ct = 4;
    else if (K >= 256 && waste128 <= 0.15 && t128 >= 2048) ct = 2;   // (SlowFast's 256 -> 64 conv_a: 117 vs 100 us here)
    else return PV_ERR_UNSUPPORTED;
  }
  const int tiles_n = (int)pv_ceil_div(cout_p8, 64 * ct);
  const long total = tiles_m * tiles_n;
  if (total <= 0 || total > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  // 256-channel tiles: 64-byte rows (BK 32) in a 4-stage ring; 128-channel tiles: full 128-byte lines (BK 64), 3 stages
  if (ct == 4) return pw ? launch8<true, 4, 32, 4>(d, tiles_n, total, s) : launch8<false, 4, 32, 4>(d, tiles_n, total, s);
  if (pv_tune("gemm8_bk", 64) == 32)
    return pw ? launch8<true, 2, 32, 4>(d, tiles_n, total, s) : launch8<false, 2, 32, 4>(d, tiles_n, total, s);
  return pw ? launch8<true, 2, 64, 3>(d, tiles_n, total, s) : launch8<false, 2, 64, 3>(d, tiles_n, total, s);
}
