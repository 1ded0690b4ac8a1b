// MFMA-bound dense convolution / linear layer: NDHWC implicit GEMM with asynchronous
// global->LDS staging (bf16 in, fp32 accumulate).
//
//   out[voxel m][channel n] = act( (sum_k X'[m][k] W[n][k]) * scale[n] + shift[n] + residual[m][n] )
//   k = tap*cin + c ;  X'[m][k] = x[voxel m shifted by tap][c]  (zero outside the image)
//
// This is the kernel for SlowFast's T x1x1 / 1x3x3 / stem / lateral convolutions and for every
// nn.Linear of MViT (layers/attention.py:102-114,425-451,541), i.e. the part of the path whose
// arithmetic intensity is above the MI355X ridge.  Structure (one workgroup = 4 waves = one
// 128 (channels) x 128 (voxels) output tile, K walked in steps of 64):
//   * both operand tiles go global -> LDS with `global_load_lds_dwordx4` (no staging registers, no
//     ds_write pass); every lane computes its own SOURCE address, which is what makes the
//     implicit-GEMM gather (taps, strides, zero padding -> a 16-byte zero page) and the K / M / N
//     tails free, and what carries the LDS swizzle: the LDS image is lane-linear, so the XOR
//     swizzle chunk ^= (row>>1)&7 that makes `ds_read_b128` conflict-free is applied to the source
//     address and again on the read (both sides or neither);
//   * two LDS buffers, one barrier per K step: the loads of step t+1 are in flight while step t
//     is multiplied; two workgroups per CU cover each other's barrier drain;
//   * v_mfma_f32_32x32x16_bf16 with the weights as the A operand (rows = output channels) and the
//     voxels as the B operand; the LDS rows of the weight tile are permuted so that a lane's 16
//     accumulator registers are 16 CONSECUTIVE channels of one voxel: the epilogue (folded BN /
//     bias, fp32-or-bf16 residual, activation) stays in registers and leaves as 16-byte stores;
//   * tiles are numbered so that consecutive workgroups (same XCD -> same L2) share the
//     activation rows and walk the weight panels.
#include <stdlib.h>
#include "pv_common.h"

__device__ __attribute__((aligned(16))) unsigned int pv_zero_page[4] = {0u, 0u, 0u, 0u};

namespace {

constexpr int kThreads = 256;
constexpr int BM = 128;   // voxels per tile
constexpr int BN = 128;   // output channels per tile
constexpr int kMaxTaps = 512;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS row rho of the weight tile holds output channel chi(rho): within each 32-row MFMA tile,
// accumulator register r of lane-half hi is MFMA row (r&3) + 8*(r>>2) + 4*hi -> channel 16*hi + r
__device__ __forceinline__ int chi(int rho) {
  return (rho & ~31) + 16 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 3) + (rho & 3);
}

// BK = 64: two workgroups per CU, fewest barriers per FLOP (long K).  BK = 32: half the LDS, three
// workgroups per CU cover each other's prologue / epilogue when K is only a few steps deep.
template <bool PW, int BK>
__global__ __launch_bounds__(kThreads, 2) void gemm_glds_kernel(const pv_conv3d_desc d, int tiles_n, int total_tiles,
                                                                float inv_cin) {
  constexpr int TILE_ELEMS = 128 * BK;                          // one operand tile (elements)
  constexpr int NJ = TILE_ELEMS * 2 / (kThreads * 16);          // 16-byte items per thread per operand tile
  constexpr int CPR = BK / 8;                                   // 16-byte chunks per tile row
  // XOR swizzle of the chunk index that makes ds_read_b128 conflict-free (rows are CPR*16 bytes)
  auto swz = [](int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  constexpr int NBUF = BK == 32 ? 4 : 2;                        // LDS ring depth
  constexpr int PD = NBUF - 1;                                  // K steps in flight ahead of the multiply
  __shared__ __attribute__((aligned(16))) bf16_t smem[NBUF * 2 * TILE_ELEMS];   // [buf][W | X][128][BK]
  __shared__ int s_tap[PW ? 1 : kMaxTaps];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;

  // XCD-aware tile order (bijective for any tile count)
  int tile;
  {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int qn = total_tiles >> 3, rn = total_tiles & 7;
    tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  }
  const int tile_n = tile % tiles_n;
  const long tile_m = tile / tiles_n;
  const long m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const long S_out = (long)d.To * d.Ho * d.Wo;
  const long M = (long)d.B * S_out;
  const int taps = d.kt * d.kh * d.kw;
  const int K = taps * d.cin;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const bf16_t* __restrict__ X = static_cast<const bf16_t*>(d.x);
  const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(pv_zero_page);

  if constexpr (!PW) {
    for (int t = tid; t < taps; t += kThreads) {
      const int dt = t / (d.kh * d.kw);
      const int r = t - dt * d.kh * d.kw;
      const int dh = r / d.kw;
      s_tap[t] = dt | (dh << 8) | ((r - dh * d.kw) << 16);
    }
    __syncthreads();
  }

  // ---- per-thread staging geometry: item g = j*256 + tid -> (row g>>3, LDS position g&7) ----
  long w_off[NJ];          // element offset of the weight row, or -1
  long x_off[NJ];          // PW: element offset of the voxel row, or -1 ; general: clip offset or -1
  int x_t[NJ], x_h[NJ], x_w[NJ];
  int kch[NJ];             // logical 8-channel chunk inside a K step carried by this item
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int g = j * kThreads + tid;
    const int row = g / CPR, pos = g % CPR;
    kch[j] = pos ^ swz(row);
    const int n = n0 + chi(row);
    w_off[j] = n < d.cout ? (long)n * K : -1;
    const long m = m0 + row;
    if (m < M) {
      const long b = m / S_out;
      const long sp = m - b * S_out;
      if constexpr (PW) {
        x_off[j] = b * d.x_bs + sp * d.ldx;
        x_t[j] = x_h[j] = x_w[j] = 0;
      } else {
        const int to = (int)(sp / (d.Ho * d.Wo));
        const int r2 = (int)(sp - (long)to * d.Ho * d.Wo);
        const int ho = r2 / d.Wo;
        x_off[j] = b * d.x_bs;
        x_t[j] = to * d.st - d.pt;
        x_h[j] = ho * d.sh - d.ph;
        x_w[j] = (r2 - ho * d.Wo) * d.sw - d.pw;
      }
    } else {
      x_off[j] = -1;
      x_t[j] = x_h[j] = x_w[j] = 0;
    }
  }

  // Source selection is done with bit masks, not `?:` -- the compiler would turn a select between two
  // pointers into two exec-masked LDS-DMA instructions, and the K loop below COUNTS the DMA
  // instructions a wave has in flight (exactly GL per step).
  const unsigned long zaddr = (unsigned long)zero;
  auto pick = [&](bool ok, const bf16_t* p) -> const bf16_t* {
    const unsigned long m = 0ul - (unsigned long)ok;
    return reinterpret_cast<const bf16_t*>(((unsigned long)p & m) | (zaddr & ~m));
  };
  auto stage = [&](int buf, int ks) {
    bf16_t* wb = smem + buf * 2 * TILE_ELEMS;
    bf16_t* xb = wb + TILE_ELEMS;
    const int k0 = ks * BK;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = k0 + kch[j] * 8;
      const bool kok = k < K;
      // weights
      __builtin_amdgcn_global_load_lds((gptr_t)pick(kok && w_off[j] >= 0, Wt + (w_off[j] >= 0 ? w_off[j] : 0) + k),
                                       (lptr_t)(wb + (j * kThreads + wave * 64) * 8), 16, 0, 0);
      // activations
      bool xok = kok && x_off[j] >= 0;
      long xo = x_off[j] >= 0 ? x_off[j] : 0;
      if constexpr (PW) {
        xo += k;
      } else {
        const int tap = (int)(((float)k + 0.5f) * inv_cin);
        const int tp = s_tap[tap < taps ? tap : 0];
        const int ti = x_t[j] + (tp & 255), hh = x_h[j] + ((tp >> 8) & 255), ww = x_w[j] + (tp >> 16);
        xok = xok && (unsigned)ti < (unsigned)d.Ti && (unsigned)hh < (unsigned)d.Hi && (unsigned)ww < (unsigned)d.Wi;
        xo += ((long)(ti * d.Hi + hh) * d.Wi + ww) * d.ldx + (k - tap * d.cin);
      }
      __builtin_amdgcn_global_load_lds((gptr_t)pick(xok, X + xo), (lptr_t)(xb + (j * kThreads + wave * 64) * 8), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][v][r] = 0.f;

  // read-side swizzle of this lane's fragment rows (fixed for the whole kernel)
  int a_row[2], b_row[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    a_row[t] = wn * 64 + t * 32 + l31;
    b_row[t] = wm * 64 + t * 32 + l31;
  }

  const int nk = (K + BK - 1) / BK;
  // K loop: an LDS ring of NBUF step-buffers with PD = NBUF-1 steps of LDS-DMA in flight.  Each wave
  // waits only for ITS OWN loads of the step about to be multiplied (counted vmcnt: the younger
  // steps stay in flight across the barrier), then one raw barrier publishes the buffer to the
  // workgroup and, at the same time, proves that everybody is done reading the buffer that is
  // refilled next.  (A plain __syncthreads() would drain the whole DMA queue every step.)
  constexpr int GL = 2 * NJ;   // LDS-DMA instructions per thread per K step
#pragma unroll
  for (int s0 = 0; s0 < PD; ++s0)
    if (s0 < nk) stage(s0, s0);
  for (int ks = 0; ks < nk; ++ks) {
    const int rem = nk - 1 - ks;   // steps after this one (wave-uniform)
    // s_waitcnt simm16 on gfx9: vmcnt = bits [3:0] | [15:14], expcnt [6:4] and lgkmcnt [11:8] left at "no wait"
    constexpr int kWaitNone = (7 << 4) | (15 << 8);
    constexpr auto vm = [](int n) { return (n & 15) | ((n >> 4) << 14) | kWaitNone; };
    if (PD >= 3 && rem >= 2) __builtin_amdgcn_s_waitcnt(vm(2 * GL));
    else if (PD >= 2 && rem >= 1) __builtin_amdgcn_s_waitcnt(vm(GL));
    else __builtin_amdgcn_s_waitcnt(vm(0));
    __builtin_amdgcn_s_barrier();
    if (ks + PD < nk) stage((ks + PD) % NBUF, ks + PD);
    const bf16_t* wb = smem + (ks % NBUF) * 2 * TILE_ELEMS;
    const bf16_t* xb = wb + TILE_ELEMS;
    // fragment reads are software-pipelined one 16-deep sub-step ahead of the MFMAs that use them
    bf16x8 af[2][2], bfr[2][2];
    auto read_frags = [&](int slot, int s) {
      const int c = 2 * s + hi;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[slot][t] = *reinterpret_cast<const bf16x8*>(wb + a_row[t] * BK + ((c ^ swz(a_row[t])) << 3));
        bfr[slot][t] = *reinterpret_cast<const bf16x8*>(xb + b_row[t] * BK + ((c ^ swz(b_row[t])) << 3));
      }
    };
    read_frags(0, 0);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      if (s + 1 < BK / 16) read_frags((s + 1) & 1, s + 1);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int v = 0; v < 2; ++v)
          acc[a][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][a], bfr[s & 1][v], acc[a][v], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // ---- epilogue: lane owns channels cb..cb+15 of voxel m for every (channel tile a, voxel tile v) ----
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int cb = n0 + wn * 64 + a * 32 + 16 * hi;
    if (cb >= cout_p8) continue;
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = cb + r < d.cout;
      sc[r] = ok ? (d.scale ? d.scale[cb + r] : 1.f) : 0.f;
      sh[r] = ok ? (d.shift ? d.shift[cb + r] : 0.f) : 0.f;
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const long m = m0 + wm * 64 + v * 32 + l31;
      if (m >= M) continue;
      const long b = m / S_out;
      const long sp = m - b * S_out;
      float o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = acc[a][v][r] * sc[r] + sh[r];
      if (d.residual != nullptr) {
        const long ro = b * d.r_bs + sp * d.ldr + cb;
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          if (cb + h8 * 8 >= cout_p8) continue;
          float rf[8];
          if (d.r_f32) {
            Chunk8<float> rc;
            rc.load(static_cast<const float*>(d.residual) + ro + h8 * 8);
            rc.to_f32(rf);
          } else {
            Chunk8<bf16_t> rc;
            rc.load(static_cast<const bf16_t*>(d.residual) + ro + h8 * 8);
            rc.to_f32(rf);
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) o[h8 * 8 + r] += rf[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[r] = pv_apply_act(o[r], d.act);
        if (cb + r >= d.cout) o[r] = 0.f;
      }
      const long yo = b * d.y_bs + sp * d.ldy + cb;
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        if (cb + h8 * 8 >= cout_p8) continue;
        float o8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) o8[r] = o[h8 * 8 + r];
        if (d.y_f32) {
          Chunk8<float> oc;
          oc.from_f32(o8);
          oc.store(static_cast<float*>(d.y) + yo + h8 * 8);
        } else {
          Chunk8<bf16_t> oc;
          oc.from_f32(o8);
          oc.store(static_cast<bf16_t*>(d.y) + yo + h8 * 8);
        }
      }
    }
  }
}

}  // namespace

// Returns PV_OK when this kernel took the op, PV_ERR_UNSUPPORTED to fall back to the generic one.
int pv_gemm_glds_try(const pv_conv3d_desc& d, bool pw, hipStream_t s) {
  if (d.dtype != PV_BF16 || d.a_gate != nullptr || d.a_act != PV_ACT_NONE) return PV_ERR_UNSUPPORTED;
  const int taps = d.kt * d.kh * d.kw;
  if (taps > kMaxTaps || d.kt > 255 || d.kh > 255 || d.kw > 255) return PV_ERR_UNSUPPORTED;
  const long M = (long)d.B * d.To * d.Ho * d.Wo;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const long tiles_m = pv_ceil_div(M, BM);
  const int tiles_n = (int)pv_ceil_div(cout_p8, BN);
  const long total = tiles_m * tiles_n;
  if (total <= 0 || total > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  const float inv_cin = 1.0f / (float)d.cin;
  static const int bk_env = getenv("PV_GEMM_BK") ? atoi(getenv("PV_GEMM_BK")) : 0;
  const int K = taps * d.cin;
  (void)K;
  const bool bk32 = bk_env == 32;   // measured: the 64-deep step wins at every K once fragment reads are pipelined
  dim3 grid((unsigned)total), block(kThreads);
  if (pw && bk32) hipLaunchKernelGGL((gemm_glds_kernel<true, 32>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin);
  else if (pw) hipLaunchKernelGGL((gemm_glds_kernel<true, 64>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin);
  else if (bk32) hipLaunchKernelGGL((gemm_glds_kernel<false, 32>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin);
  else hipLaunchKernelGGL((gemm_glds_kernel<false, 64>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin);
  PV_LAUNCH_CHECK();
  return PV_OK;
}
