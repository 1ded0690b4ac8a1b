// Pointwise conv / Linear on a HANDFUL of rows (at most 4 voxels per clip): the classification heads --
// X3D's post_conv 432 -> 2048 and proj 2048 -> 400 on one pooled voxel per clip (models/x3d.py:480-494, models/head.py:376-382),
// SlowFast's proj 2304 -> 400 (models/slowfast.py:345-361), MViT's head Linear on the cls rows (models/head.py:538-559).
//
// On the tiled GEMM such a layer is one or two row tiles and a long serial K loop on a few workgroups (X3D-M's proj: 4 tiles,
// 32 K-steps, 29 us).  Here the reduction is what is parallel:
//   * a workgroup owns 32 output channels, its 8 waves split K (wave w takes the 16-channel K-steps w, w+8, ...);
//   * both MFMA operands come STRAIGHT from global memory in operand layout (v_mfma_f32_32x32x16_bf16: A = 32 filter rows x 16 k,
//     B = 32 activation rows x 16 k; a lane's 8 k-values are 16 contiguous bytes of a filter / activation row), through buffer
//     resources whose range check supplies the zeros of ragged rows, channels and K tails: no LDS staging, no bounds branches;
//   * 8 K-steps of loads are issued before their MFMAs (the kernel is one memory round trip deep, not 32);
//   * the eight partial accumulators are joined through LDS by wave 0, which applies folded BN / bias, the activation and stores.
#include "pv_common.h"

namespace {

constexpr int kHeadThreads = 512;
constexpr int kHeadWaves = kHeadThreads / 64;
constexpr int kHeadU = 8;   // K-steps in flight per wave

template <int NB>   // 32-row tiles of activation rows (1: <= 32 rows, 2: <= 64)
__global__ __launch_bounds__(kHeadThreads) void head_rows_kernel(const pv_conv3d_desc d, int M, int S_out) {
  __shared__ float s_red[kHeadWaves - 1][NB][16][64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int co0 = blockIdx.x * 32;
  const int K = d.cin;                          // filter row pitch = cin (multiple of 8), one tap
  const int nks = (K + 15) >> 4;
  constexpr unsigned kOOB = 0x80000000u;

  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.w), 0, (int)((unsigned)d.cout * (unsigned)K * 2u), 0x00020000);
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.x), 0, 0x7ffffff0, 0x00020000);
  // per-lane row bases (bytes); rows / channels that do not exist point out of range (read as zero)
  const int co = co0 + l31;
  const unsigned w_base = co < d.cout ? (unsigned)co * (unsigned)K * 2u : kOOB;
  unsigned x_base[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int m = blockIdx.y * 64 + nb * 32 + l31;
    const int b = m / S_out, sp = m - b * S_out;
    x_base[nb] = m < M ? (unsigned)(((long)b * d.x_bs + (long)sp * d.ldx) * 2) : kOOB;
  }

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  for (int ks0 = wave; ks0 < nks; ks0 += kHeadWaves * kHeadU) {
    u32x4 af[kHeadU], bf[kHeadU][NB];
#pragma unroll
    for (int u = 0; u < kHeadU; ++u) {
      const int k = (ks0 + kHeadWaves * u) * 16 + hi * 8;
      const bool k_ok = k < K;                    // (covers K-steps past the end and the upper half of a K % 16 == 8 tail)
      const unsigned ko = (unsigned)k * 2u;
      af[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)((k_ok && w_base != kOOB) ? w_base + ko : kOOB), 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        bf[u][nb] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((k_ok && x_base[nb] != kOOB) ? x_base[nb] + ko : kOOB), 0, 0);
    }
#pragma unroll
    for (int u = 0; u < kHeadU; ++u)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[u]), __builtin_bit_cast(bf16x8, bf[u][nb]),
                                                           acc[nb], 0, 0, 0);
  }

  // join the K slices: lane-linear through LDS (bank-conflict free), wave 0 finishes
  if (wave > 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_red[wave - 1][nb][r][lane] = acc[nb][r];
  }
  __syncthreads();
  if (wave != 0) return;
  const int cout_p8 = pv_round_up(d.cout, 8);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int m = blockIdx.y * 64 + nb * 32 + l31;
    if (m >= M) continue;
    const int b = m / S_out, sp = m - b * S_out;
    const long yo = (long)b * d.y_bs + (long)sp * d.ldy;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = co0 + 8 * (r >> 2) + 4 * hi + (r & 3);      // C/D layout of the 32x32 MFMA: row = 8 (r/4) + 4 (lane/32) + r%4
      if (c >= cout_p8) continue;
      float v = acc[nb][r];
#pragma unroll
      for (int w = 0; w < kHeadWaves - 1; ++w) v += s_red[w][nb][r][lane];
      if (c < d.cout) {
        v = v * (d.scale ? d.scale[c] : 1.f) + (d.shift ? d.shift[c] : 0.f);
        v = pv_apply_act(v, d.act);
      } else {
        v = 0.f;                                                 // padding channels are exactly 0
      }
      if (d.y_f32) static_cast<float*>(d.y)[yo + c] = v;
      else static_cast<bf16_t*>(d.y)[yo + c] = (bf16_t)v;
    }
  }
}

}  // namespace

// Returns PV_OK when this kernel took the op, PV_ERR_UNSUPPORTED to leave it to the tiled kernels.
int pv_head_rows_try(const pv_conv3d_desc& d, hipStream_t s) {
  if (d.dtype != PV_BF16 || d.a_gate != nullptr || d.a_act != PV_ACT_NONE || d.residual != nullptr || d.x2 != nullptr)
    return PV_ERR_UNSUPPORTED;
  // Routed by what ONE clip looks like (<= 4 output voxels, a long reduction), never by the batch: a batch split into sub-batches
  // (SplitBatchDeployed) must run the same kernels, in the same summation order, as the one-plan form -- row for row identical.
  const long M = (long)d.B * d.To * d.Ho * d.Wo;
  if ((long)d.To * d.Ho * d.Wo > 4 || M > 0x7fffffffL / 64 || d.cin < pv_tune("head_rows_min_cin", 512)) return PV_ERR_UNSUPPORTED;
  if ((long)d.cout * d.cin * 2 > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  if (((long)(d.B - 1) * d.x_bs + (long)d.To * d.Ho * d.Wo * d.ldx) * 2 > 0x7fffffe0L) return PV_ERR_UNSUPPORTED;   // 31-bit byte offsets
  const int S_out = d.To * d.Ho * d.Wo;
  dim3 grid((unsigned)pv_ceil_div(pv_round_up(d.cout, 8), 32), (unsigned)pv_ceil_div(M, 64)), block(kHeadThreads);   // 64-row tiles in y
  if (M <= 32) PV_LAUNCH((head_rows_kernel<1>), grid, block, 0, s, d, (int)M, S_out);
  else PV_LAUNCH((head_rows_kernel<2>), grid, block, 0, s, d, (int)M, S_out);
  PV_LAUNCH_CHECK();
  return PV_OK;
}
