// Streaming pointwise (1x1x1, stride 1) convolution for the HBM-bound widths of X3D
// (cin, cout <= a few hundred): out[voxel][n] = act( sum_k x'[voxel][k] w[n][k] * scale + shift + r ).
//
// These layers are "the same small GEMM at shrinking M": K and N are far too small to
// amortise an LDS-staged, barrier-synchronised tile pipeline, and the arithmetic intensity
// (17..133 FLOP/B) leaves the matrix cores idle -- the kernel has to behave like a streaming
// copy with an MFMA in the middle:
//   * the whole weight slab of the block (<= 128 output channels x K) is staged into LDS once,
//     then the block walks voxel groups in a grid-stride loop with NO further barriers;
//   * activations go global -> registers directly in MFMA B-operand shape: lane (n = lane&15,
//     q = lane>>4) loads the 16 bytes x[voxel n][k0 + 8q .. 8q+7]; with channels-last rows that
//     are back to back a wave's load covers one contiguous run of 16 voxels;
//   * weights are the A operand with LDS rows permuted (pairs of 16-channel tiles interleaved
//     in groups of 4) so a lane ends up with 8 contiguous output channels of its voxel: the
//     epilogue (folded BN, residual, activation) stays in registers, one 16-byte store per lane;
//   * latency is hidden by occupancy (4-8 waves per SIMD, every wave with several KB in flight)
//     plus a one-chunk software prefetch along K;
//   * squeeze-excitation: x' = swish(x * gate[b][k]) is applied to the operand registers; the
//     gate rows of the (at most two) clips a wave's voxels belong to are cached in a wave-private
//     LDS region, refreshed only when the wave crosses a clip boundary.
#include "pv_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kKC = 1;  // k-steps (of 32 channels) per register chunk

struct PwRows {  // per-lane geometry of one wave tile group: clip index and voxel within the clip (-1: none)
  int b[4];
  int sp[4];
  int sp2[4];   // voxel of the second operand (strided sampling of x2), X2V kernels only
};

// X2V: compiled with the second K operand (projection shortcut folded into conv_c); a separate variant so that
// the common kernels do not carry its registers
template <int NT, int TM, bool XFORM, int KS, bool F32, bool X2V>
__global__ __launch_bounds__(kThreads, (KS >= 14 || X2V) ? 2 : ((NT >= 4 || KS >= 4) ? 3 : 4)) void pw_stream_kernel(const pv_conv3d_desc d, int ksteps_rt, int ngroups,
                                                                int nchunks, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int ksteps = KS > 0 ? KS : ksteps_rt;
  const int Kp = ksteps * 32;
  const int WLD = Kp + 8;  // weight row stride (elements): 16 B x odd -> conflict-free ds_read_b128
  bf16_t* w_s = reinterpret_cast<bf16_t*>(smem_raw);
  float* sc_s = reinterpret_cast<float*>(smem_raw + (size_t)NT * 16 * WLD * 2);
  float* sh_s = sc_s + NT * 16;
  float* sc2_s = sh_s + NT * 16;                    // X2V only: scale of the second operand's product
  float* gate_s = sh_s + NT * 16 * (X2V ? 2 : 1);   // [4 waves][2][cin]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n16 = lane & 15;
  const int q = lane >> 4;
  const int cout_p8 = pv_round_up(d.cout, 8);
  // 1-D grid: the N-splits of one voxel chunk are consecutive workgroups of the same XCD, so the
  // activation rows are fetched from HBM once and re-read from that XCD's L2 by the other splits
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chunk = (slot / nsplit) * 8 + xcd;
  const int n0 = (slot % nsplit) * (NT * 16);
  const long S_out = (long)d.To * d.Ho * d.Wo;
  const long M = (long)d.B * S_out;

  // ---- stage weights (row r of LDS = output channel n0 + perm(r)), scale, shift ----
  {
    const bf16_t* __restrict__ Wt = static_cast<const bf16_t*>(d.w);
    const int cpr = Kp / 8;  // chunks per row
    const int w_pitch = X2V ? Kp : d.cin;   // two operands: rows are packed at the padded K
    for (int id = tid; id < NT * 16 * cpr; id += kThreads) {
      const int r = id / cpr, kc = id - r * cpr;
      const int tn = r >> 4, ii = r & 15;
      const int c = n0 + (tn >> 1) * 32 + (ii >> 2) * 8 + (tn & 1) * 4 + (ii & 3);
      bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < d.cout && kc * 8 < w_pitch) v = *reinterpret_cast<const bf16x8*>(Wt + (long)c * w_pitch + kc * 8);
      *reinterpret_cast<bf16x8*>(w_s + r * WLD + kc * 8) = v;
    }
    for (int i = tid; i < NT * 16; i += kThreads) {
      const int c = n0 + i;
      const bool ok = c < d.cout;
      sc_s[i] = ok ? (d.scale ? d.scale[c] : 1.f) : 0.f;
      sh_s[i] = ok ? (d.shift ? d.shift[c] : 0.f) : 0.f;
      if constexpr (X2V) sc2_s[i] = ok ? (d.x2_scale ? d.x2_scale[c] : 1.f) : 0.f;
    }
  }
  __syncthreads();

  const bf16_t* __restrict__ X = static_cast<const bf16_t*>(d.x);
  const bf16_t* __restrict__ X2 = X2V ? static_cast<const bf16_t*>(d.x2) : nullptr;
  const int ks1 = X2V ? (d.cin + 31) / 32 : 1 << 20;   // k-steps [ks1, ..) read the second operand
  const bool has_gate = XFORM && d.a_gate != nullptr;
  const bool has_res = d.residual != nullptr;
  float* my_gate = gate_s + wave * 2 * d.cin;
  long gate_b0 = -1;  // first clip whose gate rows are cached by this wave
  const int live_pairs = min(NT / 2, (cout_p8 - n0 + 31) / 32);  // wave-uniform
  constexpr int NP = NT / 2;
  constexpr int KSR = KS > 0 ? KS : 1;

  auto rows_of = [&](int g, PwRows& r) {
    const long m_base = ((long)g * 4 + wave) * (TM * 16);
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const long m = m_base + t * 16 + n16;
      const bool ok = g < ngroups && m < M;
      const long mm = ok ? m : 0;
      const long b = mm / S_out;
      r.b[t] = (int)b;
      r.sp[t] = ok ? (int)(mm - b * S_out) : -1;
      if constexpr (X2V) {
        const int sp = ok ? r.sp[t] : 0;
        const int to = sp / (d.Ho * d.Wo), r2 = sp - to * d.Ho * d.Wo;
        const int ho = r2 / d.Wo, wo = r2 - ho * d.Wo;
        r.sp2[t] = ((to * d.x2_st) * d.x2_Hi + ho * d.x2_sh) * d.x2_Wi + wo * d.x2_sw;
      }
    }
  };
  auto load_x = [&](bf16x8 (&dst)[KSR][TM], const PwRows& r, int ks0) {
#pragma unroll
    for (int kk = 0; kk < KSR; ++kk) {
      if (X2V && ks0 + kk >= ks1) {   // second operand (wave-uniform): channel k2 of the strided x2 voxel
        const int k2 = (ks0 + kk - ks1) * 32 + q * 8;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          if (r.sp[t] >= 0 && k2 < d.x2_cin)
            dst[kk][t] = *reinterpret_cast<const bf16x8*>(X2 + (long)r.b[t] * d.x2_bs + (long)r.sp2[t] * d.x2_ld + k2);
          else dst[kk][t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        continue;
      }
      const int k0 = (ks0 + kk) * 32 + q * 8;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (r.sp[t] >= 0 && k0 < d.cin)
          dst[kk][t] = *reinterpret_cast<const bf16x8*>(X + (long)r.b[t] * d.x_bs + (long)r.sp[t] * d.ldx + k0);
        else dst[kk][t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  };
  // residual chunks in the epilogue's shape: [pair][tile] -> 8 bf16 channels (raw 16 bytes)
  auto load_res = [&](f32x4 (&dst)[NP][TM][F32 ? 2 : 1], const PwRows& r) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c0 = n0 + p * 32 + q * 8;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const bool ok = has_res && r.sp[t] >= 0 && p < live_pairs && c0 < cout_p8;
        const long ro = (long)r.b[t] * d.r_bs + (long)r.sp[t] * d.ldr + c0;
        if constexpr (F32) {
          const float* rp = static_cast<const float*>(d.residual) + ro;
          dst[p][t][0] = ok ? *reinterpret_cast<const f32x4*>(rp) : f32x4{0.f, 0.f, 0.f, 0.f};
          dst[p][t][1] = ok ? *reinterpret_cast<const f32x4*>(rp + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
          dst[p][t][0] = ok ? *reinterpret_cast<const f32x4*>(static_cast<const bf16_t*>(d.residual) + ro)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  auto xform = [&](bf16x8 (&src)[KSR][TM], const PwRows& r, int ks0) {
#pragma unroll
    for (int kk = 0; kk < KSR; ++kk) {
      if (X2V && ks0 + kk >= ks1) continue;   // gate and activation belong to the first operand only
      const int k0 = (ks0 + kk) * 32 + q * 8;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)src[kk][t][j];
        if (has_gate && k0 < d.cin) {
          const float* gp = my_gate + (r.b[t] - (int)gate_b0) * d.cin + k0;
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { f[j] *= g0[j]; f[4 + j] *= g1[j]; }
        }
        if (d.a_act == PV_ACT_SWISH) {
          // X3D's conv_c: swish(x g) on the packed fp32 pipe, two channels per instruction (the exponential and the reciprocal
          // stay per element); same operations as pv_sigmoid, so the same bits as the generic path below
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const f32x2 sv = {f[j], f[j + 1]};
            const f32x2 ea = sv * -1.44269504088896340736f;
            f32x2 den;
            den[0] = __builtin_amdgcn_exp2f(ea[0]);
            den[1] = __builtin_amdgcn_exp2f(ea[1]);
            den = den + 1.0f;
            f32x2 rc;
            rc[0] = __builtin_amdgcn_rcpf(den[0]);
            rc[1] = __builtin_amdgcn_rcpf(den[1]);
            const f32x2 o = sv * rc;
            f[j] = o[0];
            f[j + 1] = o[1];
          }
        } else {
          pv_apply_act_n<true>(f, d.a_act & 15);      // (bit 4: the A/B knob "pw_pk_swish" = 0 sends Swish down the generic path)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) src[kk][t][j] = (bf16_t)f[j];
      }
    }
  };
  // The filter fragments are loop-invariant across voxel groups; hoisting all NT x KS of them out of the
  // group loop (what LICM would do) costs more registers than the kernel has, so the LDS offset is made
  // opaque once per group and the reads stay where they are used.
  int w_opaque = 0;
  f32x4 acc2[X2V ? NT : 1][X2V ? TM : 1];   // X2V: the second operand's product, joined in the epilogue
  auto mma = [&](f32x4 (&acc)[NT][TM], const bf16x8 (&src)[KSR][TM], int ks0) {
#pragma unroll
    for (int kk = 0; kk < KSR; ++kk) {
      if constexpr (X2V) {
        if (ks0 + kk >= ks1) {   // wave-uniform
#pragma unroll
          for (int a = 0; a < NT; ++a) {
            if ((a >> 1) < live_pairs) {
              const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w_s + (a * 16 + n16) * WLD + (ks0 + kk) * 32 + q * 8);
#pragma unroll
              for (int t = 0; t < TM; ++t)
                acc2[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, src[kk][t], acc2[a][t], 0, 0, 0);
            }
          }
          continue;
        }
      }
      // long reductions: keep step kk's filter reads behind step kk-1's (the scheduler otherwise clusters
      // all NT x KS LDS reads at the top of the group and spills hundreds of registers)
      if (NT * KSR > 8) asm volatile("" : "+v"(w_opaque));
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        if ((a >> 1) < live_pairs) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w_s + w_opaque + (a * 16 + n16) * WLD + (ks0 + kk) * 32 + q * 8);
#pragma unroll
          for (int t = 0; t < TM; ++t)
            acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, src[kk][t], acc[a][t], 0, 0, 0);
        }
      }
    }
  };
  auto refresh_gate = [&](int g) {
    if (!has_gate || g >= ngroups) return;
    const long b_first = (((long)g * 4 + wave) * (TM * 16)) / S_out;  // wave-uniform
    if (b_first < d.B && b_first != gate_b0) {
      for (int i = lane; i < 2 * d.cin; i += 64) {
        const long bb = b_first + (i >= d.cin ? 1 : 0);
        my_gate[i] = bb < d.B ? d.a_gate[bb * d.cin + (i >= d.cin ? i - d.cin : i)] : 0.f;
      }
      gate_b0 = b_first;
    }
  };

  PwRows cur, nxt;
  bf16x8 xf[KSR][TM];
  f32x4 rcur[NP][TM][F32 ? 2 : 1];
  int g = chunk < nchunks ? chunk : ngroups;   // padded blocks (grid is a multiple of 8 chunks) do nothing
  rows_of(g, cur);
  load_x(xf, cur, 0);
  for (; g < ngroups; g += nchunks) {
    load_res(rcur, cur);  // consumed by the epilogue: in flight during the transform and the MFMAs
    f32x4 acc[NT][TM];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int t = 0; t < TM; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (X2V) {
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int t = 0; t < TM; ++t) acc2[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    refresh_gate(g);
    rows_of(g + nchunks, nxt);
    asm volatile("" : "+v"(w_opaque));
    if constexpr (KS > 0) {
      if (XFORM) xform(xf, cur, 0);
      mma(acc, xf, 0);
      load_x(xf, nxt, 0);          // the operand registers are free again: prefetch the next group
    } else {
      for (int ks = 0; ks < ksteps; ++ks) {
        if (XFORM) xform(xf, cur, ks);
        bf16x8 xn[KSR][TM];
        if (ks + 1 < ksteps) load_x(xn, cur, ks + 1);
        mma(acc, xf, ks);
        if (ks + 1 < ksteps) {
#pragma unroll
          for (int t = 0; t < TM; ++t) xf[0][t] = xn[0][t];
        }
      }
      load_x(xf, nxt, 0);
    }

    // ---- epilogue ----
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p >= live_pairs) break;
      const int cl = p * 32 + q * 8;   // channel within the block's slab
      const int c0 = n0 + cl;
      if (c0 >= cout_p8) continue;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc_s + cl), s1 = *reinterpret_cast<const f32x4*>(sc_s + cl + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh_s + cl), h1 = *reinterpret_cast<const f32x4*>(sh_s + cl + 4);
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (cur.sp[t] < 0) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[2 * p][t][j] * s0[j] + h0[j];
          v[4 + j] = acc[2 * p + 1][t][j] * s1[j] + h1[j];
        }
        if constexpr (X2V) {
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(sc2_s + cl), t1 = *reinterpret_cast<const f32x4*>(sc2_s + cl + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] += acc2[2 * p][t][j] * t0[j];
            v[4 + j] += acc2[2 * p + 1][t][j] * t1[j];
          }
        }
        if (has_res) {
          if constexpr (F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += rcur[p][t][0][j]; v[4 + j] += rcur[p][t][1][j]; }
          } else {
            const bf16x8 rb = __builtin_bit_cast(bf16x8, rcur[p][t][0]);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rb[j];
          }
        }
        pv_apply_act_n<true>(v, d.act);
        if (c0 + 8 > d.cout) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (c0 + j >= d.cout) v[j] = 0.f;
        }
        const long yo = (long)cur.b[t] * d.y_bs + (long)cur.sp[t] * d.ldy + c0;
        if constexpr (F32) {
          Chunk8<float> oc;
          oc.from_f32(v);
          oc.store(static_cast<float*>(d.y) + yo);
        } else {
          Chunk8<bf16_t> oc;
          oc.from_f32(v);
          oc.store(static_cast<bf16_t*>(d.y) + yo);
        }
      }
    }
    cur = nxt;
  }
}

// Workgroups of `kern` one CU holds at a time, at most 4: what the RUNTIME says for this variant's registers and this launch's
// LDS.  (Until round 6 the grid was sized from the LDS footprint alone, 4 per CU where it allowed; the variants that compile
// to 146-162 VGPRs -- the two-operand ones, <4, 2, true, 4> -- hold 3, so a quarter of the grid ran as a second, mostly empty round.)
inline long resident_per_cu(const void* kern, size_t lds) {
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kThreads, lds) != hipSuccess || occ < 1) {
    (void)hipGetLastError();
    const long by_lds = lds > 0 ? (160 * 1024) / (long)(lds + 1024) : 4;
    return by_lds < 1 ? 1 : (by_lds > 4 ? 4 : by_lds);
  }
  return occ > 4 ? 4 : occ;
}

template <int NT, int TM, bool XFORM, int KS, bool F32>
int launch_pw_f(const pv_conv3d_desc& d, int ksteps, size_t lds, hipStream_t s) {
  const long M = (long)d.B * d.To * d.Ho * d.Wo;
  const long ngroups = pv_ceil_div(M, 4 * TM * 16);
  const int nsplit = (int)pv_ceil_div(pv_round_up(d.cout, 8), NT * 16);
  if (ngroups > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  if (d.x2 != nullptr) {   // second K operand: the variants the residual blocks of X3D / ResNets need
    if constexpr (!F32 && (KS == 0 || KS == 3)) {
      const long M2 = (long)d.B * d.To * d.Ho * d.Wo;
      const long ngroups2 = pv_ceil_div(M2, 4 * TM * 16);
      const int nsplit2 = (int)pv_ceil_div(pv_round_up(d.cout, 8), NT * 16);
      if (ngroups2 > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
      auto kern2 = pw_stream_kernel<NT, TM, XFORM, KS, false, true>;
      if (lds > 64 * 1024)
        PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const long per_cu2 = resident_per_cu(reinterpret_cast<const void*>(kern2), lds);
      const long resident2 = 256 * per_cu2;
      long nchunks2 = pv_ceil_div(resident2, nsplit2);
      if (nchunks2 > ngroups2) nchunks2 = ngroups2;
      const long blocks2 = pv_ceil_div(nchunks2, 8) * 8 * nsplit2;
      PV_LAUNCH(kern2, dim3((unsigned)blocks2), dim3(kThreads), lds, s, d, ksteps, (int)ngroups2, (int)nchunks2, nsplit2);
      pv_note_kernel("pw_stream_kernel");   // (launched through a function pointer: PV_LAUNCH saw only the variable)
      PV_LAUNCH_CHECK();
      return PV_OK;
    } else {
      return PV_ERR_UNSUPPORTED;
    }
  }
  auto kern = pw_stream_kernel<NT, TM, XFORM, KS, F32, false>;
  if (lds > 64 * 1024)
    PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // one resident generation of workgroups (what LDS and the 4-waves-per-SIMD register budget admit
  // per CU); the rest is the grid-stride loop, so the weight slab is staged once per resident
  // workgroup, not once per 64 voxels
  const long per_cu = resident_per_cu(reinterpret_cast<const void*>(kern), lds);
  const long resident = 256 * per_cu;
  long nchunks = pv_ceil_div(resident, nsplit);
  if (nchunks > ngroups) nchunks = ngroups;
  const long blocks = pv_ceil_div(nchunks, 8) * 8 * nsplit;
  PV_LAUNCH(kern, dim3((unsigned)blocks), dim3(kThreads), lds, s, d, ksteps, (int)ngroups, (int)nchunks, nsplit);
  pv_note_kernel("pw_stream_kernel");   // (launched through a function pointer: PV_LAUNCH saw only the variable)
  PV_LAUNCH_CHECK();
  return PV_OK;
}

template <int NT, int TM, bool XFORM, int KS>
int launch_pw_k(const pv_conv3d_desc& d, int ksteps, size_t lds, hipStream_t s) {
  if constexpr (!XFORM) {   // fp32 output / residual: the residual stream of the bf16 MViT plan
    if (d.y_f32) return launch_pw_f<NT, TM, XFORM, KS, true>(d, ksteps, lds, s);
  }
  return launch_pw_f<NT, TM, XFORM, KS, false>(d, ksteps, lds, s);
}

template <int NT, int TM, bool XFORM>
int launch_pw_x(const pv_conv3d_desc& d, int ksteps, size_t lds, hipStream_t s) {
  // whole-K register residency (with cross-group prefetch) only where it does not spill
  if (d.x2 != nullptr && ksteps != 3) return launch_pw_k<NT, TM, XFORM, 0>(d, ksteps, lds, s);   // two-operand variants: KS 3 and generic
  if (ksteps == 1) return launch_pw_k<NT, TM, XFORM, 1>(d, ksteps, lds, s);
  if (ksteps == 2) return launch_pw_k<NT, TM, XFORM, 2>(d, ksteps, lds, s);
  if (ksteps == 3) return launch_pw_k<NT, TM, XFORM, 3>(d, ksteps, lds, s);
  if (ksteps == 4) return launch_pw_k<NT, TM, XFORM, 4>(d, ksteps, lds, s);
  if constexpr (TM == 1) {   // X3D res4 / res5 conv_c: 216 and 432 input channels, whole K in registers
    if (ksteps == 7) return launch_pw_k<NT, TM, XFORM, 7>(d, ksteps, lds, s);
    if (ksteps == 14) return launch_pw_k<NT, TM, XFORM, 14>(d, ksteps, lds, s);
  }
  return launch_pw_k<NT, TM, XFORM, 0>(d, ksteps, lds, s);
}

template <int NT, int TM>
int launch_pw(const pv_conv3d_desc& d, int ksteps, size_t lds, hipStream_t s) {
  if (d.a_gate != nullptr || d.a_act != PV_ACT_NONE) {
    if (d.a_act == PV_ACT_SWISH && !pv_tune("pw_pk_swish", 1)) {
      pv_conv3d_desc d2 = d;
      d2.a_act = PV_ACT_SWISH | 16;
      return launch_pw_x<NT, TM, true>(d2, ksteps, lds, s);
    }
    return launch_pw_x<NT, TM, true>(d, ksteps, lds, s);
  }
  return launch_pw_x<NT, TM, false>(d, ksteps, lds, s);
}

}  // namespace

// Geometry-only test for the second K operand (pointers ignored; a squeeze-excitation gate is assumed
// present, which is the larger LDS footprint).
int pv_pwconv_x2_supported(const pv_conv3d_desc& d) {
  if (d.dtype != PV_BF16 || d.y_f32 || d.cin <= 0 || d.cin % 8) return 0;
  if (d.x2_cin <= 0 || d.x2_cin % 8 || d.x2_ld < d.x2_cin || d.x2_st < 1 || d.x2_sh < 1 || d.x2_sw < 1) return 0;
  if ((long)d.To * d.Ho * d.Wo < 64) return 0;
  const int ksteps = (d.cin + 31) / 32 + (d.x2_cin + 31) / 32;
  if (ksteps > 8) return 0;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const int NT = cout_p8 <= 32 ? 2 : (cout_p8 <= 64 ? 4 : 8);
  const size_t lds = (size_t)NT * 16 * (ksteps * 32 + 8) * 2 + (size_t)3 * NT * 16 * 4 + (size_t)4 * 2 * d.cin * 4;
  return lds <= 96 * 1024;
}

// Returns PV_OK when the streaming kernel took the op, PV_ERR_UNSUPPORTED to let the caller
// fall back to the generic implicit-GEMM kernel.
int pv_pwconv_stream_try(const pv_conv3d_desc& d, hipStream_t s) {
  if (d.dtype != PV_BF16) return PV_ERR_UNSUPPORTED;
  // fp32 I/O only in the combination the MViT plan uses: fp32 output with an (optional) fp32 residual
  if ((d.r_f32 != 0) != (d.residual != nullptr && d.y_f32) || (d.y_f32 && (d.a_gate || d.a_act != PV_ACT_NONE)))
    return PV_ERR_UNSUPPORTED;
  const long S_out = (long)d.To * d.Ho * d.Wo;
  if (d.a_gate && S_out < 64) return PV_ERR_UNSUPPORTED;  // a 64-voxel wave tile must span <= 2 clips
  const int cout_p8 = pv_round_up(d.cout, 8);
  if (d.x2 && (d.x2_cin <= 0 || d.x2_cin % 8 || d.x2_ld < d.x2_cin || d.x2_st < 1 || d.x2_sh < 1 || d.x2_sw < 1 || d.y_f32))
    return PV_ERR_UNSUPPORTED;
  const int ksteps = (d.cin + 31) / 32 + (d.x2 ? (d.x2_cin + 31) / 32 : 0);
  if (ksteps > 8 && ksteps != 14) return PV_ERR_UNSUPPORTED;
  int NT = cout_p8 <= 32 ? 2 : (cout_p8 <= 64 ? 4 : 8);
  auto lds_of = [&](int nt) {
    return (size_t)nt * 16 * (ksteps * 32 + 8) * 2 + (size_t)(d.x2 ? 3 : 2) * nt * 16 * 4 + (d.a_gate ? (size_t)4 * 2 * d.cin * 4 : 0);
  };
  size_t lds = lds_of(NT);
  if (lds > 96 * 1024) return PV_ERR_UNSUPPORTED;
  if (NT == 2) return launch_pw<2, 2>(d, ksteps, lds, s);
  if (NT == 4) return launch_pw<4, 2>(d, ksteps, lds, s);
  return launch_pw<8, 1>(d, ksteps, lds, s);
}
