// Row-resident fused MLP of a MultiScaleBlock (gfx950):
//
//   y = R + b2 + W2 . act( W1 . xn + b1 ),   xn = LayerNorm(x) (computed here) or a bf16 operand tensor
//
// i.e. norm2 -> Mlp.fc1 -> GELU -> Mlp.fc2 -> + residual of pytorchvideo/layers/attention.py:102-114,750-757 in ONE
// launch: the hidden tensor (4 x the token width, the largest tensor of the block) never leaves the chip, the fp32
// token stream is read once (it is both the LayerNorm input and the residual) and written once.  Per block of
// MViT-B (M = 25 096 tokens, 384 -> 1536 -> 384) that is 77 MB of HBM traffic instead of 270 MB for
// LayerNorm + fc1 + fc2 as three launches.
//
// Mapping on CDNA4 (one workgroup = 4 waves = 128 token rows, one wave per SIMD, up to ~400 of the 512 registers):
//   * a wave owns 32 rows for the whole kernel.  Its normalised rows live in REGISTERS as the B operands of
//     v_mfma_f32_32x32x16_bf16 (C/16 fragments of 8 bf16), its output tile Y[32 rows][Cout] as Cout/32 accumulator
//     blocks (the residual + b2 are loaded INTO the accumulators before the first MFMA);
//   * the hidden dimension is walked in blocks of 32 units.  Phase A: D[32 units][32 rows] = W1blk . xn^T (C/16 MFMAs, the
//     weight fragments come from LDS).  D + b1 -> GELU -> bf16 is, register for register, the B operand of phase B
//     (the C/D layout of the 32x32 MFMA holds 16 units of one row per lane; the K order of phase B is permuted on the
//     host to that layout), so the hidden activations never touch LDS either.  Phase B: Y[ob] += W2blk[ob] . H for
//     every 32-channel output block (2 MFMAs each);
//   * only weights go through LDS: the host packs, per hidden block, the exact LDS image
//     [W1: C/16 fragments | W2: Cout/32 x 2 fragments | b1] (1 KB per fragment, lane-linear), so staging is a linear
//     `global_load_lds` stream (no address arithmetic, perfectly coalesced, L2-resident after the first workgroup),
//     double buffered: block hb+1 lands while block hb is multiplied; one barrier per hidden block;
//   * output rows are permuted (chi, as in pv_gemm.hip) so that a lane's 16 accumulator registers are 16 consecutive
//     channels of one row: residual loads and stores are 64-byte runs per lane.
//
// K order of phase A (host packing and the in-register fragments agree): fragment ks, lane half hi, element j is
// channel 32*(ks>>1) + 16*hi + 8*(ks&1) + j -- chosen so that a lane's fragment channels are exactly the channels of
// its accumulator registers; the fp32 row read once therefore serves the LayerNorm AND initialises the accumulators.
#include "pv_common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kB1Bytes = 256;   // b1 of one hidden block: [2 lane halves][16] fp32 = 128 B, padded to one 4-byte DMA piece

template <int KS, int NOB> struct MlpGeom {
  static constexpr int W1B = KS * 1024;                 // bytes of the W1 image of one hidden block
  static constexpr int W2B = NOB * 2 * 1024;
  static constexpr int STAGE = W1B + W2B + kB1Bytes;
};

// KS = C / 16 (even), NOB = Cout / 32, LN: x is the fp32 stream (LayerNorm here, residual = x, needs C == Cout)
template <int KS, int NOB, bool LN, int MINW>
__global__ __launch_bounds__(256, MINW) void mlp_rows_kernel(const pv_mlp_desc d) {
  using G = MlpGeom<KS, NOB>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G::STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long m = (long)blockIdx.x * 128 + wave * 32 + l31;
  const bool ok = m < d.M;
  const long mm = ok ? m : 0;
  const int NH = d.H >> 5;
  constexpr auto vm = [](int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); };
  // the widest LayerNorm variant sits at the edge of the 512-register file: one phase-A accumulator chain instead of two
  constexpr bool ONE_D = false;

  const unsigned char* wsrc = static_cast<const unsigned char*>(d.w12);
  auto stage = [&](int hb, int buf) {
    const unsigned char* src = wsrc + (long)hb * G::STAGE;
    unsigned char* dst = smem + buf * G::STAGE;
    constexpr int P = KS + 2 * NOB;
#pragma unroll
    for (int p0 = 0; p0 < P; p0 += 4) {
      const int p = p0 + wave;
      if (p < P)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 1024 + lane * 16), (lptr_t)(dst + p * 1024), 16, 0, 0);
    }
    if (wave == 0)   // b1 block: 64 lanes x 4 bytes
      __builtin_amdgcn_global_load_lds((gptr_t)(src + P * 1024 + lane * 4), (lptr_t)(dst + P * 1024), 4, 0, 0);
  };
  stage(0, 0);

  // ---- prologue: operand fragments and accumulator initialisation --------------------------------------------
  bf16x8 bx[KS];
  f32x16 Y[NOB];
  if constexpr (LN) {
    static_assert(KS == 2 * NOB, "LayerNorm mode needs C == Cout");
    const float* xr = static_cast<const float*>(d.x) + mm * d.ldx + 16 * hi;
    // Pass 1: row statistics, one shifted pass (shift = the row's first element, the same for both lane halves: the
    // sums of (x - shift) and (x - shift)^2 do not cancel catastrophically however far the row's mean is from 0).
    // The row is NOT kept: 192 fp32 values per lane next to the fragments and the accumulators would not fit the
    // vector file; pass 2 reads it again (L2 hit) 32 channels at a time.
    const float shift0 = static_cast<const float*>(d.x)[mm * d.ldx];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NOB; ++q) {
      const f32x4* p4 = reinterpret_cast<const f32x4*>(xr + 32 * q);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = p4[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[e] - shift0; s1 += t; s2 += t * t; }
      }
      if ((q & 3) == 3) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }   // at most 4 groups of loads in flight
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float inv_c = 1.0f / (float)(16 * KS);
    const float mu_s = s1 * inv_c;                       // mean - shift
    const float mean = shift0 + mu_s;
    const float var = fmaxf(s2 * inv_c - mu_s * mu_s, 0.f);
    const float rstd = rsqrtf(var + d.ln_eps);
#pragma unroll
    for (int q = 0; q < NOB; ++q) {
      const f32x4* g4 = reinterpret_cast<const f32x4*>(d.ln_gamma + 32 * q + 16 * hi);
      const f32x4* b4 = reinterpret_cast<const f32x4*>(d.ln_beta + 32 * q + 16 * hi);
      const f32x4* p4 = reinterpret_cast<const f32x4*>(xr + 32 * q);
      float xn[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = p4[g], gg = g4[g], bb = b4[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) xn[4 * g + e] = (v[e] - mean) * rstd * gg[e] + bb[e];
      }
      {
        // The fragments take a detour through LDS (this wave's own 2 KB per 32-channel group, inside stage buffer 1,
        // which the weight stream does not touch before the first barrier of the main loop).  Values that vector
        // arithmetic produces start life in the ArchVGPR half of the register file; 96 long-lived fragment registers
        // made that way plus the 192 accumulators made hipcc carry 270-540 registers through scratch, whereas
        // fragments that come out of a LOAD (as in the bf16-operand mode) allocate cleanly (432 registers, no scratch).
        constexpr int GQ = (NOB % 2 == 0) ? NOB / 2 : 1;      // groups per round trip: 4 waves x GQ x 2 KB <= one stage buffer
        static_assert(4 * GQ * 2048 <= G::STAGE, "LayerNorm staging does not fit the idle stage buffer");
        bf16x8 t0, t1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { t0[j] = (bf16_t)xn[j]; t1[j] = (bf16_t)xn[8 + j]; }
        unsigned char* lp = smem + G::STAGE + wave * (GQ * 2048) + (q % GQ) * 2048 + lane * 32;
        *reinterpret_cast<bf16x8*>(lp) = t0;
        *reinterpret_cast<bf16x8*>(lp + 16) = t1;
        if (q % GQ == GQ - 1) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (also keeps the compiler from forwarding the stores)
#pragma unroll
          for (int qq = q - (GQ - 1); qq <= q; ++qq) {
            const unsigned char* rp = smem + G::STAGE + wave * (GQ * 2048) + (qq % GQ) * 2048 + lane * 32;
            bx[2 * qq] = *reinterpret_cast<const bf16x8*>(rp);
            bx[2 * qq + 1] = *reinterpret_cast<const bf16x8*>(rp + 16);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next round overwrites the same 2 KB slots
        }
      }
      // one 32-channel group at a time: without the fence the scheduler hoists every gamma / beta load of the row
      // (2 x 192 registers) above the arithmetic and spills
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    // Pass 3: the residual (= the row itself) straight from memory into the accumulator blocks.  It must be a LOAD, not
    // the values of pass 2: accumulators produced by vector arithmetic start life in the ArchVGPR half of the file, which
    // the fragments already fill, and the compiler then carries all 192 of them through scratch (measured: 540 spilled
    // registers).  The clobber keeps it from merging these loads with pass 2's.  b2 is added in the epilogue.
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < NOB; ++q) {
      const f32x4* p4 = reinterpret_cast<const f32x4*>(xr + 32 * q);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = p4[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) Y[q][4 * g + e] = v[e];
      }
    }
  } else {
    const bf16_t* xr = static_cast<const bf16_t*>(d.x) + mm * d.ldx + 16 * hi;
#pragma unroll
    for (int q = 0; q < KS / 2; ++q) {
      bx[2 * q] = *reinterpret_cast<const bf16x8*>(xr + 32 * q);
      bx[2 * q + 1] = *reinterpret_cast<const bf16x8*>(xr + 32 * q + 8);
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[ob][r] = 0.f;
      if (d.residual != nullptr) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(d.residual + mm * d.ldr + 32 * ob + 16 * hi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = p4[g];
#pragma unroll
          for (int e = 0; e < 4; ++e) Y[ob][4 * g + e] = v[e];
        }
      }
      if (d.b2 != nullptr) {
        const f32x4* c4 = reinterpret_cast<const f32x4*>(d.b2 + 32 * ob + 16 * hi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 cc = c4[g];
#pragma unroll
          for (int e = 0; e < 4; ++e) Y[ob][4 * g + e] += cc[e];
        }
      }
    }
  }

  // ---- hidden blocks ------------------------------------------------------------------------------------------
  for (int hb = 0; hb < NH; ++hb) {
    __builtin_amdgcn_s_waitcnt(vm(0));      // this wave's pieces of block hb have landed ...
    __builtin_amdgcn_s_barrier();           // ... everybody's have, and everybody is done reading the other buffer
    if (hb + 1 < NH) stage(hb + 1, (hb + 1) & 1);
    const unsigned char* w1s = smem + (hb & 1) * G::STAGE + lane * 16;
    const unsigned char* w2s = w1s + G::W1B;
    const float* b1s = reinterpret_cast<const float*>(smem + (hb & 1) * G::STAGE + G::W1B + G::W2B) + 16 * hi;
    f32x16 D0, D1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { D0[r] = 0.f; D1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(w1s + ks * 1024);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(w1s + (ks + 1) * 1024);
      D0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bx[ks], D0, 0, 0, 0);
      if constexpr (ONE_D) D0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx[ks + 1], D0, 0, 0, 0);
      else D1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx[ks + 1], D1, 0, 0, 0);
    }
    float h[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) h[4 * g + e] = D0[4 * g + e] + D1[4 * g + e] + bb[e];
    }
    pv_apply_act_n<true, 16>(h, d.act);
    bf16x8 hf0, hf1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { hf0[j] = (bf16_t)h[j]; hf1[j] = (bf16_t)h[8 + j]; }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(w2s + (ob * 2) * 1024);
      Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hf0, Y[ob], 0, 0, 0);
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(w2s + (ob * 2 + 1) * 1024);
      Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hf1, Y[ob], 0, 0, 0);
    }
  }

  // ---- epilogue: 16 consecutive channels per lane and output block ---------------------------------------------
  if (ok) {
    float* yr = static_cast<float*>(d.y) + m * d.ldy + 16 * hi;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      f32x4* p4 = reinterpret_cast<f32x4*>(yr + 32 * ob);
      if constexpr (LN) {
        const f32x4* c4 = reinterpret_cast<const f32x4*>(d.b2 + 32 * ob + 16 * hi);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 cc = c4[g];
          p4[g] = f32x4{Y[ob][4 * g] + cc[0], Y[ob][4 * g + 1] + cc[1], Y[ob][4 * g + 2] + cc[2], Y[ob][4 * g + 3] + cc[3]};
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) p4[g] = f32x4{Y[ob][4 * g], Y[ob][4 * g + 1], Y[ob][4 * g + 2], Y[ob][4 * g + 3]};
      }
    }
  }
}

template <int KS, int NOB, int MINW> int launch(const pv_mlp_desc& d, hipStream_t s) {
  const unsigned grid = (unsigned)pv_ceil_div(d.M, 128);
  if (d.ln_gamma != nullptr) {
    if constexpr (KS == 2 * NOB) hipLaunchKernelGGL((mlp_rows_kernel<KS, NOB, true, MINW>), dim3(grid), dim3(256), 0, s, d);
    else return PV_ERR_UNSUPPORTED;
  } else {
    hipLaunchKernelGGL((mlp_rows_kernel<KS, NOB, false, MINW>), dim3(grid), dim3(256), 0, s, d);
  }
  PV_LAUNCH_CHECK();
  return PV_OK;
}

int check(const pv_mlp_desc& d) {
  if (!d.x || !d.w12 || !d.y || d.M <= 0 || d.M > 0x7fffffffL) return PV_ERR_INVALID;
  if (d.dtype != PV_BF16) return PV_ERR_UNSUPPORTED;
  if (d.C <= 0 || d.C % 32 || d.H <= 0 || d.H % 32 || d.Cout <= 0 || d.Cout % 32) return PV_ERR_UNSUPPORTED;
  const bool ln = d.ln_gamma != nullptr;
  if (ln && (d.ln_beta == nullptr || d.b2 == nullptr || d.C != d.Cout || d.residual != nullptr)) return PV_ERR_INVALID;
  if (d.ldx < d.C || d.ldx % (ln ? 4 : 8) || d.ldy < d.Cout || d.ldy % 4) return PV_ERR_INVALID;
  if (d.residual && (d.ldr < d.Cout || d.ldr % 4)) return PV_ERR_INVALID;
  return PV_OK;
}

}  // namespace

extern "C" int pv_mlp_rows_supported(const pv_mlp_desc* d) {
  if (!d || check(*d) != PV_OK) return 0;
  const int c = d->C, o = d->Cout;
  return (c == 96 && o == 192) || (c == 96 && o == 96) || (c == 192 && o == 192) || (c == 192 && o == 384) ||
         (c == 384 && o == 384);
}

extern "C" int pv_mlp_rows(const pv_mlp_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const int rc = check(*dp);
  if (rc != PV_OK) return rc;
  const pv_mlp_desc& d = *dp;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d.C == 96 && d.Cout == 96) return launch<6, 3, 2>(d, s);
  if (d.C == 96 && d.Cout == 192) return launch<6, 6, 2>(d, s);
  if (d.C == 192 && d.Cout == 192) return launch<12, 6, 1>(d, s);
  if (d.C == 192 && d.Cout == 384) return launch<12, 12, 1>(d, s);
  if (d.C == 384 && d.Cout == 384) return launch<24, 12, 1>(d, s);
  return PV_ERR_UNSUPPORTED;
}
