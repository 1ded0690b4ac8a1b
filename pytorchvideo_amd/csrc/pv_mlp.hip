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
// Round 3 built this on 32-row wave tiles (v_mfma_f32_32x32x16_bf16, 192 accumulator + 96 operand registers per wave, ONE
// wave per SIMD); round 6 replaced it by the 16-row kernel below (two waves per SIMD, the LayerNorm in one pass over the row):
// 98.6 -> 78.2 us at 25 096 x 384 -> 1536 -> 384, 133.8 -> 130.9 us at 100 360 x 192 -> 768 -> 192, 252 -> 240 us at
// 401 416 x 96 -> 384 -> 192 (profiles/r6/bench_mlp_call5.txt); the 32-row kernel is gone from the library.
// The LayerNorm + Linear kernel at the end of this file (norm1 + q|k|v) keeps the 32-row mapping.
#include "pv_common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA issued from inline asm inside the MFMA streams: hipcc models `__builtin_amdgcn_global_load_lds` as an LDS access
// and drains lgkmcnt to 0 at the next use of ANY ds_read result (every other DMA piece in the fragment-ring loops below cost
// a full LDS round trip); an asm statement is opaque to the waitcnt pass.  The data is ordered for the readers by the
// explicit vmcnt(0) + s_barrier at the top of the next block, nothing else (MI355X_MICROARCH.md, LDS-DMA notes).
__device__ __forceinline__ void dma16_asm(const unsigned char* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void dma4_asm(const unsigned char* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_offset(const void* p) {   // generic pointer into LDS -> byte offset inside the workgroup's LDS
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) unsigned char*)p;
}

constexpr int kB1Bytes = 256;   // b1 of one hidden block: [2 lane halves][16] fp32 = 128 B, padded to one 4-byte DMA piece

// ---------------------------------------------------------------------------------------------------------------
// The fused MLP on 16-ROW wave tiles, TWO waves per SIMD.
//
// The 32-row kernel of round 3 was bound by instruction ISSUE, not by the matrix pipe (PMC, round 5: MFMA busy 22 %,
// issue-stalled 36 %, parked 26 %): a wave owned 32 rows x 384 channels of fp32 accumulators (192 registers) + 96 registers of
// operand fragments, so only ONE wave fit a SIMD and nothing covered its ~9 non-MFMA instructions per MFMA, its LDS round trips
// or its barrier waits.  Here a wave owns 16 rows on v_mfma_f32_16x16x32_bf16: Y[16 rows][Cout] is Cout/16 x 4 = 96
// accumulator registers, the normalised rows are C/32 x 4 = 48, everything fits 256 registers and a 512-thread workgroup
// (8 waves = the same 128 rows per CU) puts two waves on every SIMD: one multiplies while its partner computes GELUs,
// waits for LDS or sits at the barrier.  Price: a 1 KB weight fragment now feeds a 16-cycle MFMA instead of a 32-cycle
// one -- the LDS read rate per FLOP doubles (at the matrix peak the fragment reads alone would need 227 of the port's
// 256 B/clk), so this kernel is LDS-port-bound somewhere above half of the MFMA peak instead of issue-bound at a fifth.
//
// Roles (16x16x32: A[m = l&15][k = 8 (l>>4) + j], B[k = 8 (l>>4) + j][n = l&15], D[m = 4 (l>>4) + r][n = l&15]):
//   lane (g = l>>4, n = l&15) holds token row n of the wave's 16;  M = weight rows, K = channels (phase A) / hidden units (B).
//   bx[ks]   = xn[n][32 ks + 8 g + j]                       (phase-A B operand, C/32 fragments, registers for the whole kernel)
//   D_uh     = hidden units 16 uh + 4 g + r of token n      (two 16-unit halves of a 32-unit hidden block, two MFMA chains)
//   hp       = bf16 [D_0[0..3] | D_1[0..3]] after the activation: register for register the phase-B B operand when the
//              host orders fc2's K dimension as  k = 8 g + j  <->  unit (j < 4 ? 4 g + j : 16 + 4 g + j - 4)
//   Y[ob][r] = channel 32 (ob>>1) + 8 g + 4 (ob&1) + r of token n: the blocks (2 ks, 2 ks + 1) of a lane are the 8 consecutive
//              channels of its fragment bx[ks] -- the fp32 row read once serves LayerNorm, residual and the accumulators,
//              and the four lanes of a token cover a full 128-byte line of the fp32 stream per block pair.
// LDS image of hidden block j (host: emit_mvit.pack_mlp_weights):
//   [f = 2 ks + uh < C/16][l < 64][j8 < 8] bf16  W1[32 j + 16 uh + (l&15)][32 ks + 8 (l>>4) + j8]
//   [ob < Cout/16][l < 64][j8 < 8]         bf16  W2[32 (ob>>1) + 8 (m>>2) + 4 (ob&1) + (m&3)][32 (j-1) + unit(l>>4, j8)],  m = l&15
//   [u < 32] fp32 b1[32 j + u], then 128 bytes of padding
template <int KS2, int NOB16> struct Mlp16Geom {
  static constexpr int W1B = 2 * KS2 * 1024;
  static constexpr int W2B = NOB16 * 1024;
  static constexpr int STAGE = W1B + W2B + kB1Bytes;
  static constexpr int P = 2 * KS2 + NOB16;           // 1 KB pieces (= fragments) per hidden block
  static constexpr int NPW = (P + 7) / 8;             // pieces per wave
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// ABL (development variant of the library only, timing builds with WRONG results, tools/bench_mlp.py): 1 no activation,
// 2 no weight streaming in the loop, 3 no phase B, 4 no phase A, 5 no barrier, 6 no LDS fragment reads, 7 no MFMA at all
template <int KS2, int NOB16, bool LN, int ACT, int ABL = 0>
__global__ __launch_bounds__(512, 2) void mlp_rows16_kernel(const pv_mlp_desc d) {
  using G = Mlp16Geom<KS2, NOB16>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * G::STAGE + NOB16 * 192];   // + gamma | beta of the next block's norm1 (d.yn) + b2 (LayerNorm mode: added in the epilogue)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, g = lane >> 4;
  const long m = (long)blockIdx.x * 128 + wave * 16 + n16;
  const bool ok = m < d.M;
  const long mm = ok ? m : 0;
  const int NH = d.H >> 5;
  constexpr auto vm = [](int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); };
  constexpr int P = G::P, NPW = G::NPW;

  const unsigned char* wsrc = static_cast<const unsigned char*>(d.w12);
  auto stage = [&](int hb, int buf) {
    const unsigned char* src = wsrc + (long)hb * G::STAGE;
    unsigned char* dst = smem + buf * G::STAGE;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int p = 8 * i + wave < P ? 8 * i + wave : P - 1;     // every wave issues the same number of pieces (counted waits)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 1024 + lane * 16), (lptr_t)(dst + p * 1024), 16, 0, 0);
    }
    if (wave == 0)   // b1 block: 64 lanes x 4 bytes
      __builtin_amdgcn_global_load_lds((gptr_t)(src + P * 1024 + lane * 4), (lptr_t)(dst + P * 1024), 4, 0, 0);
  };
  stage(0, 0);
  stage(1, 1);      // (H = 32: block 1 is the padding block)

  // gamma | beta of the next block's norm1 -> LDS (read in the epilogue between global stores: a global load there is
  // serialised behind the stores by the compiler, see the 32-row kernel)
  static_assert(NOB16 * 8 <= 512, "the gamma | beta preload is one f32x4 per thread");
  if (d.yn != nullptr && tid < NOB16 * 8) {
    const int k = tid < NOB16 * 4 ? tid : tid - NOB16 * 4;
    reinterpret_cast<f32x4*>(smem + 3 * G::STAGE)[tid] = reinterpret_cast<const f32x4*>(tid < NOB16 * 4 ? d.nn_gamma : d.nn_beta)[k];
  }
  if (LN && tid >= 256 && tid < 256 + NOB16 * 4)      // b2 -> LDS (threads of the second half: the first loads gamma | beta)
    reinterpret_cast<f32x4*>(smem + 3 * G::STAGE + NOB16 * 128)[tid - 256] = reinterpret_cast<const f32x4*>(d.b2)[tid - 256];

  // ---- prologue: operand fragments and accumulator initialisation (residual + b2 go INTO the accumulators) ----
  bf16x8 bx[KS2];
  f32x4 Y[NOB16];
  if constexpr (LN) {
    static_assert(2 * KS2 == NOB16, "LayerNorm mode needs C == Cout");
    // ONE pass over the fp32 row: its quarter (the token's four lanes hold a quarter row each) is loaded straight INTO the
    // accumulators (Y = x is the residual), the statistics are taken from those registers (mean, then centred squares: no
    // cancellation problem, no shift), the normalised values become the operand fragments; b2 is added in the epilogue, from
    // LDS (adding it here made hipcc keep old and new accumulators side by side and spill 150 registers).  (The 32-row
    // kernel reads the row three times: with 192 accumulators + 96 fragment registers it cannot keep it.  H = 32 sweep,
    // tools/r6/mlp_h_sweep.py: 40-47 us of a 100 us launch were prologue + epilogue, all workgroups in it at the same time.)
    const float* xr = static_cast<const float*>(d.x) + mm * d.ldx + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      Y[2 * ks] = *reinterpret_cast<const f32x4*>(xr + 32 * ks);
      Y[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xr + 32 * ks + 4);
    }
    float s1 = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB16; ++ob) s1 += (Y[ob][0] + Y[ob][1]) + (Y[ob][2] + Y[ob][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float inv_c = 1.0f / (float)(32 * KS2);
    const float mean = s1 * inv_c;
    float s2 = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB16; ++ob)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = Y[ob][e] - mean; s2 += t * t; }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    float rstd = rsqrtf(s2 * inv_c + d.ln_eps);
    // the parameter loads below do not depend on the statistics: without this fence the scheduler hoists all 72 of them (288
    // registers) above the reductions and carries them through scratch
    asm volatile("" : "+v"(rstd) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      const int c0 = 32 * ks + 8 * g;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(d.ln_gamma + c0), g1 = *reinterpret_cast<const f32x4*>(d.ln_gamma + c0 + 4);
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(d.ln_beta + c0), e1 = *reinterpret_cast<const f32x4*>(d.ln_beta + c0 + 4);
      bf16x8 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[e] = (bf16_t)((Y[2 * ks][e] - mean) * rstd * g0[e] + e0[e]);
        t[4 + e] = (bf16_t)((Y[2 * ks + 1][e] - mean) * rstd * g1[e] + e1[e]);
      }
      {
        // pin the fragment HERE: its only use is in the main loop, and LLVM otherwise sinks all the arithmetic below the last
        // parameter load (every gamma / beta value of the row live at once: 56 registers through scratch)
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t pin = __builtin_bit_cast(u32x4_t, t);
        asm volatile("" : "+v"(pin) :: "memory");
        bx[ks] = __builtin_bit_cast(bf16x8, pin);
      }
    }
  } else {
    const bf16_t* xr = static_cast<const bf16_t*>(d.x) + mm * d.ldx + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) bx[ks] = *reinterpret_cast<const bf16x8*>(xr + 32 * ks);
#pragma unroll
    for (int ob = 0; ob < NOB16; ++ob) {
      const int c0 = 32 * (ob >> 1) + 8 * g + 4 * (ob & 1);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (d.residual != nullptr) v = *reinterpret_cast<const f32x4*>(d.residual + mm * d.ldr + c0);
      if (d.b2 != nullptr) v += *reinterpret_cast<const f32x4*>(d.b2 + c0);
      Y[ob] = v;
    }
  }

  // ---- hidden blocks: iteration hb multiplies phase A of block hb, then phase B of block hb - 1 with the activation of
  //      block hb sliced between those MFMAs (the image is packed for it: block j = [W1(j) | W2(j-1) | b1(j)]) ----
  __builtin_amdgcn_s_waitcnt(vm(0));     // every load the compiler knows of is complete: its waits must not land inside the loop, where they would drain the LDS-DMA prefetches it cannot see
  const unsigned smem_lds = __builtin_amdgcn_readfirstlane(lds_offset(smem));
  constexpr int NF = P;                               // fragments per block: W1 (2 KS2, order f = 2 ks + uh), then W2 (NOB16)
  constexpr int PF = NF < 6 ? NF : 6;                 // fragment ring (the partner wave covers most of an LDS round trip already)
  int cur = 0;
  bf16x8 hp;
#pragma unroll
  for (int j8 = 0; j8 < 8; ++j8) hp[j8] = (bf16_t)0.f;
  for (int hb = 0; hb <= NH; ++hb) {
    if constexpr (ABL != 2) {
      if (wave == 0) __builtin_amdgcn_s_waitcnt(vm(NPW + 1));
      else __builtin_amdgcn_s_waitcnt(vm(NPW));
    }
    if constexpr (ABL != 5) __builtin_amdgcn_s_barrier();
    const int nxt = cur == 0 ? 2 : cur - 1;   // (hb + 2) % 3
    const unsigned char* nsrc = wsrc + (long)(hb + 2) * G::STAGE;      // (two blocks of padding behind block NH: no branch)
    const unsigned ndst_lds = smem_lds + nxt * G::STAGE;
    const unsigned char* ws = smem + cur * G::STAGE + lane * 16;
    const float* b1s = reinterpret_cast<const float*>(smem + cur * G::STAGE + G::W1B + G::W2B) + 4 * g;
    cur = cur == 2 ? 0 : cur + 1;
    bf16x8 ring[PF];
#pragma unroll
    for (int f = 0; f < PF; ++f) ring[f] = *reinterpret_cast<const bf16x8*>(ws + f * 1024);
    f32x4 D0 = *reinterpret_cast<const f32x4*>(b1s), D1 = *reinterpret_cast<const f32x4*>(b1s + 16);   // b1 is the C operand of the first MFMAs
    __builtin_amdgcn_sched_barrier(0);
    constexpr int STEP = NF / NPW;          // a DMA piece of block hb + 2 every STEP fragments
    auto dma = [&](int f) {
      if constexpr (ABL == 2) return;
      if (f % STEP == 0 && f / STEP < NPW) {
        const int pc = 8 * (f / STEP) + wave;
        const int pq = (P % 8 == 0 || pc < P) ? pc : P - 1;      // every wave issues NPW pieces (the counted wait above relies on it)
        dma16_asm(nsrc + pq * 1024 + lane * 16, ndst_lds + pq * 1024);
      }
      if (f == NF - 1 && wave == 0) dma4_asm(nsrc + P * 1024 + lane * 4, ndst_lds + P * 1024);   // b1 block: 64 lanes x 4 bytes
    };
    // phase A of block hb: D_uh[16 hidden units][16 rows]
#pragma unroll
    for (int f = 0; f < 2 * KS2; ++f) {
      const bf16x8 afrag = ring[f % PF];
      if (ABL != 6 && f + PF < NF) ring[f % PF] = *reinterpret_cast<const bf16x8*>(ws + (f + PF) * 1024);
      if constexpr (ABL != 4 && ABL != 7) {
        if (f & 1) D1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bx[f >> 1], D1, 0, 0, 0);
        else D0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bx[f >> 1], D0, 0, 0, 0);
      } else {
        asm volatile("" :: "v"(afrag), "v"(bx[f >> 1]));
      }
      dma(f);
      __builtin_amdgcn_sched_barrier(0);
    }
    // phase B of block hb - 1, the activation of block hb (4 pairs x 3 stages = 12 units) spread over its MFMA slots
    f32x2_t gx[4], gax[4], gdn[4], gxx[4], gt[4], ge[4];
    unsigned hn[4];
    auto act_unit = [&](int u) {
      const int q = u / 3, st = u % 3;
      const f32x2_t dv = q < 2 ? f32x2_t{D0[2 * q], D0[2 * q + 1]} : f32x2_t{D1[2 * (q - 2)], D1[2 * (q - 2) + 1]};
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      if constexpr (ACT == PV_ACT_GELU && ABL != 1) {
        // pv_gelu_fast2 (A&S 7.1.25 erf, |error| <= 2.5e-5) cut into three stages, the same operations in the same order
        if (st == 0) {
          gx[q] = dv;
          gax[q][0] = fabsf(dv[0]);
          gax[q][1] = fabsf(dv[1]);
          gdn[q] = (gax[q] * 0.70710678118654752440f) * 0.47047f + 1.0f;
          gxx[q] = dv * dv;
        } else if (st == 1) {
          gt[q][0] = __builtin_amdgcn_rcpf(gdn[q][0]);
          gt[q][1] = __builtin_amdgcn_rcpf(gdn[q][1]);
          const f32x2_t a = gxx[q] * -0.72134752044448170368f;
          ge[q][0] = __builtin_amdgcn_exp2f(a[0]);
          ge[q][1] = __builtin_amdgcn_exp2f(a[1]);
        } else {
          const f32x2_t t = gt[q];
          const f32x2_t poly = t * (0.3480242f + t * (-0.0958798f + t * 0.7478556f));
          const f32x2_t gq = (gx[q] + gax[q] * (1.0f - poly * ge[q])) * 0.5f;
          const bf16x2_t pk = {(bf16_t)gq[0], (bf16_t)gq[1]};
          hn[q] = __builtin_bit_cast(unsigned, pk);
        }
      } else if (st == 0) {
        float r0 = dv[0], r1 = dv[1];
        if constexpr (ACT == PV_ACT_NONE || ABL == 1) {}
        else if constexpr (ACT == PV_ACT_RELU) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); }
        else { r0 = pv_apply_act(r0, d.act); r1 = pv_apply_act(r1, d.act); }
        const bf16x2_t pk = {(bf16_t)r0, (bf16_t)r1};
        hn[q] = __builtin_bit_cast(unsigned, pk);
      }
    };
#pragma unroll
    for (int ob = 0; ob < NOB16; ++ob) {
      const int f = 2 * KS2 + ob;
      const bf16x8 afrag = ring[f % PF];
      if (ABL != 6 && f + PF < NF) ring[f % PF] = *reinterpret_cast<const bf16x8*>(ws + (f + PF) * 1024);
      if constexpr (ABL != 3 && ABL != 7) Y[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, hp, Y[ob], 0, 0, 0);
      else asm volatile("" :: "v"(afrag), "v"(hp));
#pragma unroll
      for (int u = ob * 12 / NOB16; u < (ob + 1) * 12 / NOB16; ++u) act_unit(u);
      dma(f);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      hp = __builtin_bit_cast(bf16x8, u32x4_t{hn[0], hn[1], hn[2], hn[3]});
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: 8 consecutive channels per lane and block pair, a full 128-byte line per token and pair ----
  __builtin_amdgcn_s_waitcnt(vm(0));        // the padding block's LDS-DMA must not outlive the workgroup
  if constexpr (LN) {
    const f32x4* c2 = reinterpret_cast<const f32x4*>(smem + 3 * G::STAGE + NOB16 * 128);      // [Cout] b2
#pragma unroll
    for (int ob = 0; ob < NOB16; ++ob) Y[ob] += c2[8 * (ob >> 1) + 2 * g + (ob & 1)];           // channels 32 (ob>>1) + 8 g + 4 (ob&1) ..
  }
  if (ok) {
    float* yr = static_cast<float*>(d.y) + m * d.ldy + 8 * g;
#pragma unroll
    for (int ob = 0; ob < NOB16; ++ob) *reinterpret_cast<f32x4*>(yr + 32 * (ob >> 1) + 4 * (ob & 1)) = Y[ob];
  }
  // ---- norm1 of the NEXT MultiScaleBlock (layers/attention.py:729-737) from the rows this wave still holds ----
  if (d.yn != nullptr) {      // wave-uniform
    float s1 = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB16; ++ob) s1 += (Y[ob][0] + Y[ob][1]) + (Y[ob][2] + Y[ob][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    const float inv_c = 1.0f / (float)(16 * NOB16);
    const float mean = s1 * inv_c;
    float s2 = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB16; ++ob)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float t = Y[ob][r] - mean; s2 += t * t; }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 * inv_c + d.nn_eps);
    if (ok) {
      bf16_t* nr = static_cast<bf16_t*>(d.yn) + m * d.ldyn + 8 * g;
      const f32x4* gam = reinterpret_cast<const f32x4*>(smem + 3 * G::STAGE);      // [Cout] gamma, then [Cout] beta
      const f32x4* bet = gam + NOB16 * 4;
      static_assert(NOB16 % 2 == 0, "output blocks come in pairs");
#pragma unroll
      for (int pq = 0; pq < NOB16 / 2; ++pq) {
        bf16x8 o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 gg = gam[8 * pq + 2 * g + h], bb = bet[8 * pq + 2 * g + h];     // channels 32 pq + 8 g + 4 h ..
#pragma unroll
          for (int e = 0; e < 4; ++e) o[4 * h + e] = (bf16_t)((Y[2 * pq + h][e] - mean) * rstd * gg[e] + bb[e]);
        }
        *reinterpret_cast<bf16x8*>(nr + 32 * pq) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm + Linear on token rows:  y[m][:] = act( W . LayerNorm(x[m][:]) + b )  -- norm1 -> the fused q|k|v Linear of a
// MultiScaleBlock (layers/attention.py:729-737 norm1, :425-451 _qkv_proj) in one launch.  Same mapping as above, phase A
// only: the normalised rows are MFMA B operands in registers, the weight streams through LDS one 32-channel output block
// at a time (host-packed image, rows permuted by chi so that a lane's 16 accumulator registers are 16 consecutive
// channels), the epilogue adds the bias and stores 32 bytes of bf16 per lane and block.  The fp32 stream is read once
// (twice from L2 for the statistics), the bf16 operand tensor LayerNorm used to write and the GEMM to read is gone.
template <int KS, int MINW>
__global__ __launch_bounds__(256, MINW) void ln_linear_rows_kernel(const pv_ln_linear_desc d) {
  constexpr int STAGE = KS * 1024 + kB1Bytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE];      // three stage buffers: the LDS-DMA of block nb + 2 is issued while block nb is multiplied (with two, the last pieces of a block are requested ~100 cycles before the wait and their L2 latency is exposed)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long m = (long)blockIdx.x * 128 + wave * 32 + l31;
  const bool ok = m < d.M;
  const long mm = ok ? m : 0;
  const int NB = d.N >> 5;
  constexpr auto vm = [](int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); };

  const unsigned char* wsrc = static_cast<const unsigned char*>(d.wb);
  auto stage = [&](int nb, int buf) {
    const unsigned char* src = wsrc + (long)nb * STAGE;
    unsigned char* dst = smem + buf * STAGE;
#pragma unroll
    for (int p0 = 0; p0 < KS; p0 += 4) {
      const int p = p0 + wave < KS ? p0 + wave : KS - 1;   // every wave issues the same number of pieces (counted waits)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 1024 + lane * 16), (lptr_t)(dst + p * 1024), 16, 0, 0);
    }
    if (wave == (KS & 3))   // bias block: 64 lanes x 4 bytes, issued by the wave with the fewest fragments
      __builtin_amdgcn_global_load_lds((gptr_t)(src + KS * 1024 + lane * 4), (lptr_t)(dst + KS * 1024), 4, 0, 0);
  };

  // ---- LayerNorm of this lane's half row -> MFMA B fragments (two passes over the row: shifted statistics, then the normalisation; the fragments take a detour through LDS because values vector arithmetic produces start life in the ArchVGPR half of a 512-register file and hipcc then carries them through scratch)
  bf16x8 bx[KS];
  {
    const float* xr = static_cast<const float*>(d.x) + mm * d.ldx + 16 * hi;
    const float shift0 = static_cast<const float*>(d.x)[mm * d.ldx];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < KS / 2; ++q) {
      const f32x4* p4 = reinterpret_cast<const f32x4*>(xr + 32 * q);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = p4[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[e] - shift0; s1 += t; s2 += t * t; }
      }
      if ((q & 3) == 3) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float inv_c = 1.0f / (float)(16 * KS);
    const float mu_s = s1 * inv_c;
    const float mean = shift0 + mu_s;
    const float rstd = rsqrtf(fmaxf(s2 * inv_c - mu_s * mu_s, 0.f) + d.ln_eps);
    constexpr int NQ = KS / 2;
    constexpr int GQ2 = (NQ % 2 == 0) ? NQ / 2 : 1;           // groups per LDS round trip: 4 waves x GQ2 x 2 KB of the (still idle) stage buffers
    static_assert(NQ % GQ2 == 0 && 4 * GQ2 * 2048 <= 2 * STAGE, "LayerNorm staging does not fit");
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
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
      bf16x8 t0, t1;
#pragma unroll
      for (int j = 0; j < 8; ++j) { t0[j] = (bf16_t)xn[j]; t1[j] = (bf16_t)xn[8 + j]; }
      unsigned char* lp = smem + wave * (GQ2 * 2048) + (q % GQ2) * 2048 + lane * 32;
      *reinterpret_cast<bf16x8*>(lp) = t0;
      *reinterpret_cast<bf16x8*>(lp + 16) = t1;
      if (q % GQ2 == GQ2 - 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int qq = q - (GQ2 - 1); qq <= q; ++qq) {
          const unsigned char* rp = smem + wave * (GQ2 * 2048) + (qq % GQ2) * 2048 + lane * 32;
          bx[2 * qq] = *reinterpret_cast<const bf16x8*>(rp);
          bx[2 * qq + 1] = *reinterpret_cast<const bf16x8*>(rp + 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  __builtin_amdgcn_s_barrier();     // every wave is done with its LayerNorm staging area: the weight stream may use the buffers
  stage(0, 0);
  stage(1, 1);
  bf16_t* yr = static_cast<bf16_t*>(d.y) + m * d.ldy + 16 * hi;
  // a wave with at least one row in range issues both store instructions of a block (partially masked or not); a wave
  // entirely past the last row issues none (the compiler branches around them on exec == 0) and must not count them
  const bool wave_stores = __builtin_amdgcn_ballot_w64(ok) != 0ul;
  const unsigned smem_lds = __builtin_amdgcn_readfirstlane(lds_offset(smem));
  constexpr int PF = KS < 8 ? KS : 8;             // fragment ring: reads run PF fragments ahead of the MFMA that consumes them
  constexpr int NPW = (KS + 3) / 4;
  int cur = 0;
  for (int nb = 0; nb < NB; ++nb) {
    // Issue order of this wave's vector-memory operations: ... DMA(nb), stores(nb - 2), DMA(nb + 1), stores(nb - 1).
    // Block nb has landed once at most the operations issued after its pieces are still in flight (in-order return).
    {
      const int dma_next = (wave == (KS & 3)) ? NPW + 1 : NPW;
      const int st = wave_stores ? 2 : 0;
      const int allow = dma_next + (nb >= 1 ? st : 0) + (nb >= 2 ? st : 0);
      // (s_waitcnt takes an immediate: the handful of possible counts are separate instructions)
      if (allow >= NPW + 5) __builtin_amdgcn_s_waitcnt(vm(NPW + 5));
      else if (allow == NPW + 4) __builtin_amdgcn_s_waitcnt(vm(NPW + 4));
      else if (allow == NPW + 3) __builtin_amdgcn_s_waitcnt(vm(NPW + 3));
      else if (allow == NPW + 2) __builtin_amdgcn_s_waitcnt(vm(NPW + 2));
      else if (allow == NPW + 1) __builtin_amdgcn_s_waitcnt(vm(NPW + 1));
      else __builtin_amdgcn_s_waitcnt(vm(NPW));
    }
    __builtin_amdgcn_s_barrier();
    const int nxt = cur == 0 ? 2 : cur - 1;                          // (nb + 2) % 3
    const unsigned char* nsrc = wsrc + (long)(nb + 2) * STAGE;      // (two blocks of padding behind the last: no branch)
    const unsigned ndst_lds = smem_lds + nxt * STAGE;
    const unsigned char* ws = smem + cur * STAGE + lane * 16;
    const float* bs = reinterpret_cast<const float*>(smem + cur * STAGE + KS * 1024) + 16 * hi;
    cur = cur == 2 ? 0 : cur + 1;
    bf16x8 ring[PF];
#pragma unroll
    for (int f = 0; f < PF; ++f) ring[f] = *reinterpret_cast<const bf16x8*>(ws + f * 1024);
    f32x4 bb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bb[g] = *reinterpret_cast<const f32x4*>(bs + 4 * g);
    f32x16 D0, D1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { D0[r] = 0.f; D1[r] = 0.f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < KS; ++f) {
      const bf16x8 afrag = ring[f % PF];
      if (f + PF < KS) ring[f % PF] = *reinterpret_cast<const bf16x8*>(ws + (f + PF) * 1024);
      if (f & 1) D1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bx[f], D1, 0, 0, 0);
      else D0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bx[f], D0, 0, 0, 0);
      if (f % (KS / NPW) == 0 && f / (KS / NPW) < NPW) {
        const int pc = 4 * (f / (KS / NPW)) + wave;
        const int pq = (KS % 4 == 0 || pc < KS) ? pc : KS - 1;      // every wave issues NPW pieces (counted waits)
        dma16_asm(nsrc + pq * 1024 + lane * 16, ndst_lds + pq * 1024);
      }
      if (f == KS - 1 && wave == (KS & 3)) dma4_asm(nsrc + KS * 1024 + lane * 4, ndst_lds + KS * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    float h[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) h[4 * g + e] = D0[4 * g + e] + D1[4 * g + e] + bb[g][e];
    if (d.act != PV_ACT_NONE) pv_apply_act_n<true, 16>(h, d.act);
    bf16x8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0[j] = (bf16_t)h[j]; o1[j] = (bf16_t)h[8 + j]; }
    if (ok) {
      *reinterpret_cast<bf16x8*>(yr + 32 * nb) = o0;
      *reinterpret_cast<bf16x8*>(yr + 32 * nb + 8) = o1;
    }
  }
  __builtin_amdgcn_s_waitcnt(vm(0));        // the padding block's LDS-DMA must not outlive the workgroup
}

// (Round 4's two variants of this kernel -- two row sets per wave, +-0 on MViT-B, and the attention output projection +
// residual join on a row-resident kernel, -4 % against the tiled GEMM -- were measured, documented in DESIGN 7 and removed
// in round 5: the product library carries no switched-off kernels.)

template <int KS, int MINW> int launch_ln_linear(const pv_ln_linear_desc& d, hipStream_t s) {
  PV_LAUNCH((ln_linear_rows_kernel<KS, MINW>), dim3((unsigned)pv_ceil_div(d.M, 128)), dim3(256), 0, s, d);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

int check_ln_linear(const pv_ln_linear_desc& d) {
  if (!d.x || !d.wb || !d.y || d.M <= 0 || d.M > 0x7fffffffL) return PV_ERR_INVALID;
  if (d.dtype != PV_BF16) return PV_ERR_UNSUPPORTED;
  if (d.C <= 0 || d.C % 32 || d.N <= 0 || d.N % 32) return PV_ERR_UNSUPPORTED;
  if (!d.ln_gamma || !d.ln_beta) return PV_ERR_INVALID;
  if (d.ldx < d.C || d.ldx % 4 || d.ldy < d.N || d.ldy % 8) return PV_ERR_INVALID;
  return PV_OK;
}

template <int KS2, int NOB16> int launch16(const pv_mlp_desc& d, hipStream_t s) {
  const unsigned grid = (unsigned)pv_ceil_div(d.M, 128);
  const bool ln = d.ln_gamma != nullptr;
#define PV16_GO(ACTv)                                                                                                    \
  do {                                                                                                                   \
    if (ln) {                                                                                                            \
      if constexpr (2 * KS2 == NOB16) PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, true, ACTv>), dim3(grid), dim3(512), 0, s, d); \
      else return PV_ERR_UNSUPPORTED;                                                                                    \
    } else {                                                                                                             \
      PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, false, ACTv>), dim3(grid), dim3(512), 0, s, d);                           \
    }                                                                                                                    \
  } while (0)
#ifdef PV_DEV_ABLATION   // timing builds (1-7 give WRONG results): development variant of the library only
  if constexpr (KS2 == 12 && NOB16 == 24) {
    const int abl = pv_tune("mlp_abl", 0);
    if (abl && ln && d.act == PV_ACT_GELU) {
      switch (abl) {
        case 1: PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, true, PV_ACT_GELU, 1>), dim3(grid), dim3(512), 0, s, d); break;
        case 2: PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, true, PV_ACT_GELU, 2>), dim3(grid), dim3(512), 0, s, d); break;
        case 3: PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, true, PV_ACT_GELU, 3>), dim3(grid), dim3(512), 0, s, d); break;
        case 4: PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, true, PV_ACT_GELU, 4>), dim3(grid), dim3(512), 0, s, d); break;
        case 5: PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, true, PV_ACT_GELU, 5>), dim3(grid), dim3(512), 0, s, d); break;
        case 6: PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, true, PV_ACT_GELU, 6>), dim3(grid), dim3(512), 0, s, d); break;
        default: PV_LAUNCH((mlp_rows16_kernel<KS2, NOB16, true, PV_ACT_GELU, 7>), dim3(grid), dim3(512), 0, s, d); break;
      }
      PV_LAUNCH_CHECK();
      return PV_OK;
    }
  }
#endif
  if (d.act == PV_ACT_GELU) PV16_GO(PV_ACT_GELU);
  else PV16_GO(-1);
#undef PV16_GO
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
  if (d.yn && (!d.nn_gamma || !d.nn_beta || d.ldyn < d.Cout || d.ldyn % 8)) return PV_ERR_INVALID;
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
  if (d.C == 96 && d.Cout == 96) return launch16<3, 6>(d, s);
  if (d.C == 96 && d.Cout == 192) return launch16<3, 12>(d, s);
  if (d.C == 192 && d.Cout == 192) return launch16<6, 12>(d, s);
  if (d.C == 192 && d.Cout == 384) return launch16<6, 24>(d, s);
  if (d.C == 384 && d.Cout == 384) return launch16<12, 24>(d, s);
  return PV_ERR_UNSUPPORTED;
}

extern "C" int pv_ln_linear_rows_supported(const pv_ln_linear_desc* d) {
  if (!d || check_ln_linear(*d) != PV_OK) return 0;
  return d->C == 96 || d->C == 192 || d->C == 384 || d->C == 768;
}

extern "C" int pv_ln_linear_rows(const pv_ln_linear_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const int rc = check_ln_linear(*dp);
  if (rc != PV_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dp->C) {
    case 96: return launch_ln_linear<6, 2>(*dp, s);
    case 192: return launch_ln_linear<12, 2>(*dp, s);
    case 384: return launch_ln_linear<24, 2>(*dp, s);
    case 768: return launch_ln_linear<48, 1>(*dp, s);
    default: return PV_ERR_UNSUPPORTED;
  }
}
