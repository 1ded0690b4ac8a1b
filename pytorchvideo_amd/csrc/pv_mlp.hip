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

template <int KS, int NOB> struct MlpGeom {
  static constexpr int W1B = KS * 1024;                 // bytes of the W1 image of one hidden block
  static constexpr int W2B = NOB * 2 * 1024;
  static constexpr int STAGE = W1B + W2B + kB1Bytes;
};

// KS = C / 16 (even), NOB = Cout / 32, LN: x is the fp32 stream (LayerNorm here, residual = x, needs C == Cout)
// ABL: ablation builds for tools/bench_mlp.py (timing only, wrong results): 1 no activation, 2 no weight streaming, 3 no phase B,
// 4 no phase A, 5 no barrier, 6 no LDS fragment reads; 7 / 8 are correct variants kept for A/B: one phase-A accumulator chain,
// activation stages pinned into the MFMA gaps
// ACT: the activation between the two Linears as a compile-time constant (PV_ACT_GELU for MViT), or -1 = read d.act at
// run time (any pv_act; a branch tree per element inside the MFMA stream, slower).
template <int KS, int NOB, bool LN, int MINW, int ACT, int ABL = 0>
__global__ __launch_bounds__(256, MINW) void mlp_rows_kernel(const pv_mlp_desc d) {
  using G = MlpGeom<KS, NOB>;
  // THREE stage buffers: the LDS-DMA of block hb + 2 is issued while block hb is multiplied.  With two (prefetch distance
  // one block) the last pieces of a block are issued ~100 cycles before the wait at the top of the next iteration and their
  // whole L2 latency (~1.1 us) is exposed: measured 2.6 us per hidden block for 1.1 us of MFMA + GELU work.
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * G::STAGE + 3 * NOB * 128];   // + b2 (LayerNorm mode: read in the epilogue) + gamma | beta of the NEXT block's norm1 (d.yn)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long m = (long)blockIdx.x * 128 + wave * 32 + l31;
  const bool ok = m < d.M;
  const long mm = ok ? m : 0;
  const int NH = d.H >> 5;
  constexpr auto vm = [](int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); };
  // Two phase-A accumulator chains.  (With two, hipcc parks two of the Y blocks in ArchVGPRs during phase A and moves them
  // back -- 64 v_accvgpr moves per hidden block -- yet ONE chain measured 2 % slower: 114.6 vs 112.3 us at 384 -> 1536 -> 384.)
  constexpr bool ONE_D = (ABL == 7);
  constexpr bool PIN = (ABL == 8);      // activation stages tied into the phase-B MFMA gaps (measured 117.0 us against 112.3 with the stages left where hipcc sinks them, below the last MFMA: a gap hides ~5 instructions, a stage plus the fragment read, the wait and the DMA piece are 10-12)

  const unsigned char* wsrc = static_cast<const unsigned char*>(d.w12);
  auto stage = [&](int hb, int buf) {
    const unsigned char* src = wsrc + (long)hb * G::STAGE;
    unsigned char* dst = smem + buf * G::STAGE;
    constexpr int P = KS + 2 * NOB;
#pragma unroll
    for (int p0 = 0; p0 < P; p0 += 4) {
      const int p = p0 + wave < P ? p0 + wave : P - 1;     // every wave issues the same number of pieces (counted waits)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 1024 + lane * 16), (lptr_t)(dst + p * 1024), 16, 0, 0);
    }
    if (wave == 0)   // b1 block: 64 lanes x 4 bytes
      __builtin_amdgcn_global_load_lds((gptr_t)(src + P * 1024 + lane * 4), (lptr_t)(dst + P * 1024), 4, 0, 0);
  };
  stage(0, 0);
  stage(1, 1);      // (H = 32: block 1 is the padding block)

  // gamma | beta of the next block's norm1 -> LDS (read in the epilogue between global stores, like b2 below)
  static_assert(NOB * 16 <= 256, "the gamma | beta preload is one f32x4 per thread of a 256-thread workgroup");
  if (d.yn != nullptr && tid < NOB * 16) {
    const int k = tid < NOB * 8 ? tid : tid - NOB * 8;
    reinterpret_cast<f32x4*>(smem + 3 * G::STAGE + NOB * 128)[tid] =
        reinterpret_cast<const f32x4*>(tid < NOB * 8 ? d.nn_gamma : d.nn_beta)[k];
  }

  // ---- prologue: operand fragments and accumulator initialisation --------------------------------------------
  bf16x8 bx[KS];
  f32x16 Y[NOB];
  if constexpr (LN) {
    static_assert(KS == 2 * NOB, "LayerNorm mode needs C == Cout");
    // b2 -> LDS: the epilogue adds it while storing, and a GLOBAL load between global stores is serialised by the compiler
    // (it cannot prove y and b2 do not alias: one L2 round trip per 64 bytes stored -- 48 of them, measured ~25 us per launch)
    if (tid < NOB * 8) reinterpret_cast<f32x4*>(smem + 3 * G::STAGE)[tid] = reinterpret_cast<const f32x4*>(d.b2)[tid];
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
        // The fragments take a detour through LDS (this wave's own 2 KB per 32-channel group, inside stage buffer 2,
        // which the weight stream does not touch before the first barrier of the main loop).  Values that vector
        // arithmetic produces start life in the ArchVGPR half of the register file; 96 long-lived fragment registers
        // made that way plus the 192 accumulators made hipcc carry 270-540 registers through scratch, whereas
        // fragments that come out of a LOAD (as in the bf16-operand mode) allocate cleanly (432 registers, no scratch).
        constexpr int GQ = (NOB % 2 == 0) ? NOB / 2 : 1;      // groups per round trip: 4 waves x GQ x 2 KB <= one stage buffer
        static_assert(4 * GQ * 2048 <= G::STAGE, "LayerNorm staging does not fit the idle stage buffer");
        bf16x8 t0, t1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { t0[j] = (bf16_t)xn[j]; t1[j] = (bf16_t)xn[8 + j]; }
        unsigned char* lp = smem + 2 * G::STAGE + wave * (GQ * 2048) + (q % GQ) * 2048 + lane * 32;
        *reinterpret_cast<bf16x8*>(lp) = t0;
        *reinterpret_cast<bf16x8*>(lp + 16) = t1;
        if (q % GQ == GQ - 1) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (also keeps the compiler from forwarding the stores)
#pragma unroll
          for (int qq = q - (GQ - 1); qq <= q; ++qq) {
            const unsigned char* rp = smem + 2 * G::STAGE + wave * (GQ * 2048) + (qq % GQ) * 2048 + lane * 32;
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
  // One wave per SIMD: nothing hides a latency unless the instruction stream does.  The NF = C/16 + 2 Cout/32 weight
  // fragments of a hidden block are read from LDS through a ring of PF registers sets, PF fragments ahead of the MFMA
  // that consumes them (left to itself hipcc emits read, read, lgkmcnt(0), mfma, mfma: every pair of MFMAs then waits
  // a full LDS round trip -- measured 4.4x the MFMA time), and the LDS-DMA pieces of the NEXT block are issued a few
  // MFMAs apart instead of as one burst at the top.  sched_barrier(0) after every step pins that order; the waitcnt
  // pass still emits counted lgkmcnt waits.
  // Every load the COMPILER knows of is complete before the loop: the accumulator blocks come straight from global loads,
  // and hipcc would otherwise place its waits for them (vmcnt(36) ... vmcnt(0)) inside the loop body, where they also drain
  // the LDS-DMA prefetches it cannot see (inline asm) -- every iteration (measured: 41 % of the wave cycles parked).
  __builtin_amdgcn_s_waitcnt(vm(0));
  const unsigned smem_lds = __builtin_amdgcn_readfirstlane(lds_offset(smem));
  constexpr int NF = KS + 2 * NOB;
  constexpr int PF = NF < 8 ? NF : 8;
  constexpr int P = KS + 2 * NOB;                     // 1 KB pieces per block (the b1 piece is extra)
  constexpr int NPW = (P + 3) / 4;                    // pieces per wave
  auto frag_off = [](int f) { return f < KS ? f * 1024 : G::W1B + (((f - KS) % NOB) * 2 + (f - KS) / NOB) * 1024; };
  // Software pipeline over the hidden blocks (the host image is packed for it: block j = [W1(j) | W2(j-1) | b1(j)], j = 0..NH,
  // W2(-1) = W1(NH) = 0): iteration j multiplies phase A of block j, then phase B of block j-1 -- whose B operand, the
  // activations of block j-1, was finished in iteration j-1 -- and slices the bias + activation + bf16 conversion of block j
  // between those phase-B MFMAs.  The activation's ~200 VALU instructions then issue in the shadow of MFMAs that do not
  // depend on them instead of stalling the matrix pipe between phase A and phase B (ablation: 29 of 127 us per launch).
  int cur = 0;                                // stage buffer of block j (j % 3, kept without a division)
  bf16x8 hp0, hp1;                            // activations of the previous hidden block (phase B operand)
#pragma unroll
  for (int j8 = 0; j8 < 8; ++j8) { hp0[j8] = (bf16_t)0.f; hp1[j8] = (bf16_t)0.f; }
  for (int hb = 0; hb <= NH; ++hb) {
    // this wave's pieces of block hb have landed (in-order return: exactly the pieces of block hb + 1 may stay in flight) ...
    if constexpr (ABL != 2) {
      if (wave == 0) __builtin_amdgcn_s_waitcnt(vm(NPW + 1));
      else __builtin_amdgcn_s_waitcnt(vm(NPW));
    }
    if constexpr (ABL != 5) __builtin_amdgcn_s_barrier();   // ... everybody's have, and everybody is done reading the buffer refilled next
    // (the image carries TWO blocks of padding behind block NH: the prefetch of block hb + 2 needs no branch -- a branch
    //  per piece splits the loop body into basic blocks and costs a full lgkmcnt(0) drain at every join)
    const int nxt = cur == 0 ? 2 : cur - 1;   // (hb + 2) % 3
    const unsigned char* nsrc = wsrc + (long)(hb + 2) * G::STAGE;
    const unsigned ndst_lds = smem_lds + nxt * G::STAGE;
    const unsigned char* ws = smem + cur * G::STAGE + lane * 16;
    const float* b1s = reinterpret_cast<const float*>(smem + cur * G::STAGE + G::W1B + G::W2B) + 16 * hi;
    cur = cur == 2 ? 0 : cur + 1;
    bf16x8 ring[PF];
#pragma unroll
    for (int f = 0; f < PF; ++f) ring[f] = *reinterpret_cast<const bf16x8*>(ws + frag_off(f));
    // b1 of this block is the C operand of the first phase-A MFMA (the four 16-byte reads land in the tuple the matrix
    // instruction reads: no accumulator initialisation, no bias add); the second chain starts from the zero literal
    f32x16 Bv;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 q4 = *reinterpret_cast<const f32x4*>(b1s + 4 * g);
      Bv[4 * g] = q4[0]; Bv[4 * g + 1] = q4[1]; Bv[4 * g + 2] = q4[2]; Bv[4 * g + 3] = q4[3];
    }
    const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 D0, D1;
    __builtin_amdgcn_sched_barrier(0);
    constexpr int STEP = NF / NPW;          // a DMA piece of block hb + 2 every STEP fragments
    auto dma = [&](int f) {
      if constexpr (ABL == 2) return;
      if (f % STEP == 0 && f / STEP < NPW) {
        const int pc = 4 * (f / STEP) + wave;
        const int pq = (P % 4 == 0 || pc < P) ? pc : P - 1;      // every wave issues NPW pieces (the counted wait above relies on it)
        dma16_asm(nsrc + pq * 1024 + lane * 16, ndst_lds + pq * 1024);
      }
      if (f == NF - 1 && wave == 0) dma4_asm(nsrc + P * 1024 + lane * 4, ndst_lds + P * 1024);   // b1 block: 64 lanes x 4 bytes
    };
    // phase A of block hb: D[32 hidden units][32 rows]
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 afrag = ring[ks % PF];
      if (ABL != 6 && ks + PF < NF) ring[ks % PF] = *reinterpret_cast<const bf16x8*>(ws + frag_off(ks + PF));
      if constexpr (ABL != 4) {
        if (!ONE_D && (ks & 1)) D1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bx[ks], ks == 1 ? kZero16 : D1, 0, 0, 0);
        else D0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bx[ks], ks == 0 ? Bv : D0, 0, 0, 0);
      } else if (ks < 2) {
        if (ks & 1) D1 = kZero16; else D0 = Bv;
      }
      dma(ks);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the phase-A accumulator is read by vector arithmetic only: named as an ArchVGPR operand here, hipcc allocates it there
    //  for its whole life instead of borrowing the AGPR range of a Y block and moving that block out and back every iteration)
    if constexpr (PIN) asm volatile("" : "+v"(D0));
    // phase B of block hb - 1 (Y[ob] += W2blk[ob] . H(hb-1)), the activation of block hb in its shadow.
    // The activation is PINNED into the phase-B slots: its results are only consumed at the bottom of the loop, so without a
    // tie hipcc sinks the whole computation below the last MFMA (round-3 ISA: 200 VALU instructions per block in the loop
    // latch, none in the MFMA shadow).  GELU of the 16 hidden units of a lane = 8 pairs (packed fp32 math) x 3 stages of
    // 5-7 issue slots; one stage per MFMA gap at NOB = 12 (a gap hides about 5 single-issue instructions,
    // MI355X_MICROARCH.md), each stage ending in an empty asm that names its outputs.
    constexpr int UNITS = 24 / (2 * NOB);      // stage units per phase-B MFMA (NOB = 12 / 6 / 3 -> 1 / 2 / 4)
    static_assert(UNITS * 2 * NOB == 24, "8 pairs x 3 stages over the phase-B slots");
    f32x2 gx[8], gax[8], gdn[8], gxx[8], gt[8], ge[8];
    unsigned hn[8];                            // activations of block hb as packed bf16 pairs: pair q = hidden units 2q, 2q + 1 of the lane
    auto act_unit = [&](int u) {
      const int q = u / 3, st = u % 3;
      if constexpr (ACT == PV_ACT_GELU && ABL != 1) {
        if (st == 0) {
          gx[q] = f32x2{D0[2 * q], D0[2 * q + 1]};
          if constexpr (!ONE_D) gx[q] += f32x2{D1[2 * q], D1[2 * q + 1]};
          gax[q][0] = fabsf(gx[q][0]);
          gax[q][1] = fabsf(gx[q][1]);
          gdn[q] = (gax[q] * 0.70710678118654752440f) * 0.47047f + 1.0f;
          gxx[q] = gx[q] * gx[q];
          if constexpr (PIN) asm volatile("" : "+v"(gx[q]), "+v"(gax[q]), "+v"(gdn[q]), "+v"(gxx[q]));
        } else if (st == 1) {
          gt[q][0] = __builtin_amdgcn_rcpf(gdn[q][0]);
          gt[q][1] = __builtin_amdgcn_rcpf(gdn[q][1]);
          const f32x2 a = gxx[q] * -0.72134752044448170368f;
          ge[q][0] = __builtin_amdgcn_exp2f(a[0]);
          ge[q][1] = __builtin_amdgcn_exp2f(a[1]);
          if constexpr (PIN) asm volatile("" : "+v"(gt[q]), "+v"(ge[q]));
        } else {
          const f32x2 t = gt[q];
          const f32x2 poly = t * (0.3480242f + t * (-0.0958798f + t * 0.7478556f));
          const f32x2 gq = (gx[q] + gax[q] * (1.0f - poly * ge[q])) * 0.5f;
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          const bf16x2_t pk = {(bf16_t)gq[0], (bf16_t)gq[1]};
          hn[q] = __builtin_bit_cast(unsigned, pk);
          if constexpr (PIN) asm volatile("" : "+v"(hn[q]));
        }
      } else {
        if (st == 0) {
          f32x2 v = f32x2{D0[2 * q], D0[2 * q + 1]};
          if constexpr (!ONE_D) v += f32x2{D1[2 * q], D1[2 * q + 1]};
          float r0 = v[0], r1 = v[1];
          if constexpr (ABL == 1 || ACT == PV_ACT_NONE) {}
          else if constexpr (ACT == PV_ACT_RELU) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); }
          else { r0 = pv_apply_act(r0, d.act); r1 = pv_apply_act(r1, d.act); }
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          const bf16x2_t pk = {(bf16_t)r0, (bf16_t)r1};
          hn[q] = __builtin_bit_cast(unsigned, pk);
          asm volatile("" : "+v"(hn[q]));
        }
      }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        const int t = i * NOB + ob;
        const int f = KS + t;
        const bf16x8 afrag = ring[f % PF];
        if (ABL != 6 && f + PF < NF) ring[f % PF] = *reinterpret_cast<const bf16x8*>(ws + frag_off(f + PF));
        if constexpr (ABL != 3) Y[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, i == 0 ? hp0 : hp1, Y[ob], 0, 0, 0);
        else asm volatile("" :: "v"(afrag), "v"(hp0), "v"(hp1));
#pragma unroll
        for (int u = t * UNITS; u < (t + 1) * UNITS; ++u) act_unit(u);
        dma(f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      hp0 = __builtin_bit_cast(bf16x8, u32x4_t{hn[0], hn[1], hn[2], hn[3]});
      hp1 = __builtin_bit_cast(bf16x8, u32x4_t{hn[4], hn[5], hn[6], hn[7]});
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: 16 consecutive channels per lane and output block ---------------------------------------------
  __builtin_amdgcn_s_waitcnt(vm(0));        // the padding block's LDS-DMA must not outlive the workgroup
  if (ok) {
    float* yr = static_cast<float*>(d.y) + m * d.ldy + 16 * hi;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      f32x4* p4 = reinterpret_cast<f32x4*>(yr + 32 * ob);
      if constexpr (LN) {
        const f32x4* c4 = reinterpret_cast<const f32x4*>(smem + 3 * G::STAGE) + 8 * ob + 4 * hi;
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
  // ---- norm1 of the NEXT MultiScaleBlock (layers/attention.py:729-737) on the rows this wave still holds: the bf16 GEMM
  //      operand the next block's q|k|v projection reads, instead of a LayerNorm launch that reads the stream back (round 4).
  //      A row's Cout values sit in two lanes (l, l + 32): in-lane sums + one exchange; two passes (mean, then centred squares).
  if (d.yn != nullptr) {      // wave-uniform
    float s1 = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
      if constexpr (LN) {
        const f32x4* c4 = reinterpret_cast<const f32x4*>(smem + 3 * G::STAGE) + 8 * ob + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 cc = c4[g];
#pragma unroll
          for (int e = 0; e < 4; ++e) Y[ob][4 * g + e] += cc[e];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s1 += Y[ob][r];
    }
    s1 += __shfl_xor(s1, 32, 64);
    const float inv_c = 1.0f / (float)(32 * NOB);
    const float mean = s1 * inv_c;
    float s2 = 0.f;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float t = Y[ob][r] - mean; s2 += t * t; }
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 * inv_c + d.nn_eps);
    if (ok) {
      bf16_t* nr = static_cast<bf16_t*>(d.yn) + m * d.ldyn + 16 * hi;
      const f32x4* gam = reinterpret_cast<const f32x4*>(smem + 3 * G::STAGE + NOB * 128);
      const f32x4* bet = gam + NOB * 8;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        bf16x8 o0, o1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 gg = gam[8 * ob + 4 * hi + g], bb = bet[8 * ob + 4 * hi + g];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = (Y[ob][4 * g + e] - mean) * rstd * gg[e] + bb[e];
            if (g < 2) o0[4 * g + e] = (bf16_t)v; else o1[4 * (g - 2) + e] = (bf16_t)v;
          }
        }
        *reinterpret_cast<bf16x8*>(nr + 32 * ob) = o0;
        *reinterpret_cast<bf16x8*>(nr + 32 * ob + 8) = o1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the same fused MLP on 16-ROW wave tiles -- TWO waves per SIMD (d.layout == PV_MLP_LAYOUT_ROWS16).
//
// The 32-row kernel above is bound by instruction ISSUE, not by the matrix pipe (PMC, round 5: MFMA busy 22 %, issue-stalled
// 36 %, parked 26 %): a wave owns 32 rows x 384 channels of fp32 accumulators (192 registers) + 96 registers of operand
// fragments, so only ONE wave fits a SIMD and nothing covers its ~9 non-MFMA instructions per MFMA, its LDS round trips or
// its barrier waits.  Here a wave owns 16 rows on v_mfma_f32_16x16x32_bf16: Y[16 rows][Cout] is Cout/16 x 4 = 96
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
// LDS image of hidden block j (host: emit_mvit.pack_mlp_weights(layout=16); same size as the 32-row image):
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
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE];      // three stage buffers: see mlp_rows_kernel
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

  // ---- LayerNorm of this lane's half row -> MFMA B fragments (see mlp_rows_kernel for the two passes and the LDS detour)
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
  constexpr int PF = KS < 8 ? KS : 8;             // fragment ring (see mlp_rows_kernel)
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

template <int KS, int NOB, int MINW, int ACT> int launch_act(const pv_mlp_desc& d, hipStream_t s) {
  const unsigned grid = (unsigned)pv_ceil_div(d.M, 128);
#ifdef PV_DEV_ABLATION   // ablation / A-B builds (1-6 give WRONG results): development variant of the library only (csrc/build.py --variant dev)
  if constexpr (KS == 24 && NOB == 12 && ACT == PV_ACT_GELU) {
    const int abl = pv_tune("mlp_abl", 0);      // tools/bench_mlp.py only: timing builds with wrong results
    if (abl && d.ln_gamma != nullptr) {
      switch (abl) {
        case 1: PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT, 1>), dim3(grid), dim3(256), 0, s, d); break;
        case 2: PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT, 2>), dim3(grid), dim3(256), 0, s, d); break;
        case 3: PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT, 3>), dim3(grid), dim3(256), 0, s, d); break;
        case 4: PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT, 4>), dim3(grid), dim3(256), 0, s, d); break;
        case 5: PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT, 5>), dim3(grid), dim3(256), 0, s, d); break;
        case 7: PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT, 7>), dim3(grid), dim3(256), 0, s, d); break;
        case 8: PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT, 8>), dim3(grid), dim3(256), 0, s, d); break;
        default: PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT, 6>), dim3(grid), dim3(256), 0, s, d); break;
      }
      PV_LAUNCH_CHECK();
      return PV_OK;
    }
  }
#endif
  if (d.ln_gamma != nullptr) {
    if constexpr (KS == 2 * NOB) PV_LAUNCH((mlp_rows_kernel<KS, NOB, true, MINW, ACT>), dim3(grid), dim3(256), 0, s, d);
    else return PV_ERR_UNSUPPORTED;
  } else {
    PV_LAUNCH((mlp_rows_kernel<KS, NOB, false, MINW, ACT>), dim3(grid), dim3(256), 0, s, d);
  }
  PV_LAUNCH_CHECK();
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

template <int KS, int NOB, int MINW> int launch(const pv_mlp_desc& d, hipStream_t s) {
  if (d.act == PV_ACT_GELU) return launch_act<KS, NOB, MINW, PV_ACT_GELU>(d, s);
  return launch_act<KS, NOB, MINW, -1>(d, s);
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
  if (d.layout != PV_MLP_LAYOUT_ROWS32 && d.layout != PV_MLP_LAYOUT_ROWS16) return PV_ERR_INVALID;
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
  if (d.layout == PV_MLP_LAYOUT_ROWS16) {
    if (d.C == 96 && d.Cout == 96) return launch16<3, 6>(d, s);
    if (d.C == 96 && d.Cout == 192) return launch16<3, 12>(d, s);
    if (d.C == 192 && d.Cout == 192) return launch16<6, 12>(d, s);
    if (d.C == 192 && d.Cout == 384) return launch16<6, 24>(d, s);
    if (d.C == 384 && d.Cout == 384) return launch16<12, 24>(d, s);
    return PV_ERR_UNSUPPORTED;
  }
  if (d.C == 96 && d.Cout == 96) return launch<6, 3, 2>(d, s);
  // two workgroups per CU where registers (<= 256 per wave) and LDS (3 stages each) allow: the narrow variants spend as long
  // loading the residual rows and storing the result as multiplying, and a lone workgroup overlaps neither with anything
  if (d.C == 96 && d.Cout == 192) return pv_tune("mlp_minw", 2) >= 2 ? launch<6, 6, 2>(d, s) : launch<6, 6, 1>(d, s);
  if (d.C == 192 && d.Cout == 192) return pv_tune("mlp_minw", 2) >= 2 ? launch<12, 6, 2>(d, s) : launch<12, 6, 1>(d, s);
  if (d.C == 192 && d.Cout == 384) return launch<12, 12, 1>(d, s);
  if (d.C == 384 && d.Cout == 384) return launch<24, 12, 1>(d, s);
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
