// Fused pooled attention for MViT (gfx950):  o = softmax((q*scale) k^T) v [+ q]
// (reference: pytorchvideo/layers/attention.py:531-539) without materialising the scores.
//
// Tokens are channels-last: row n of batch item b holds heads*D channels, head h at channel
// offset h*D, so the reference's permute/reshape between (B,N,h*D) and (B,h,N,D) is free.
//
// Work decomposition: one workgroup = 4 waves = 128 query rows of one (batch, head); each
// wave owns 32 query rows for the whole key loop.  K/V tiles of KT keys are staged through
// LDS once per workgroup and shared by the four waves.
//
// Matrix-core mapping (v_mfma_f32_32x32x16_bf16, "swapped" product so that all softmax state
// is lane-private):
//   S^T[key][q] = sum_d K[key][d] * Q[q][d]      A = K rows (LDS), B = Q rows (registers)
//     -> lane (q = lane&31, hi = lane>>5) holds 16 keys of ONE query: row max / row sum are
//        in-lane reductions plus a single exchange with lane^32.
//   O^T[d][q]   = sum_k V[k][d] * P[q][k]        A = V^T rows (LDS), B = P (registers)
//     -> the B operand wants, for lane (q,hi), eight k's per 16-key slot; the accumulator of
//        the first product already holds keys {4hi..4hi+3, 8+4hi..8+4hi+3} (+16 per slot).
//        Since a contraction may enumerate k in any order as long as A and B agree, V^T is
//        simply read in that same order (two 8-byte LDS reads) and P never moves between
//        lanes.  O^T keeps one query per lane, so the online-softmax rescale is a per-lane
//        scalar multiply.
// V is transposed while it is staged (4 keys x 8 channels per thread -> eight 8-byte LDS
// stores); K rows are padded to 16*odd+... bytes so that ds_read_b128 is bank-conflict free.
// fp32 (parity mode) uses v_mfma_f32_32x32x2_f32 with the same lane-private structure.
#include <math.h>
#include <type_traits>
#include "pv_common.h"

int pv_attn_w64_try(const pv_attention_desc& d, hipStream_t s);   // pv_attn64.hip; PV_ERR_UNSUPPORTED = not its case

namespace {

constexpr int kThreads = 256;
constexpr int kQB = 128;  // query rows per workgroup

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <typename T, int D> struct AttnCfg;
template <int D> struct AttnCfg<bf16_t, D> {
  static constexpr int KT = 64;          // keys per tile
  static constexpr int NBUF = 2;
  static constexpr int KLD = D + 8;      // K row stride (elements): 16B * odd -> conflict-free b128
  static constexpr int VLD = KT + 4;     // V^T row stride (elements): 34 dwords
  static constexpr int K_ELEMS = KT * KLD;
  static constexpr int V_ELEMS = D * VLD;
};
template <int D> struct AttnCfg<float, D> {
  static constexpr int KT = 32;
  static constexpr int NBUF = 1;
  static constexpr int KLD = D + 1;      // odd dword stride: column reads conflict-free
  static constexpr int VLD = D;          // row-major V, lanes run over d
  static constexpr int K_ELEMS = KT * KLD;
  static constexpr int V_ELEMS = KT * VLD;
};

template <typename T, int D>
__global__ __launch_bounds__(kThreads, 2) void attn_kernel(const pv_attention_desc d, int nqb, int total) {
  using Cfg = AttnCfg<T, D>;
  constexpr bool kBf16 = sizeof(T) == 2;
  constexpr int KT = Cfg::KT, NBUF = Cfg::NBUF, KLD = Cfg::KLD, VLD = Cfg::VLD;
  constexpr int NSUB = KT / 32;   // 32-key sub-tiles per tile
  constexpr int NDB = D / 32;     // 32-channel output blocks
  __shared__ __attribute__((aligned(16))) T smem[NBUF * (Cfg::K_ELEMS + Cfg::V_ELEMS)];

  // XCD-aware order: consecutive work items (same batch/head -> same K/V) share an XCD's L2
  int w;
  {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int qn = total >> 3, rn = total & 7;
    w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  }
  const int bh = w / nqb;
  const int qb = w - bh * nqb;
  const int b = bh / d.heads;
  const int h = bh - b * d.heads;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  const T* __restrict__ Q = static_cast<const T*>(d.q) + (long)b * d.q_bs + h * D;
  const T* __restrict__ K = static_cast<const T*>(d.k) + (long)b * d.k_bs + h * D;
  const T* __restrict__ V = static_cast<const T*>(d.v) + (long)b * d.v_bs + h * D;
  T* __restrict__ O = static_cast<T*>(d.o) + (long)b * d.o_bs + h * D;

  const int q_row = qb * kQB + wave * 32 + l31;
  const bool q_ok = q_row < d.Nq;
  const int ntiles = (d.Nk + KT - 1) / KT;
  const float sc = d.scale * 1.44269504088896340736f;  // softmax in the exp2 domain

  // ---- Q fragments (B operand), resident in registers for the whole key loop ----
  constexpr int NQF = kBf16 ? D / 16 : D / 2;
  bf16x8 qf[kBf16 ? NQF : 1];
  float qs[kBf16 ? 1 : NQF];
  if constexpr (kBf16) {
#pragma unroll
    for (int ks = 0; ks < NQF; ++ks) {
      if (q_ok) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (long)q_row * d.ldq + ks * 16 + hi * 8);
      else qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  } else {
#pragma unroll
    for (int s = 0; s < NQF; ++s) qs[s] = q_ok ? (float)Q[(long)q_row * d.ldq + 2 * s + hi] : 0.f;
  }

  // ---- staging registers ----
  constexpr int KCH = kBf16 ? (KT * D / 8 + kThreads - 1) / kThreads : (KT * D / 4) / kThreads;
  bf16x8 kreg_h[kBf16 ? KCH : 1];
  bf16x8 vreg_h[kBf16 ? 4 : 1];
  f32x4 kreg_f[kBf16 ? 1 : KCH];
  f32x4 vreg_f[kBf16 ? 1 : KCH];

  auto load_tile = [&](int t) {
    const int key0 = t * KT;
    if constexpr (kBf16) {
      constexpr int CPR = D / 8;  // chunks per row
#pragma unroll
      for (int i = 0; i < KCH; ++i) {
        const int c = tid + i * kThreads;
        const int row = c / CPR, col = c - row * CPR;
        const int key = key0 + row;
        if (c < KT * CPR && key < d.Nk) kreg_h[i] = *reinterpret_cast<const bf16x8*>(K + (long)key * d.ldk + col * 8);
        else kreg_h[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
      // V: thread = (key group of 4, channel chunk of 8)
      const int kg = tid & 15, dc = tid >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = key0 + kg * 4 + i;
        if (dc < CPR && key < d.Nk) vreg_h[i] = *reinterpret_cast<const bf16x8*>(V + (long)key * d.ldv + dc * 8);
        else vreg_h[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    } else {
      constexpr int CPR = D / 4;
#pragma unroll
      for (int i = 0; i < KCH; ++i) {
        const int c = tid + i * kThreads;
        const int row = c / CPR, col = c - row * CPR;
        const int key = key0 + row;
        if (key < d.Nk) {
          kreg_f[i] = *reinterpret_cast<const f32x4*>(K + (long)key * d.ldk + col * 4);
          vreg_f[i] = *reinterpret_cast<const f32x4*>(V + (long)key * d.ldv + col * 4);
        } else {
          kreg_f[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          vreg_f[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };

  auto store_tile = [&](int buf) {
    T* ks_ = smem + buf * (Cfg::K_ELEMS + Cfg::V_ELEMS);
    T* vs_ = ks_ + Cfg::K_ELEMS;
    if constexpr (kBf16) {
      constexpr int CPR = D / 8;
#pragma unroll
      for (int i = 0; i < KCH; ++i) {
        const int c = tid + i * kThreads;
        const int row = c / CPR, col = c - row * CPR;
        if (c < KT * CPR) *reinterpret_cast<bf16x8*>(ks_ + row * KLD + col * 8) = kreg_h[i];
      }
      const int kg = tid & 15, dc = tid >> 4;
      if (dc < CPR) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16x4 t4 = {vreg_h[0][j], vreg_h[1][j], vreg_h[2][j], vreg_h[3][j]};
          *reinterpret_cast<bf16x4*>(vs_ + (dc * 8 + j) * VLD + kg * 4) = t4;
        }
      }
    } else {
      constexpr int CPR = D / 4;
#pragma unroll
      for (int i = 0; i < KCH; ++i) {
        const int c = tid + i * kThreads;
        const int row = c / CPR, col = c - row * CPR;
#pragma unroll
        for (int j = 0; j < 4; ++j) ks_[row * KLD + col * 4 + j] = kreg_f[i][j];
        *reinterpret_cast<f32x4*>(vs_ + row * VLD + col * 4) = vreg_f[i];
      }
    }
  };

  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  load_tile(0);
  for (int t = 0; t < ntiles; ++t) {
    const int buf = (NBUF == 2) ? (t & 1) : 0;
    if (NBUF == 1 && t > 0) __syncthreads();  // everyone is done reading the single buffer
    store_tile(buf);
    __syncthreads();
    if (t + 1 < ntiles) load_tile(t + 1);

    const T* ks_ = smem + buf * (Cfg::K_ELEMS + Cfg::V_ELEMS);
    const T* vs_ = ks_ + Cfg::K_ELEMS;

    // ---- S^T = K Q^T ----  (K fragments are read two k-steps ahead of the MFMAs that use them)
    f32x16 s[NSUB];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
    if constexpr (kBf16) {
      constexpr int NKS = D / 16;
      bf16x8 kf[3][NSUB];
      auto read_k = [&](int slot, int ks) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
          kf[slot][sub] = *reinterpret_cast<const bf16x8*>(ks_ + (sub * 32 + l31) * KLD + ks * 16 + hi * 8);
      };
      read_k(0, 0);
      if (NKS > 1) read_k(1, 1);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 2 < NKS) read_k((ks + 2) % 3, ks + 2);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks % 3][sub], qf[ks], s[sub], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
        for (int ss = 0; ss < D / 2; ++ss) {
          const float a = (float)ks_[(sub * 32 + l31) * KLD + 2 * ss + hi];
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qs[ss], s[sub], 0, 0, 0);
        }
      }
    }

    // ---- online softmax (lane-private: this lane's query, 16*NSUB of the tile's keys) ----
    // exp2 domain; raw v_exp_f32 (arguments are <= kDefer, results never need denormal scaling).
    // Deferred rescale: the running max is only advanced (and O, l rescaled) when some lane's tile
    // max exceeds it by more than kDefer -- otherwise P = exp2(x - m_stale) <= 2^kDefer is still
    // exact to bf16 precision and the O accumulators never leave the matrix-core register file.
    constexpr float kDefer = 8.0f;
    if ((t + 1) * KT > d.Nk) {   // last tile: mask the keys past Nk (wave-uniform branch)
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((t * KT + sub * 32 + crow(r, hi)) >= d.Nk) s[sub][r] = -INFINITY;
    }
    float mx = -1e30f;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
    mx *= sc;   // sc > 0: max commutes with the scaling
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (__any(mx > m_run + kDefer)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    float psum = 0.f;
    const float neg_m = -m_run;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[sub][r], sc, neg_m));
        s[sub][r] = p;
        psum += p;
      }
    l_run += psum;

    // ---- O^T += V^T P^T ----  (P -> bf16 first, which frees the score registers; V^T fragments are
    //      read two MFMAs ahead)
    if constexpr (kBf16) {
      bf16x8 pb[NSUB * 2];
#pragma unroll
      for (int i = 0; i < NSUB * 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[i][j] = (bf16_t)s[i >> 1][(i & 1) * 8 + j];
      constexpr int NF = NSUB * 2 * NDB;   // fragment sequence: i = (sub*2 + k2) * NDB + db
      bf16x8 vf[3];
      auto read_v = [&](int slot, int i) {
        const int sk = i / NDB, db = i - sk * NDB;
        const T* vp = vs_ + (db * 32 + l31) * VLD + sk * 16 + hi * 4;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vp);
        const bf16x4 up = *reinterpret_cast<const bf16x4*>(vp + 8);
        vf[slot] = bf16x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
      };
      read_v(0, 0);
      read_v(1, 1);
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        if (i + 2 < NF) read_v((i + 2) % 3, i + 2);
        const int sk = i / NDB, db = i - sk * NDB;
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % 3], pb[sk], o[db], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = sub * 32 + crow(r, hi);
#pragma unroll
          for (int db = 0; db < NDB; ++db) {
            const float a = (float)vs_[key * VLD + db * 32 + l31];
            o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[sub][r], o[db], 0, 0, 0);
          }
        }
    }
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l (+ q) ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    T* orow = O + (long)q_row * d.ldo;
    const T* qrow = Q + (long)q_row * d.ldq;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = o[db][g * 4 + j] * inv;
        if (d.residual_q) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)qrow[d0 + j];
        }
        if constexpr (kBf16) {
          bf16x4 ov = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>(orow + d0) = ov;
        } else {
          *reinterpret_cast<f32x4*>(orow + d0) = f32x4{v[0], v[1], v[2], v[3]};
        }
      }
  }
}

// bf16, software-pipelined over key tiles: the S^T = K Q^T MFMAs of tile t+1 are issued in the same instruction
// stream as the exponentials of tile t (two score register sets, sA / sB, swapped by unrolling the tile loop twice),
// so the matrix pipe works while the VALU / transcendental unit does the softmax -- a wave is in-order, an MFMA only
// overlaps VALU work that follows it in program order and does not depend on it.  K and V live in separate double
// buffers because their lifetimes now differ by half a step: K(t+1) is written before the tile's one barrier and
// read right after it, V(t+1) is written after the barrier (PV(t-1) was the last reader of that buffer) and read
// one barrier later.  The lane<->lane+32 max exchange is a v_permlane32_swap (VALU) instead of an LDS bpermute.
template <int D>
__global__ __launch_bounds__(kThreads, 2) void attn_pipe_kernel(const pv_attention_desc d, int nqb, int total) {
  using T = bf16_t;
  using Cfg = AttnCfg<bf16_t, D>;
  constexpr int KT = Cfg::KT, KLD = Cfg::KLD, VLD = Cfg::VLD;
  constexpr int NSUB = KT / 32, NDB = D / 32, NKS = D / 16, CPR = D / 8;
  static_assert(NSUB == 2, "two 32-key sub-tiles per tile");
  __shared__ __attribute__((aligned(16))) T smem[2 * (Cfg::K_ELEMS + Cfg::V_ELEMS)];
  T* const kb0 = smem;
  T* const kb1 = smem + Cfg::K_ELEMS;
  T* const vb0 = smem + 2 * Cfg::K_ELEMS;
  T* const vb1 = vb0 + Cfg::V_ELEMS;

  int w;
  {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int qn = total >> 3, rn = total & 7;
    w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  }
  const int bh = w / nqb;
  const int qb = w - bh * nqb;
  const int b = bh / d.heads;
  const int h = bh - b * d.heads;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  const T* __restrict__ Q = static_cast<const T*>(d.q) + (long)b * d.q_bs + h * D;
  const T* __restrict__ K = static_cast<const T*>(d.k) + (long)b * d.k_bs + h * D;
  const T* __restrict__ V = static_cast<const T*>(d.v) + (long)b * d.v_bs + h * D;
  T* __restrict__ O = static_cast<T*>(d.o) + (long)b * d.o_bs + h * D;

  const int q_row = qb * kQB + wave * 32 + l31;
  const bool q_ok = q_row < d.Nq;
  const int ntiles = (d.Nk + KT - 1) / KT;
  const float sc = d.scale * 1.44269504088896340736f;

  bf16x8 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (q_ok) qf[ks] = *reinterpret_cast<const bf16x8*>(Q + (long)q_row * d.ldq + ks * 16 + hi * 8);
    else qf[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }

  // K / V rows through buffer resources: keys past Nk (and idle staging threads) read zeros by the range check,
  // no exec-mask branches around the loads (the tile offset goes into the VGPR offset: the SGPR offset of a raw
  // buffer access is not range-checked)
  constexpr int KCH = (KT * CPR + kThreads - 1) / kThreads;
  constexpr unsigned kOOB = 0x80000000u;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(K), 0, (int)(((long)(d.Nk - 1) * d.ldk + D) * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(V), 0, (int)(((long)(d.Nk - 1) * d.ldv + D) * 2), 0x00020000);
  unsigned koff[KCH], voff;
#pragma unroll
  for (int i = 0; i < KCH; ++i) {
    const int c = tid + i * kThreads;
    const int row = c / CPR, col = c - row * CPR;
    koff[i] = c < KT * CPR ? (unsigned)(row * d.ldk + col * 8) * 2u : kOOB;
  }
  {
    const int kg = tid & 15, dc = tid >> 4;
    voff = dc < CPR ? (unsigned)(kg * 4 * d.ldv + dc * 8) * 2u : kOOB;
  }
  const unsigned ktile_b = (unsigned)(KT * d.ldk) * 2u, vtile_b = (unsigned)(KT * d.ldv) * 2u, vrow_b = (unsigned)d.ldv * 2u;
  bf16x8 kreg[KCH], vreg[4];
  auto load_k = [&](int t) {
#pragma unroll
    for (int i = 0; i < KCH; ++i)
      kreg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rk, (int)(koff[i] + (unsigned)t * ktile_b), 0, 0));
  };
  auto load_v = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      vreg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, (int)(voff + (unsigned)i * vrow_b + (unsigned)t * vtile_b), 0, 0));
  };
  auto store_k = [&](T* ks_) {
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
      const int c = tid + i * kThreads;
      const int row = c / CPR, col = c - row * CPR;
      if ((KT * CPR) % kThreads == 0 || c < KT * CPR) *reinterpret_cast<bf16x8*>(ks_ + row * KLD + col * 8) = kreg[i];
    }
  };
  // staging threads without a channel chunk (tid/16 >= D/8) hold zeros and drop them into the 8-byte pad at the end of
  // a V^T row: no exec-mask branch, so the stores can sit between the PV MFMAs without splitting the block
  const int v_col = (tid >> 4) < CPR ? (tid & 15) * 4 : KT;
  const int v_dc = (tid >> 4) < CPR ? (tid >> 4) : CPR - 1;
  auto store_v_part = [&](T* vs_, int j0, int j1) {
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      bf16x4 t4 = {vreg[0][j], vreg[1][j], vreg[2][j], vreg[3][j]};
      *reinterpret_cast<bf16x4*>(vs_ + (v_dc * 8 + j) * VLD + v_col) = t4;
    }
  };
  auto store_v = [&](T* vs_) { store_v_part(vs_, 0, 8); };
  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  f32x16 sA[NSUB], sB[NSUB];
  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // one tile step: softmax + PV of tile t (scores in sc_), scores of tile t+1 into sn
  // One tile step: softmax + PV of tile t (scores in sc_); MORE: there is a tile t+1 and its scores are produced
  // here, into sn.  Between the rescale branch and the next barrier a step is straight-line code -- with a branch in
  // between, LLVM sinks the exponentials below it, next to their use in the PV product.  That is why the key mask of a
  // ragged last tile is applied by the step that CONSUMES the scores, before its rescale decision, not by the step that
  // produced them.
  float mx_next = -1e30f;   // scaled row max of the scores the NEXT step will consume (computed under this step's PV)
  auto row_max = [&](f32x16 (&sx)[NSUB]) {
    float mx = -1e30f;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sx[sub][r]);
    return mx;
  };
  auto step = [&](auto more_c, int t, f32x16 (&sc_)[NSUB], f32x16 (&sn)[NSUB]) {
    constexpr bool more = decltype(more_c)::value;
    T* const kn = ((t + 1) & 1) ? kb1 : kb0;
    T* const vn = ((t + 1) & 1) ? vb1 : vb0;
    const T* const vc = (t & 1) ? vb1 : vb0;
    if constexpr (more) {
      store_k(kn);                       // K(t+1): loaded one step ago
      load_k(t + 2);                     // ... and its registers go straight back to memory for K(t+2) (past the
                                         // last tile the range check returns zeros; nobody reads them)
    }
    __syncthreads();
    if ((t + 1) * KT > d.Nk) {   // ragged last tile: keys >= Nk out of the softmax (zero K rows scored 0), exact row max
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((t * KT + sub * 32 + crow(r, hi)) >= d.Nk) sc_[sub][r] = -1e30f;
      mx_next = row_max(sc_);
    }

    // ---- deferred rescale (see attn_kernel); the row max came out of the previous step's PV phase ----
    constexpr float kDefer = 8.0f;
    {
      float mx = mx_next * sc;
      const unsigned mu = __builtin_bit_cast(unsigned, mx);
      auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
      mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
      if (__any(mx > m_run + kDefer)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
    }
    const float neg_m = -m_run;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 sc2 = {sc, sc}, nm2 = {neg_m, neg_m};
    f32x2 ps = {0.f, 0.f};
    auto exp_range = [&](int v0, int v1) {   // even-aligned pairs: v_pk_fma_f32 / v_pk_add_f32 halve the plain VALU issue
#pragma unroll
      for (int v = v0; v < v1; v += 2) {
        f32x2 x = {sc_[v >> 4][v & 15], sc_[v >> 4][(v & 15) + 1]};
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(sc2), "v"(nm2));   // (hipcc scalarises the builtin)
        x[0] = __builtin_amdgcn_exp2f(x[0]);
        x[1] = __builtin_amdgcn_exp2f(x[1]);
        sc_[v >> 4][v & 15] = x[0];
        sc_[v >> 4][(v & 15) + 1] = x[1];
        ps += x;
      }
    };
    if constexpr (more) {
      // ---- S^T(t+1) = K Q^T on the matrix pipe, exp2 of tile t on the VALU, interleaved in program order ----
      bf16x8 kf[2][NSUB];
      auto read_k = [&](int slot, int ks) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
          kf[slot][sub] = *reinterpret_cast<const bf16x8*>(kn + (sub * 32 + l31) * KLD + ks * 16 + hi * 8);
      };
      read_k(0, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 1 < NKS) read_k((ks + 1) & 1, ks + 1);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
          sn[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks & 1][sub], qf[ks], ks == 0 ? kZero16 : sn[sub], 0, 0, 0);
          const int i = ks * NSUB + sub;                       // NKS * NSUB slices of the 16 pairs of exponentials
          exp_range(2 * (i * 16 / (NKS * NSUB)), 2 * ((i + 1) * 16 / (NKS * NSUB)));
          __builtin_amdgcn_sched_barrier(0);                   // keep the slice behind ITS MFMA (hipcc would sink all
        }                                                      // 32 exponentials below the last MFMA otherwise)
      }
    } else {
      exp_range(0, 32);
    }
    l_run += ps[0] + ps[1];

    // ---- O^T += V^T P^T on the matrix pipe; under it, on the VALU / LDS: P -> bf16, V(t+1) transposed into its
    //      buffer (then V(t+2) requested), and the row max of the scores just produced ----
    bf16x8 pb[NSUB * 2];
    constexpr int NF = NSUB * 2 * NDB;
    bf16x8 vf[3];
    auto read_v = [&](int slot, int i) {
      const int sk = i / NDB, db = i - sk * NDB;
      const T* vp = vc + (db * 32 + l31) * VLD + sk * 16 + hi * 4;
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vp);
      const bf16x4 up = *reinterpret_cast<const bf16x4*>(vp + 8);
      vf[slot] = bf16x8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
    };
    read_v(0, 0);
    read_v(1, 1);
    float mxa = -1e30f, mxb = -1e30f;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (i + 2 < NF) read_v((i + 2) % 3, i + 2);
      const int sk = i / NDB, db = i - sk * NDB;
      if (db == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[sk][j] = (bf16_t)sc_[sk >> 1][(sk & 1) * 8 + j];
      }
      o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % 3], pb[sk], o[db], 0, 0, 0);
      if constexpr (more) {
        // the side work in 8 parts (one V^T row group + 4 score registers each) spread over iterations I0..L
        constexpr int I0 = NF >= 12 ? 2 : 0, L = NF >= 12 ? NF - 2 : NF - 1, CNT = L - I0 + 1;
        if (i >= I0 && i <= L) {
          const int j0 = (i - I0) * 8 / CNT, j1 = (i - I0 + 1) * 8 / CNT;
          store_v_part(vn, j0, j1);
#pragma unroll
          for (int r = j0 * 2; r < j1 * 2; ++r) {
            mxa = fmaxf(mxa, sn[0][r]);
            mxb = fmaxf(mxb, sn[1][r]);
          }
          asm volatile("" : "+v"(mxa), "+v"(mxb));   // pin the partial maxima to this slice (see below)
        }
        if (i == (L + 1 < NF ? L + 1 : NF - 1)) load_v(t + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    mx_next = fmaxf(mxa, mxb);
    asm volatile("" : "+v"(mx_next));   // pin: without a use HERE LLVM sinks the whole max chain below the next barrier
  };

  // prologue: tile 0 staged, tile 1 requested, the scores of tile 0 computed the plain way
  load_k(0);
  load_v(0);
  store_k(kb0);
  store_v(vb0);
  if (ntiles > 1) {
    load_k(1);
    load_v(1);
  }
  __syncthreads();
  {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const bf16x8 kfr = *reinterpret_cast<const bf16x8*>(kb0 + (sub * 32 + l31) * KLD + ks * 16 + hi * 8);
        sA[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[ks], ks == 0 ? kZero16 : sA[sub], 0, 0, 0);
      }
    mx_next = row_max(sA);   // (a ragged single tile is masked, and its max redone, at the top of its step)
  }
  {
    using Y = std::integral_constant<bool, true>;
    using N = std::integral_constant<bool, false>;
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
      step(Y{}, t, sA, sB);
      step(Y{}, t + 1, sB, sA);
    }
    if (ntiles - t == 2) {       // the current scores are in sA
      step(Y{}, t, sA, sB);
      step(N{}, t + 1, sB, sA);
    } else {
      step(N{}, t, sA, sB);
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    T* orow = O + (long)q_row * d.ldo;
    const T* qrow = Q + (long)q_row * d.ldq;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = o[db][g * 4 + j] * inv;
        if (d.residual_q) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)qrow[d0 + j];
        }
        bf16x4 ov = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(orow + d0) = ov;
      }
  }
}

template <typename T, int D> int launch_attn(const pv_attention_desc& d, hipStream_t s) {
  const int nqb = (d.Nq + kQB - 1) / kQB;
  const long total = (long)d.B * d.heads * nqb;
  if (total <= 0 || total > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  if constexpr (sizeof(T) == 2 && D <= 96) {   // (D = 128 would spill: two score sets + 64 O accumulators)
    const long kv_bytes = ((long)(d.Nk + 192) * (d.ldk > d.ldv ? d.ldk : d.ldv) + D) * 2;   // 32-bit offsets, 2 tiles past the end
    if (pv_tune("attn_pipe", 1) && kv_bytes < 0x7fffffffL) {
      PV_LAUNCH((attn_pipe_kernel<D>), dim3((unsigned)total), dim3(kThreads), 0, s, d, nqb, (int)total);
      PV_LAUNCH_CHECK();
      return PV_OK;
    }
  }
  PV_LAUNCH((attn_kernel<T, D>), dim3((unsigned)total), dim3(kThreads), 0, s, d, nqb, (int)total);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

template <typename T> int launch_attn_d(const pv_attention_desc& d, hipStream_t s) {
  switch (d.head_dim) {
    case 32: return launch_attn<T, 32>(d, s);
    case 64: return launch_attn<T, 64>(d, s);
    case 96: return launch_attn<T, 96>(d, s);
    case 128: return launch_attn<T, 128>(d, s);
    default: return PV_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int pv_attention(const pv_attention_desc* dp, pv_stream_t stream) {
  if (!dp) return PV_ERR_INVALID;
  const pv_attention_desc& d = *dp;
  if (!d.q || !d.k || !d.v || !d.o) return PV_ERR_INVALID;
  if (d.B <= 0 || d.heads <= 0 || d.head_dim <= 0 || d.Nq <= 0 || d.Nk <= 0) return PV_ERR_INVALID;
  const int align = d.dtype == PV_BF16 ? 8 : 4;  // 16-byte vector accesses
  if (d.ldq % align || d.ldk % align || d.ldv % align || d.ldo % align) return PV_ERR_INVALID;
  if (d.q_bs % align || d.k_bs % align || d.v_bs % align || d.o_bs % align) return PV_ERR_INVALID;
  const int width = d.heads * d.head_dim;
  if (d.ldq < width || d.ldk < width || d.ldv < width || d.ldo < width) return PV_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  {   // bf16, head dim 96 (MViT): the one-wave-per-SIMD kernel of pv_attn64.hip
    const int r = pv_attn_w64_try(d, s);
    if (r != PV_ERR_UNSUPPORTED) return r;
  }
  if (d.dtype == PV_BF16) return launch_attn_d<bf16_t>(d, s);
  if (d.dtype == PV_F32) return launch_attn_d<float>(d, s);
  return PV_ERR_UNSUPPORTED;
}
