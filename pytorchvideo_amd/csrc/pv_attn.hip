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
#include "pv_common.h"

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

template <typename T, int D> int launch_attn(const pv_attention_desc& d, hipStream_t s) {
  const int nqb = (d.Nq + kQB - 1) / kQB;
  const long total = (long)d.B * d.heads * nqb;
  if (total <= 0 || total > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((attn_kernel<T, D>), dim3((unsigned)total), dim3(kThreads), 0, s, d, nqb, (int)total);
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
  if (d.dtype == PV_BF16) return launch_attn_d<bf16_t>(d, s);
  if (d.dtype == PV_F32) return launch_attn_d<float>(d, s);
  return PV_ERR_UNSUPPORTED;
}
