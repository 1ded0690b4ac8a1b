// Fused pooled attention for MViT, bf16, one wave per SIMD with the whole register file (gfx950):
//   o = softmax((q*scale) k^T) v [+ q]      (reference: pytorchvideo/layers/attention.py:531-539)
//
// Same mathematics and the same lane-private softmax as attn_pipe_kernel (pv_attn.hip): swapped products
// S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16_bf16, one query per lane, deferred rescale, exp2 domain.
// What differs is the mapping on the CU:
//   * a workgroup = 4 waves = 256 query rows of one (batch, head), ONE wave per SIMD, each wave owning 64 query rows
//     (two 32-row blocks): every K / V fragment read from LDS feeds two MFMAs, and the ~400 registers a wave may use
//     hold both blocks' O^T (96), two score sets (128), Q (48) and fragment rings read several MFMAs ahead;
//   * V is staged ROW-major ([4 keys][32 channels] blocks of 256 contiguous bytes) and read with
//     ds_read_b64_tr_b16, the LDS transpose read: no transposing stores, no v_perm;
//   * a tile step is 48 MFMAs in two phases -- S^T(t+1) beside the exponentials of tile t, then PV(t) beside the row
//     maxima of tile t+1 -- with ONE barrier between them; K lives in a 2-slot ring, V in a 3-slot ring, so that
//     fragments of the next phase can be requested before the barrier that ends this one;
//   * staging through registers (T14 of the guide: loads issued a whole step before their ds_write).
#include <type_traits>
#include "pv_common.h"

int pv_attn_w64_try(const pv_attention_desc& d, hipStream_t s);   // called by pv_attention (pv_attn.hip)

#ifdef PV_DEV_ABLATION
// dev library only: clock stamps of wave 0 of every workgroup (tools/r6/attn_stamps.py reads them through pv_dev_attn_stamps)
__device__ unsigned long long pv_attn_stamp_buf[8 * 8192];
#define PV_STAMP(i)                                                                                          \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 8192) pv_attn_stamp_buf[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
extern "C" int pv_dev_attn_stamps(unsigned long long* out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return PV_ERR_HIP;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pv_attn_stamp_buf), sizeof(unsigned long long) * (size_t)(n < 8 * 8192 ? n : 8 * 8192)) != hipSuccess) return PV_ERR_HIP;
  return PV_OK;
}
#else
#define PV_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int KT = 64;         // keys per tile

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// the exponential pipeline's batches (see exp_pipe): the 32*NQB elements per lane and tile over 19*NQB MFMA slices, in pairs
template <int NQB> __host__ __device__ constexpr int exp_bound(int g) {
  return g <= 0 ? 0 : (g >= 19 * NQB ? 32 * NQB : 2 * ((g * 16 * NQB) / (19 * NQB)));
}

// The S^T MFMAs, written out: with more than 256 registers per wave hipcc selects the accumulator-file form of every MFMA,
// and the 64 scores per lane and tile would each cost a v_accvgpr_read before the softmax can touch them (PMC, first
// version of this kernel: the VALU was busy 43 cycles per 32-cycle MFMA, a sixth of it those reads).  Here the
// destination is arithmetic registers ("v"), the resident Q fragment comes from the accumulator half ("a").  hipcc's
// hazard recogniser does not look inside inline assembly: the consumers of these results are placed 16+ MFMAs behind
// the last one in the tile loop, and the prologue waits explicitly.
// (AG = false, the 256-register forms: no accumulator registers at all -- one "a" constraint anywhere makes hipcc split
//  the wave's 256 registers 128 / 128 and spill the arithmetic half)
template <bool AG> __device__ __forceinline__ void mfma_s_first(f32x16& dst, const bf16x8& a, const bf16x8& b) {
  if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(dst) : "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(dst) : "v"(a), "v"(b));
}
template <bool AG> __device__ __forceinline__ void mfma_s_acc(f32x16& dst, const bf16x8& a, const bf16x8& b) {
  if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dst) : "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dst) : "v"(a), "v"(b));
}

// ABL (dev library only, -DPV_DEV_ABLATION; WRONG results, timing only): bit 0 no exponentials / sums / conversions,
// bit 1 no barrier inside the step, bit 2 no staging (loads and LDS stores), bit 3 no S^T MFMAs, bit 4 no PV MFMAs,
// bit 5 no fragment reads, bit 6 no row maxima
// NQB: 32-row query blocks per wave (2: the one-wave-per-SIMD form described above; 1: half the registers, two waves per
// SIMD -- every stall of one wave is covered by the other, at the price of reading each fragment for 32 rows only).
// NW: waves per workgroup (NQB = 1: 8 waves = one workgroup per CU sharing its tiles, the first four waves staging K and the
// other four V; or 4 waves = two workgroups per CU, each staging its own tiles, one's prologue / epilogue beside the
// other's tile loop).
template <int D, int NQB, int NW, int ABL = 0>
__global__ __launch_bounds__(64 * NW, NQB == 2 ? 1 : 2) void attn_w64_kernel(const pv_attention_desc d, int nqb, int total) {
  using T = bf16_t;
  constexpr int kThreads = 64 * NW;
  constexpr int kRowsWG = 32 * NQB * NW;
  constexpr bool SPLIT = NW == 8;                 // K staged by waves 0-3, V by waves 4-7
  constexpr bool AG = NQB == 2;                   // O^T and Q in the accumulator half of the register file
  constexpr int kExpSlices = 19 * NQB;
  constexpr int NE = 32 * NQB;                    // exponentials per lane and tile
  constexpr int KLD = D + 8;                      // K row stride (elements): 16 B * odd -> conflict-free ds_read_b128
  constexpr int NKS = D / 16, NDB = D / 32;
  constexpr int K_BYTES = KT * KLD * 2, V_BYTES = KT * D * 2;
  constexpr int NKF = 2 * NKS;                    // K fragments per tile (sub, ks)
  constexpr int NVF = 4 * NDB;                    // V fragments per tile (sk, db)
  constexpr int KPRE = NQB == 2 ? 5 : 4, KRING = KPRE + 1;   // K fragments requested before their phase / ring depth (the 256-register forms have no room for more)
  constexpr int VPRE = NQB == 2 ? 5 : 4, VRING = VPRE + 1;
  constexpr int KLEAD = 2 * NQB, VLEAD = 4 * NQB; // MFMA slices between the last pre-read and the end of its phase
  static_assert(NKF >= KPRE && NVF >= VPRE, "rings assume at least KPRE fragments per tile");
  __shared__ __attribute__((aligned(16))) char smem[2 * K_BYTES + 3 * V_BYTES];
  char* const ksm = smem;
  char* const vsm = smem + 2 * K_BYTES;

  PV_STAMP(0);
  int w;
  {   // XCD-aware order: consecutive work items (same batch / head -> same K / V) share an XCD's L2
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int qn = total >> 3, rn = total & 7;
    w = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
  }
  const int bh = w / nqb;
  const int qblk = w - bh * nqb;
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

  const int ntiles = (d.Nk + KT - 1) / KT;
  const float sc = d.scale * 1.44269504088896340736f;   // softmax in the exp2 domain

  // ---- Q fragments (B operands), resident for the whole key loop ----
  int q_row[NQB];
  bool q_ok[NQB];
  bf16x8 qf[NQB][NKS];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    q_row[qb] = qblk * kRowsWG + wave * (32 * NQB) + qb * 32 + l31;
    q_ok[qb] = q_row[qb] < d.Nq;
    const int r = q_ok[qb] ? q_row[qb] : d.Nq - 1;   // (rows past the end: a valid row, never stored)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(Q + (long)r * d.ldq + ks * 16 + hi * 8);
  }
  // (Q is only ever the "a" operand of mfma_s_*: it lives in the accumulator half of the file; the arithmetic
  //  registers hold the two score sets and the softmax)

  // ---- staging: thread = (key s_tid/4, 16-byte chunk s_tid%4 of each 32-channel block); keys past Nk read as zeros
  //      (zeros; their scores are masked in the last tile, their P is 0).  SPLIT: threads 0-255 stage K, 256-511 V. ----
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int s_tid = tid & 255;
  const bool role_v = SPLIT && __builtin_amdgcn_readfirstlane(tid) >= 256;   // wave-uniform, and known to be (a per-lane select of
                                                                              // the buffer resource puts every load into a waterfall loop)
  const int s_key = s_tid >> 2, s_cq = s_tid & 3;
  __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(K), 0, (int)(((long)(d.Nk - 1) * d.ldk + D) * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(V), 0, (int)(((long)(d.Nk - 1) * d.ldv + D) * 2), 0x00020000);
  // two register sets each (by step parity): a tile is requested TWO steps before its ds_write -- the workgroups of a
  // (batch, head) run in lockstep on one XCD and miss its L2 together, so the latency to cover is HBM's, not the L2's
  // (with one set, loads one step ahead, removing the staging from a timing build saved 15 % of the kernel)
  u32x4 kreg[2][NDB], vreg[SPLIT ? 1 : 2][NDB];     // SPLIT: kreg is the role's tile (K or V), vreg unused
  const int k_st = s_key * (KLD * 2) + s_cq * 16;                                  // + i * 64
  const int v_st = (((s_key >> 4) * 4 + ((s_key >> 2) & 3)) * NDB) * 256 + (s_key & 3) * 64 + s_cq * 16;   // + i * 256
  // (the tile offset goes into the VGPR offset, which the buffer's range check covers -- an SGPR offset is not checked:
  //  keys past Nk read zeros)
  const unsigned k_off0 = (unsigned)(s_key * d.ldk + s_cq * 8) * 2u, v_off0 = (unsigned)(s_key * d.ldv + s_cq * 8) * 2u;
  const unsigned k_tile_b = (unsigned)(KT * d.ldk) * 2u, v_tile_b = (unsigned)(KT * d.ldv) * 2u;
  auto load_from = [&](u32x4 (&dst)[NDB], __amdgpu_buffer_rsrc_t r, unsigned off0, unsigned tile_b, int t) __attribute__((always_inline)) {
    const unsigned off = off0 + (unsigned)t * tile_b;
#pragma unroll
    for (int i = 0; i < NDB; ++i) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + (unsigned)i * 64u), 0, 0);
  };
  // !SPLIT: load_k / load_v / store_k / store_v act on every thread.  SPLIT: load_kv(set, tk, tv) / store_kv(set, kslot,
  // vslot, i) act on the role's tile, chosen by wave-uniform selects (no branch inside the tile loop).
  auto load_k = [&](int set, int t) __attribute__((always_inline)) { load_from(kreg[set], rk, k_off0, k_tile_b, t); };
  auto load_v = [&](int set, int t) __attribute__((always_inline)) { load_from(vreg[SPLIT ? 0 : set], rv, v_off0, v_tile_b, t); };
  auto store_k = [&](int set, int slot, int i) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(ksm + slot * K_BYTES + k_st + i * 64) = kreg[set][i];
  };
  auto store_v = [&](int set, int slot, int i) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(vsm + slot * V_BYTES + v_st + i * 256) = vreg[SPLIT ? 0 : set][i];
  };
  const int st_sel = role_v ? v_st : k_st;
  const int stride_sel = role_v ? 256 : 64;
  auto load_kv = [&](int set, int tk, int tv) __attribute__((always_inline)) {
    load_from(kreg[set], role_v ? rv : rk, role_v ? v_off0 : k_off0, role_v ? v_tile_b : k_tile_b, role_v ? tv : tk);
  };
  auto store_kv = [&](int set, int kslot, int vslot, int i) __attribute__((always_inline)) {
    const int region = role_v ? 2 * K_BYTES + vslot * V_BYTES : kslot * K_BYTES;
    *reinterpret_cast<u32x4*>(smem + region + st_sel + i * stride_sel) = kreg[set][i];
  };

  // ---- fragment reads ----
  // K (A operand of S^T): lane (l31, hi) reads key row sub*32 + l31, channels ks*16 + hi*8 .. +8
  const int k_rd = l31 * (KLD * 2) + hi * 16;
  // V (A operand of O^T, through the transpose read): a 16-lane group reads one [4 keys][16 channels] block, lane i of
  // the group supplying the address of row i/4, channels 4*(i%4)..+4 and receiving column i; the two groups of a half
  // wave take the two channel halves of a 32-channel block, hi selects the key quad (keys 4hi.. and 8+4hi.. of a
  // 16-key slot: the order the score accumulators already hold, see pv_attn.hip)
  const int v_rd = hi * (NDB * 256) + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  bf16x8 kfr[KRING], vfr[VRING];
  // (the slot's base address is formed once per step and hidden from the compiler, so that every fragment address is
  //  that register + an immediate: otherwise hipcc re-associates slot + constant on the scalar unit and spends one
  //  v_add per read)
  auto lds_base = [&](char* p) __attribute__((always_inline)) {
    unsigned a = (unsigned)(unsigned long)p;   // LDS addresses are 32-bit
    asm volatile("" : "+v"(a));
    return (__attribute__((address_space(3))) char*)(unsigned long)a;
  };
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  auto read_k = [&](lds_ptr_t kb, int f) __attribute__((always_inline)) {   // f = sub * NKS + ks
    const int sub = f / NKS, ks = f - sub * NKS;
    kfr[f % KRING] = *reinterpret_cast<__attribute__((address_space(3))) const bf16x8*>(kb + sub * 32 * (KLD * 2) + ks * 32);
  };
  auto read_v = [&](lds_ptr_t vb, int f) __attribute__((always_inline)) {   // f = sk * NDB + db
    const int sk = f / NDB, db = f - sk * NDB;
    lds_ptr_t p = vb + ((sk * 4) * NDB + db) * 256;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)p);
    const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(p + 2 * NDB * 256));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
    vfr[f % VRING] = __builtin_bit_cast(bf16x8, both);
  };

  f32x16 o[NQB][NDB];
  float m_run[NQB], l_run[NQB];
  float mx_next[NQB];                    // raw (unscaled) row maxima of the scores the next step consumes
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    m_run[qb] = -1e30f;
    l_run[qb] = 0.f;
    mx_next[qb] = -1e30f;
  }
  f32x16 sA[NQB][2], sB[NQB][2];         // two score sets [q block][32-key sub-tile], swapped by unrolling the tile loop twice
  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // One tile step: softmax + PV of tile t (scores in sc_); MORE: the scores of tile t+1 are produced here, into sn.
  //   phase 1: S^T(t+1) MFMAs | exp2 of tile t, P -> bf16 | first V(t) fragments
  //   barrier  (K(t+3) / V(t+2) written in phase 2 of the previous step become visible; everybody is done with K(t+1))
  //   phase 2: PV(t) MFMAs | rest of the exponentials | row maxima of S(t+1) | K(t+3), V(t+2) regs -> LDS, K(t+5), V(t+4)
  //            requested | first K(t+2) fragments
  auto step = [&](auto more_c, auto par_c, int t, f32x16 (&sc_)[NQB][2], f32x16 (&sn)[NQB][2]) __attribute__((always_inline)) {
    constexpr bool more = decltype(more_c)::value;
    constexpr int par = decltype(par_c)::value;   // t & 1: the staging register set of this step
    const lds_ptr_t kslot = lds_base(ksm + ((t + 1) & 1) * K_BYTES + k_rd);   // K(t+1)
    const lds_ptr_t vslot = lds_base(vsm + (t % 3) * V_BYTES + v_rd);         // V(t)
    const lds_ptr_t knext = lds_base(ksm + (t & 1) * K_BYTES + k_rd);         // K(t+2)
    if (__builtin_expect((t + 1) * KT > d.Nk, 0)) {   // ragged last tile: keys >= Nk out of the softmax, exact row max
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        float mx = -1e30f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if ((t * KT + sub * 32 + crow(r, hi)) >= d.Nk) sc_[qb][sub][r] = -1e30f;
            mx = fmaxf(mx, sc_[qb][sub][r]);
          }
        mx_next[qb] = mx;
      }
    }
    // ---- deferred rescale: the running max advances only when some lane's tile max exceeds it by more than kDefer ----
    constexpr float kDefer = 8.0f;
    float mxs[NQB];
    bool grow = false;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      const float mx = mx_next[qb] * sc;   // sc > 0
      const unsigned mu = __builtin_bit_cast(unsigned, mx);
      auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
      mxs[qb] = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
      grow = grow || mxs[qb] > m_run[qb] + kDefer;
    }
    if (__builtin_expect(__any(grow), 0)) {   // (cold: the register allocator must not pay for it on the common path)
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        const float m_new = fmaxf(m_run[qb], mxs[qb]);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
        // O^T lives in the accumulator half of the register file and must stay there on the common path: written as
        // `o *= alpha` hipcc moves all of it into arithmetic registers at the head of EVERY step (for this rare block)
        // and spills the score copies to make room.  Element by element through scratch registers instead, with the
        // waits the hazard recogniser cannot see inside inline assembly (MFMA result -> accumulator read, accumulator
        // write -> MFMA operand).
        if constexpr (!AG) {
#pragma unroll
          for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
          continue;
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 4) {
            float e0 = o[qb][i][r], e1 = o[qb][i][r + 1], e2 = o[qb][i][r + 2], e3 = o[qb][i][r + 3], t0, t1, t2, t3;
            asm volatile(
                "v_accvgpr_read_b32 %4, %0\n\tv_accvgpr_read_b32 %5, %1\n\tv_accvgpr_read_b32 %6, %2\n\tv_accvgpr_read_b32 %7, %3\n\t"
                "s_nop 0\n\t"
                "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8\n\t"
                "s_nop 0\n\t"
                "v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                : "+a"(e0), "+a"(e1), "+a"(e2), "+a"(e3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "v"(alpha));
            o[qb][i][r] = e0;
            o[qb][i][r + 1] = e1;
            o[qb][i][r + 2] = e2;
            o[qb][i][r + 3] = e3;
          }
        asm volatile("s_nop 7");
      }
    }
    float neg_m[NQB], ps[NQB][2];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      neg_m[qb] = -m_run[qb];
      ps[qb][0] = ps[qb][1] = 0.f;
    }
    bf16x8 pb[NQB][4];   // [q block][16-key slot]
    // element e of the tile's NE exponentials per lane: slot sk = e / (8 NQB), then q block, then the 8 values of the fragment
    auto exp_range = [&](int e0, int e1) __attribute__((always_inline)) {
      if constexpr (ABL & 1) return;
#pragma unroll
      for (int e = e0; e < e1; e += 2) {
        const int sk = e / (8 * NQB), qb = (e >> 3) % NQB, j = e & 7;
        const int sub = sk >> 1, r = (sk & 1) * 8 + j;
        const float p0 = __builtin_amdgcn_exp2f(fmaf(sc_[qb][sub][r], sc, neg_m[qb]));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(sc_[qb][sub][r + 1], sc, neg_m[qb]));
        ps[qb][0] += p0;
        ps[qb][1] += p1;
        asm volatile("" : "+v"(ps[qb][0]), "+v"(ps[qb][1]));   // (hipcc otherwise keeps all the values alive and sums them at the end)
        pb[qb][sk][j] = (bf16_t)p0;
        pb[qb][sk][j + 1] = (bf16_t)p1;
      }
    };
    // The same work as a three-stage pipeline over the MFMA slices of a step (g = 0 .. NS1 + NS2 - 1): slice g computes
    // the exponent arguments of batch g, the exponentials of batch g-1 and the sums / conversions of batch g-2 -- no
    // instruction waits for the one before it (a wave alone on its SIMD is in-order: fma -> exp -> add back to back
    // stalls twice per element).  Batch g = elements [exp_bound(g), exp_bound(g+1)), NE elements over kExpSlices slices;
    // slot sk is complete two slices after its last batch, before PV MFMA NS1 + NQB*NDB*sk needs it.
    float xv[NE], pv[NE];
    auto exp_pipe = [&](int g) __attribute__((always_inline)) {
      if constexpr (ABL & 1) return;
#pragma unroll
      for (int e = exp_bound<NQB>(g - 2); e < exp_bound<NQB>(g - 1); e += 2) {
        const int sk = e / (8 * NQB), qb = (e >> 3) % NQB, j = e & 7;
        ps[qb][0] += pv[e];
        ps[qb][1] += pv[e + 1];
        asm volatile("" : "+v"(ps[qb][0]), "+v"(ps[qb][1]));
        pb[qb][sk][j] = (bf16_t)pv[e];
        pb[qb][sk][j + 1] = (bf16_t)pv[e + 1];
      }
#pragma unroll
      for (int e = exp_bound<NQB>(g - 1); e < exp_bound<NQB>(g); ++e) pv[e] = __builtin_amdgcn_exp2f(xv[e]);
#pragma unroll
      for (int e = exp_bound<NQB>(g); e < exp_bound<NQB>(g + 1); ++e) {
        const int sk = e / (8 * NQB), qb = (e >> 3) % NQB, j = e & 7;
        xv[e] = fmaf(sc_[qb][sk >> 1][(sk & 1) * 8 + j], sc, neg_m[qb]);
      }
    };
    constexpr int NS1 = more ? NQB * NKF : 0;    // MFMAs of phase 1
    constexpr int NS2 = NQB * NVF;               // MFMAs of phase 2
    if constexpr (more) {
#pragma unroll
      for (int i = 0; i < NS1; ++i) {
        const int f = i / NQB, qb = i % NQB;
        const int sub = f / NKS, ks = f - sub * NKS;
        if (!(ABL & 32) && qb == 0 && f + KPRE < NKF) read_k(kslot, f + KPRE);
        if constexpr (!(ABL & 8)) {
          if (ks == 0) mfma_s_first<AG>(sn[qb][sub], kfr[f % KRING], qf[qb][ks]);
          else mfma_s_acc<AG>(sn[qb][sub], kfr[f % KRING], qf[qb][ks]);
        }
        exp_pipe(i);
        // first fragments of V(t), early enough to have landed when the barrier's lgkmcnt(0) is reached
        if (!(ABL & 32) && i >= NS1 - VPRE - VLEAD && i < NS1 - VLEAD) read_v(vslot, i - (NS1 - VPRE - VLEAD));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      exp_range(0, 8 * NQB);
#pragma unroll
      for (int f = 0; f < VPRE; ++f) read_v(vslot, f);
    }
    if constexpr (!(ABL & 2)) __syncthreads();
    {
      constexpr int EB = 8 * NQB;                // (last step only: exponentials already done)
      constexpr int NE2 = NE - EB;               // ... still to do, under the PV MFMAs of slots 0 .. 2 (slot sk due at NQB*NDB*sk)
      constexpr int ESL = 3 * NQB * NDB - NQB;   // spread over this many PV MFMAs
      float mxa[NQB];
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) mxa[qb] = -1e30f;
#pragma unroll
      for (int i = 0; i < NS2; ++i) {
        const int f = i / NQB, qb = i % NQB;
        const int sk = f / NDB, db = f - sk * NDB;
        if (!(ABL & 32) && qb == 0 && f + VPRE < NVF) read_v(vslot, f + VPRE);
        if constexpr (!(ABL & 16)) o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[f % VRING], pb[qb][sk], o[qb][db], 0, 0, 0);
        if constexpr (more) exp_pipe(NS1 + i);
        else if (i < ESL) exp_range(EB + 2 * (i * (NE2 / 2) / ESL), EB + 2 * ((i + 1) * (NE2 / 2) / ESL));
        if constexpr (more) {
          // staging: this thread's K(t+3) / V(t+2) registers (requested two steps ago) into LDS, then the registers go
          // back out for K(t+5) / V(t+4)
          if constexpr (!(ABL & 4)) {
            if constexpr (SPLIT) {
              if (i >= 2 && i < 2 + NDB) store_kv(par, (t + 1) & 1, (t + 2) % 3, i - 2);
              if (i == 2 + NDB) load_kv(par, t + 5, t + 4);
            } else {
              if (i >= 2 && i < 2 + NDB) store_k(par, (t + 1) & 1, i - 2);
              if (i == 2 + NDB) load_k(par, t + 5);
              if (i >= 4 + NDB && i < 4 + 2 * NDB) store_v(par, (t + 2) % 3, i - 4 - NDB);
              if (i == 4 + 2 * NDB) load_v(par, t + 4);
            }
          }
          // row maxima of S(t+1) over the last slots (behind the last batch of the exponential pipeline)
          constexpr int M0 = kExpSlices + 2 - NS1, MC = NS2 - M0;
          static_assert(MC > 0 && 4 + 2 * NDB < NS2, "phase 2 has room for the staging and the maxima");
          if (!(ABL & 64) && i >= M0) {
            const int a0 = 2 * ((i - M0) * 16 / MC), a1 = 2 * ((i - M0 + 1) * 16 / MC);   // 32 (sub, r) positions in pairs, every q block each
#pragma unroll
            for (int a = a0; a < a1; a += 2)
#pragma unroll
              for (int qq = 0; qq < NQB; ++qq)   // (written out: hipcc puts a canonicalising v_max in front of fmaxf on every asm result)
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mxa[qq]) : "v"(sn[qq][a >> 4][a & 15]), "v"(sn[qq][a >> 4][(a & 15) + 1]));
#pragma unroll
            for (int qq = 0; qq < NQB; ++qq) asm volatile("" : "+v"(mxa[qq]));   // (the chain stays in its slice)
          }
          // first fragments of K(t+2) (visible since this step's barrier)
          if (!(ABL & 32) && i >= NS2 - KPRE - KLEAD && i < NS2 - KLEAD) read_k(knext, i - (NS2 - KPRE - KLEAD));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int qq = 0; qq < NQB; ++qq) {
        mx_next[qq] = mxa[qq];
        asm volatile("" : "+v"(mx_next[qq]));
      }
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) l_run[qb] += ps[qb][0] + ps[qb][1];
  };

  // ---- prologue: tiles 0 and 1 staged, K(2) behind them; the scores of tile 0 the plain way ----
  if constexpr (SPLIT) {
    load_kv(0, 0, 0);
    load_kv(1, 1, 1);
#pragma unroll
    for (int i = 0; i < NDB; ++i) store_kv(0, 0, 0, i);
    load_kv(0, 2, 2);
#pragma unroll
    for (int i = 0; i < NDB; ++i) store_kv(1, 1, 1, i);
    load_kv(1, 4, 3);
  } else {
    load_k(0, 0);
    load_v(0, 0);
    load_k(1, 1);
    load_v(1, 1);
#pragma unroll
    for (int i = 0; i < NDB; ++i) { store_k(0, 0, i); store_v(0, 0, i); }
    load_k(0, 2);
    load_v(0, 2);
#pragma unroll
    for (int i = 0; i < NDB; ++i) { store_k(1, 1, i); store_v(1, 1, i); }
    load_k(1, 4);
    load_v(1, 3);
  }
  PV_STAMP(1);
  __syncthreads();
  PV_STAMP(2);
#pragma unroll
  for (int f = 0; f < NKF; ++f) {
    const int sub = f / NKS, ks = f - sub * NKS;
    const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(ksm + k_rd + sub * 32 * (KLD * 2) + ks * 32);
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      if (ks == 0) mfma_s_first<AG>(sA[qb][sub], kf0, qf[qb][ks]);
      else mfma_s_acc<AG>(sA[qb][sub], kf0, qf[qb][ks]);
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");   // MFMA result -> VALU read (see mfma_s_first)
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    float mx = -1e30f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sA[qb][sub][r]);
    mx_next[qb] = mx;
    if (KT <= d.Nk) {   // the running max starts at tile 0's (unless that tile is ragged: the step masks it first), so the
                        // first step does not take the rescale block for an O that is still zero
      const float ms = mx * sc;
      const unsigned mu = __builtin_bit_cast(unsigned, ms);
      auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
      m_run[qb] = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    }
  }
  __syncthreads();                       // everybody has read K(0): its slot takes K(2)
  // entering step 0: set 0 = K(3) | V(2), set 1 = K(4) | V(3)  (step t stores K(t+3), V(t+2) from set t&1)
  if constexpr (SPLIT) {
    if (!role_v) {                       // (a wave-uniform branch, outside the tile loop)
#pragma unroll
      for (int i = 0; i < NDB; ++i) store_k(0, 0, i);
      load_k(0, 3);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NDB; ++i) store_k(0, 0, i);
    load_k(0, 3);
  }
#pragma unroll
  for (int f = 0; f < KPRE; ++f) read_k(lds_base(ksm + K_BYTES + k_rd), f);   // first fragments of K(1)
  {
    using Y = std::integral_constant<bool, true>;
    using N = std::integral_constant<bool, false>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    PV_STAMP(3);
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
      step(Y{}, P0{}, t, sA, sB);
      step(Y{}, P1{}, t + 1, sB, sA);
    }
    PV_STAMP(4);
    if (ntiles - t == 2) {
      step(Y{}, P0{}, t, sA, sB);
      step(N{}, P1{}, t + 1, sB, sA);
    } else {
      step(N{}, P0{}, t, sA, sB);
    }
  }

  PV_STAMP(5);
  // ---- epilogue: O[q][d] = O^T[d][q] / l (+ q), whole rows at a time ----
  // A lane holds 4 channels of each of 24 (block, group) pieces of ITS query: stored from there, every store of a wave
  // touches 32 rows (and the residual's reads likewise) -- measured 14 k cycles per workgroup, a fifth of a 13-tile
  // workgroup.  Instead each wave turns its fp32 block through the (now free) tile rings, ER rows at a time: 16-byte LDS
  // writes by (query, channel quad), then 8-channel chunks read back in row order, + q in fp32, ONE rounding, 16-byte
  // global accesses in which 12 consecutive lanes cover a row.
  __syncthreads();                                   // every wave is done with the K / V rings
  constexpr int EPITCH = D * 4 + 16;                 // fp32 row + 16 B: rows 4 banks apart, 16-byte aligned
  constexpr int ER = NW == 4 ? 32 : 16;              // rows of a q block per pass
  static_assert(NW * ER * EPITCH <= 2 * K_BYTES + 3 * V_BYTES, "epilogue staging fits the rings");
  char* const ebase = smem + wave * (ER * EPITCH);
  constexpr int CH = D / 8;                          // 8-channel chunks per row
  static_assert((ER * CH) % 64 == 0, "whole waves of chunks");
  auto wave_sync = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int pass = 0; pass < 32 / ER; ++pass) {
      if (ER == 32 || (l31 / ER) == pass) {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {o[qb][db][g * 4 + 0] * inv, o[qb][db][g * 4 + 1] * inv, o[qb][db][g * 4 + 2] * inv, o[qb][db][g * 4 + 3] * inv};
            *reinterpret_cast<f32x4*>(ebase + (l31 % ER) * EPITCH + (db * 32 + 8 * g + 4 * hi) * 4) = v;
          }
      }
      wave_sync();
      const int row0 = qblk * kRowsWG + wave * (32 * NQB) + qb * 32 + pass * ER;
#pragma unroll
      for (int i = 0; i < ER * CH / 64; ++i) {
        const int id = i * 64 + lane;
        const int r = id / CH, c = id - r * CH;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ebase + r * EPITCH + c * 32);
        const f32x4 up = *reinterpret_cast<const f32x4*>(ebase + r * EPITCH + c * 32 + 16);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
        const int row = row0 + r;
        if (row < d.Nq) {
          if (d.residual_q) {
            const bf16x8 qv = *reinterpret_cast<const bf16x8*>(Q + (long)row * d.ldq + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)qv[j];
          }
          bf16x8 ov;
#pragma unroll
          for (int j = 0; j < 8; ++j) ov[j] = (bf16_t)v[j];
          *reinterpret_cast<bf16x8*>(O + (long)row * d.ldo + c * 8) = ov;
        }
      }
      wave_sync();   // the next pass reuses the staging rows
    }
  }
  asm volatile("s_waitcnt vmcnt(0)");
  PV_STAMP(6);
}

template <int NQB, int NW> int launch_w64(const pv_attention_desc& d, hipStream_t s) {
  constexpr int kRows = 32 * NQB * NW;
  const int nqb = (d.Nq + kRows - 1) / kRows;
  const long total = (long)d.B * d.heads * nqb;
  if (total <= 0 || total > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
#ifdef PV_DEV_ABLATION
  if (NW == 4) switch (pv_tune("attn_abl", 0)) {
#define PV_ABL_CASE(a) case a: PV_LAUNCH((attn_w64_kernel<96, NQB, 4, a>), dim3((unsigned)total), dim3(256), 0, s, d, nqb, (int)total); PV_LAUNCH_CHECK(); return PV_OK;
    PV_ABL_CASE(1) PV_ABL_CASE(2) PV_ABL_CASE(4) PV_ABL_CASE(8) PV_ABL_CASE(16) PV_ABL_CASE(32) PV_ABL_CASE(64) PV_ABL_CASE(24) PV_ABL_CASE(65) PV_ABL_CASE(103)
#undef PV_ABL_CASE
    default: break;
  }
#endif
  PV_LAUNCH((attn_w64_kernel<96, NQB, NW>), dim3((unsigned)total), dim3(64 * NW), 0, s, d, nqb, (int)total);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

}  // namespace

int pv_attn_w64_try(const pv_attention_desc& d, hipStream_t s) {
  if (d.dtype != PV_BF16 || d.head_dim != 96) return PV_ERR_UNSUPPORTED;
  // 0: pv_attn.hip's kernels; 1: 64 rows per wave, one wave per SIMD; 2 / 3: 32 rows per wave, two waves per SIMD as one 8-wave
  // workgroup / two 4-wave workgroups per CU; 4: form 1 while its 256-row items fit two rounds of the chip, else form 3.
  // Default 3: in the micro-benchmark (batch 8, one kernel on the chip) form 1 wins the shapes with <= 3137 queries and the
  // mix (4) the sum; in the model's default form -- two sub-batches of 4 as parallel graph branches -- form 3 wins every
  // time: 1270 (attn_pipe_kernel) / 1286 (4) / 1287 (2) / 1297 (3) clips/s over three interleaved rounds
  // (profiles/r6/model_ab_attn_w64_forms_call79.txt): its workgroups are half the size and leave registers for the
  // other branch's kernels.
  const int form = pv_tune("attn_w64", 3);
  if (!form) return PV_ERR_UNSUPPORTED;
  const long kv_bytes = ((long)d.Nk * (d.ldk > d.ldv ? d.ldk : d.ldv) + d.head_dim) * 2;   // 32-bit buffer offsets
  if (kv_bytes >= 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  if (form == 2) return launch_w64<1, 8>(d, s);
  if (form == 3) return launch_w64<1, 4>(d, s);
  if (form == 4) {   // by size: the one-wave form while its 256-row items fit two rounds of the chip
    const long items = (long)d.B * d.heads * ((d.Nq + 255) / 256);
    return items <= pv_tune("attn_w64_items", 512) ? launch_w64<2, 4>(d, s) : launch_w64<1, 4>(d, s);
  }
  return launch_w64<2, 4>(d, s);
}
