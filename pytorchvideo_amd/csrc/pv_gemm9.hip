// MFMA-bound dense convolution / linear layer, 256 x 256 tile with an eight-phase main loop (round 5).
//
//   out[voxel m][channel n] = act( (sum_k X'[m][k] W[n][k]) * scale[n] + shift[n] + residual[m][n] )
//   k = tap*cin + c ;  X'[m][k] = x[voxel m shifted by tap][c]  (zero outside the image)
//
// Why a third GEMM kernel.  Counters and ablations of the two older ones (pv_gemm.hip: 128 x 128 tile, two LDS buffers,
// `vmcnt(0)` + barrier per K step; pv_gemm8.hip: 256-wide tiles on a ring of 32-deep stages) say the same thing from two
// sides: a CU moves 64 bytes per clock from L2 into LDS and multiplies 4096 bf16 FLOP per clock, so a tile needs MORE than
// 64 FLOP per staged byte -- the 128 x 128 tile has exactly 64 -- and the staged rows must be whole 128-byte lines: a
// 32-deep stage fetches every line twice, half at a time, through a 32 KB L1 that has long lost the first half (the
// 256 x 256 x 32 ring measured 960 TFLOP/s on 32768 x 4096 x 4096, a third of the matrix pipe).  This kernel is the
// structure /opt/skills/guides/cdna_hip_programming.md 5 describes for that regime (T2 + T3 + T4 + T5), rebuilt around
// the implicit-GEMM staging of this library:
//   * 256 x 256 x 64 tile, 512 threads = 8 waves as 4 (voxels) x 2 (channels); a wave owns 64 voxels x 128 channels =
//     8 accumulator blocks of v_mfma_f32_32x32x16_bf16 (weights are the A operand with permuted rows, so a lane's 16
//     accumulator registers are 16 consecutive channels of one voxel: epilogue in registers, 16-byte stores);
//   * the K step of 64 is multiplied in FOUR PHASES, one quadrant (64 channels x 32 voxels x K 64 = 8 MFMAs) each, in snake
//     order (a0,v0) (a0,v1) (a1,v1) (a1,v0): a phase re-reads only the operand half that changed (12 / 4 / 8 / 0
//     `ds_read_b128`), and -- the point of quadrant phases -- the LDS rows of a K tile are RETIRED progressively: the
//     channel half a0 and voxel half v0 are last read in phase 0, v1 in phase 1, a1 in phase 2;
//   * LDS = 2 K tiles x 4 units of 16 KB (channel halves AE / AO, voxel halves BE / BO; 128 rows of one full 128-byte line
//     each), filled by LDS-DMA, ONE unit (two `global_load_lds_dwordx4` per thread) per phase, each unit as soon as its
//     rows are two phases dead:  phase 0 -> BO(t+1), phase 1 -> AO(t+1), phase 2 -> AE(t+2), phase 3 -> BE(t+2).  Every
//     unit is requested five to six phases (>= 2.5 k cycles) before its first read; the wait of each phase is the counted
//     `vmcnt(8)` (the four youngest units stay in flight across the barriers), never 0;
//   * the two halves of the workgroup (waves 0-3 / 4-7: one of each per SIMD) run the same stream ONE BARRIER APART: a
//     phase is  [fragment reads, DMA issue, counted wait] B1 [8 MFMAs at raised priority] B2, so on every SIMD one wave
//     feeds the matrix pipe while its partner reads LDS and issues loads;
//   * staging arithmetic per DMA is a 64-bit add and a select: a K step of 64 lies inside ONE tap (the input width is a
//     multiple of 64), so tap, frame / row / column shift and weight column are wave-uniform scalars; a thread keeps, per
//     tile, the byte offsets of its 4 + 4 staging rows and a 24-bit window mask per voxel row (which dt / dh / dw land
//     inside the image); out-of-image taps and the exhausted stream read a 16-byte zero page; M / N tails read a clamped
//     row whose results are never stored;
//   * persistent workgroups (one per CU) walk an XCD-aware tile list and the DMA stream runs on ACROSS tiles: the first
//     six units of the next tile are in flight during the epilogue; the epilogue's stores are counted into the waits of the
//     K step that follows it (masked stores get an out-of-range buffer offset: every wave issues the same number).
// LDS swizzle as in the older kernels (chunk ^= (row >> 1) & 7 on the DMA's SOURCE address and on the read: conflict-free
// `ds_read_b128`, both sides or neither).
#include <stdlib.h>
#include "pv_common.h"

__device__ __attribute__((aligned(16))) unsigned int pv_zero_page9[4] = {0u, 0u, 0u, 0u};

namespace {

constexpr int kThreads9 = 512;
constexpr int BT9 = 256;              // tile edge: voxels and channels
constexpr int UNIT9 = 128 * 64;       // elements of one staging unit (128 rows x 64 K, 16 KB)
// LDS (elements): [AE0 | AO0 | AE1 | AO1 | BE0 | BO0 | BE1 | BO1] -- K tile parity P, half h: the channel units at
// (2 P + h) * UNIT9, the voxel units 4 * UNIT9 further; every fragment read is then one per-lane base register + a 16-bit
// immediate (the main loop is written once per parity), no address arithmetic and no second set of address registers
constexpr int LDS9_ELEMS = 8 * UNIT9;   // 128 KB
__device__ __forceinline__ constexpr int unit_a(int P, int a) { return (2 * P + a) * UNIT9; }
__device__ __forceinline__ constexpr int unit_b(int P, int v) { return (4 + 2 * P + v) * UNIT9; }
constexpr int kTab9 = LDS9_ELEMS * 2;       // byte offset of the epilogue tables: [tile parity][scale 256 f32 | shift 256 f32]
constexpr int kGeo9 = kTab9 + 2 * 2048;          // implicit-GEMM staging geometry: 32 bytes per thread
constexpr int kLds9Bytes = kGeo9 + kThreads9 * 32;
constexpr int kRegionB9 = 4 * UNIT9 * 2;   // byte offset of the voxel units (folded into the read base: immediates stay < 64 K)

typedef const __attribute__((address_space(1))) void* gptr9_t;
typedef __attribute__((address_space(3))) void* lptr9_t;

// LDS row -> channel inside a 32-row MFMA tile (accumulator register r of lane-half hi is channel 16*hi + r)
__device__ __forceinline__ int chi9(int rho) {
  return (rho & ~31) + 16 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 3) + (rho & 3);
}

struct Geom9 {         // per-thread staging rows of one output tile (DMA j of a unit covers unit rows 64 j + 8 wave + lane / 8)
  unsigned a_row;      // BYTE offset in w of the weight row of (a = 0, j = 0); (a, j) is 64 a + 128 j rows further, clamped at use
  unsigned b_row;      // PW: byte offset in x of the voxel row of (v = 0, j = 0); (v, j) is 32 v + 128 j rows further, clamped at use
  // !PW: the four voxel rows' window origins (element offsets in x, negative for padding rows) and window masks (bit dt | bit
  // 8 + dh | bit 16 + dw set when that tap lies inside the image) do not fit the register budget next to 128 accumulators and
  // 64 fragment registers (the compiler spilled accumulators in the loop header): they live in 32 bytes of LDS per thread,
  // [v][off j0, off j1, mask j0, mask j1], written per tile and fetched by one ds_read_b128 in the phase that issues the voxel
  // half (each thread reads only what it wrote: no synchronisation)
};

template <bool PW, bool YF32, int VAR>
__global__ __launch_bounds__(kThreads9) void gemm_quad_kernel(const pv_conv3d_desc d, int tiles_n, int total_tiles, int tap_rot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem9_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem9_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int S_out = d.To * d.Ho * d.Wo;
  const long M = (long)d.B * S_out;
  const int K = d.kt * d.kh * d.kw * d.cin;
  const int nk = K >> 6;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const int dil_t = d.dil_t > 1 ? d.dil_t : 1, dil_h = d.dil_h > 1 ? d.dil_h : 1, dil_w = d.dil_w > 1 ? d.dil_w : 1;
  const bf16_t* __restrict__ X = static_cast<const bf16_t*>(d.x);
  const char* __restrict__ Wb = static_cast<const char*>(d.w);
  const unsigned long zaddr = (unsigned long)reinterpret_cast<const bf16_t*>(pv_zero_page9);

  // ---- staging rows of this thread (the same for every unit): DMA j covers unit rows 64 j + 8 wave + lane / 8 ----
  auto tile_origin = [&](int it, long& m0, int& n0) __attribute__((always_inline)) {   // XCD-aware tile order (bijective for any tile count)
    const int xcd = it & 7, slot = it >> 3;
    const int qn = total_tiles >> 3, rn = total_tiles & 7;
    const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
    m0 = (long)(tile / tiles_n) * BT9;
    n0 = (tile % tiles_n) * BT9;
  };
  // K chunk (8 elements) that lands on LDS position lane % 8 of this thread's staging rows: the swizzle key (row >> 1) & 7 is
  // the same for both DMAs of a unit (rows 64 apart) -- one register for all eight staging rows
  const int chunk8 = ((lane & 7) ^ (((8 * wave + (lane >> 3)) >> 1) & 7)) * 8;
  // row pitches and last rows in bytes (wave-uniform): a staging row is  min(row0 + step, last) + chunk + K offset  -- an add, a
  // min and an add per DMA, no multiply (a 64-bit multiply-add per DMA cost the eight-phase loop 6-13 %: profiles/r5)
  const unsigned a_pitch = (unsigned)K * 2u, a_last = (unsigned)(d.cout - 1) * a_pitch;
  const unsigned b_pitch = (unsigned)d.ldx * 2u, b_last = (unsigned)(M - 1) * b_pitch;
  int* geo = reinterpret_cast<int*>(smem9_raw + kGeo9) + tid * 8;   // (!PW) this thread's staging geometry
  int iss_c0 = 0, iss_dt = 0, iss_dh = 0, iss_dw = 0;   // channel offset inside the tap, tap coordinates (wave-uniform)
  unsigned iss_wk = 0;                                  // byte offset of the stream's K tile inside a weight row
  auto geom_of = [&](int it, Geom9& g) __attribute__((always_inline)) {
    long m0;
    int n0;
    tile_origin(it, m0, n0);
    iss_c0 = iss_dh = iss_dw = 0;
    // temporal-tap rotation (tap_rot: temporal stride 1, whole tiles inside one frame -- host check): the tile of output frame t
    // starts its reduction at tap (pt - t) mod kt and wraps, so that in step i EVERY concurrent tile reads an input frame with
    // index = i (mod kt): the kt consumers of a frame fetch it at the same moment instead of kt times through a 4 MB L2
    // (SlowFast res4 conv_a: 32 tiles per XCD touch 7.3 MB of frames per tap; PMC fetch 2.95 x the input without this)
    int rot = 0;
    if constexpr (!PW) {
      if (tap_rot) {
        const unsigned sp = (unsigned)m0 % (unsigned)S_out;
        rot = (d.pt - (int)(sp / (unsigned)(d.Ho * d.Wo))) % d.kt;
        rot = rot < 0 ? rot + d.kt : rot;
      }
    }
    iss_dt = rot;
    iss_wk = (unsigned)(rot * d.kh * d.kw * d.cin) * 2u;
    const int rho0 = 8 * wave + (lane >> 3);   // unit row of DMA 0 (< 64)
    g.a_row = (unsigned)(n0 + 32 * (rho0 >> 5) + chi9(rho0 & 31)) * a_pitch;
    if constexpr (PW) {
      g.b_row = (unsigned)((int)m0 + 64 * (rho0 >> 5) + (rho0 & 31)) * b_pitch;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          long m = m0 + 128 * j + 64 * (rho0 >> 5) + 32 * v + (rho0 & 31);
          m = m < M ? m : M - 1;                              // M tail: a clamped row, never stored
          const unsigned b = (unsigned)m / (unsigned)S_out;
          const unsigned sp = (unsigned)m - b * (unsigned)S_out;
          const unsigned to = sp / (unsigned)(d.Ho * d.Wo);
          const unsigned r2 = sp - to * (unsigned)(d.Ho * d.Wo);
          const unsigned ho = r2 / (unsigned)d.Wo;
          const int t0 = (int)to * d.st - d.pt, h0 = (int)ho * d.sh - d.ph, w0 = (int)(r2 - ho * (unsigned)d.Wo) * d.sw - d.pw;
          geo[v * 4 + j] = (int)((long)b * d.x_bs + ((long)(t0 * d.Hi + h0) * d.Wi + w0) * d.ldx) + chunk8;
          unsigned msk = 0u;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (q < d.kt && (unsigned)(t0 + q * dil_t) < (unsigned)d.Ti) msk |= 1u << q;
            if (q < d.kh && (unsigned)(h0 + q * dil_h) < (unsigned)d.Hi) msk |= 1u << (8 + q);
            if (q < d.kw && (unsigned)(w0 + q * dil_w) < (unsigned)d.Wi) msk |= 1u << (16 + q);
          }
          geo[v * 4 + 2 + j] = (int)msk;
        }
    }
  };

  // ---- issue side: the DMA stream, K tile by K tile across output tiles (all of this state is wave-uniform) ----
  Geom9 g;
  int iss_it = blockIdx.x, iss_ku = 0, iss_jt = 0;   // work item, K tile inside it, this workgroup's tile count
  bool iss_live = iss_it < total_tiles;
  geom_of(iss_live ? iss_it : 0, g);

  // source selection with bit masks, not `?:` (a select between two pointers becomes two exec-masked DMAs)
  auto pick = [&](bool ok, unsigned long p) __attribute__((always_inline)) -> const bf16_t* {
    const unsigned long m = 0ul - (unsigned long)ok;
    return reinterpret_cast<const bf16_t*>((p & m) | (zaddr & ~m));
  };
  // jsel: 0 / 1 = that DMA of the unit only, -1 = both
  auto issue_a = [&](int a, int unit, int jsel) __attribute__((always_inline)) {   // channel half a of the stream's K tile -> LDS unit `unit`
    const unsigned kc = iss_wk + (unsigned)(chunk8 * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (jsel >= 0 && jsel != j) continue;
      unsigned off = g.a_row + (unsigned)(64 * a + 128 * j) * a_pitch;
      off = (off < a_last ? off : a_last) + kc;              // N tail: a clamped row, zeroed in the epilogue
      __builtin_amdgcn_global_load_lds((gptr9_t)pick(iss_live, (unsigned long)(Wb + off)),
                                       (lptr9_t)(smem + unit + (j * 8 + wave) * 512), 16, 0, 0);
    }
  };
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto issue_b = [&](int v, int unit, int jsel, const i32x4& gq) __attribute__((always_inline)) {   // voxel half v (gq: its geometry)
    if constexpr (PW) {
      const unsigned kc = (unsigned)(iss_ku * 128 + chunk8 * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (jsel >= 0 && jsel != j) continue;
        unsigned off = g.b_row + (unsigned)(32 * v + 128 * j) * b_pitch;
        off = (off < b_last ? off : b_last) + kc;            // M tail: a clamped row, never stored
        __builtin_amdgcn_global_load_lds((gptr9_t)pick(iss_live, (unsigned long)(reinterpret_cast<const char*>(X) + off)),
                                         (lptr9_t)(smem + unit + (j * 8 + wave) * 512), 16, 0, 0);
      }
    } else {
      // element offset of the K tile's tap + channel block and the mask bits that must be set for this tap (wave-uniform)
      const int uni = ((iss_dt * dil_t * d.Hi + iss_dh * dil_h) * d.Wi + iss_dw * dil_w) * d.ldx + iss_c0;
      const unsigned sel = (1u << iss_dt) | (1u << (8 + iss_dh)) | (1u << (16 + iss_dw));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (jsel >= 0 && jsel != j) continue;
        const bool ok = iss_live && ((unsigned)gq[2 + j] & sel) == sel;
        __builtin_amdgcn_global_load_lds((gptr9_t)pick(ok, (unsigned long)(X + (long)(gq[j] + uni))),
                                         (lptr9_t)(smem + unit + (j * 8 + wave) * 512), 16, 0, 0);
      }
    }
  };
  auto load_geo = [&](int v) __attribute__((always_inline)) -> i32x4 {
    if constexpr (PW) {
      return i32x4{0, 0, 0, 0};
    } else {
      // (the address is rebuilt from the thread index behind an empty asm: as a loop-invariant register it was the one value the
      // compiler spilled, and its reload -- a scratch load + vmcnt(0) in the loop header -- drained the DMA pipeline every K-tile pair)
      unsigned t = threadIdx.x;
      asm volatile("" : "+v"(t));
      return *reinterpret_cast<const i32x4*>(smem9_raw + kGeo9 + t * 32u + (unsigned)v * 16u);
    }
  };
  // folded BatchNorm / bias tables of the stream's tile -> LDS, one 4-byte DMA per thread: wave w < 4 carries scale[64 w ..
  // 64 w + 63] of the tile's 256 channels, w >= 4 the shift (channels past cout read a clamped entry: zeroed in the epilogue).
  // The global round trips of 2 x 64 table loads per thread were a third of the per-tile fixed cost (epilogue reads: ds_read).
  // Extra DMAs inside the counted windows only make the waits stricter (an older unit completes earlier), never weaker.
  auto issue_tables = [&]() __attribute__((always_inline)) {
    const float* tab = wave < 4 ? d.scale : d.shift;
    if (tab != nullptr && iss_live) {
      long m0;
      int n0;
      tile_origin(iss_it, m0, n0);
      int n = n0 + 64 * (wave & 3) + lane;
      n = n < d.cout ? n : d.cout - 1;
      __builtin_amdgcn_global_load_lds((gptr9_t)(tab + n), (lptr9_t)(smem9_raw + kTab9 + (iss_jt & 1) * 2048 + wave * 256), 4, 0, 0);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {   // next K tile of the stream
    ++iss_ku;
    iss_wk += 128u;
    if constexpr (!PW) {
      iss_c0 += 64;
      if (iss_c0 == d.cin) {
        iss_c0 = 0;
        if (++iss_dw == d.kw) {
          iss_dw = 0;
          if (++iss_dh == d.kh) {
            iss_dh = 0;
            if (++iss_dt == d.kt) { iss_dt = 0; iss_wk = 0u; }   // (a rotated reduction wraps to tap 0 = weight column 0)
          }
        }
      }
    }
    if (iss_ku == nk) {
      iss_ku = 0;
      iss_wk = 0u;
      iss_it += gridDim.x;
      ++iss_jt;
      iss_live = iss_it < total_tiles;
      if (iss_live) geom_of(iss_it, g);
      issue_tables();
    }
  };

  // ---- read side: this lane's fragment position inside a unit, as a BYTE offset from the LDS base ----
  // A rows: wn*64 + ta*32 + (lane & 31) (ta = 1: + 32 rows = + 4096 bytes, same swizzle); B rows: wm*32 + (lane & 31).
  // K slice s reads chunk (2 s + hi) ^ swz(row) = (hi ^ swz(row)) ^ 2 s: ONE register per operand holds the position of
  // slice 0, slices 1-3 are that register ^ (s * 32) -- recomputed at every read (an empty asm keeps the compiler from
  // hoisting the three variants into registers of their own: the kernel sits at the 256-register limit)
  unsigned rd_a0, rd_b0;
  {
    const int ra = (wave & 1) * 64 + (lane & 31), rb = (wave >> 1) * 32 + (lane & 31);
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) unsigned char*)smem9_raw);
    rd_a0 = lds0 + (unsigned)(ra * 128 + (((lane >> 5) ^ ((ra >> 1) & 7)) << 4));
    rd_b0 = lds0 + (unsigned)(kRegionB9 + rb * 128 + (((lane >> 5) ^ ((rb >> 1) & 7)) << 4));
  }
  typedef const __attribute__((address_space(3))) bf16x8* lfrag9_t;
#define PV9_RD(BASE, S, BYTES) (*(lfrag9_t)(unsigned long)(((BASE) ^ ((S) * 32u)) + (unsigned)(BYTES)))

  // s_waitcnt simm16 on gfx9: vmcnt = bits [3:0] | [15:14]; expcnt [6:4] "no wait"; lgkmcnt [11:8] = 0
  constexpr auto vml = [](int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4); };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned kOOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      d.y, 0, (int)((unsigned)d.B * (unsigned)d.y_bs * (YF32 ? 4u : 2u)), 0x00020000);   // stores per thread per tile: 32 (fp32) / 16

  // ---- prologue: the first tile's tables, AE BE BO AO of K tile 0, AE BE of K tile 1 ----
  issue_tables();
  issue_a(0, unit_a(0, 0), -1);
  issue_b(0, unit_b(0, 0), -1, load_geo(0));
  issue_b(1, unit_b(0, 1), -1, load_geo(1));
  issue_a(1, unit_a(0, 1), -1);
  advance();
  issue_a(0, unit_a(1, 0), -1);
  issue_b(0, unit_b(1, 0), -1, load_geo(0));
  __builtin_amdgcn_s_waitcnt(vml(8));   // AE, BE of K tile 0: this thread's share
  __builtin_amdgcn_s_barrier();
  const bool half_b = wave >= 4;        // wave-uniform
  if (half_b) __builtin_amdgcn_s_barrier();   // the offset between the two halves
  __builtin_amdgcn_sched_barrier(0);

  bool stores_behind = false;   // the previous tile's stores sit behind the units the first K tile's waits cover (wave-uniform)
  int jt = 0;                   // tiles finished by this workgroup (table parity)
  f32x16 acc[4][2];

  // one phase:  [reads] [DMA] wait | B1 | 8 MFMAs | B2.   FIRST: the K tile that follows an epilogue.
  // DM (variant bit 1): the phase's two DMAs are issued AMONG its MFMAs (after the 2nd and the 4th) instead of before B1 -- an
  // LDS-DMA costs the issuing wave 60-180 cycles, which the matrix pipe hides when the wave has MFMAs in flight; the counted wait
  // then comes BEFORE this phase's DMAs and leaves three units (6 DMAs) in flight instead of four.
#define PV9_WAIT_B1(FIRST)                                                                          \
  do {                                                                                              \
    if ((FIRST) && stores_behind) __builtin_amdgcn_s_waitcnt(vml((DM ? 6 : 8) + (YF32 ? 32 : 16)));  \
    else __builtin_amdgcn_s_waitcnt(vml(DM ? 6 : 8));                                               \
    __builtin_amdgcn_s_barrier();                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#define PV9_M1(AF, A0, V, BF, S, TA)                                                                \
  acc[(A0) + (TA)][V] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[S][TA], BF[S], acc[(A0) + (TA)][V], 0, 0, 0)
#define PV9_SLOT(N, ISS)                                                                             \
  do {                                                                                              \
    if (DM && ((N) == 1 || (N) == 3)) {                                                             \
      __builtin_amdgcn_sched_barrier(0); ISS((N) == 1 ? 0 : 1); __builtin_amdgcn_sched_barrier(0);   \
    }                                                                                               \
  } while (0)
  // ISS(j): issue DMA j (0 / 1) of the phase's unit, ISS(-1): both
#define PV9_PHASE(FIRST, ISS, AF, A0, V, BF)                                                        \
  do {                                                                                              \
    if (!DM) { ISS(-1); }                                                                           \
    PV9_WAIT_B1(FIRST);                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    PV9_M1(AF, A0, V, BF, 0, 0); PV9_SLOT(0, ISS);                                                  \
    PV9_M1(AF, A0, V, BF, 0, 1); PV9_SLOT(1, ISS);                                                  \
    PV9_M1(AF, A0, V, BF, 1, 0); PV9_SLOT(2, ISS);                                                  \
    PV9_M1(AF, A0, V, BF, 1, 1); PV9_SLOT(3, ISS);                                                  \
    PV9_M1(AF, A0, V, BF, 2, 0); PV9_SLOT(4, ISS);                                                  \
    PV9_M1(AF, A0, V, BF, 2, 1); PV9_SLOT(5, ISS);                                                  \
    PV9_M1(AF, A0, V, BF, 3, 0); PV9_SLOT(6, ISS);                                                  \
    PV9_M1(AF, A0, V, BF, 3, 1); PV9_SLOT(7, ISS);                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    __builtin_amdgcn_s_barrier();                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#define PV9_READ_A(AF, P, A)                                                                                              \
  do {                                                                                                                    \
    asm volatile("" : "+v"(rd_a0));                                                                                       \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                         \
      _Pragma("unroll") for (int ta = 0; ta < 2; ++ta) AF[s][ta] = PV9_RD(rd_a0, s, unit_a(P, A) * 2 + ta * 4096);        \
  } while (0)
#define PV9_READ_B(BF, P, V)                                                                                              \
  do {                                                                                                                    \
    asm volatile("" : "+v"(rd_b0));                                                                                       \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) BF[s] = PV9_RD(rd_b0, s, unit_b(P, V) * 2 - kRegionB9);                 \
  } while (0)
  // One K tile of LDS parity P (compile-time: every LDS address is a per-lane base + an immediate).  Fragment reads per phase:
  // 12 / 4 / 8 / 0 (reading the next K tile's first channel half one phase early, 4 / 4 / 8 / 8, needs a second 32-register
  // fragment set and measured 5-12 % SLOWER in every shape: profiles/r5/bench_gemm_quad_v3_variants.txt).
  // FIRST: the K tile that follows an epilogue (its waits have the previous tile's stores behind the units they cover).
#define PV9_KTILE(P, FIRST)                                                                                               \
  do {                                                                                                                    \
    bf16x8 af[4][2], b0[4], b1[4];   /* a channel half, voxel halves v0 / v1 of the K tile (4 K slices each) */           \
    /* phase 0: (a0, v0); BO of the next K tile */                                                                        \
    i32x4 gq = load_geo(1);                                                                                               \
    PV9_READ_B(b0, P, 0);                                                                                                 \
    PV9_READ_A(af, P, 0);                                                                                                 \
    PV9_PHASE(FIRST, ISS_P0_##P, af, 0, 0, b0);                                                                           \
    /* phase 1: (a0, v1); AO of the next K tile */                                                                        \
    PV9_READ_B(b1, P, 1);                                                                                                 \
    PV9_PHASE(FIRST, ISS_P1_##P, af, 0, 1, b1);                                                                           \
    /* phase 2: the stream moves on to the K tile after next (a new output tile's staging rows are computed HERE, where    \
       only the accumulators and the voxel fragments are live); (a1, v1); AE of that K tile -- this parity's AE, last      \
       read two phases ago */                                                                                             \
    advance();                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    PV9_READ_A(af, P, 1);                                                                                                 \
    PV9_PHASE(FIRST, ISS_P2_##P, af, 2, 1, b1);                                                                           \
    /* phase 3: (a1, v0); BE of the K tile after next */                                                                  \
    gq = load_geo(0);                                                                                                     \
    PV9_PHASE(FIRST, ISS_P3_##P, af, 2, 0, b0);                                                                           \
  } while (0)
#define ISS_P0_0(J) issue_b(1, unit_b(1, 1), J, gq)
#define ISS_P1_0(J) issue_a(1, unit_a(1, 1), J)
#define ISS_P2_0(J) issue_a(0, unit_a(0, 0), J)
#define ISS_P3_0(J) issue_b(0, unit_b(0, 0), J, gq)
#define ISS_P0_1(J) issue_b(1, unit_b(0, 1), J, gq)
#define ISS_P1_1(J) issue_a(1, unit_a(0, 1), J)
#define ISS_P2_1(J) issue_a(0, unit_a(1, 0), J)
#define ISS_P3_1(J) issue_b(0, unit_b(1, 0), J, gq)

  constexpr bool DM = (VAR & 2) != 0;     // DMAs among the MFMAs
  const int nkp = nk >> 1;   // K tiles come in pairs (K % 128 == 0, host check): every output tile starts on LDS parity 0
  for (int it = blockIdx.x; it < total_tiles; it += gridDim.x, ++jt) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][v][r] = 0.f;

    PV9_KTILE(0, true);
    PV9_KTILE(1, false);
    for (int kp = 1; kp < nkp; ++kp) {
      PV9_KTILE(0, false);
      PV9_KTILE(1, false);
    }

    // The first half has just passed its last B2; the second half is still multiplying its last quadrant.  One extra barrier
    // here (matched by that half's last B2) and one at the END of the second half's epilogue (matched by the first half's next
    // B1) let both halves run their epilogues side by side instead of one after the other, and restore the offset.
    if (!half_b) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wn = wave & 1, wm = wave >> 1;
    // ---- epilogue: lane owns channels cb..cb+15 of voxel m for every (channel tile a, voxel tile v) ----
    long m0;
    int n0;
    tile_origin(it, m0, n0);
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
    const float* tabs = reinterpret_cast<const float*>(smem9_raw + kTab9 + (jt & 1) * 2048);
    typedef f32x4 res_t[2][2][2];   // [v][h8][half]: 8 channels as 2 x f32x4 (fp32) or 1 x 16 bytes (bf16)
    auto load_res = [&](int a, res_t& res) __attribute__((always_inline)) {
      if (d.residual == nullptr) return;
      const int cb = n0 + wn * 128 + a * 32 + 16 * hi;
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
    };
    auto finish = [&](int a, const res_t& res) __attribute__((always_inline)) {
      const int cl = wn * 128 + a * 32 + 16 * hi;   // channel inside the tile
      const int cb = n0 + cl;
      if (d.scale != nullptr) {
        f32x4 sc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sc[q] = *reinterpret_cast<const f32x4*>(tabs + cl + 4 * q);
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] *= sc[r >> 2][r & 3];
      }
      if (d.shift != nullptr) {
        f32x4 sh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sh[q] = *reinterpret_cast<const f32x4*>(tabs + 256 + cl + 4 * q);
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] += sh[r >> 2][r & 3];
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
      __builtin_amdgcn_sched_barrier(0);
    };
    // (requesting the residual rows of tile a + 1 before tile a is finished -- a second 32-register set -- measured nothing and
    // pushed the epilogue into scratch: one set)
    // (literal channel-tile indices: inside a loop the lambdas' `a` is a run-time index when the accumulators are first split
    // into registers, and half of them then live in scratch for the whole kernel)
    {
      res_t r0;
      load_res(0, r0);
      finish(0, r0);
      load_res(1, r0);
      finish(1, r0);
      load_res(2, r0);
      finish(2, r0);
      load_res(3, r0);
      finish(3, r0);
    }
    // ... then nothing but stores (E_BF16 / E_F32 of them, whatever is masked)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int cb = n0 + wn * 128 + a * 32 + 16 * hi;
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const unsigned yo = (unsigned)(e_b[v] * d.y_bs + e_sp[v] * d.ldy + cb);
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          const bool ok = e_ok[v] && cb + h8 * 8 < cout_p8;
          const int r0 = h8 * 8;
          if constexpr (YF32) {
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
    stores_behind = true;   // the four phases of the next K tile wait on units requested BEFORE these stores
    if (half_b) __builtin_amdgcn_s_barrier();   // (see above: matched by the first half's next B1)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef PV9_KTILE
#undef PV9_PHASE
#undef PV9_SLOT
#undef PV9_M1
#undef PV9_READ_A
#undef PV9_READ_B
#undef PV9_WAIT_B1
#undef PV9_RD
  if (!half_b) __builtin_amdgcn_s_barrier();   // matches the second half's offset barrier
  __builtin_amdgcn_s_waitcnt(vml(0));          // the stream's trailing (zero-page) DMAs land before the LDS is released
}

template <bool PW, bool YF32, int VAR>
int launch9(const pv_conv3d_desc& d, int tiles_n, long total, hipStream_t s) {
  const size_t lds = (size_t)kLds9Bytes;   // 128 KB of units + 4 KB of epilogue tables
  auto kern = gemm_quad_kernel<PW, YF32, VAR>;
  PV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long resident = 256;   // one workgroup per CU
  dim3 grid((unsigned)(total < resident ? total : resident)), block(kThreads9);
  // tap rotation: only where a whole tile lies inside one output frame and frames map to taps one to one
  const int tap_rot = !PW && d.kt > 1 && d.st == 1 && d.dil_t <= 1 && (d.Ho * d.Wo) % BT9 == 0 && pv_tune("gemm9_tap_rot", 1);
  PV_LAUNCH(kern, grid, block, lds, s, d, tiles_n, (int)total, tap_rot);
  pv_note_kernel("gemm_quad_kernel");   // (launched through a function pointer: PV_LAUNCH saw only the variable)
  PV_LAUNCH_CHECK();
  return PV_OK;
}

}  // namespace

// Can this kernel run the geometry at all?  (pointers are not looked at)
static bool gemm9_geometry_ok(const pv_conv3d_desc& d, long& M, long& K, long& tiles_m, int& tiles_n) {
  if (d.dtype != PV_BF16 || d.a_gate != nullptr || d.a_act != PV_ACT_NONE || d.x2 != nullptr) return false;
  if (d.kt > 8 || d.kh > 8 || d.kw > 8) return false;                                // 8-bit window masks per axis
  // dilated convs (detection backbones' res5) stay on the 128 x 128 kernel, whose dilation path has kernel tests
  // (test_dilated_dense_conv); the window masks here fold dil_* in, but no test forces a dilated descriptor onto these tiles
  if (d.dil_t > 1 || d.dil_h > 1 || d.dil_w > 1) return false;
  const int taps = d.kt * d.kh * d.kw;
  if (d.cin % 64 != 0 || ((long)taps * d.cin) % 128 != 0) return false;             // a K step of 64 inside one tap; K tiles in pairs
  M = (long)d.B * d.To * d.Ho * d.Wo;
  K = (long)taps * d.cin;
  const int cout_p8 = pv_round_up(d.cout, 8);
  if (K < 256) return false;                                                         // a first and a last pair of K tiles
  // 31-bit element offsets into x, 32-bit byte offsets into w, 31-bit byte offsets in the store descriptor
  if (M > 0x7fffffffL || (long)d.B * d.x_bs > 0x7fffffffL || ((long)d.cout + 256) * K * 2 > 0xffffffffL ||
      (M + 256) * d.ldx * 2 > 0xffffffffL)
    return false;
  if ((long)d.B * d.y_bs * (d.y_f32 ? 4 : 2) > 0x7fffffffL) return false;
  tiles_m = pv_ceil_div(M, BT9);
  tiles_n = (int)pv_ceil_div(cout_p8, BT9);
  return tiles_m * tiles_n > 0 && tiles_m * tiles_n < 0x3fffffffL;
}

int pv_gemm9h_try(const pv_conv3d_desc& d, bool pw, hipStream_t s);   // pv_gemm9h.hip: the same loop on 128 x 256 tiles

// Returns PV_OK when this kernel took the op, PV_ERR_UNSUPPORTED to leave it to the older GEMM kernels.
int pv_gemm9_try(const pv_conv3d_desc& d, bool pw, hipStream_t s) {
  const int mode = pv_tune("gemm9", 1);   // 0 off, 1 heuristic, 2 wherever the kernel can run
  if (mode == 0) return PV_ERR_UNSUPPORTED;
  {
    // layers whose 256 x 256 tiles do not spread over the chip (SlowFast res4 / res5: 128 / 64 of them) go to the half-height
    // form first; it declines what it cannot run (K % 192, tile count, padding).  Not where this kernel has one round of >= 120
    // tiles and the half-height tiles would need a second, mostly empty one (MViT-B fc1 of the 768-wide blocks: 156 tiles here,
    // 300 there: measured 27 vs 28 us)
    const long Mv = (long)d.B * d.To * d.Ho * d.Wo, tn = pv_ceil_div((long)pv_round_up(d.cout, 8), BT9);
    const long t256 = pv_ceil_div(Mv, BT9) * tn, th = pv_ceil_div(Mv, BT9 / 2) * tn;
    const bool one_round_here = t256 >= pv_tune("gemm9_min_tiles", 120) && th > 256;
    const long c8 = pv_round_up(d.cout, 8);
    const bool padded_here = (double)(tn * BT9 - c8) > 0.15 * (double)c8;   // 128-channel layers: the transposed half tile
    if (mode != 2 && ((t256 < pv_tune("gemm9h_below", 200) && !one_round_here) || padded_here || pv_tune("gemm9h", 1) == 2)) {
      const int r = pv_gemm9h_try(d, pw, s);
      if (r != PV_ERR_UNSUPPORTED) return r;
    }
  }
  long M, K, tiles_m;
  int tiles_n;
  if (!gemm9_geometry_ok(d, M, K, tiles_m, tiles_n)) return PV_ERR_UNSUPPORTED;
  const int cout_p8 = pv_round_up(d.cout, 8);
  const long tiles = tiles_m * tiles_n;
  const long total = tiles;
  if (mode == 1) {
    // one workgroup per CU: the work items must fill the chip, and a 256-channel tile must not be mostly padding
    const double waste = (double)((long)tiles_n * BT9 - cout_p8) / (double)cout_p8;
    // measured on SlowFast-R50 / MViT-B (profiles/r5/model_ab_quad_final.txt): 120 beats 200 beats 300 -- even on half of the
    // CUs (SlowFast res4: 128 tiles) the deeper pipeline is worth more than the idle half costs
    const long min_tiles = pv_tune("gemm9_min_tiles", 120);
    if (total < min_tiles || waste > 0.15) return PV_ERR_UNSUPPORTED;
  }
  // the pointwise form addresses voxel row m at x + m * ldx: batch items must follow each other without a gap
  const bool rows = pw && d.x_bs == (long)d.To * d.Ho * d.Wo * d.ldx;
  // variant (pv_tune "gemm9_var"): 0 = a phase's two DMAs before its first barrier; 2 = among its MFMAs (default from six K
  // tiles on: +5 ... 12 % on long reductions, -6 % on four-K-tile layers).  Measured and removed in round 5 (profiles/r5/):
  // reading the next K tile's first channel half a phase early (bit 0: 5-12 % slower everywhere), every wave of a half issuing
  // behind a different MFMA (bit 2: spills inside the loop), and the reduction split over workgroups for layers with fewer
  // than 200 tiles (pairwise sums through L2 with tickets: correct and bit-reproducible, but 20-45 % SLOWER than the 128 x 128
  // kernel on SlowFast's res4 / res5 shapes -- the implicit-GEMM form of it kept an accumulator block in scratch).
  int var = pv_tune("gemm9_var", -1);
  if (var < 0) var = K >= 384 ? 2 : 0;
#define PV9_GO(PWv, YFv)                                                  \
  if (var == 2) return launch9<PWv, YFv, 2>(d, tiles_n, total, s);        \
  return launch9<PWv, YFv, 0>(d, tiles_n, total, s);
  if (d.y_f32) {
    if (rows) { PV9_GO(true, true) } else { PV9_GO(false, true) }
  } else {
    if (rows) { PV9_GO(true, false) } else { PV9_GO(false, false) }
  }
#undef PV9_GO
}
