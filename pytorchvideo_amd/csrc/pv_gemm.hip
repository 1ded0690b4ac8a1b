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
constexpr int BM = 128;   // voxels per tile (64 in the VT = 1 variant)
constexpr int BN = 128;   // output channels per tile
constexpr int kMaxTaps = 512;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS row rho of the weight tile holds output channel chi(rho): within each 32-row MFMA tile,
// accumulator register r of lane-half hi is MFMA row (r&3) + 8*(r>>2) + 4*hi -> channel 16*hi + r
__device__ __forceinline__ int chi(int rho) {
  return (rho & ~31) + 16 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 3) + (rho & 3);
}

struct GemmGeom {   // per-thread staging geometry of one output tile
  long w_off[4];    // element offset of the weight row carried by item j, or -1
  long x_off[4];    // PW: element offset of the voxel row, or -1 ; general: clip offset or -1
  int x_t[4], x_h[4], x_w[4];
  long x_sp[4];     // temporal fast path: element offset of (t*st - pt, h, w) inside the clip (may be negative: masked by x_t)
  long m0;
  int n0;
  int rot;          // temporal-tap rotation of this tile (tap_rot kernels), else 0
};

// Persistent workgroups (two per CU): each walks a strided list of output tiles.  While the last K step
// of a tile is multiplied and its epilogue runs, the LDS-DMA for the first K step of the NEXT tile is
// already in flight, and the next tile's address arithmetic is done under the current tile's MFMAs --
// for the short-K layers of MViT (K = 384: six steps) the per-tile prologue / epilogue is otherwise
// as long as the K loop itself.
// VT = voxel tiles (32 rows) per wave: 2 -> 128 x 128 output tiles; 1 -> 64 x 128, for layers whose tile count
// sits just above a multiple of the resident workgroups (the second, nearly empty round costs a full tile time).
template <bool PW, int VT, int ABL = 0>   // ABL: ablation builds for tools/bench_gemm.py (1 no loads, 2 no MFMA, 3 no epilogue)
__global__ __launch_bounds__(kThreads, 2) void gemm_glds_kernel(const pv_conv3d_desc d, int tiles_n, int total_tiles,
                                                                float inv_cin, int tap_rot) {
  constexpr int BK = 64;
  constexpr int TILE_ELEMS = 128 * BK;                          // one operand tile (elements)
  constexpr int NJ = TILE_ELEMS * 2 / (kThreads * 16);          // 16-byte items per thread per operand tile (4)
  constexpr int NJX = NJ * VT / 2;                               // ... of the voxel tile (64 * VT rows)
  constexpr int BMV = 64 * VT;                                  // voxels per output tile
  constexpr int CPR = BK / 8;                                   // 16-byte chunks per tile row
  // XOR swizzle of the chunk index that makes ds_read_b128 conflict-free (rows are 128 bytes)
  auto swz = [](int row) { return (row >> 1) & 7; };
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * 2 * TILE_ELEMS];   // [buf][W | X][128][64]
  __shared__ int s_tap[PW ? 1 : kMaxTaps];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;

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
      s_tap[t] = (dt * (d.dil_t > 1 ? d.dil_t : 1)) | ((dh * (d.dil_h > 1 ? d.dil_h : 1)) << 8) |
                 (((r - dh * d.kw) * (d.dil_w > 1 ? d.dil_w : 1)) << 16);   // voxel offsets, dilation folded in
    }
    __syncthreads();
  }

  // item g = j*256 + tid -> (row g/8, LDS position g%8); the logical chunk it carries is tile-independent
  int kch[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int g = j * kThreads + tid;
    kch[j] = (g % CPR) ^ swz(g / CPR);
  }

  // XCD-aware tile order (bijective for any tile count): work item `it` -> tile
  auto geom_of = [&](int it, GemmGeom& gg) {
    const int xcd = it & 7, slot = it >> 3;
    const int qn = total_tiles >> 3, rn = total_tiles & 7;
    const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
    const int tile_n = tile % tiles_n;
    gg.m0 = (long)(tile / tiles_n) * BMV;
    gg.n0 = tile_n * BN;
    gg.rot = 0;
    if constexpr (!PW) {
      // (kt,1,1) = (3,1,1) convs (SlowFast's temporal conv_a, models/resnet.py:98-105): the tiles that run concurrently on an
      // XCD lie on different frames of the same clips, and each input frame is the operand of THREE of them (as tap t-1, t,
      // t+1) at three different moments of their K loops -- by then it has left the XCD's 4 MB L2 (profiles/r4/calib_fetch.md:
      // L2 fills = 3.0x the input, and the layer runs 1.4-1.6x slower than the plain GEMM with the same K that really reads
      // 3x the bytes).  Rotating the tap order by the tile's frame index makes every tile read, in loop phase j, the one
      // frame of its three with index == j (mod 3): the three consumers of a frame fetch it at the same time.
      if ((tap_rot & 1) && gg.m0 < M) {
        const long sp0 = gg.m0 - (long)((unsigned)gg.m0 / (unsigned)S_out) * S_out;
        const int to0 = (int)((unsigned)sp0 / (unsigned)(d.Ho * d.Wo));
        gg.rot = (1 + 2 * to0) % 3;      // == (1 - to0) mod 3
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = (j * kThreads + tid) / CPR;
      const int n = gg.n0 + chi(row);
      gg.w_off[j] = n < d.cout ? (long)n * K : -1;
      const long m = gg.m0 + row;
      if (j < NJX && m < M) {
        // M < 2^31 (host check): 32-bit divisions, an order of magnitude cheaper than the 64-bit sequence
        const long b = (long)((unsigned)m / (unsigned)S_out);
        const long sp = m - b * S_out;
        if constexpr (PW) {
          gg.x_off[j] = b * d.x_bs + sp * d.ldx;
          gg.x_sp[j] = 0;
          gg.x_t[j] = gg.x_h[j] = gg.x_w[j] = 0;
        } else {
          const int to = (int)((unsigned)sp / (unsigned)(d.Ho * d.Wo));
          const int r2 = (int)(sp - (long)to * d.Ho * d.Wo);
          const int ho = r2 / d.Wo;
          gg.x_off[j] = b * d.x_bs;
          gg.x_sp[j] = ((long)((to * d.st - d.pt) * d.Hi + (ho * d.sh - d.ph)) * d.Wi + ((r2 - ho * d.Wo) * d.sw - d.pw)) * d.ldx;
          gg.x_t[j] = to * d.st - d.pt;
          gg.x_h[j] = ho * d.sh - d.ph;
          gg.x_w[j] = (r2 - ho * d.Wo) * d.sw - d.pw;
        }
      } else {
        gg.x_off[j] = -1;
        gg.x_sp[j] = 0;
        gg.x_t[j] = gg.x_h[j] = gg.x_w[j] = 0;
      }
    }
  };

  // Source selection is done with bit masks, not `?:` -- the compiler would turn a select between two
  // pointers into two exec-masked LDS-DMA instructions.
  const unsigned long zaddr = (unsigned long)zero;
  auto pick = [&](bool ok, const bf16_t* p) -> const bf16_t* {
    const unsigned long m = 0ul - (unsigned long)ok;
    return reinterpret_cast<const bf16_t*>(((unsigned long)p & m) | (zaddr & ~m));
  };
  auto stage = [&](int buf, const GemmGeom& gg, int ks) {
    bf16_t* wb = smem + buf * 2 * TILE_ELEMS;
    bf16_t* xb = wb + TILE_ELEMS;
    const int k0 = ks * BK;
    // Temporal fast path ((kt,1,1) convs whose input width is a multiple of the K step: SlowFast's conv_a, the lateral
    // fusions; flag bit 1): a K step lies inside ONE tap, so the tap, its weight-column shift and its frame offset are
    // wave-uniform and computed once per step -- the per-chunk tap decoding (a float multiply, an LDS look-up, three bounds
    // tests and a 64-bit multiply chain) was what kept the (3,1,1) layer 1.3x behind the plain GEMM of the same K
    // (profiles/r4/calib_fetch_times*.txt: B vs C).
    const bool tmode = !PW && (tap_rot & 2);
    // ... and its generalisation to any tap shape (flag bit 2, pv_tune "gemm_umode"; round 5: validated through the model
    // suites, on by default): with the input width a multiple of the K step the tap (dt, dh, dw) of a step is still wave-uniform;
    // what stays per chunk are the three bounds tests of the row's window position.
    const bool umode = !PW && (tap_rot & 4);
    int tap0_u = 0, tap_u = 0, udt = 0, udh = 0, udw = 0;
    long tap_off_u = 0;
    if (tmode) {
      tap0_u = (int)(((float)k0 + 0.5f) * inv_cin);
      tap_u = tap0_u + gg.rot;
      tap_u -= tap_u >= 3 && gg.rot ? 3 : 0;
      tap_off_u = (long)(tap_u * (d.dil_t > 1 ? d.dil_t : 1)) * d.Hi * d.Wi * d.ldx;
    } else if (umode) {
      tap0_u = tap_u = (int)(((float)k0 + 0.5f) * inv_cin);
      const int tp = s_tap[tap_u < taps ? tap_u : 0];
      udt = tp & 255;
      udh = (tp >> 8) & 255;
      udw = tp >> 16;
      tap_off_u = ((long)(udt * d.Hi + udh) * d.Wi + udw) * d.ldx;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = k0 + kch[j] * 8;
      const bool kok = k < K;
      int kw_col = k, tap = 0, tap0 = 0;     // weight column / tap whose operand this chunk carries / tap position in the loop
      if constexpr (!PW) {
        if (tmode || umode) {
          tap0 = tap0_u;
          tap = tap_u;
        } else {
          tap0 = (int)(((float)k + 0.5f) * inv_cin);
          tap = tap0 + gg.rot;                 // (rot != 0 only for three taps)
          tap -= tap >= 3 && gg.rot ? 3 : 0;
        }
        kw_col = k + (tap - tap0) * d.cin;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)pick(kok && gg.w_off[j] >= 0, Wt + (gg.w_off[j] >= 0 ? gg.w_off[j] : 0) + kw_col),
                                       (lptr_t)(wb + (j * kThreads + wave * 64) * 8), 16, 0, 0);
      if (j >= NJX) continue;   // the voxel tile has fewer rows than the filter tile in the VT = 1 variant
      bool xok = kok && gg.x_off[j] >= 0;
      long xo = gg.x_off[j] >= 0 ? gg.x_off[j] : 0;
      if constexpr (PW) {
        xo += k;
      } else if (tmode) {
        const int ti = gg.x_t[j] + tap * (d.dil_t > 1 ? d.dil_t : 1);
        xok = xok && (unsigned)ti < (unsigned)d.Ti;
        xo += gg.x_sp[j] + tap_off_u + (k - tap0 * d.cin);
      } else if (umode) {
        const int ti = gg.x_t[j] + udt, hh = gg.x_h[j] + udh, ww = gg.x_w[j] + udw;
        xok = xok && (unsigned)ti < (unsigned)d.Ti && (unsigned)hh < (unsigned)d.Hi && (unsigned)ww < (unsigned)d.Wi;
        xo += gg.x_sp[j] + tap_off_u + (k - tap0 * d.cin);
      } else {
        const int tp = s_tap[tap < taps ? tap : 0];
        const int ti = gg.x_t[j] + (tp & 255), hh = gg.x_h[j] + ((tp >> 8) & 255), ww = gg.x_w[j] + (tp >> 16);
        xok = xok && (unsigned)ti < (unsigned)d.Ti && (unsigned)hh < (unsigned)d.Hi && (unsigned)ww < (unsigned)d.Wi;
        xo += ((long)(ti * d.Hi + hh) * d.Wi + ww) * d.ldx + (k - tap0 * d.cin);
      }
      __builtin_amdgcn_global_load_lds((gptr_t)pick(xok, X + xo), (lptr_t)(xb + (j * kThreads + wave * 64) * 8), 16, 0, 0);
    }
  };

  // read-side swizzle of this lane's fragment rows (fixed for the whole kernel)
  int a_row[2], b_row[VT];
#pragma unroll
  for (int t = 0; t < 2; ++t) a_row[t] = wn * 64 + t * 32 + l31;
#pragma unroll
  for (int t = 0; t < VT; ++t) b_row[t] = wm * 32 * VT + t * 32 + l31;
  // s_waitcnt simm16 on gfx9: vmcnt = bits [3:0] | [15:14]; expcnt [6:4] and lgkmcnt [11:8] left at "no wait"
  constexpr auto vm = [](int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned kOOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      d.y, 0, (int)((unsigned)d.B * (unsigned)d.y_bs * (d.y_f32 ? 4u : 2u)), 0x00020000);
  bool first_wait_stores = false;   // wave-uniform

  const int nk = (K + BK - 1) / BK;
  GemmGeom cur, nxt;
  int it = blockIdx.x;
  geom_of(it, cur);
  stage(0, cur, 0);
  int gs = 0;   // global K-step counter: LDS buffer parity runs on across tiles
  for (; it < total_tiles; it += gridDim.x) {
    const bool has_next = it + (int)gridDim.x < total_tiles;
    if (has_next) geom_of(it + gridDim.x, nxt);

    f32x16 acc[2][VT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int v = 0; v < VT; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][v][r] = 0.f;

    for (int ks = 0; ks < nk; ++ks, ++gs) {
      // this wave's share of the step's tiles has landed; the barrier publishes the buffer and proves
      // that every wave is done reading the other one, which is refilled next
      if (first_wait_stores) {
        // the LDS-DMA of this step was issued BEFORE the previous tile's stores (in-order return):
        // leave exactly those stores in flight
        if (d.y_f32) __builtin_amdgcn_s_waitcnt(vm(8 * VT));
        else __builtin_amdgcn_s_waitcnt(vm(4 * VT));
        first_wait_stores = false;
      } else {
        __builtin_amdgcn_s_waitcnt(vm(0));
      }
      __builtin_amdgcn_s_barrier();
      if constexpr (ABL != 1) {
      if (ks + 1 < nk) stage((gs + 1) & 1, cur, ks + 1);
      else if (has_next) stage((gs + 1) & 1, nxt, 0);   // first step of the next tile: in flight during the epilogue
      }
      const bf16_t* wb = smem + (gs & 1) * 2 * TILE_ELEMS;
      const bf16_t* xb = wb + TILE_ELEMS;
      // fragment reads are software-pipelined one 16-deep sub-step ahead of the MFMAs that use them
      bf16x8 af[2][2], bfr[2][VT];
      auto read_frags = [&](int slot, int s) {
        const int c = 2 * s + hi;
#pragma unroll
        for (int t = 0; t < 2; ++t)
          af[slot][t] = *reinterpret_cast<const bf16x8*>(wb + a_row[t] * BK + ((c ^ swz(a_row[t])) << 3));
#pragma unroll
        for (int t = 0; t < VT; ++t)
          bfr[slot][t] = *reinterpret_cast<const bf16x8*>(xb + b_row[t] * BK + ((c ^ swz(b_row[t])) << 3));
      };
      if constexpr (ABL == 2) continue;
      read_frags(0, 0);
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        if (s + 1 < BK / 16) read_frags((s + 1) & 1, s + 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int v = 0; v < VT; ++v)
            acc[a][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][a], bfr[s & 1][v], acc[a][v], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }

    // ---- epilogue: lane owns channels cb..cb+15 of voxel m for every (channel tile a, voxel tile v) ----
    // Output goes through a buffer descriptor: masked stores get an out-of-range offset and are dropped
    // by the hardware, so every wave issues EXACTLY kStores store instructions per tile -- the next
    // tile's first wait can then be a counted vmcnt that leaves them in flight.
    // every load of the epilogue (residual rows, scale / shift) is consumed before the first store is
    // issued -- a load consumed after a store would make the compiler drain the store queue
    if constexpr (ABL == 3) {
      float keep = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) keep += acc[a][v][r];
      if (keep == 1.2345e-30f) static_cast<float*>(d.y)[0] = keep;   // keeps the accumulators live
      first_wait_stores = false;
      cur = nxt;
      continue;
    }
    long e_b[VT], e_sp[VT];
    bool e_ok[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) {
      const long m = cur.m0 + wm * 32 * VT + v * 32 + l31;
      e_ok[v] = m < M;
      const long mm = e_ok[v] ? m : 0;
      e_b[v] = (long)((unsigned)mm / (unsigned)S_out);
      e_sp[v] = mm - e_b[v] * S_out;
    }
    // finish the accumulators in place, one channel tile at a time (scale/shift and residual loads of
    // tile a are consumed before tile a+1's are issued; no store has been issued yet) ...
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int cb = cur.n0 + wn * 64 + a * 32 + 16 * hi;
      f32x4 res[VT][2][2];   // [v][h8][half]: 8 channels as 2 x f32x4 (fp32) or 1 x 16 bytes (bf16)
      if (d.residual != nullptr) {
#pragma unroll
        for (int v = 0; v < VT; ++v)
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
      }
      // Every choice below (scale or bias only, residual kind, activation, ragged last channel tile) is
      // wave-uniform and decided once per tile around straight-line 16-element loops -- the epilogue is
      // a fixed cost per tile that rivals the MFMA time of the short-K (K = 192 / 384) MViT layers.
      // (scale pass, then shift pass: one 16-register table live at a time)
      if (d.scale != nullptr) {
        float sc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = cb + r < d.cout ? d.scale[cb + r] : 0.f;
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] *= sc[r];
      }
      if (d.shift != nullptr) {
        float sh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sh[r] = cb + r < d.cout ? d.shift[cb + r] : 0.f;
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] += sh[r];
      }
      if (d.residual != nullptr) {
        if (d.r_f32) {
#pragma unroll
          for (int v = 0; v < VT; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][v][r] += res[v][r >> 3][(r >> 2) & 1][r & 3];
        } else {
#pragma unroll
          for (int v = 0; v < VT; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][v][r] += (float)__builtin_bit_cast(bf16x8, res[v][r >> 3][0])[r & 7];
        }
      }
      if (d.act == PV_ACT_RELU) {
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] = fmaxf(acc[a][v][r], 0.f);
      } else if (d.act == PV_ACT_GELU) {
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] = pv_gelu_fast(acc[a][v][r]);
      } else if (d.act == PV_ACT_SWISH) {
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] *= pv_sigmoid(acc[a][v][r]);
      } else if (d.act == PV_ACT_SIGMOID) {
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] = pv_sigmoid(acc[a][v][r]);
      }
      if (cb + 16 > d.cout) {   // ragged last channel tile: the padding up to the 8-multiple is written as zeros
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][v][r] = cb + r < d.cout ? acc[a][v][r] : 0.f;
      }
    }
    // ... then nothing but stores
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int cb = cur.n0 + wn * 64 + a * 32 + 16 * hi;
#pragma unroll
      for (int v = 0; v < VT; ++v) {
        const unsigned yo = (unsigned)(e_b[v] * d.y_bs + e_sp[v] * d.ldy + cb);   // elements (< 2^31 bytes: host check)
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          const bool ok = e_ok[v] && cb + h8 * 8 < cout_p8;
          const int r0 = h8 * 8;
          if (d.y_f32) {
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
    first_wait_stores = true;
    cur = nxt;
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
  const int tiles_n = (int)pv_ceil_div(cout_p8, BN);
  // two persistent workgroups per CU (64 KB of LDS each)
  const long resident = 2 * 256;
  // 64-row tiles only where 128-row tiles leave more than half of the resident slots empty (one workgroup on
  // fewer than every CU): there the halved tiles double the parallelism for free.  Measured on one box: choosing
  // them whenever they pack the rounds better costs MViT-B 4 % (a 64-row tile re-stages the same 128-row filter
  // tile for half the MFMA work); this rule gains SlowFast-R50 1.4 % and MViT-B 0.7 %.
  const long t128 = pv_ceil_div(M, 128) * tiles_n;
  const int force_vt = pv_tune("gemm_vt", 0);
  // (half-height tiles where only the LAST round of 128-row tiles is thin measured -0.7 % on MViT-B in round 4 and are gone)
  const int vt = force_vt ? force_vt : (2 * t128 <= resident ? 1 : 2);
  const long tiles_m = pv_ceil_div(M, 64 * vt);
  const long total = tiles_m * tiles_n;
  if (total <= 0 || total > 0x7fffffffL || M > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
  if ((long)d.B * d.y_bs * (d.y_f32 ? 4 : 2) > 0x7fffffffL) return PV_ERR_UNSUPPORTED;   // 31-bit buffer offsets
  const float inv_cin = 1.0f / (float)d.cin;
  // temporal-tap rotation (see geom_of): pure (3,1,1) convs, stride / dilation 1, whole tiles inside one frame
  int tap_rot = (!pw && d.kt == 3 && d.kh == 1 && d.kw == 1 && d.st == 1 && d.pt == 1 && d.dil_t <= 1 && d.To >= 3 &&
                 ((long)d.Ho * d.Wo) % (64 * vt) == 0 && pv_tune("gemm_tap_rot", 1)) ? 1 : 0;
  // flag bit 1: the temporal fast path of stage() -- (kt,1,1) taps, unit spatial stride, no spatial padding, input width a
  // multiple of the 64-wide K step (a step then lies inside one tap)
  if (!pw && d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && d.cin % 64 == 0 && d.kt > 1 &&
      pv_tune("gemm_tmode", 1))
    tap_rot |= 2;
  else if (!pw && taps > 1 && d.cin % 64 == 0 && pv_tune("gemm_umode", 1))
    tap_rot = 4;      // uniform-tap staging does not apply the rotation: drop bit 0 rather than compute a rotation nobody uses
  dim3 grid((unsigned)(total < resident ? total : resident)), block(kThreads);
#ifdef PV_DEV_ABLATION   // timing builds that skip loads / MFMAs / the epilogue (WRONG results): development variant of the library only
  const int abl = pv_tune("gemm_abl", 0);
  if (abl && vt == 2) {
    if (abl == 1) { if (pw) PV_LAUNCH((gemm_glds_kernel<true, 2, 1>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot);
                    else PV_LAUNCH((gemm_glds_kernel<false, 2, 1>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot); }
    if (abl == 2) { if (pw) PV_LAUNCH((gemm_glds_kernel<true, 2, 2>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot);
                    else PV_LAUNCH((gemm_glds_kernel<false, 2, 2>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot); }
    if (abl == 3) { if (pw) PV_LAUNCH((gemm_glds_kernel<true, 2, 3>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot);
                    else PV_LAUNCH((gemm_glds_kernel<false, 2, 3>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot); }
    PV_LAUNCH_CHECK();
    return PV_OK;
  }
#endif
  if (vt == 1) {
    if (pw) PV_LAUNCH((gemm_glds_kernel<true, 1>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot);
    else PV_LAUNCH((gemm_glds_kernel<false, 1>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot);
  } else {
    if (pw) PV_LAUNCH((gemm_glds_kernel<true, 2>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot);
    else PV_LAUNCH((gemm_glds_kernel<false, 2>), grid, block, 0, s, d, tiles_n, (int)total, inv_cin, tap_rot);
  }
  PV_LAUNCH_CHECK();
  return PV_OK;
}
