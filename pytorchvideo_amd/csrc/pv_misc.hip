// Small HBM-bound kernels of the forward path: squeeze-excitation gate, pooling, layout
// ingest/egress, LayerNorm, row softmax / mean, positional-encoding add, residual add.
// All of them move 8-channel chunks (16 B of bf16) with lanes running over channels first.
#include <float.h>
#include "pv_common.h"

namespace {

constexpr int kThreads = 256;

// --------------------------------------------------------------------- SE gate
// one block per batch item; psum[b][blk][c_p] -> gate[b][c_p]
__global__ __launch_bounds__(kThreads) void se_gate_kernel(const pv_se_gate_desc d) {
  extern __shared__ float s[];
  float* s_mean = s;                 // [c_p]
  float* s_hid = s + d.c_p;          // [cr]
  float* s_part = s + d.c_p + d.cr;  // [groups][c_p]
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* ps = d.psum + (long)b * d.nblk * d.c_p;
  // rows of the partial-sum table are split over `groups` thread groups; each group walks its
  // rows with coalesced loads over the channels (4 independent loads in flight per thread)
  const int cw = d.c_p < kThreads ? d.c_p : kThreads;   // channels covered per pass
  const int groups = kThreads / cw;
  const int g = tid / cw, cl = tid - g * cw;
  for (int c0 = 0; c0 < d.c_p; c0 += cw) {
    const int c = c0 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (g < groups && c < d.c_p) {
      int k = g;
      for (; k + 3 * groups < d.nblk; k += 4 * groups) {
        a0 += ps[(long)k * d.c_p + c];
        a1 += ps[(long)(k + groups) * d.c_p + c];
        a2 += ps[(long)(k + 2 * groups) * d.c_p + c];
        a3 += ps[(long)(k + 3 * groups) * d.c_p + c];
      }
      for (; k < d.nblk; k += groups) a0 += ps[(long)k * d.c_p + c];
      s_part[g * d.c_p + c] = (a0 + a1) + (a2 + a3);
    }
  }
  __syncthreads();
  for (int c = tid; c < d.c_p; c += kThreads) {
    float a = 0.f;
    for (int gg = 0; gg < groups; ++gg) a += s_part[gg * d.c_p + c];
    s_mean[c] = a * d.inv_count;
  }
  __syncthreads();
  // fc1 + relu: one wave per hidden unit
  const int lane = tid & 63, wave = tid >> 6;
  for (int r = wave; r < d.cr; r += kThreads / 64) {
    float a = 0.f;
    for (int c = lane; c < d.C; c += 64) a += d.w1[(long)r * d.C + c] * s_mean[c];
    a = pv_wave_sum(a);
    if (lane == 0) s_hid[r] = fmaxf(a + (d.b1 ? d.b1[r] : 0.f), 0.f);
  }
  __syncthreads();
  for (int c = tid; c < d.c_p; c += kThreads) {
    float g2 = 0.f;
    if (c < d.C) {
      float a = d.b2 ? d.b2[c] : 0.f;
      for (int r = 0; r < d.cr; ++r) a += d.w2[(long)c * d.cr + r] * s_hid[r];
      g2 = pv_sigmoid(a);
    }
    d.gate[(long)b * d.c_p + c] = g2;
  }
}

// --------------------------------------------------------------------- pooling
// small windows: one thread per (output voxel, chunk)
template <typename T>
__global__ __launch_bounds__(kThreads) void pool_direct_kernel(const pv_pool3d_desc d, long total) {
  const int CG = pv_round_up(d.C, 8) / 8;
  const long id = (long)blockIdx.x * kThreads + threadIdx.x;
  if (id >= total) return;
  const int cg = (int)(id % CG);
  long v = id / CG;
  const int wo = (int)(v % d.Wo); v /= d.Wo;
  const int ho = (int)(v % d.Ho); v /= d.Ho;
  const int to = (int)(v % d.To);
  const int b = (int)(v / d.To);
  const T* X = static_cast<const T*>(d.x) + (long)b * d.x_bs + (long)d.n_prefix * d.ldx + cg * 8;
  float a[8];
  const bool is_max = d.mode == PV_POOL_MAX;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = is_max ? -FLT_MAX : 0.f;
  for (int dt = 0; dt < d.kt; ++dt) {
    const int ti = to * d.st - d.pt + dt;
    if ((unsigned)ti >= (unsigned)d.Ti) continue;
    for (int dh = 0; dh < d.kh; ++dh) {
      const int hi = ho * d.sh - d.ph + dh;
      if ((unsigned)hi >= (unsigned)d.Hi) continue;
      for (int dw = 0; dw < d.kw; ++dw) {
        const int wi = wo * d.sw - d.pw + dw;
        if ((unsigned)wi >= (unsigned)d.Wi) continue;
        Chunk8<T> c;
        c.load(X + ((long)(ti * d.Hi + hi) * d.Wi + wi) * d.ldx);
        float f[8];
        c.to_f32(f);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = is_max ? fmaxf(a[j], f[j]) : a[j] + f[j];
      }
    }
  }
  if (!is_max) {
    const float inv = 1.f / (float)(d.kt * d.kh * d.kw);  // count_include_pad=True
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] *= inv;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (cg * 8 + j >= d.C) a[j] = 0.f;
  Chunk8<T> o;
  o.from_f32(a);
  o.store(static_cast<T*>(d.y) + (long)b * d.y_bs + (long)d.n_prefix * d.ldy +
          ((long)(to * d.Ho + ho) * d.Wo + wo) * d.ldy + cg * 8);
}

// The windows the models use between layers -- 3x3x3 (MViT's skip-path max pool, layers/attention.py:677-679,720-727) and
// 1x3x3 (SlowFast / ResNet stem pools, models/stem.py:98-104) -- with the taps unrolled: every load of a temporal slice is issued
// unconditionally from a clamped address (no branch per tap: the runtime loops above wait for each load behind its own bounds
// test), validity is applied as a select.  One output voxel x 8-channel chunk per thread, as above.
template <typename T, int KT, int KH, int KW>
__global__ __launch_bounds__(kThreads) void pool_window_kernel(const pv_pool3d_desc d, long total) {
  const int CG = pv_round_up(d.C, 8) / 8;
  const long id = (long)blockIdx.x * kThreads + threadIdx.x;
  if (id >= total) return;
  const int cg = (int)(id % CG);
  long v = id / CG;
  const int wo = (int)(v % d.Wo); v /= d.Wo;
  const int ho = (int)(v % d.Ho); v /= d.Ho;
  const int to = (int)(v % d.To);
  const int b = (int)(v / d.To);
  const T* X = static_cast<const T*>(d.x) + (long)b * d.x_bs + (long)d.n_prefix * d.ldx + cg * 8;
  const bool is_max = d.mode == PV_POOL_MAX;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = is_max ? -FLT_MAX : 0.f;
  const int h0 = ho * d.sh - d.ph, w0 = wo * d.sw - d.pw;
  int hoff[KH], woff[KW];
  bool hok[KH], wok[KW];
#pragma unroll
  for (int dh = 0; dh < KH; ++dh) {
    const int hi = h0 + dh;
    hok[dh] = (unsigned)hi < (unsigned)d.Hi;
    hoff[dh] = (hok[dh] ? hi : 0) * d.Wi;
  }
#pragma unroll
  for (int dw = 0; dw < KW; ++dw) {
    const int wi = w0 + dw;
    wok[dw] = (unsigned)wi < (unsigned)d.Wi;
    woff[dw] = wok[dw] ? wi : 0;
  }
#pragma unroll
  for (int dt = 0; dt < KT; ++dt) {
    const int ti = to * d.st - d.pt + dt;
    const bool tok = (unsigned)ti < (unsigned)d.Ti;
    const T* P = X + (long)(tok ? ti : 0) * d.Hi * d.Wi * d.ldx;
    Chunk8<T> c[KH][KW];
#pragma unroll
    for (int dh = 0; dh < KH; ++dh)
#pragma unroll
      for (int dw = 0; dw < KW; ++dw) c[dh][dw].load(P + (long)(hoff[dh] + woff[dw]) * d.ldx);
#pragma unroll
    for (int dh = 0; dh < KH; ++dh)
#pragma unroll
      for (int dw = 0; dw < KW; ++dw) {
        const bool ok = tok && hok[dh] && wok[dw];
        float f[8];
        c[dh][dw].to_f32(f);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = is_max ? fmaxf(a[j], ok ? f[j] : -FLT_MAX) : a[j] + (ok ? f[j] : 0.f);
      }
  }
  if (!is_max) {
    const float inv = 1.f / (float)(KT * KH * KW);  // count_include_pad=True
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] *= inv;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (cg * 8 + j >= d.C) a[j] = 0.f;
  Chunk8<T> o;
  o.from_f32(a);
  o.store(static_cast<T*>(d.y) + (long)b * d.y_bs + (long)d.n_prefix * d.ldy +
          ((long)(to * d.Ho + ho) * d.Wo + wo) * d.ldy + cg * 8);
}

// large windows: one block per output voxel (and chunk slab), taps split over threads
template <typename T>
__global__ __launch_bounds__(kThreads) void pool_reduce_kernel(const pv_pool3d_desc d, int cgb) {
  __shared__ float s_red[kThreads * 8];
  const int CG = pv_round_up(d.C, 8) / 8;
  const int tid = threadIdx.x;
  const int splits = kThreads / cgb;
  const int cgl = tid % cgb, sp = tid / cgb;
  const int cg = blockIdx.y * cgb + cgl;
  long v = blockIdx.x;
  const int wo = (int)(v % d.Wo); v /= d.Wo;
  const int ho = (int)(v % d.Ho); v /= d.Ho;
  const int to = (int)(v % d.To);
  const int b = (int)(v / d.To);
  const bool is_max = d.mode == PV_POOL_MAX;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = is_max ? -FLT_MAX : 0.f;
  const int taps = d.kt * d.kh * d.kw;
  if (cg < CG && sp < splits) {
    const T* X = static_cast<const T*>(d.x) + (long)b * d.x_bs + (long)d.n_prefix * d.ldx + cg * 8;
    for (int t = sp; t < taps; t += splits) {
      const int dt = t / (d.kh * d.kw);
      const int r = t - dt * d.kh * d.kw;
      const int dh = r / d.kw, dw = r - dh * d.kw;
      const int ti = to * d.st - d.pt + dt, hi = ho * d.sh - d.ph + dh, wi = wo * d.sw - d.pw + dw;
      if ((unsigned)ti >= (unsigned)d.Ti || (unsigned)hi >= (unsigned)d.Hi || (unsigned)wi >= (unsigned)d.Wi)
        continue;
      Chunk8<T> c;
      c.load(X + ((long)(ti * d.Hi + hi) * d.Wi + wi) * d.ldx);
      float f[8];
      c.to_f32(f);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = is_max ? fmaxf(a[j], f[j]) : a[j] + f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s_red[tid * 8 + j] = a[j];
  __syncthreads();
  if (sp == 0 && cg < CG) {
    for (int s2 = 1; s2 < splits; ++s2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float o = s_red[(s2 * cgb + cgl) * 8 + j];
        a[j] = is_max ? fmaxf(a[j], o) : a[j] + o;
      }
    }
    if (!is_max) {
      const float inv = 1.f / (float)taps;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] *= inv;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cg * 8 + j >= d.C) a[j] = 0.f;
    Chunk8<T> o;
    o.from_f32(a);
    o.store(static_cast<T*>(d.y) + (long)b * d.y_bs + (long)d.n_prefix * d.ldy +
            ((long)(to * d.Ho + ho) * d.Wo + wo) * d.ldy + cg * 8);
  }
}

// copy the n_prefix leading rows (cls token) of every batch item
template <typename T>
__global__ void pool_prefix_kernel(const pv_pool3d_desc d) {
  const int CG = pv_round_up(d.C, 8) / 8;
  const int total = d.B * d.n_prefix * CG;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int cg = id % CG;
  const int r = (id / CG) % d.n_prefix;
  const int b = id / (CG * d.n_prefix);
  Chunk8<T> c;
  c.load(static_cast<const T*>(d.x) + (long)b * d.x_bs + (long)r * d.ldx + cg * 8);
  c.store(static_cast<T*>(d.y) + (long)b * d.y_bs + (long)r * d.ldy + cg * 8);
}

// --------------------------------------------------------------------- layout
template <typename S> __device__ __forceinline__ float ld_as_f32(const S* p) { return (float)*p; }

// NCDHW (contiguous) -> NDHWC padded.  Thread = (voxel fastest, chunk).
template <typename S, typename T>
__global__ __launch_bounds__(kThreads) void ingest_kernel(const pv_layout_desc d, long nvox) {
  const int CG = d.c_p / 8;
  const long id = (long)blockIdx.x * kThreads + threadIdx.x;
  if (id >= nvox * CG) return;
  const long vox = id % nvox;
  const int cg = (int)(id / nvox);
  const long S3 = (long)d.T * d.H * d.W;
  const int b = (int)(vox / S3);
  const long sp = vox - (long)b * S3;
  // source voxel: frame t of the destination reads frame t_index[t] of the source clip
  const long HW = (long)d.H * d.W;
  const int t = (int)(sp / HW);
  const long S3s = (long)(d.t_index ? d.src_T : d.T) * HW;
  const long sps = (long)(d.t_index ? d.t_index[t] : t) * HW + (sp - (long)t * HW);
  const S* src = static_cast<const S*>(d.src) + (long)b * d.C * S3s + sps;
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    float v = c < d.C ? ld_as_f32(src + (long)c * S3s) : 0.f;
    if (c < d.C && d.ch_scale) v = v * d.ch_scale[c] + (d.ch_shift ? d.ch_shift[c] : 0.f);
    f[j] = v;
  }
  Chunk8<T> o;
  o.from_f32(f);
  o.store(static_cast<T*>(d.dst) + (long)b * d.bs + sp * d.ld + cg * 8);
}

// NCDHW -> NDHWC with 4 channels per voxel (first-layer layout of pv_stem.hip): thread = voxel
template <typename S>
__global__ __launch_bounds__(kThreads) void ingest_c4_kernel(const pv_layout_desc d, long nvox) {
  const long vox = (long)blockIdx.x * kThreads + threadIdx.x;
  if (vox >= nvox) return;
  const long S3 = (long)d.T * d.H * d.W;
  const long b = vox / S3;
  const long sp = vox - b * S3;
  const long HW = (long)d.H * d.W;
  const int t = (int)(sp / HW);
  const long S3s = (long)(d.t_index ? d.src_T : d.T) * HW;
  const long sps = (long)(d.t_index ? d.t_index[t] : t) * HW + (sp - (long)t * HW);
  const S* src = static_cast<const S*>(d.src) + b * d.C * S3s + sps;
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = j < d.C ? ld_as_f32(src + (long)j * S3s) : 0.f;
    if (j < d.C && d.ch_scale) v = v * d.ch_scale[j] + (d.ch_shift ? d.ch_shift[j] : 0.f);
    o[j] = (bf16_t)v;
  }
  *reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(d.dst) + b * d.bs + sp * 4) = o;
}

// The same with EIGHT W-adjacent voxels per thread (the frame size is a multiple of 8 voxels and the source planes are
// 16-byte aligned: every BASELINE geometry): per channel plane one 16-byte (bf16) / 8-byte (uint8) / 2 x 16-byte
// (fp32) load instead of eight scalar ones, and 64 contiguous bytes stored -- the scalar kernel moves 2 bytes per
// load instruction and reaches 2.6 TB/s (138 us for X3D-M's 32 clips, 3.8 % of its forward).
template <typename S>
__global__ __launch_bounds__(kThreads) void ingest_c4_vec8_kernel(const pv_layout_desc d, long ngroups) {
  const long grp = (long)blockIdx.x * kThreads + threadIdx.x;
  if (grp >= ngroups) return;
  const long HW = (long)d.H * d.W;
  const long S3 = (long)d.T * HW;
  const long vox = grp * 8;
  const long b = vox / S3;
  const long sp = vox - b * S3;          // multiple of 8, and so is HW: the eight voxels share frame t
  const int t = (int)(sp / HW);
  const long S3s = (long)(d.t_index ? d.src_T : d.T) * HW;
  const long sps = (long)(d.t_index ? d.t_index[t] : t) * HW + (sp - (long)t * HW);
  const S* src = static_cast<const S*>(d.src) + b * d.C * S3s + sps;
  float f[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < d.C) {
      const S* p = src + (long)j * S3s;
      if constexpr (sizeof(S) == 2) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[j][i] = (float)v[i];
      } else if constexpr (sizeof(S) == 4) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[j][i] = lo[i]; f[j][4 + i] = hi[i]; }
      } else {
        typedef unsigned char u8x8 __attribute__((ext_vector_type(8)));
        const u8x8 v = *reinterpret_cast<const u8x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[j][i] = (float)v[i];
      }
      if (d.ch_scale) {
        const float a = d.ch_scale[j], c = d.ch_shift ? d.ch_shift[j] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[j][i] = f[j][i] * a + c;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[j][i] = 0.f;
    }
  }
  bf16_t* dst = static_cast<bf16_t*>(d.dst) + b * d.bs + sp * 4;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {     // two voxels = one 16-byte chunk of the first-layer layout
    const bf16x8 o = {(bf16_t)f[0][i], (bf16_t)f[1][i], (bf16_t)f[2][i], (bf16_t)f[3][i],
                      (bf16_t)f[0][i + 1], (bf16_t)f[1][i + 1], (bf16_t)f[2][i + 1], (bf16_t)f[3][i + 1]};
    *reinterpret_cast<bf16x8*>(dst + i * 4) = o;
  }
}

template <typename T, typename S>
__global__ __launch_bounds__(kThreads) void egress_kernel(const pv_layout_desc d, long nvox) {
  const int CG = d.c_p / 8;
  const long id = (long)blockIdx.x * kThreads + threadIdx.x;
  if (id >= nvox * CG) return;
  const long vox = id % nvox;
  const int cg = (int)(id / nvox);
  const long S3 = (long)d.T * d.H * d.W;
  const int b = (int)(vox / S3);
  const long sp = vox - (long)b * S3;
  Chunk8<T> c;
  c.load(static_cast<const T*>(d.src) + (long)b * d.bs + sp * d.ld + cg * 8);
  float f[8];
  c.to_f32(f);
  S* dst = static_cast<S*>(d.dst) + (long)b * d.C * S3 + sp;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = cg * 8 + j;
    if (ch < d.C) dst[(long)ch * S3] = (S)f[j];
  }
}

// --------------------------------------------------------------------- row ops
// LayerNorm: one wave per row, two-pass statistics in registers (C <= 64*8*MAXC).
template <typename TI, typename T, int MAXC>
__global__ __launch_bounds__(kThreads) void layernorm_kernel(const pv_rows_desc d) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  const int CG = pv_round_up(d.C, 8) / 8;
  const TI* x = static_cast<const TI*>(d.x) + row * d.ldx;
  float f[MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cg = lane + i * 64;
    if (cg < CG) {
      Chunk8<TI> c;
      c.load(x + cg * 8);
      c.to_f32(f[i]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[i][j];  // padding channels are zero
  }
  const float mean = pv_wave_sum(s) / (float)d.C;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (lane + i * 64) * 8 + j;
      const float dlt = (c < d.C) ? f[i][j] - mean : 0.f;
      v += dlt * dlt;
    }
  }
  const float rstd = rsqrtf(pv_wave_sum(v) / (float)d.C + d.eps);
  T* y = static_cast<T*>(d.y) + row * d.ldy;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cg = lane + i * 64;
    if (cg < CG) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        o[j] = (c < d.C) ? (f[i][j] - mean) * rstd * (d.gamma ? d.gamma[c] : 1.f) + (d.beta ? d.beta[c] : 0.f) : 0.f;
      }
      Chunk8<T> oc;
      oc.from_f32(o);
      oc.store(y + cg * 8);
    }
  }
}

// Narrow rows (C <= 256, e.g. MViT's 96/192-wide stream and per-head norms): G = 16 or 32 lanes per
// row, 4 or 2 rows per wave, so a 96-channel row keeps 12 of 16 lanes busy instead of 12 of 64.
template <typename TI, typename T, int G>
__global__ __launch_bounds__(kThreads) void layernorm16_kernel(const pv_rows_desc d) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, l16 = lane % G;
  const int CG = pv_round_up(d.C, 8) / 8;
  constexpr int RPW = 64 / G;                       // rows per wave per iteration
  const long wave_id = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const long wave_stride = (long)gridDim.x * (kThreads / 64);
  // this lane's slice of gamma / beta is the same for every row it will see: load it once.  With a
  // periodic table (g_period rows) that still holds, because the grid is a multiple of g_period and a
  // lane group's row index advances by multiples of 8 * gridDim.x.
  const int prow = d.g_period > 0 ? (int)((wave_id * RPW + sub) % d.g_period) * d.C : 0;
  float gm[8], bt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = l16 * 8 + j;
    gm[j] = (c < d.C && d.gamma) ? d.gamma[prow + c] : 1.f;
    bt[j] = (c < d.C && d.beta) ? d.beta[prow + c] : 0.f;
  }
  // grid-stride over row groups, two groups in flight per wave
  for (long g0 = wave_id; g0 * RPW < d.rows; g0 += 2 * wave_stride) {
    float f[2][8];
    long row[2];
    bool act[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      row[u] = (g0 + u * wave_stride) * RPW + sub;
      act[u] = row[u] < d.rows && l16 < CG;
      if (act[u]) {
        Chunk8<TI> c;
        c.load(static_cast<const TI*>(d.x) + row[u] * d.ldx + l16 * 8);
        c.to_f32(f[u]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[u][j] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[u][j];
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mean = s / (float)d.C;
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dlt = (l16 * 8 + j < d.C) ? f[u][j] - mean : 0.f;
        v += dlt * dlt;
      }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const float rstd = rsqrtf(v / (float)d.C + d.eps);
      if (act[u]) {
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = (l16 * 8 + j < d.C) ? (f[u][j] - mean) * rstd * gm[j] + bt[j] : 0.f;
        Chunk8<T> oc;
        oc.from_f32(o8);
        oc.store(static_cast<T*>(d.y) + row[u] * d.ldy + l16 * 8);
      }
    }
  }
}

// fp32 rows -> `T` rows (the fp32 residual stream of the bf16 MViT plan): a lane owns FOUR channels
// (one 16-byte fp32 load, one 8-byte bf16 store), so every load instruction of a wave covers a dense
// run of memory; NL loads per lane cover rows up to 64*4*NL channels; G lanes per row.
template <typename T, int G, int NL>
__global__ __launch_bounds__(kThreads) void layernorm_f32in_kernel(const pv_rows_desc d) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, lg = lane % G;
  constexpr int RPW = 64 / G;
  const long wave_id = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const long wave_stride = (long)gridDim.x * (kThreads / 64);
  const int C4 = pv_round_up(d.C, 8) / 4;   // 4-channel chunks per (padded) row
  f32x4 gm[NL], bt[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int c0 = (lg + i * G) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gm[i][j] = (c0 + j < d.C && d.gamma) ? d.gamma[c0 + j] : 1.f;
      bt[i][j] = (c0 + j < d.C && d.beta) ? d.beta[c0 + j] : 0.f;
    }
  }
  for (long g0 = wave_id; g0 * RPW < d.rows; g0 += 2 * wave_stride) {
    f32x4 f[2][NL];
    long row[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      row[u] = (g0 + u * wave_stride) * RPW + sub;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int ch = lg + i * G;
        if (row[u] < d.rows && ch < C4)
          f[u][i] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(d.x) + row[u] * d.ldx + ch * 4);
        else
          f[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NL; ++i) s += (f[u][i][0] + f[u][i][1]) + (f[u][i][2] + f[u][i][3]);   // padding is zero
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mean = s / (float)d.C;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dlt = ((lg + i * G) * 4 + j < d.C) ? f[u][i][j] - mean : 0.f;
          v += dlt * dlt;
        }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const float rstd = rsqrtf(v / (float)d.C + d.eps);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int ch = lg + i * G;
        if (row[u] < d.rows && ch < C4) {
          float o4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o4[j] = (ch * 4 + j < d.C) ? (f[u][i][j] - mean) * rstd * gm[i][j] + bt[i][j] : 0.f;
          T* yp = static_cast<T*>(d.y) + row[u] * d.ldy + ch * 4;
          if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(yp) = bf16x4{(bf16_t)o4[0], (bf16_t)o4[1], (bf16_t)o4[2], (bf16_t)o4[3]};
          else *reinterpret_cast<f32x4*>(yp) = f32x4{o4[0], o4[1], o4[2], o4[3]};
        }
      }
    }
  }
}

// softmax over channels of each row (head activation); one wave per row, generic C
template <typename T>
__global__ __launch_bounds__(kThreads) void softmax_rows_kernel(const pv_rows_desc d) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (row >= d.rows) return;
  const T* x = static_cast<const T*>(d.x) + row * d.ldx;
  T* y = static_cast<T*>(d.y) + row * d.ldy;
  float mx = -FLT_MAX;
  for (int c = lane; c < d.C; c += 64) mx = fmaxf(mx, (float)x[c]);
  mx = pv_wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < d.C; c += 64) s += __expf((float)x[c] - mx);
  s = pv_wave_sum(s);
  const float inv = 1.f / s;
  for (int c = lane; c < d.C; c += 64) y[c] = (T)(__expf((float)x[c] - mx) * inv);
}

// Video-level ensembling: one wave per clip.  The wave of the FIRST clip of a video in this batch folds
// every clip of that video (in index order) into the video's score row, so no two waves touch the same
// row and the result does not depend on scheduling.
__global__ __launch_bounds__(kThreads) void ensemble_kernel(const pv_ensemble_desc d) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (i >= d.N) return;
  const int v = d.video_index[i];
  if (v < 0 || v >= d.V) return;
  for (int j = 0; j < i; ++j)
    if (d.video_index[j] == v) return;   // an earlier clip's wave owns this video
  float* acc = d.accum + (long)v * d.C;
  int cnt = 0;
  for (int j = i; j < d.N; ++j) {
    if (d.video_index[j] != v) continue;
    const float* x = d.logits + (long)j * d.ld;
    float mx = -FLT_MAX;
    for (int c = lane; c < d.C; c += 64) mx = fmaxf(mx, x[c]);
    mx = pv_wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < d.C; c += 64) s += __expf(x[c] - mx);
    s = pv_wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < d.C; c += 64) {
      const float p = __expf(x[c] - mx) * inv;
      acc[c] = d.mode == 1 ? fmaxf(acc[c], p) : acc[c] + p;   // same lane re-reads what it wrote: ordered
    }
    ++cnt;
  }
  if (lane == 0) d.counts[v] += cnt;
}

// y[b][c] = mean over rows_per_batch rows (fp32 out); thread per (b, c)
template <typename T>
__global__ __launch_bounds__(kThreads) void mean_rows_kernel(const pv_rows_desc d, int nb) {
  const long id = (long)blockIdx.x * kThreads + threadIdx.x;
  if (id >= (long)nb * d.C) return;
  const int c = (int)(id % d.C);
  const int b = (int)(id / d.C);
  const T* x = static_cast<const T*>(d.x) + (long)b * d.rows_per_batch * d.ldx + c;
  float s = 0.f;
  for (int r = 0; r < d.rows_per_batch; ++r) s += (float)x[(long)r * d.ldx];
  static_cast<float*>(d.y)[(long)b * d.ldy + c] = s / (float)d.rows_per_batch;
}

// tokens[b][0] = cls + pos_class ; tokens[b][1+t*HW+s] += pos_spatial[s] + pos_temporal[t]
template <typename T>
__global__ __launch_bounds__(kThreads) void posenc_kernel(const pv_posenc_desc d, long total) {
  const int c_p = pv_round_up(d.C, 8);
  const int CG = c_p / 8;
  const long id = (long)blockIdx.x * kThreads + threadIdx.x;
  if (id >= total) return;
  const int cg = (int)(id % CG);
  long r = id / CG;
  const int has_cls = d.cls_token != nullptr;
  const long rows = (long)d.T * d.HW + has_cls;
  const int b = d.cls_only ? (int)r : (int)(r / rows);          // cls_only: one work item per (clip, chunk)
  const long n = d.cls_only ? 0 : r - (long)b * rows;
  T* x = static_cast<T*>(d.x) + ((long)b * rows + n) * d.ld + cg * 8;
  float f[8];
  if (has_cls && n == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      float v = 0.f;
      if (c < d.C) {
        v = d.cls_token[c];
        if (d.pos_temporal != nullptr) { if (d.pos_class) v += d.pos_class[c]; }
        else v += d.pos_spatial[c];  // full table: row 0 belongs to cls
      }
      f[j] = v;
    }
  } else {
    Chunk8<T> cch;
    cch.load(x);
    cch.to_f32(f);
    const long g = n - has_cls;
    const int t = (int)(g / d.HW), s = (int)(g - (long)t * d.HW);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      if (c < d.C) {
        if (d.pos_temporal != nullptr)
          f[j] += d.pos_spatial[(long)s * d.C + c] + d.pos_temporal[(long)t * d.C + c];
        else
          f[j] += d.pos_spatial[(g + has_cls) * d.C + c];
      }
    }
  }
  Chunk8<T> o;
  o.from_f32(f);
  o.store(x);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void add_act_kernel(const pv_add_desc d, long total) {
  const int CG = pv_round_up(d.C, 8) / 8;
  const long id = (long)blockIdx.x * kThreads + threadIdx.x;
  if (id >= total) return;
  const int cg = (int)(id % CG);
  const long r = id / CG;
  Chunk8<T> a, b;
  a.load(static_cast<const T*>(d.a) + r * d.lda + cg * 8);
  b.load(static_cast<const T*>(d.b) + r * d.ldb + cg * 8);
  float fa[8], fb[8];
  a.to_f32(fa);
  b.to_f32(fb);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    fa[j] = pv_apply_act(fa[j] + fb[j], d.act);
    if (cg * 8 + j >= d.C) fa[j] = 0.f;
  }
  a.from_f32(fa);
  a.store(static_cast<T*>(d.y) + r * d.ldy + cg * 8);
}

// The same gate for the sizes X3D uses (C <= 448, reduced width <= 32): this kernel is a chain of four
// dependent phases on a handful of bytes, so its duration is the sum of their memory latencies.  Every
// FC weight a thread will need is requested at kernel entry, together with the partial sums, so the
// phases after the first run out of registers.
constexpr int kSeMaxCh = 2;     // channels per thread: c_p <= 512
// CJ: 64-channel strides covering C (<= 7); CRP: reduced width rounded up to 8 / 16 / 32
template <int CJ, int CRP>
__global__ __launch_bounds__(kThreads) void se_gate_fast_kernel(const pv_se_gate_desc d) {
  constexpr int kSeMaxC64 = CJ;
  constexpr int kSeMaxR = CRP / 4;   // hidden units per wave (4 waves)
  extern __shared__ float s[];
  float* s_mean = s;                 // [c_p]
  float* s_hid = s + d.c_p;          // [32]
  float* s_part = s + d.c_p + 32;    // [groups][c_p]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // ---- prefetch: fc1 rows (wave = hidden unit), fc2 rows (thread = channel), biases ----
  float w1r[kSeMaxR][kSeMaxC64], b1r[kSeMaxR];
#pragma unroll
  for (int i = 0; i < kSeMaxR; ++i) {
    const int r = wave + i * 4;
    const bool rok = r < d.cr;
    b1r[i] = (rok && d.b1) ? d.b1[r] : 0.f;
#pragma unroll
    for (int j = 0; j < kSeMaxC64; ++j) {
      const int c = lane + j * 64;
      const bool ok = rok && c < d.C;
      const float v = d.w1[ok ? (long)r * d.C + c : 0];
      w1r[i][j] = ok ? v : 0.f;
    }
  }
  float w2r[kSeMaxCh][CRP], b2r[kSeMaxCh];
#pragma unroll
  for (int i = 0; i < kSeMaxCh; ++i) {
    const int c = tid + i * kThreads;
    const bool cok = c < d.C;
    b2r[i] = (cok && d.b2) ? d.b2[c] : 0.f;
#pragma unroll
    for (int r = 0; r < CRP; ++r) {
      const bool ok = cok && r < d.cr;
      const float v = d.w2[ok ? (long)c * d.cr + r : 0];
      w2r[i][r] = ok ? v : 0.f;
    }
  }
  // ---- mean over T,H,W from the per-tile partial sums ----
  const float* ps = d.psum + (long)b * d.nblk * d.c_p;
  const int cw = d.c_p < kThreads ? d.c_p : kThreads;
  const int groups = kThreads / cw;
  const int g = tid / cw, cl = tid - g * cw;
  for (int c0 = 0; c0 < d.c_p; c0 += cw) {
    const int c = c0 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (g < groups && c < d.c_p) {
      int k = g;
      for (; k + 3 * groups < d.nblk; k += 4 * groups) {
        a0 += ps[(long)k * d.c_p + c];
        a1 += ps[(long)(k + groups) * d.c_p + c];
        a2 += ps[(long)(k + 2 * groups) * d.c_p + c];
        a3 += ps[(long)(k + 3 * groups) * d.c_p + c];
      }
      for (; k < d.nblk; k += groups) a0 += ps[(long)k * d.c_p + c];
      s_part[g * d.c_p + c] = (a0 + a1) + (a2 + a3);
    }
  }
  __syncthreads();
  for (int c = tid; c < d.c_p; c += kThreads) {
    float a = 0.f;
    for (int gg = 0; gg < groups; ++gg) a += s_part[gg * d.c_p + c];
    s_mean[c] = a * d.inv_count;
  }
  __syncthreads();
  // ---- fc1 + relu ----
  float mj[kSeMaxC64];
#pragma unroll
  for (int j = 0; j < kSeMaxC64; ++j) mj[j] = lane + j * 64 < d.C ? s_mean[lane + j * 64] : 0.f;
#pragma unroll
  for (int i = 0; i < kSeMaxR; ++i) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < kSeMaxC64; ++j) a += w1r[i][j] * mj[j];
    a = pv_wave_sum(a);
    if (lane == 0 && wave + i * 4 < 32) s_hid[wave + i * 4] = wave + i * 4 < d.cr ? fmaxf(a + b1r[i], 0.f) : 0.f;
  }
  __syncthreads();
  // ---- fc2 + sigmoid ----
  float hid[CRP];
#pragma unroll
  for (int r = 0; r < CRP; ++r) hid[r] = s_hid[r];
#pragma unroll
  for (int i = 0; i < kSeMaxCh; ++i) {
    const int c = tid + i * kThreads;
    if (c < d.c_p) {
      float a = b2r[i];
#pragma unroll
      for (int r = 0; r < CRP; ++r) a += w2r[i][r] * hid[r];
      d.gate[(long)b * d.c_p + c] = c < d.C ? pv_sigmoid(a) : 0.f;
    }
  }
}

inline unsigned blocks_for(long total) { return (unsigned)pv_ceil_div(total, kThreads); }

}  // namespace

// ======================================================================= C ABI
extern "C" int pv_se_gate(const pv_se_gate_desc* d, pv_stream_t stream) {
  if (!d || !d->psum || !d->gate || !d->w1 || !d->w2) return PV_ERR_INVALID;
  if (d->B <= 0 || d->C <= 0 || d->cr <= 0 || d->nblk <= 0 || d->c_p < d->C || d->c_p % 8) return PV_ERR_INVALID;
  const int cw = d->c_p < kThreads ? d->c_p : kThreads;
  if (d->C <= 448 && d->cr <= 32 && d->c_p <= kThreads * kSeMaxCh) {
    const size_t lds_f = sizeof(float) * ((size_t)d->c_p + 32 + (size_t)(kThreads / cw) * d->c_p);
    const int cj = (d->C + 63) / 64;
    dim3 grid(d->B), block(kThreads);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cj <= 1 && d->cr <= 8) PV_LAUNCH((se_gate_fast_kernel<1, 8>), grid, block, lds_f, st, *d);
    else if (cj <= 2 && d->cr <= 8) PV_LAUNCH((se_gate_fast_kernel<2, 8>), grid, block, lds_f, st, *d);
    else if (cj <= 4 && d->cr <= 16) PV_LAUNCH((se_gate_fast_kernel<4, 16>), grid, block, lds_f, st, *d);
    else PV_LAUNCH((se_gate_fast_kernel<7, 32>), grid, block, lds_f, st, *d);
    PV_LAUNCH_CHECK();
    return PV_OK;
  }
  const size_t lds = sizeof(float) * ((size_t)d->c_p + d->cr + (size_t)(kThreads / cw) * d->c_p);
  PV_LAUNCH(se_gate_kernel, dim3(d->B), dim3(kThreads), lds, static_cast<hipStream_t>(stream), *d);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

template <typename T> static int pool_launch(const pv_pool3d_desc& d, hipStream_t s) {
  const int CG = pv_round_up(d.C, 8) / 8;
  const int taps = d.kt * d.kh * d.kw;
  const long nvox = (long)d.B * d.To * d.Ho * d.Wo;
  if (taps >= 64) {
    // 8 chunks (128 contiguous bytes of a voxel) per block, the window split 32 ways: the head pools
    // (16x7x7 over 432 channels, 8x7x7 over 2048) have only B..4B output voxels, so the parallelism has
    // to come from the channel slabs and the taps
    const int cgb = CG < 8 ? CG : 8;
    if (nvox > 0x7fffffffL) return PV_ERR_UNSUPPORTED;
    dim3 grid((unsigned)nvox, (unsigned)pv_ceil_div(CG, cgb));
    PV_LAUNCH(pool_reduce_kernel<T>, grid, dim3(kThreads), 0, s, d, cgb);
  } else {
    const long total = nvox * CG;
    const bool win = pv_tune("pool_window", 1) != 0;
    if (win && d.kt == 3 && d.kh == 3 && d.kw == 3) PV_LAUNCH((pool_window_kernel<T, 3, 3, 3>), dim3(blocks_for(total)), dim3(kThreads), 0, s, d, total);
    else if (win && d.kt == 1 && d.kh == 3 && d.kw == 3) PV_LAUNCH((pool_window_kernel<T, 1, 3, 3>), dim3(blocks_for(total)), dim3(kThreads), 0, s, d, total);
    else PV_LAUNCH(pool_direct_kernel<T>, dim3(blocks_for(total)), dim3(kThreads), 0, s, d, total);
  }
  PV_LAUNCH_CHECK();
  if (d.n_prefix > 0) {
    const int total = d.B * d.n_prefix * CG;
    PV_LAUNCH(pool_prefix_kernel<T>, dim3(blocks_for(total)), dim3(kThreads), 0, s, d);
    PV_LAUNCH_CHECK();
  }
  return PV_OK;
}

extern "C" int pv_pool3d(const pv_pool3d_desc* dp, pv_stream_t stream) {
  if (!dp || !dp->x || !dp->y) return PV_ERR_INVALID;
  const pv_pool3d_desc& d = *dp;
  if (d.B <= 0 || d.C <= 0 || d.To <= 0 || d.Ho <= 0 || d.Wo <= 0) return PV_ERR_INVALID;
  if (d.ldx % 8 || d.ldy % 8 || d.x_bs % 8 || d.y_bs % 8) return PV_ERR_INVALID;
  if (d.kt < 1 || d.kh < 1 || d.kw < 1 || d.st < 1 || d.sh < 1 || d.sw < 1) return PV_ERR_INVALID;
  if ((d.Ti + 2 * d.pt - d.kt) / d.st + 1 != d.To || (d.Hi + 2 * d.ph - d.kh) / d.sh + 1 != d.Ho ||
      (d.Wi + 2 * d.pw - d.kw) / d.sw + 1 != d.Wo)
    return PV_ERR_INVALID;
  if (d.mode != PV_POOL_MAX && d.mode != PV_POOL_AVG) return PV_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d.dtype == PV_BF16) return pool_launch<bf16_t>(d, s);
  if (d.dtype == PV_F32) return pool_launch<float>(d, s);
  return PV_ERR_UNSUPPORTED;
}

static int layout_check(const pv_layout_desc* d) {
  if (!d || !d->src || !d->dst) return PV_ERR_INVALID;
  if (d->B <= 0 || d->C <= 0 || d->T <= 0 || d->H <= 0 || d->W <= 0) return PV_ERR_INVALID;
  if (d->c_p % 8 || d->c_p < d->C || d->ld % 8 || d->ld < d->c_p || d->bs % 8) return PV_ERR_INVALID;
  return PV_OK;
}

extern "C" int pv_ingest_ncdhw(const pv_layout_desc* d, pv_stream_t stream) {
  if (d && d->src && d->dst && d->c_p == 4 && d->ld == 4 && d->C >= 1 && d->C <= 4 && d->dst_dtype == PV_BF16 &&
      d->B > 0 && d->T > 0 && d->H > 0 && d->W > 0 && d->bs % 4 == 0) {
    const long nvox = (long)d->B * d->T * d->H * d->W;
    dim3 grid(blocks_for(nvox)), block(kThreads);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d->t_index && d->src_T <= 0) return PV_ERR_INVALID;
    const int esz = d->src_dtype == PV_F32 ? 4 : (d->src_dtype == PV_BF16 ? 2 : 1);
    const long HW = (long)d->H * d->W;
    if (HW % 8 == 0 && (HW * esz) % 16 == 0 && ((uintptr_t)d->src % 16) == 0 && ((uintptr_t)d->dst % 16) == 0 && d->bs % 8 == 0) {
      const long ngroups = nvox / 8;
      dim3 g8(blocks_for(ngroups));
      if (d->src_dtype == PV_F32) PV_LAUNCH(ingest_c4_vec8_kernel<float>, g8, block, 0, s, *d, ngroups);
      else if (d->src_dtype == PV_BF16) PV_LAUNCH(ingest_c4_vec8_kernel<bf16_t>, g8, block, 0, s, *d, ngroups);
      else if (d->src_dtype == PV_U8) PV_LAUNCH(ingest_c4_vec8_kernel<unsigned char>, g8, block, 0, s, *d, ngroups);
      else return PV_ERR_UNSUPPORTED;
      PV_LAUNCH_CHECK();
      return PV_OK;
    }
    if (d->src_dtype == PV_F32) PV_LAUNCH(ingest_c4_kernel<float>, grid, block, 0, s, *d, nvox);
    else if (d->src_dtype == PV_BF16) PV_LAUNCH(ingest_c4_kernel<bf16_t>, grid, block, 0, s, *d, nvox);
    else if (d->src_dtype == PV_U8) PV_LAUNCH(ingest_c4_kernel<unsigned char>, grid, block, 0, s, *d, nvox);
    else return PV_ERR_UNSUPPORTED;
    PV_LAUNCH_CHECK();
    return PV_OK;
  }
  const int v = layout_check(d);
  if (v != PV_OK) return v;
  const long nvox = (long)d->B * d->T * d->H * d->W;
  const long total = nvox * (d->c_p / 8);
  dim3 grid(blocks_for(total)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->t_index && d->src_T <= 0) return PV_ERR_INVALID;
  if (d->src_dtype == PV_U8 && d->dst_dtype == PV_F32)
    PV_LAUNCH((ingest_kernel<unsigned char, float>), grid, block, 0, s, *d, nvox);
  else if (d->src_dtype == PV_U8 && d->dst_dtype == PV_BF16)
    PV_LAUNCH((ingest_kernel<unsigned char, bf16_t>), grid, block, 0, s, *d, nvox);
  else if (d->src_dtype == PV_F32 && d->dst_dtype == PV_F32)
    PV_LAUNCH((ingest_kernel<float, float>), grid, block, 0, s, *d, nvox);
  else if (d->src_dtype == PV_F32 && d->dst_dtype == PV_BF16)
    PV_LAUNCH((ingest_kernel<float, bf16_t>), grid, block, 0, s, *d, nvox);
  else if (d->src_dtype == PV_BF16 && d->dst_dtype == PV_BF16)
    PV_LAUNCH((ingest_kernel<bf16_t, bf16_t>), grid, block, 0, s, *d, nvox);
  else if (d->src_dtype == PV_BF16 && d->dst_dtype == PV_F32)
    PV_LAUNCH((ingest_kernel<bf16_t, float>), grid, block, 0, s, *d, nvox);
  else
    return PV_ERR_UNSUPPORTED;
  PV_LAUNCH_CHECK();
  return PV_OK;
}

extern "C" int pv_egress_ncdhw(const pv_layout_desc* d, pv_stream_t stream) {
  const int v = layout_check(d);
  if (v != PV_OK) return v;
  const long nvox = (long)d->B * d->T * d->H * d->W;
  const long total = nvox * (d->c_p / 8);
  dim3 grid(blocks_for(total)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // src = NDHWC side (src_dtype), dst = NCDHW side (dst_dtype)
  if (d->src_dtype == PV_F32 && d->dst_dtype == PV_F32)
    PV_LAUNCH((egress_kernel<float, float>), grid, block, 0, s, *d, nvox);
  else if (d->src_dtype == PV_BF16 && d->dst_dtype == PV_F32)
    PV_LAUNCH((egress_kernel<bf16_t, float>), grid, block, 0, s, *d, nvox);
  else if (d->src_dtype == PV_BF16 && d->dst_dtype == PV_BF16)
    PV_LAUNCH((egress_kernel<bf16_t, bf16_t>), grid, block, 0, s, *d, nvox);
  else if (d->src_dtype == PV_F32 && d->dst_dtype == PV_BF16)
    PV_LAUNCH((egress_kernel<float, bf16_t>), grid, block, 0, s, *d, nvox);
  else
    return PV_ERR_UNSUPPORTED;
  PV_LAUNCH_CHECK();
  return PV_OK;
}

static int rows_check(const pv_rows_desc* d) {
  if (!d || !d->x || !d->y || d->rows <= 0 || d->C <= 0) return PV_ERR_INVALID;
  return PV_OK;
}

// BatchNorm (eval) on token rows: y = act(x * gamma + beta), 8 channels per thread, rows over the grid.
// TX = float: fp32 stream in, T out (the block norms); TX = T: same type, in place allowed (the pre-pooling norm).
template <typename TX, typename T>
__global__ __launch_bounds__(kThreads) void affine_rows_kernel(const pv_rows_desc d, long total, int CG) {
  for (long id = (long)blockIdx.x * kThreads + threadIdx.x; id < total; id += (long)gridDim.x * kThreads) {
    const long row = id / CG;
    const int c0 = (int)(id - row * CG) * 8;
    const bool skip = d.rows_per_batch > 0 && (int)(row % d.rows_per_batch) < d.n_prefix;
    const TX* xp = static_cast<const TX*>(d.x) + row * d.ldx + c0;
    T* yp = static_cast<T*>(d.y) + row * d.ldy + c0;
    if (skip && static_cast<const void*>(xp) == static_cast<const void*>(yp)) continue;
    float v[8];
    Chunk8<TX> in;
    in.load(xp);
    in.to_f32(v);
    if (!skip) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = c0 + j < d.C;
        const float g = ok ? (d.gamma ? d.gamma[c0 + j] : 1.f) : 0.f;
        const float b = ok ? (d.beta ? d.beta[c0 + j] : 0.f) : 0.f;
        v[j] = v[j] * g + b;
      }
      pv_apply_act_n<sizeof(T) == 2>(v, d.act);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (c0 + j >= d.C) v[j] = 0.f;
    }
    Chunk8<T> o;
    o.from_f32(v);
    o.store(yp);
  }
}

extern "C" int pv_affine_rows(const pv_rows_desc* d, pv_stream_t stream) {
  int v = rows_check(d);
  if (v != PV_OK) return v;
  if (d->ldx % 8 || d->ldy % 8 || d->g_period != 0 || d->n_prefix < 0 || d->rows_per_batch < 0) return PV_ERR_INVALID;
  if (d->n_prefix > 0 && d->rows_per_batch <= 0) return PV_ERR_INVALID;
  const int CG = pv_round_up(d->C, 8) / 8;
  const long total = d->rows * CG;
  long nb = pv_ceil_div(total, kThreads);
  nb = nb < 8192 ? nb : 8192;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid((unsigned)nb), block(kThreads);
  if (d->dtype == PV_BF16 && d->x_f32) PV_LAUNCH((affine_rows_kernel<float, bf16_t>), grid, block, 0, s, *d, total, CG);
  else if (d->dtype == PV_BF16) PV_LAUNCH((affine_rows_kernel<bf16_t, bf16_t>), grid, block, 0, s, *d, total, CG);
  else if (d->dtype == PV_F32) PV_LAUNCH((affine_rows_kernel<float, float>), grid, block, 0, s, *d, total, CG);
  else return PV_ERR_UNSUPPORTED;
  PV_LAUNCH_CHECK();
  return PV_OK;
}

extern "C" int pv_layernorm(const pv_rows_desc* d, pv_stream_t stream) {
  int v = rows_check(d);
  if (v != PV_OK) return v;
  if (d->ldx % 8 || d->ldy % 8) return PV_ERR_INVALID;
  const int CG = pv_round_up(d->C, 8) / 8;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->g_period < 0 || d->g_period > 16) return PV_ERR_INVALID;
  if (d->g_period > 1 && (CG > 32 || d->x_f32 || 16 % d->g_period)) return PV_ERR_UNSUPPORTED;   // narrow-row kernel only
#define PV_LN16(G)                                                                                              \
  do {                                                                                                          \
    long nb16 = pv_ceil_div(d->rows, (kThreads / 64) * (64 / G) * 2);                                          \
    nb16 = nb16 < 4096 ? nb16 : 4096;                                                                           \
    if (d->g_period > 1) nb16 = pv_ceil_div(nb16, d->g_period) * d->g_period;                                   \
    dim3 grid16((unsigned)nb16), block16(kThreads);                                                             \
    if (d->dtype == PV_BF16 && d->x_f32)                                                                        \
      PV_LAUNCH((layernorm16_kernel<float, bf16_t, G>), grid16, block16, 0, s, *d);                    \
    else if (d->dtype == PV_BF16)                                                                               \
      PV_LAUNCH((layernorm16_kernel<bf16_t, bf16_t, G>), grid16, block16, 0, s, *d);                   \
    else if (d->dtype == PV_F32)                                                                                \
      PV_LAUNCH((layernorm16_kernel<float, float, G>), grid16, block16, 0, s, *d);                     \
    else                                                                                                        \
      return PV_ERR_UNSUPPORTED;                                                                                \
    PV_LAUNCH_CHECK();                                                                                          \
    return PV_OK;                                                                                               \
  } while (0)
  if (d->dtype == PV_BF16 && d->x_f32 && CG <= 128) {   // fp32 stream -> bf16 operand: 4 channels per lane
    const int C4 = CG * 2;
#define PV_LNF(G, NL)                                                                                  \
  do {                                                                                                 \
    const long nb = pv_ceil_div(d->rows, (kThreads / 64) * (64 / G) * 2);                              \
    PV_LAUNCH((layernorm_f32in_kernel<bf16_t, G, NL>), dim3((unsigned)(nb < 4096 ? nb : 4096)), \
                       dim3(kThreads), 0, s, *d);                                                      \
    PV_LAUNCH_CHECK();                                                                                 \
    return PV_OK;                                                                                      \
  } while (0)
    if (C4 <= 16) PV_LNF(16, 1);
    if (C4 <= 32) PV_LNF(32, 1);
    if (C4 <= 64) PV_LNF(64, 1);
    if (C4 <= 128) PV_LNF(64, 2);
    PV_LNF(64, 4);
#undef PV_LNF
  }
  if (CG <= 16) PV_LN16(16);
  if (CG <= 32) PV_LN16(32);
#undef PV_LN16
  dim3 grid((unsigned)pv_ceil_div(d->rows, kThreads / 64)), block(kThreads);
#define PV_LN(TI, T, MAXC) PV_LAUNCH((layernorm_kernel<TI, T, MAXC>), grid, block, 0, s, *d)
  if (d->dtype == PV_BF16 && d->x_f32) {
    if (CG <= 64) PV_LN(float, bf16_t, 1); else if (CG <= 128) PV_LN(float, bf16_t, 2);
    else if (CG <= 256) PV_LN(float, bf16_t, 4); else return PV_ERR_UNSUPPORTED;
  } else if (d->dtype == PV_BF16) {
    if (CG <= 64) PV_LN(bf16_t, bf16_t, 1); else if (CG <= 128) PV_LN(bf16_t, bf16_t, 2);
    else if (CG <= 256) PV_LN(bf16_t, bf16_t, 4); else return PV_ERR_UNSUPPORTED;
  } else if (d->dtype == PV_F32) {
    if (CG <= 64) PV_LN(float, float, 1); else if (CG <= 128) PV_LN(float, float, 2);
    else if (CG <= 256) PV_LN(float, float, 4); else return PV_ERR_UNSUPPORTED;
  } else return PV_ERR_UNSUPPORTED;
#undef PV_LN
  PV_LAUNCH_CHECK();
  return PV_OK;
}

extern "C" int pv_softmax_rows(const pv_rows_desc* d, pv_stream_t stream) {
  int v = rows_check(d);
  if (v != PV_OK) return v;
  dim3 grid((unsigned)pv_ceil_div(d->rows, kThreads / 64)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->dtype == PV_BF16) PV_LAUNCH(softmax_rows_kernel<bf16_t>, grid, block, 0, s, *d);
  else if (d->dtype == PV_F32) PV_LAUNCH(softmax_rows_kernel<float>, grid, block, 0, s, *d);
  else return PV_ERR_UNSUPPORTED;
  PV_LAUNCH_CHECK();
  return PV_OK;
}

extern "C" int pv_ensemble_scores(const pv_ensemble_desc* d, pv_stream_t stream) {
  if (!d || !d->logits || !d->video_index || !d->accum || !d->counts) return PV_ERR_INVALID;
  if (d->N <= 0 || d->C <= 0 || d->V <= 0 || d->ld < d->C || (d->mode != 0 && d->mode != 1)) return PV_ERR_INVALID;
  PV_LAUNCH(ensemble_kernel, dim3((unsigned)pv_ceil_div(d->N, kThreads / 64)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), *d);
  PV_LAUNCH_CHECK();
  return PV_OK;
}

extern "C" int pv_mean_rows(const pv_rows_desc* d, pv_stream_t stream) {
  int v = rows_check(d);
  if (v != PV_OK) return v;
  if (d->rows_per_batch <= 0 || d->rows % d->rows_per_batch) return PV_ERR_INVALID;
  const int nb = (int)(d->rows / d->rows_per_batch);
  dim3 grid(blocks_for((long)nb * d->C)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->dtype == PV_BF16) PV_LAUNCH(mean_rows_kernel<bf16_t>, grid, block, 0, s, *d, nb);
  else if (d->dtype == PV_F32) PV_LAUNCH(mean_rows_kernel<float>, grid, block, 0, s, *d, nb);
  else return PV_ERR_UNSUPPORTED;
  PV_LAUNCH_CHECK();
  return PV_OK;
}

extern "C" int pv_add_posenc(const pv_posenc_desc* d, pv_stream_t stream) {
  if (!d || !d->x || !d->pos_spatial) return PV_ERR_INVALID;
  if (d->B <= 0 || d->T <= 0 || d->HW <= 0 || d->C <= 0 || d->ld % 8) return PV_ERR_INVALID;
  const long rows = (long)d->T * d->HW + (d->cls_token ? 1 : 0);
  if (d->cls_only && !d->cls_token) return PV_ERR_INVALID;
  const long total = (long)d->B * (d->cls_only ? 1 : rows) * (pv_round_up(d->C, 8) / 8);
  dim3 grid(blocks_for(total)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->dtype == PV_BF16) PV_LAUNCH(posenc_kernel<bf16_t>, grid, block, 0, s, *d, total);
  else if (d->dtype == PV_F32) PV_LAUNCH(posenc_kernel<float>, grid, block, 0, s, *d, total);
  else return PV_ERR_UNSUPPORTED;
  PV_LAUNCH_CHECK();
  return PV_OK;
}

extern "C" int pv_add_act(const pv_add_desc* d, pv_stream_t stream) {
  if (!d || !d->a || !d->b || !d->y || d->rows <= 0 || d->C <= 0) return PV_ERR_INVALID;
  if (d->lda % 8 || d->ldb % 8 || d->ldy % 8) return PV_ERR_INVALID;
  const long total = d->rows * (pv_round_up(d->C, 8) / 8);
  dim3 grid(blocks_for(total)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d->dtype == PV_BF16) PV_LAUNCH(add_act_kernel<bf16_t>, grid, block, 0, s, *d, total);
  else if (d->dtype == PV_F32) PV_LAUNCH(add_act_kernel<float>, grid, block, 0, s, *d, total);
  else return PV_ERR_UNSUPPORTED;
  PV_LAUNCH_CHECK();
  return PV_OK;
}
