// RoIAlign on a channels-last feature map, optionally fused with the whole-window max pool that
// follows it in the detection head (ResNetRoIHead.forward, models/head.py:437-482;
// roi_layer = torchvision.ops.RoIAlign, pool_spatial = MaxPool2d(resolution, stride=1)).
//
// One workgroup per (box, slab of 64 chunks = 512 channels).  A lane owns one 8-channel chunk, so the
// four neighbours of a bilinear sample are four 16-byte (bf16) loads that the 64 lanes of a wave issue
// as one contiguous 1 KB row each; the sample coordinates and weights are wave-uniform.  The four waves
// of the group take the output bins round-robin; with pool_max they join through 8 KB of LDS and the
// [R, C, ph, pw] tensor is never written.  The feature map of a clip (16x16x2304 bf16 = 1.2 MB for
// SlowFast-R50 at 256^2) stays in L2 across its boxes: HBM traffic is the map once plus one row per box.
#include <float.h>
#include "pv_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / PV_WAVE;
constexpr int kSlab = 64;  // chunks per workgroup = lanes per wave

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <typename T>
__global__ __launch_bounds__(kThreads) void roi_align_kernel(const pv_roi_align_desc d) {
  __shared__ float s_max[kWaves][kSlab * 8];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & (PV_WAVE - 1);
  const int wave = uniform(threadIdx.x >> 6);
  const int CG = pv_round_up(d.C, 8) / 8;
  const int cg = blockIdx.y * kSlab + lane;
  const bool live = cg < CG;

  // ---- box geometry (torchvision roi_align: aligned=0 clamps the roi size to >= 1 feature pixel)
  const float* box = d.boxes + (long)n * 5;
  const int b = uniform((int)box[0]);
  const float off = d.aligned ? 0.5f : 0.0f;
  const float start_w = box[1] * d.spatial_scale - off;
  const float start_h = box[2] * d.spatial_scale - off;
  const float end_w = box[3] * d.spatial_scale - off;
  const float end_h = box[4] * d.spatial_scale - off;
  float roi_w = end_w - start_w, roi_h = end_h - start_h;
  if (!d.aligned) {
    roi_w = fmaxf(roi_w, 1.0f);
    roi_h = fmaxf(roi_h, 1.0f);
  }
  const float bin_h = roi_h / (float)d.ph, bin_w = roi_w / (float)d.pw;
  const int grid_h = uniform(d.sampling_ratio > 0 ? d.sampling_ratio : (int)ceilf(roi_h / (float)d.ph));
  const int grid_w = uniform(d.sampling_ratio > 0 ? d.sampling_ratio : (int)ceilf(roi_w / (float)d.pw));
  const int n_samples = grid_h * grid_w;
  const float inv_count = 1.0f / (float)(n_samples > 1 ? n_samples : 1);
  const bool box_ok = b >= 0 && b < d.B;

  const T* X = static_cast<const T*>(d.x) + (long)(box_ok ? b : 0) * d.x_bs + (long)cg * 8;
  const int bins = d.ph * d.pw;
  float best[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) best[j] = -FLT_MAX;

  for (int bin = wave; bin < bins; bin += kWaves) {
    const int ph = bin / d.pw, pw = bin - ph * d.pw;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (box_ok) {
      for (int iy = 0; iy < grid_h; ++iy) {
        float y = start_h + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)grid_h;
        if (y < -1.0f || y > (float)d.H) continue;  // sample outside the map contributes 0
        y = fmaxf(y, 0.0f);
        int y_lo = (int)y, y_hi;
        if (y_lo >= d.H - 1) {
          y_hi = y_lo = d.H - 1;
          y = (float)y_lo;
        } else {
          y_hi = y_lo + 1;
        }
        const float ly = y - (float)y_lo, hy = 1.0f - ly;
        for (int ix = 0; ix < grid_w; ++ix) {
          float x = start_w + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)grid_w;
          if (x < -1.0f || x > (float)d.W) continue;
          x = fmaxf(x, 0.0f);
          int x_lo = (int)x, x_hi;
          if (x_lo >= d.W - 1) {
            x_hi = x_lo = d.W - 1;
            x = (float)x_lo;
          } else {
            x_hi = x_lo + 1;
          }
          const float lx = x - (float)x_lo, hx = 1.0f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          if (live) {
            Chunk8<T> c1, c2, c3, c4;
            c1.load(X + ((long)y_lo * d.W + x_lo) * d.ldx);
            c2.load(X + ((long)y_lo * d.W + x_hi) * d.ldx);
            c3.load(X + ((long)y_hi * d.W + x_lo) * d.ldx);
            c4.load(X + ((long)y_hi * d.W + x_hi) * d.ldx);
            float f1[8], f2[8], f3[8], f4[8];
            c1.to_f32(f1), c2.to_f32(f2), c3.to_f32(f3), c4.to_f32(f4);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w1 * f1[j] + w2 * f2[j] + w3 * f3[j] + w4 * f4[j];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j] *= inv_count;
      if (cg * 8 + j >= d.C) acc[j] = 0.f;  // padding channels stay zero
    }
    if (d.pool_max) {
#pragma unroll
      for (int j = 0; j < 8; ++j) best[j] = fmaxf(best[j], acc[j]);
    } else if (live) {
      Chunk8<T> o;
      o.from_f32(acc);
      o.store(static_cast<T*>(d.y) + ((long)n * bins + bin) * d.ldy + (long)cg * 8);
    }
  }

  if (d.pool_max) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s_max[wave][lane * 8 + j] = best[j];
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
      for (int w = 1; w < kWaves; ++w)
#pragma unroll
        for (int j = 0; j < 8; ++j) best[j] = fmaxf(best[j], s_max[w][lane * 8 + j]);
      Chunk8<T> o;
      o.from_f32(best);
      o.store(static_cast<T*>(d.y) + (long)n * d.ldy + (long)cg * 8);
    }
  }
}

}  // namespace

extern "C" int pv_roi_align(const pv_roi_align_desc* dp, pv_stream_t stream) {
  if (!dp || !dp->x || !dp->boxes || !dp->y) return PV_ERR_INVALID;
  const pv_roi_align_desc& d = *dp;
  if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.C <= 0 || d.R <= 0 || d.ph <= 0 || d.pw <= 0) return PV_ERR_INVALID;
  if (d.ldx % 8 || d.ldy % 8 || d.x_bs % 8 || d.ldx < d.C || d.ldy < d.C) return PV_ERR_INVALID;
  if (!(d.spatial_scale > 0.f)) return PV_ERR_INVALID;
  if (d.dtype != PV_F32 && d.dtype != PV_BF16) return PV_ERR_INVALID;
  const int CG = pv_round_up(d.C, 8) / 8;
  dim3 grid((unsigned)d.R, (unsigned)pv_ceil_div(CG, kSlab));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d.dtype == PV_BF16)
    PV_LAUNCH(roi_align_kernel<bf16_t>, grid, dim3(kThreads), 0, s, d);
  else
    PV_LAUNCH(roi_align_kernel<float>, grid, dim3(kThreads), 0, s, d);
  PV_LAUNCH_CHECK();
  return PV_OK;
}
