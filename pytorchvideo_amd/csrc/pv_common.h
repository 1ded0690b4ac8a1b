// Shared device helpers for the gfx950 kernels (wave64, MFMA, bf16/f32 8-element chunks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pv_mi355x.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PV_WAVE 64

int pv_set_hip_error(hipError_t e, const char* what);  // pv_plan.hip
// development knobs (kernel routing A/Bs): set through the C ABI (pv_tune_set), never read from the environment
int pv_tune(const char* key, int dflt);                // pv_plan.hip
#define PV_HIP_CHECK(expr)                                   \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return pv_set_hip_error(_e, #expr); \
  } while (0)
#define PV_LAUNCH_CHECK() PV_HIP_CHECK(hipGetLastError())
// Every kernel launch goes through PV_LAUNCH: it notes the kernel's symbol (a pointer store) so that pv_plan_profile can
// report which kernel an op was routed to -- bench.py's `roofline` is per kernel SYMBOL, not per op label.
void pv_note_kernel(const char* name);                 // pv_plan.hip
#define PV_LAUNCH(kernel, ...)              \
  do {                                      \
    pv_note_kernel(#kernel);                \
    hipLaunchKernelGGL(kernel, __VA_ARGS__); \
  } while (0)

// ---- an 8-channel chunk: the unit every kernel moves (16 B of bf16, 32 B of f32) ----
template <typename T> struct Chunk8;
template <> struct alignas(16) Chunk8<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void zero() {
    v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
  __device__ __forceinline__ void to_f32(float (&f)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  __device__ __forceinline__ void from_f32(const float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)f[i];
  }
};
template <> struct alignas(16) Chunk8<float> {
  f32x4 lo, hi;
  __device__ __forceinline__ void zero() {
    lo = f32x4{0.f, 0.f, 0.f, 0.f};
    hi = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void load(const float* p) {
    lo = *reinterpret_cast<const f32x4*>(p);
    hi = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<f32x4*>(p) = lo;
    *reinterpret_cast<f32x4*>(p + 4) = hi;
  }
  __device__ __forceinline__ void to_f32(float (&f)[8]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = lo[i]; f[4 + i] = hi[i]; }
  }
  __device__ __forceinline__ void from_f32(const float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { lo[i] = f[i]; hi[i] = f[4 + i]; }
  }
};

// 1/(1+e^-x) with the hardware reciprocal (1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float pv_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float pv_apply_act(float v, int act) {
  switch (act) {
    case PV_ACT_RELU: return fmaxf(v, 0.0f);
    case PV_ACT_SWISH: return v * pv_sigmoid(v);
    case PV_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case PV_ACT_SIGMOID: return pv_sigmoid(v);
    default: return v;
  }
}

// GELU (erf form) for bf16 epilogues: erf by Abramowitz & Stegun 7.1.25 (|error| <= 2.5e-5, two
// transcendentals, ~12 instructions against ~40 for erff) -- an order of magnitude below the bf16
// rounding of the stored result.  gelu(x) = h + |h| erf(|x|/sqrt2), h = x/2.
__device__ __forceinline__ float pv_gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.47047f * z);
  const float poly = t * (0.3480242f + t * (-0.0958798f + t * 0.7478556f));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // exp(-z^2) = 2^(-x^2/2 * log2 e)
  const float h = 0.5f * x;
  return h + fabsf(h) * (1.0f - poly * e);
}

// Two elements at once on the packed fp32 pipe (v_pk_mul / v_pk_fma_f32: one issue slot per PAIR; the reciprocal and the
// exponential stay per element): 17 issue slots per pair against 24 -- the fused MLP kernel is issue-bound, not
// MFMA-bound (pv_mlp.hip).  Same operations in the same order as pv_gelu_fast (0.5 (x + |x| (1 - poly e)) = h + |h| (...)
// exactly: the scaling by 0.5 is exact), so both give the same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pv_gelu_fast2(f32x2 x) {
  f32x2 ax;
  ax[0] = fabsf(x[0]);
  ax[1] = fabsf(x[1]);
  const f32x2 z = ax * 0.70710678118654752440f;
  const f32x2 dn = z * 0.47047f + 1.0f;
  f32x2 t;
  t[0] = __builtin_amdgcn_rcpf(dn[0]);
  t[1] = __builtin_amdgcn_rcpf(dn[1]);
  const f32x2 poly = t * (0.3480242f + t * (-0.0958798f + t * 0.7478556f));
  const f32x2 a = (x * x) * -0.72134752044448170368f;
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(a[0]);
  e[1] = __builtin_amdgcn_exp2f(a[1]);
  return (x + ax * (1.0f - poly * e)) * 0.5f;
}

// Activation of a small register array with ONE wave-uniform branch around straight-line loops (a
// per-element switch on a runtime act code makes the compiler emit a branch tree per element).
// FAST: bf16 kernels may use the cheap GELU; fp32 kernels keep erff.
template <bool FAST, int N> __device__ __forceinline__ void pv_apply_act_n(float (&v)[N], int act) {
  if (act == PV_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = fmaxf(v[j], 0.0f);
  } else if (act == PV_ACT_SWISH) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] *= pv_sigmoid(v[j]);
  } else if (act == PV_ACT_GELU) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = FAST ? pv_gelu_fast(v[j]) : 0.5f * v[j] * (1.0f + erff(v[j] * 0.70710678118654752440f));
  } else if (act == PV_ACT_SIGMOID) {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = pv_sigmoid(v[j]);
  }
}

__host__ __device__ __forceinline__ int pv_round_up(int v, int m) { return (v + m - 1) / m * m; }
__host__ __device__ __forceinline__ long pv_ceil_div(long a, long b) { return (a + b - 1) / b; }

__device__ __forceinline__ float pv_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float pv_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
