// Is  v_pk_fma_f32 vD, vA, vD, vC op_sel_hi:[1,0,1]  (the upper lane multiplies by the LOWER half of vD, the register pair the
// instruction itself overwrites) safe on gfx950 when LDS data is returning into the register file and another kernel keeps the
// CU's LDS busy?  hipcc's SLP vectoriser produced exactly this form in bottleneck_block_kernel's one-channel-per-lane variant
// (round 6), whose output column 0 of a segment was not reproducible beside stride-2 pwdw_plane_kernel / the stem kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_fma_inplace.hip -o /tmp/pk_fma_inplace && /tmp/pk_fma_inplace
// Prints, per form (in place / separate destination), the number of threads whose chained result differs from the host's, alone
// and beside the LDS-heavy kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <bool INPLACE>
__global__ __launch_bounds__(512) void victim(float* out, const float* seed, int iters) {
  __shared__ float lds[512 * 9];
  const int tid = threadIdx.x;
  for (int i = tid; i < 512 * 9; i += 512) lds[i] = seed[(blockIdx.x * 7 + i) & 4095];
  __syncthreads();
  f2 xy = {seed[tid & 4095], 0.25f};
  const f2 w = {0.75f, -0.5f}, c = {0.125f, 0.0625f};
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
    // nine LDS reads in flight, as the stencil has them, consumed AFTER the packed op
    float l[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) l[k] = lds[((tid + it * 31 + k * 57) % 512) * 9 + k];
    if (INPLACE) {
      asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[1,0,1]" : "+v"(xy) : "v"(w), "v"(c));
    } else {
      f2 r;
      asm volatile("v_pk_fma_f32 %0, %2, %1, %3 op_sel_hi:[1,0,1]" : "=&v"(r) : "v"(xy), "v"(w), "v"(c));
      xy = r;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) keep += l[k];
    // keep the chain bounded and dependent on both lanes:  x <- 0.5 * (x' + y'),  y <- y'
    xy.x = 0.5f * (xy.x + xy.y);
  }
  out[blockIdx.x * 512 + tid] = xy.x + xy.y;
  if (keep == 123456.f) out[0] = keep;
}

__global__ __launch_bounds__(256) void lds_noise(float* sink, int iters) {
  __shared__ float buf[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) buf[i] = (float)i;
  __syncthreads();
  float a = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 8
    for (int k = 0; k < 32; ++k) a += buf[(tid * 33 + k * 257 + it) & 8191];
    buf[(tid * 17 + it) & 8191] = a;
  }
  if (a == 1.5f) sink[0] = a;
}

static float host_chain(float x0, int iters) {
  float x = x0, y = 0.25f;
  for (int it = 0; it < iters; ++it) {
    const float nx = fmaf(0.75f, x, 0.125f), ny = fmaf(-0.5f, x, 0.0625f);
    y = ny;
    x = 0.5f * (nx + ny);
  }
  return x + y;
}

int main() {
  const int blocks = 2048, iters = 400;
  std::vector<float> seed(4096);
  for (int i = 0; i < 4096; ++i) seed[i] = 0.001f * (float)((i * 37) % 1999) - 1.f;
  float *d_seed, *d_out, *d_sink;
  hipMalloc(&d_seed, 4096 * 4); hipMalloc(&d_out, blocks * 512 * 4); hipMalloc(&d_sink, 4096);
  hipMemcpy(d_seed, seed.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipStream_t s0, s1;
  hipStreamCreate(&s0); hipStreamCreate(&s1);
  std::vector<float> out(blocks * 512), want(512);
  for (int t = 0; t < 512; ++t) want[t] = host_chain(seed[t], iters);
  for (int beside = 0; beside < 2; ++beside)
    for (int inplace = 1; inplace >= 0; --inplace)
      for (int rep = 0; rep < 3; ++rep) {
        if (beside) lds_noise<<<4096, 256, 0, s1>>>(d_sink, 3000);
        if (inplace) victim<true><<<blocks, 512, 0, s0>>>(d_out, d_seed, iters);
        else victim<false><<<blocks, 512, 0, s0>>>(d_out, d_seed, iters);
        hipDeviceSynchronize();
        hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (size_t i = 0; i < out.size(); ++i) bad += out[i] != want[i & 511];
        printf("%s, %s: %ld of %zu threads differ from the host's chain\n", inplace ? "in place (vD is also src1, upper lane reads vD.lo)" : "separate destination",
               beside ? "beside the LDS-heavy kernel" : "alone on the chip", bad, out.size());
      }
  return 0;
}
