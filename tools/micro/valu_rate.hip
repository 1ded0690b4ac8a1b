// Issue cost of the VALU instructions the attention softmax is made of (one wave alone on a SIMD, 16 independent chains):
// cycles per instruction = (s_memtime delta) / (iterations x 16).  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void k(float* out, long long* cyc, int iters, float seed) {
  float v[16];
  float w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f; w[i] = seed * 0.5f + i; }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define X_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define X_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
#define X_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define X_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
#define X_CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define X_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[(i) & 14])) : "v"(*reinterpret_cast<double*>(&w[(i) & 14])));
#define X_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define X_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(w[i]));
    if (OP == 0) { REP16(X_EXP) }
    if (OP == 1) { REP16(X_FMA) }
    if (OP == 2) { REP16(X_ADD) }
    if (OP == 3) { REP16(X_MAX3) }
    if (OP == 4) { REP16(X_CVT) }
    if (OP == 5) { REP16(X_PKADD) }
    if (OP == 6) { REP16(X_RCP) }
    if (OP == 7) { REP16(X_MOV) }
    if (OP == 8) { X_FMA(0) X_EXP(1) X_FMA(2) X_EXP(3) X_FMA(4) X_EXP(5) X_FMA(6) X_EXP(7) X_FMA(8) X_EXP(9) X_FMA(10) X_EXP(11) X_FMA(12) X_EXP(13) X_FMA(14) X_EXP(15) }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP> void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 4000;
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 0.25f);
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 0.25f);
  hipDeviceSynchronize();
  std::vector<long long> h(threads / 64);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  printf("%-22s %4d threads: s_memtime ticks per instruction and wave:", name, threads);
  for (size_t i = 0; i < h.size() && i < 8; ++i) printf(" %.2f", (double)h[i] / (iters * 16.0));
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("clockRate %d kHz, wall clock rate %d kHz (s_memtime counts at a fixed rate: scale to the shader clock)\n", p.clockRate, p.clockInstructionRate);
  for (int threads : {64, 512}) {   // 512 threads = 8 waves = two per SIMD
    run<0>("v_exp_f32", threads); run<1>("v_fma_f32", threads); run<2>("v_add_f32", threads); run<3>("v_max3_f32", threads);
    run<4>("v_cvt_pk_bf16_f32", threads); run<5>("v_pk_add_f32", threads); run<6>("v_rcp_f32", threads); run<7>("v_mov_b32", threads);
    run<8>("fma/exp alternating", threads);
  }
  return 0;
}
