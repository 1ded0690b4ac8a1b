// Micro-benchmark: how fast can 64-lane waves write a [rows][ld] bf16 matrix when a store instruction
// covers (a) 64 x 16 B contiguous, (b) 32 rows x 2 pieces of 16 B (the GEMM epilogue's pattern),
// (c) 16 rows x 64 B (the streaming pointwise kernel's), (d) 8 rows x 128 B.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o gpurun_out/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// rows_per_inst rows, each lane-group of (64/rows_per_inst) lanes writes contiguous 16-B chunks of one row
template <int RPI>
__global__ __launch_bounds__(256) void k(u32x4* y, long rows, int ld16 /* row pitch in 16-B units */, int width16) {
  constexpr int LPR = 64 / RPI;          // lanes per row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r_in = lane / LPR, c_in = lane % LPR;
  const long nrb = rows / RPI;           // row blocks
  const int ncb = width16 / LPR;         // column blocks
  const long total = nrb * ncb;
  for (long t = (long)blockIdx.x * 4 + wave; t < total; t += (long)gridDim.x * 4) {
    const long rb = t / ncb; const int cb = (int)(t - rb * ncb);
    const long row = rb * RPI + r_in;
    u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
    y[row * ld16 + cb * LPR + c_in] = v;
  }
}
// the GEMM epilogue: lane (l31, hi) -> row l31, 16-B piece at column (hi*2 + h8) * 16 B, 4 store instructions
__global__ __launch_bounds__(256) void k_gemm(u32x4* y, long rows, int ld16, int width16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long nrb = rows / 32; const int ncb = width16 / 8;   // 8 pieces (64 channels) per wave column block
  const long total = nrb * ncb;
  for (long t = (long)blockIdx.x * 4 + wave; t < total; t += (long)gridDim.x * 4) {
    const long rb = t / ncb; const int cb = (int)(t - rb * ncb);
    const long row = rb * 32 + l31;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
        y[row * ld16 + cb * 8 + a * 4 + hi * 2 + h8] = v;
      }
  }
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
  const long rows = 401408; const int width = 576;          // MViT block-1 qkv output (bf16)
  const int ld16 = width * 2 / 16, width16 = ld16;          // 72 16-B units per row
  u32x4* y; hipMalloc(&y, rows * ld16 * 16); 
  const double gb = rows * (double)ld16 * 16 / 1e9;
  const int grid = 2048;
  float t;
  t = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, y, rows, ld16, width16 / 64 * 64); });
  printf("64x16B contiguous (1 row/inst, 64 of 72 cols): %.3f ms  %.0f GB/s\n", t, gb * 64 / 72 / t * 1e3);
  t = timeit([&] { hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, y, rows, ld16, width16); });
  printf("8 rows x 128 B per inst: %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
  t = timeit([&] { hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, y, rows, ld16, width16); });
  printf("16 rows x 64 B per inst: %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
  t = timeit([&] { hipLaunchKernelGGL(k<32>, dim3(grid), dim3(256), 0, 0, y, rows, ld16, width16); });
  printf("32 rows x 32 B per inst: %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
  t = timeit([&] { hipLaunchKernelGGL(k<64>, dim3(grid), dim3(256), 0, 0, y, rows, ld16, width16); });
  printf("64 rows x 16 B per inst: %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
  t = timeit([&] { hipLaunchKernelGGL(k_gemm, dim3(grid), dim3(256), 0, 0, y, rows, ld16, width16); });
  printf("GEMM epilogue pattern (32 rows x 2 x 16 B, 4 inst): %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
  return 0;
}
