// Calibration of SQ_INST_LEVEL_VMEM (rocprofv3 --pmc): waves that keep exactly CH dependent-load chains in flight, one wave per
// workgroup.  The kernel times its own loads (cycle counter around the loop), so  LEVEL / INSTS  of the counter pass can be put
// beside a latency known in cycles, and  LEVEL / WAVE_CYCLES  beside the CH instructions that are outstanding all the time.
//   hipcc --offload-arch=gfx950 -O3 vmem_level.hip -o vmem_level.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH>
__global__ void chase_kernel(const unsigned* __restrict__ next, unsigned* out, long long* cyc, int iters, unsigned mask) {
  unsigned p[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) p[c] = ((blockIdx.x * 64u + threadIdx.x) * 9973u + c * 7919u * 4099u) & mask;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) p[c] = __builtin_nontemporal_load(next + p[c]);
  }
  const long long t1 = __builtin_readcyclecounter();
  unsigned acc = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) acc ^= p[c];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const unsigned N = 1u << 26;   // 256 MB of indices: every load misses the caches
  std::vector<unsigned> h(N);
  unsigned long long x = 88172645463325252ull;
  for (unsigned i = 0; i < N; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (unsigned)(x >> 20) & (N - 1); }
  unsigned *next, *out; long long* cyc;
  const int blocks = 1024, iters = 2000;
  hipMalloc(&next, (size_t)N * 4); hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
  hipMemcpy(next, h.data(), (size_t)N * 4, hipMemcpyHostToDevice);
  std::vector<long long> hc(blocks);
  auto report = [&](const char* name, int ch) {
    hipDeviceSynchronize();
    hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (long long v : hc) s += (double)v;
    printf("%s: %d chains per wave, %d waves: %.0f cycles per loop trip (cycle counter), %.0f per load instruction if the chains overlap fully\n",
           name, ch, blocks, s / blocks / iters, s / blocks / iters);
  };
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(chase_kernel<1>, dim3(blocks), dim3(64), 0, 0, next, out, cyc, iters, N - 1); report("chase_kernel<1>", 1);
    hipLaunchKernelGGL(chase_kernel<4>, dim3(blocks), dim3(64), 0, 0, next, out, cyc, iters, N - 1); report("chase_kernel<4>", 4);
  }
  return 0;
}
