// Microbenchmark: throughput of LDS atomics on gfx950 (drives the Schur-accumulation design).
// Each wave issues N atomic adds per lane to a 64 KiB LDS array; address patterns: linear (conflict-free),
// random, and "few" (many lanes on the same address).  Reports wave-instructions per microsecond per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <typename T> __device__ void add(T* p, T v);
template <> __device__ void add<double>(double* p, double v) { unsafeAtomicAdd(p, v); }
template <> __device__ void add<float>(float* p, float v) { unsafeAtomicAdd(p, v); }
template <> __device__ void add<unsigned long long>(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
template <> __device__ void add<unsigned>(unsigned* p, unsigned v) { atomicAdd(p, v); }

template <typename T, int MODE>
__global__ void __launch_bounds__(256) k(const int* __restrict__ idx, int iters, T* out) {
  __shared__ T sh[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) sh[i] = T(0);
  __syncthreads();
  const int lane = threadIdx.x;
  int a = (MODE == 0) ? lane : idx[blockIdx.x * 256 + lane];
  if (MODE == 2) a &= 15;  // 16 distinct addresses per block
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) add<T>(&sh[(a + u * 37) & 8191], T(1));
    a = (a * 5 + 1) & 8191;
    if (MODE == 0) a = (lane + it * 256) & 8191;
    if (MODE == 2) a &= 15;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

template <typename T, int MODE>
void run(const char* name, const int* d_idx) {
  T* d_out; hipMalloc(&d_out, 4096 * sizeof(T));
  const int grid = 256 * 4, iters = 200;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<T, MODE><<<grid, 256>>>(d_idx, 10, d_out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<T, MODE><<<grid, 256>>>(d_idx, iters, d_out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double wave_instr = (double)grid * 4 * iters * 16;
  printf("%-28s %8.3f ms  %8.1f wave-atomics/us/CU  (%.1f cycles per wave-instr per CU @2.4GHz)\n", name, ms,
         wave_instr / (ms * 1e3) / 256, 2400.0 / (wave_instr / (ms * 1e3) / 256));
  hipFree(d_out);
}

int main() {
  std::vector<int> h(256 * 4 * 256);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) & 8191; }
  int* d_idx; hipMalloc(&d_idx, h.size() * 4); hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<double, 0>("f64 linear", d_idx); run<double, 1>("f64 random", d_idx); run<double, 2>("f64 16-addr", d_idx);
  run<unsigned long long, 0>("u64 linear", d_idx); run<unsigned long long, 1>("u64 random", d_idx); run<unsigned long long, 2>("u64 16-addr", d_idx);
  run<float, 0>("f32 linear", d_idx); run<float, 1>("f32 random", d_idx); run<float, 2>("f32 16-addr", d_idx);
  run<unsigned, 0>("u32 linear", d_idx); run<unsigned, 1>("u32 random", d_idx); run<unsigned, 2>("u32 16-addr", d_idx);
  return 0;
}
