// Microbenchmark of the Schur pair-phase LDS pattern: every lane adds a 6x6 block into a random (ci, cj) block
// of a 96 x 113 tile of doubles (36 atomic adds to base + r*ld + c).  Variants isolate what limits the rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

constexpr int LD = 113, ROWS = 96, TILE = ROWS * LD;

template <int MODE>  // 0: ds_add_f64 random blocks, 1: ds_add_u64 random, 2: ds_add_f64 lane-private blocks (no conflicts across lanes),
                     // 3: plain stores random, 4: f64 random but c-major issue order, 5: f64, one (r,c) per instr but rows permuted per lane
__global__ void __launch_bounds__(512) k(const int* __restrict__ idx, int iters, double* out) {
  __shared__ double sh[TILE + 64];
  for (int i = threadIdx.x; i < TILE + 64; i += blockDim.x) sh[i] = 0.0;
  __syncthreads();
  const int* my = idx + (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  double v = 1.0 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
    const int sel = my[it & 7];
    int base;
    if (MODE == 2) base = ((threadIdx.x & 63) % 16) * 6 * LD + ((threadIdx.x & 63) / 16) * 28;  // spread, mostly distinct banks
    else base = (sel & 15) * 6 * LD + ((sel >> 4) & 15) * 7;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int r = (MODE == 4) ? b : a, c = (MODE == 4) ? a : b;
        double* p = &sh[base + r * LD + c];
        if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)(it + a));
        else if (MODE == 3) *reinterpret_cast<volatile double*>(p) = v;
        else unsafeAtomicAdd(p, v);
      }
    v += 1e-9;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sh[5];
}

template <int MODE> void run(const char* name, const int* d_idx, int threads) {
  double* d_out; hipMalloc(&d_out, 4096 * sizeof(double));
  const int grid = 256, iters = 200;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<grid, threads>>>(d_idx, 10, d_out); hipDeviceSynchronize();
  hipEventRecord(a); k<MODE><<<grid, threads>>>(d_idx, iters, d_out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double winstr = (double)grid * (threads / 64) * iters * 36;
  printf("%-44s threads %4d  %7.3f ms  %6.1f cycles per wave-atomic per CU (one WG per CU)\n", name, threads, ms, ms * 1e-3 * 2.4e9 / (winstr / 256));
  hipFree(d_out);
}

int main() {
  std::vector<int> h(256 * 1024 * 8);
  uint32_t s = 777; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 9) & 255; }
  int* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int th : {256, 512, 1024}) {
    run<0>("f64 atomics, random blocks (kernel pattern)", d, th);
    run<1>("u64 atomics, random blocks", d, th);
    run<2>("f64 atomics, lane-spread blocks", d, th);
    run<3>("plain stores, random blocks", d, th);
    run<4>("f64 atomics, random blocks, column-major issue", d, th);
  }
  return 0;
}
