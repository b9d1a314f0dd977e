// Standalone check + timing of the 16-wide two-stage block factorisation with the short pivot chain, next sum formed during the chain (fac16_body.h = chol_factor_block of csrc/cba_kernels.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 fac8.hip -o fac8.bin && ./fac8.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
constexpr int NB = 32;
constexpr int WAVE = 64;
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double fast_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}
#include "fac16_body.h"
__global__ void __launch_bounds__(512) kf(const double* A, double* W, int n, int ldw, int* flags, double* xinv, int reps) {
  __shared__ double sh_D[2 * NB][NB + 1], sh_T[NB][NB + 1];
  for (int rep = 0; rep < reps; ++rep) {
    for (int t = threadIdx.x; t < NB * NB; t += 512) sh_D[t / NB][t % NB] = (t / NB < n && t % NB < n) ? A[(long)(t / NB) * ldw + t % NB] : (t / NB == t % NB ? 1.0 : 0.0);
    __syncthreads();
    if (threadIdx.x < WAVE) chol_factor_block(sh_D, sh_T, n, W, ldw, flags, xinv);
    __syncthreads();
    STAMP(5);
    chol_factor_store(sh_D, sh_T, n, W, ldw, xinv, threadIdx.x, 512);
    __syncthreads();
    STAMP(6);
  }
}
int main() {
  for (int n : {32, 27, 1}) {
    const int ldw = 32;
    std::vector<double> A(32 * 32, 0.0), Lref(32 * 32, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * 32 + j] = (i == j ? n + 1.0 : 0.0) + 1.0 / (1.0 + i + j) + 0.01 * ((i * 7 + j * 3) % 5 + (j * 7 + i * 3) % 5);
    for (int j = 0; j < n; ++j) { double d = A[j*32+j]; for (int t = 0; t < j; ++t) d -= Lref[j*32+t]*Lref[j*32+t]; Lref[j*32+j] = std::sqrt(d);
      for (int i = j + 1; i < n; ++i) { double v = A[i*32+j]; for (int t = 0; t < j; ++t) v -= Lref[i*32+t]*Lref[j*32+t]; Lref[i*32+j] = v / Lref[j*32+j]; } }
    double *dA, *dW, *dX; int* f;
    hipMalloc(&dA, 32 * 32 * 8); hipMalloc(&dW, 32 * 32 * 8); hipMalloc(&dX, 32 * 32 * 8); hipMalloc(&f, 16); hipMemset(f, 0, 16); hipMemset(dW, 0, 32 * 32 * 8);
    hipMemcpy(dA, A.data(), 32 * 32 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kf, dim3(1), dim3(512), 0, 0, dA, dW, n, ldw, f, dX, 1);
    std::vector<double> L(32 * 32), X(32 * 32); int flags[4];
    hipMemcpy(L.data(), dW, 32 * 32 * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, 32 * 32 * 8, hipMemcpyDeviceToHost); hipMemcpy(flags, f, 16, hipMemcpyDeviceToHost);
    double eL = 0, eX = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) eL = std::fmax(eL, std::fabs(L[i*32+j] - Lref[i*32+j]));
    // X L = I on the live part (X identity-padded)
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 32; ++k) { const double lkj = (k < n && j < n) ? (j <= k ? Lref[k*32+j] : 0.0) : (k == j ? 1.0 : 0.0); s += X[i*32+k] * lkj; } eX = std::fmax(eX, std::fabs(s - (i == j ? 1.0 : 0.0))); }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 200;
    hipLaunchKernelGGL(kf, dim3(1), dim3(512), 0, 0, dA, dW, n, ldw, f, dX, 5);
    hipEventRecord(a); hipLaunchKernelGGL(kf, dim3(1), dim3(512), 0, 0, dA, dW, n, ldw, f, dX, reps); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    { long long c[8]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)); printf("clocks: stage1 %lld  mfma23 %lld  stage4 %lld  mfma5 %lld  store %lld\n", c[1]-c[0], c[2]-c[1], c[3]-c[2], c[4]-c[3], c[6]-c[5]); }
    printf("n=%d  max |L - Lref| = %.3e  max |X L - I| = %.3e  flag %d   %.2f us per block (incl. ~0.3 us of load + barriers)\n", n, eL, eX, flags[2], ms * 1e3 / reps);
  }
  // not positive definite: flag
  { std::vector<double> A(32 * 32, 0.0); for (int i = 0; i < 32; ++i) A[i * 32 + i] = 1.0; A[5 * 32 + 5] = -1.0;
    double *dA, *dW, *dX; int* f; hipMalloc(&dA, 8192); hipMalloc(&dW, 8192); hipMalloc(&dX, 8192); hipMalloc(&f, 16); hipMemset(f, 0, 16);
    hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kf, dim3(1), dim3(512), 0, 0, dA, dW, 32, 32, f, dX, 1); int flags[4]; hipMemcpy(flags, f, 16, hipMemcpyDeviceToHost);
    printf("indefinite block: flag %d (expected 1)\n", flags[2]); }
  return 0;
}
