// flag hand-over latency between workgroups (same XCD / other XCD), with and without an 8 KB payload: sizes the persistent dense solve
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

__device__ __forceinline__ long long wall() { return wall_clock64(); }

// WG a and WG b bounce `rounds` times; payload doubles are written by the sender before its release and read by the receiver after its acquire
__global__ void __launch_bounds__(512) k_pp(int a, int b, int rounds, int payload, int* flag, double* buf, long long* out, double* sink) {
  const int me = blockIdx.x;
  if (me != a && me != b) return;
  const int tid = threadIdx.x;
  __shared__ int sh_seen;
  double acc = 0.0;
  long long t0 = 0;
  if (tid == 0) t0 = wall();
  for (int r = 0; r < rounds; ++r) {
    const bool my_turn_first = (me == a);
    for (int half = 0; half < 2; ++half) {
      const bool send = (half == 0) == my_turn_first;
      const int token = 2 * r + half + 1;
      if (send) {
        for (int i = tid; i < payload; i += 512) buf[i] = (double)(token + i);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (tid == 0) {
          long spins = 0;
          while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < token) { if (++spins > (1L << 24)) break; }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int i = tid; i < payload; i += 512) acc += buf[i];
      }
      __syncthreads();
    }
  }
  if (tid == 0) out[me == a ? 0 : 1] = wall() - t0;
  sink[me * 512 + tid] = acc;
}

int main() {
  int* flag; double* buf; long long* out; double* sink;
  CHECK(hipMalloc(&flag, 4)); CHECK(hipMalloc(&buf, 65536 * 8)); CHECK(hipMalloc(&out, 16)); CHECK(hipMalloc(&sink, 64 * 512 * 8));
  const int rounds = 2000;
  for (int payload : {0, 1024, 4096}) {
    for (int b : {8, 1, 4}) {
      CHECK(hipMemset(flag, 0, 4));
      hipLaunchKernelGGL(k_pp, dim3(16), dim3(512), 0, 0, 0, b, rounds, payload, flag, buf, out, sink);
      CHECK(hipDeviceSynchronize());
      long long h[2]; CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      printf("payload %5d doubles, WG 0 <-> WG %d: %.3f us per one-way hand-over (100 MHz clock: %lld ticks for %d round trips)\n", payload, b,
             (double)h[0] * 0.01 / (2.0 * rounds), h[0], rounds);
    }
  }
  return 0;
}
