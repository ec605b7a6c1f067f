// gather_mode_bench.hip -- does the cache policy of a 4-byte gather change the chip's gather rate?
// (plain / non-temporal / agent-scope relaxed atomic load = sc1: the latter two do not allocate in the
// CU's L1, so a missing gather need not pull a whole line through the 64 B/clk L1 fill path)
//   hipcc --offload-arch=gfx950 -O3 tools/gather_mode_bench.hip -o build/gather_mode_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void k_fill_idx(int* idx, size_t n, uint32_t table, int skew) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t h = splitmix64(i);
  uint32_t v = (uint32_t)(h % table);
  if (skew) v &= (uint32_t)((h >> 32) % table);
  idx[i] = (int)v;
}
template <int MODE>
__device__ __forceinline__ float ld(const float* p) {
  if (MODE == 1) return __builtin_nontemporal_load(p);
  if (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return *p;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const int* __restrict__ idx, const float* x, size_t n, float* out) {
  constexpr int U = 8;
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
  size_t stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    int c[U];
#pragma unroll
    for (int j = 0; j < U; j++) c[j] = __builtin_nontemporal_load(&idx[i + j * stride]);
#pragma unroll
    for (int j = 0; j < U; j++) acc += ld<MODE>(x + c[j]);
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  const size_t n = (size_t)1 << 29;
  int* idx; float* x; float* out;
  hipMalloc(&idx, n * 4); hipMalloc(&out, 64);
  const size_t maxtab = (size_t)1 << 26;
  hipMalloc(&x, maxtab * 4); hipMemset(x, 0, maxtab * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[4] = {"plain", "nontemporal", "sc1(agent)", "wg-scope"};
  for (int skew = 0; skew < 2; skew++)
    for (int lg = 18; lg <= 26; lg += 2) {
      uint32_t table = 1u << lg;
      k_fill_idx<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, table, skew);
      for (int mode = 0; mode < 4; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(a);
          if (mode == 0) k_gather<0><<<256 * 32, 256>>>(idx, x, n, out);
          if (mode == 1) k_gather<1><<<256 * 32, 256>>>(idx, x, n, out);
          if (mode == 2) k_gather<2><<<256 * 32, 256>>>(idx, x, n, out);
          if (mode == 3) k_gather<3><<<256 * 32, 256>>>(idx, x, n, out);
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("skew=%d table=%7.1f MB %-12s: %7.3f ms  %6.1f Ggather/s\n", skew, table * 4.0 / 1e6, names[mode], best, n / best / 1e6);
      }
    }
  return 0;
}
