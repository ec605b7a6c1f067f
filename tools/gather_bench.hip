// gather_bench.hip -- how fast can an MI355X do random 4-byte gathers?  (hardware ceiling for
// the multiply+reduce kernels)
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o build/gather_bench && build/gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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
  if (skew) {  // RMAT-like: and a second draw together => skew towards small ids
    uint32_t w = (uint32_t)((h >> 32) % table);
    v = v & w;
  }
  idx[i] = (int)v;
}
template <int U>
__global__ void __launch_bounds__(256) k_gather(const int* __restrict__ idx, const float* __restrict__ x, size_t n, float* out) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
  size_t stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    int c[U];
#pragma unroll
    for (int j = 0; j < U; j++) c[j] = idx[i + j * stride];
#pragma unroll
    for (int j = 0; j < U; j++) acc += x[c[j]];
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  const size_t n = (size_t)1 << 29;  // 512M gathers
  int* idx; float* x; float* out;
  hipMalloc(&idx, n * 4); hipMalloc(&out, 64);
  const size_t maxtab = (size_t)1 << 26;
  hipMalloc(&x, maxtab * 4); hipMemset(x, 0, maxtab * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int skew = 0; skew < 2; skew++)
    for (int lg = 16; lg <= 26; lg += 2) {
      uint32_t table = 1u << lg;
      k_fill_idx<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, table, skew);
      for (int u : {1, 4, 8}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(a);
          if (u == 1) k_gather<1><<<256 * 32, 256>>>(idx, x, n, out);
          if (u == 4) k_gather<4><<<256 * 32, 256>>>(idx, x, n, out);
          if (u == 8) k_gather<8><<<256 * 32, 256>>>(idx, x, n, out);
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("skew=%d table=%8.1f MB unroll=%d : %7.3f ms  %7.1f Ggather/s  (idx stream %.0f GB/s)\n", skew, table * 4.0 / 1e6, u, best,
               n / best / 1e6, n * 4.0 / best / 1e6);
      }
    }
  return 0;
}
