// rowgather_bench.hip -- bandwidth of random 512-byte row gathers (the SGD K=128 access pattern) as a function of
// the table size: does a table that fits the 256 MB Infinity Cache gather faster than one in HBM?
//   hipcc --offload-arch=gfx950 -O3 tools/rowgather_bench.hip -o build/rowgather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void k_fill_idx(int* idx, size_t n, uint32_t rows) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = (int)(splitmix64(i) % rows);
}
// a wave gathers 2 rows of 512 B per load instruction (32 lanes x float4 each), U instructions in flight
template <int U>
__global__ void __launch_bounds__(256) k_rows(const int* __restrict__ idx, const float4* __restrict__ x, size_t nrows_to_read, float* out) {
  const int lane = threadIdx.x & 63, sub = lane >> 5, part = lane & 31;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
  float acc = 0.f;
  for (size_t r = wave * 2 * U; r + 2 * U <= nrows_to_read; r += nwaves * 2 * U) {
    float4 q[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const int c = __builtin_nontemporal_load(&idx[r + 2 * j + sub]);
      q[j] = x[(size_t)c * 32 + part];
    }
#pragma unroll
    for (int j = 0; j < U; j++) acc += q[j].x + q[j].y + q[j].z + q[j].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  const size_t n = (size_t)1 << 27;  // 128 M row reads = 64 GiB moved
  int* idx; float4* x; float* out;
  hipMalloc(&idx, n * 4); hipMalloc(&out, 64);
  const size_t maxrows = (size_t)1 << 24;  // 8 GiB table
  hipMalloc(&x, maxrows * 512); hipMemset(x, 0, maxrows * 512);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int lg = 13; lg <= 24; lg++) {
    uint32_t rows = 1u << lg;
    k_fill_idx<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, rows);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a);
      k_rows<8><<<256 * 16, 256>>>(idx, x, n, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("table %8.1f MB (%8u rows of 512 B): %7.3f ms  %6.2f TB/s  %6.2f G rows/s\n", rows * 512.0 / 1e6, rows, best, n * 512.0 / best / 1e9, n / best / 1e6);
  }
  return 0;
}
