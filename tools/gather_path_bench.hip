// gather_path_bench.hip -- is there gather capacity next to the vector L1 (TCP) path?
// The production multiply is bound by the per-CU vector memory path (TA busy ~100 %, ~110 requests outstanding per
// CU).  A wave can also load through the SCALAR data cache: an index made wave-uniform (v_readlane) turns x[c] into
// s_load_dword, which does not pass the TA/TCP at all.  This measures 4-byte gathers per second for
//   V     : every gather a per-lane vector load (the baseline)
//   S     : every gather a scalar load (64 per 64 indices)
//   Mk    : per 8 index vectors of a wave, k go through the scalar path and 8-k through the vector path
// over an L2-resident and a larger table, uniform and skewed indices.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_path_bench.hip -o build/gather_path_bench
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
// one wave works on 8 consecutive 64-index vectors per step; NS of them are gathered through the scalar cache
template <int NS>
__global__ void __launch_bounds__(256) k_gather(const int* __restrict__ idx, const float* __restrict__ x, size_t n, float* out) {
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
  float acc = 0.f;
  for (size_t base = wave * (U * 64); base + U * 64 <= n; base += nwaves * (U * 64)) {
    int c[U];
#pragma unroll
    for (int j = 0; j < U; j++) c[j] = __builtin_nontemporal_load(&idx[base + j * 64 + lane]);
    float v[U];
#pragma unroll
    for (int j = NS; j < U; j++) v[j] = x[c[j]];       // vector path, all issued before anything is consumed
#pragma unroll
    for (int j = 0; j < NS; j++) {                      // scalar path: 64 uniform loads per index vector
#pragma unroll
      for (int i = 0; i < 64; i++) {
        const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane(c[j], i);
        acc += x[ci];
      }
    }
#pragma unroll
    for (int j = NS; j < U; j++) acc += v[j];
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int NS>
static float run(const int* idx, const float* x, size_t n, float* out, int wgs) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(a);
    k_gather<NS><<<wgs, 256>>>(idx, x, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best;
}
int main() {
  const size_t n = (size_t)1 << 28;
  int* idx; float* x; float* out;
  hipMalloc(&idx, n * 4); hipMalloc(&out, 64);
  const size_t maxtab = (size_t)1 << 26;
  hipMalloc(&x, maxtab * 4); hipMemset(x, 0, maxtab * 4);
  for (int skew = 0; skew < 2; skew++)
    for (int lg : {18, 22, 25}) {
      uint32_t table = 1u << lg;
      k_fill_idx<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, table, skew);
      const int wgs = 256 * 8;
      float t[6];
      t[0] = run<0>(idx, x, n, out, wgs);
      t[1] = run<1>(idx, x, n, out, wgs);
      t[2] = run<2>(idx, x, n, out, wgs);
      t[3] = run<3>(idx, x, n, out, wgs);
      t[4] = run<4>(idx, x, n, out, wgs);
      t[5] = run<8>(idx, x, n, out, wgs);
      const char* nm[6] = {"V (0/8 scalar)", "M1 (1/8)", "M2 (2/8)", "M3 (3/8)", "M4 (4/8)", "S (8/8 scalar)"};
      for (int k = 0; k < 6; k++)
        printf("skew=%d table=%7.1f MB %-15s: %7.3f ms  %6.1f Ggather/s\n", skew, table * 4.0 / 1e6, nm[k], t[k], n / t[k] / 1e6);
    }
  return 0;
}
