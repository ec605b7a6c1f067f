// gather_xcd_bench.hip -- does column-slicing the gathers per XCD help?  Each workgroup b only
// gathers table lines whose (line index % 8) == b % 8 (workgroup b is dispatched to XCD b % 8), so the
// eight 4 MB L2s hold disjoint parts of the table instead of eight copies of the same hot lines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// RMAT-like skew: id = AND of two uniform draws; sliced: force line%8 == slice
__global__ void k_fill(int* idx, size_t n, uint32_t table, int sliced, int nblocks) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t h = splitmix64(i);
  uint32_t v = ((uint32_t)(h % table)) & ((uint32_t)((h >> 32) % table));
  if (sliced) {
    // element i is consumed by workgroup ((i / 256) % nblocks): its slice is that % 8
    uint32_t blk = (uint32_t)((i / 256) % nblocks);
    uint32_t line = v >> 4;
    line = (line & ~7u) | (blk & 7u);
    v = (line << 4) | (v & 15u);
    if (v >= table) v -= 128;
  }
  idx[i] = (int)v;
}
__global__ void __launch_bounds__(256) k_gather(const int* __restrict__ idx, const float* __restrict__ x, size_t n, float* out) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
  size_t stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int c0 = idx[i], c1 = idx[i + stride], c2 = idx[i + 2 * stride], c3 = idx[i + 3 * stride];
    acc += x[c0] + x[c1] + x[c2] + x[c3];
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  const size_t n = (size_t)1 << 29;
  const int nblocks = 256 * 32;
  int* idx; float* x; float* out;
  hipMalloc(&idx, n * 4); hipMalloc(&out, 64);
  const size_t maxtab = (size_t)1 << 26;
  hipMalloc(&x, maxtab * 4); hipMemset(x, 0, maxtab * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int lg = 22; lg <= 26; lg += 2)
    for (int sliced = 0; sliced < 2; sliced++) {
      uint32_t table = 1u << lg;
      k_fill<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, table, sliced, nblocks);
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        k_gather<<<nblocks, 256>>>(idx, x, n, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      printf("table=%7.1f MB skewed %s : %7.3f ms  %7.1f Ggather/s\n", table * 4.0 / 1e6, sliced ? "XCD-sliced" : "unsliced  ", best, n / best / 1e6);
    }
  return 0;
}
