// tools/pmc_calibrate.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE counters report for two access patterns whose HBM / fabric
// bytes are known by construction (the review of round 4 asked for both factors next to profiles/pmc_traffic.json):
//   k_cal_stream:  every thread reads consecutive uint32 words of a 2 GiB array once (coalesced, non-temporal) and one word per
//                  1024 is written: bytes = 4 per word read;
//   k_cal_gather:  N random 4-byte gathers from a 1 GiB table (far larger than L2 + Infinity Cache reuse allows), indices from a
//                  hash: every gather that misses moves one 64-byte sector at least: bytes >= 64 per miss, useful bytes 4;
//   k_cal_write:   a 1 GiB array written once, coalesced: bytes = 4 per word.
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE, TCC_MISS_sum in their own passes); tools/final_profiles_r5.sh
// divides counter x 1024 by these known bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calibrate.hip -o build/pmc_calibrate
#include <stdint.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_cal_stream(const uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(&a[i]);
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_cal_gather(const uint32_t* __restrict__ t, size_t tn, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    acc += t[h % tn];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_cal_write(uint32_t* __restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (uint32_t)i;
}
int main() {
  const size_t n = 512ull << 20, tn = 256ull << 20, ng = 256ull << 20;
  uint32_t *a, *t, *out;
  OK(hipMalloc(&a, n * 4)); OK(hipMalloc(&t, tn * 4)); OK(hipMalloc(&out, 64));
  OK(hipMemset(a, 1, n * 4)); OK(hipMemset(t, 1, tn * 4));
  OK(hipDeviceSynchronize());
  for (int r = 0; r < 2; r++) {
    k_cal_stream<<<8192, 256>>>(a, n, out);
    k_cal_gather<<<8192, 256>>>(t, tn, ng, out);
    k_cal_write<<<8192, 256>>>(a, tn);
  }
  OK(hipDeviceSynchronize());
  printf("k_cal_stream: %zu bytes read per launch; k_cal_gather: %zu gathers per launch (>= %zu bytes in 64-byte sectors when every one misses); k_cal_write: %zu bytes written per launch\n",
         n * 4, ng, ng * 64, tn * 4);
  return 0;
}
