// tools/dpp_fold_test.hip -- the ordered fp32 fold of a wave's 64 products as a chain of 64 v_add_f32_dpp wave_shr:1
// (lane l takes lane l-1's running value and adds its own product; after step k lanes < k are final and stay final),
// against the v_readlane + v_add form of kernels.hpp: wave_row: same bits, half the VALU instructions.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dpp_fold_test.hip -o build/dpp_fold_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

#define S1 "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define S4 S1 S1 S1 S1
#define S16 S4 S4 S4 S4
#define S63 S16 S16 S16 S4 S4 S4 S1 S1 S1

__device__ __forceinline__ float fold_dpp(float carry, float t, int lane) {
  float a = t;
  if (lane == 0) a = carry + t;
  asm volatile(S63 "s_nop 1" : "+v"(a) : "v"(t));
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(a), 63));
}
__device__ __forceinline__ float fold_readlane(float carry, float t) {
  float acc = carry;
#pragma unroll
  for (int i = 0; i < 64; i++) acc += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(t), i));
  return acc;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_fold(const float* __restrict__ x, int64_t per_wave, float* __restrict__ out) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const float* p = x + (int64_t)w * per_wave;
  float carry = 0.f;
  float t = p[lane], tn = 0.f;
  for (int64_t k = 0; k < per_wave; k += 64) {
    if (k + 64 < per_wave) tn = p[k + 64 + lane];
    carry = MODE ? fold_dpp(carry, t, lane) : fold_readlane(carry, t);
    t = tn;
  }
  if (lane == 0) out[w] = carry;
}

int main() {
  const int waves = 256 * 32, per_wave = 64 * 512;
  const int64_t n = (int64_t)waves * per_wave;
  std::vector<float> h(n);
  uint32_t s = 12345;
  for (int64_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (1.0f / 16777216.0f) * ((s & 255) < 8 ? 1000.f : 1.f); }
  float *x, *o0, *o1;
  OK(hipMalloc(&x, n * 4)); OK(hipMalloc(&o0, waves * 4)); OK(hipMalloc(&o1, waves * 4));
  OK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
  float ms[2] = {0, 0};
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 3; rep++) {
      OK(hipEventRecord(a));
      if (mode) k_fold<1><<<waves / 4, 256>>>(x, per_wave, o1); else k_fold<0><<<waves / 4, 256>>>(x, per_wave, o0);
      OK(hipEventRecord(b)); OK(hipEventSynchronize(b));
      float t; OK(hipEventElapsedTime(&t, a, b)); ms[mode] = t;
    }
  std::vector<float> r0(waves), r1(waves);
  OK(hipMemcpy(r0.data(), o0, waves * 4, hipMemcpyDeviceToHost)); OK(hipMemcpy(r1.data(), o1, waves * 4, hipMemcpyDeviceToHost));
  int bad0 = 0, bad1 = 0;
  for (int w = 0; w < waves; w++) {
    volatile float acc = 0.f;
    for (int k = 0; k < per_wave; k++) { volatile float v = acc + h[(int64_t)w * per_wave + k]; acc = v; }
    float ref = acc;
    bad0 += memcmp(&ref, &r0[w], 4) != 0;
    bad1 += memcmp(&ref, &r1[w], 4) != 0;
  }
  printf("ordered fold of %lld floats, %d waves: v_readlane + v_add %.3f ms (%.1f G adds/s), v_add_dpp wave_shr:1 chain %.3f ms (%.1f G adds/s); waves differing from the host's serial sum: %d / %d\n",
         (long long)n, waves, ms[0], n / ms[0] * 1e-6, ms[1], n / ms[1] * 1e-6, bad0, bad1);
  return bad0 || bad1;
}
