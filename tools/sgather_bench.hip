// tools/sgather_bench.hip -- can the SCALAR memory path (s_load_dword through the scalar data cache) deliver random
// 4-byte gathers BESIDE the vector path (global_load through TA / the CU's L1)?  
//
// The multiply's gathers are bound by the per-CU L1 fill path (~0.8 G missing gathers/s per CU whatever serves the
// miss).  The scalar cache is a separate client of the L2 with 64-byte lines; if it sustains a useful rate of its own,
// a kernel could send part of its gathers that way.  Three kernels over the same pre-generated index stream:
//   V   one vector gather per index (the baseline of tools/gather_bench.hip, unroll 8)
//   S   every index through v_readlane + s_load_dword, 16 scalar loads in flight per wave
//   M   per 64 indices gathered by the vector path, NS indices gathered by the scalar path (the rest vector)
// hipcc --offload-arch=gfx950 -O3 tools/sgather_bench.hip -o build/sgather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void k_fill_idx(uint32_t* idx, size_t n, uint32_t table) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = (uint32_t)(splitmix64(i) % table) * 4u;  // byte offsets
}
__global__ void k_fill_x(float* x, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (float)(i & 1023) * 0.001f;
}

// 16 scalar gathers: offsets come out of the lanes lane0..lane0+15 of `off`
__device__ __forceinline__ float sgather16(const float* x, uint32_t off, int lane0) {
  uint32_t o[16];
#pragma unroll
  for (int j = 0; j < 16; j++) o[j] = __builtin_amdgcn_readlane(off, lane0 + j);
  float v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15;
  asm volatile(
      "s_load_dword %0, %16, %17\n s_load_dword %1, %16, %18\n s_load_dword %2, %16, %19\n s_load_dword %3, %16, %20\n"
      "s_load_dword %4, %16, %21\n s_load_dword %5, %16, %22\n s_load_dword %6, %16, %23\n s_load_dword %7, %16, %24\n"
      "s_load_dword %8, %16, %25\n s_load_dword %9, %16, %26\n s_load_dword %10, %16, %27\n s_load_dword %11, %16, %28\n"
      "s_load_dword %12, %16, %29\n s_load_dword %13, %16, %30\n s_load_dword %14, %16, %31\n s_load_dword %15, %16, %32\n"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3), "=&s"(v4), "=&s"(v5), "=&s"(v6), "=&s"(v7), "=&s"(v8), "=&s"(v9), "=&s"(v10), "=&s"(v11),
        "=&s"(v12), "=&s"(v13), "=&s"(v14), "=&s"(v15)
      : "s"(x), "s"(o[0]), "s"(o[1]), "s"(o[2]), "s"(o[3]), "s"(o[4]), "s"(o[5]), "s"(o[6]), "s"(o[7]), "s"(o[8]), "s"(o[9]), "s"(o[10]), "s"(o[11]),
        "s"(o[12]), "s"(o[13]), "s"(o[14]), "s"(o[15])
      : "memory");
  return ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)) + ((v8 + v9) + (v10 + v11)) + ((v12 + v13) + (v14 + v15));
}

// V: NV vector gathers per lane and step
template <int NV>
__global__ void __launch_bounds__(256) k_vec(const uint32_t* __restrict__ idx, const float* __restrict__ x, size_t n, float* out) {
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (size_t base = wave * (64 * NV); base + 64 * NV <= n; base += nwaves * (64 * NV)) {
    uint32_t c[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) c[j] = __builtin_nontemporal_load(&idx[base + j * 64 + lane]);
#pragma unroll
    for (int j = 0; j < NV; j++) acc += *(const float*)((const char*)x + c[j]);
  }
  if (acc == 12345.678f) out[0] = acc;
}

// M: per step NV*64 vector gathers and NS*16 scalar gathers per wave (NS = 0: V; NV = 0: S)
template <int NV, int NS>
__global__ void __launch_bounds__(256) k_mix(const uint32_t* __restrict__ idx, const float* __restrict__ x, size_t n, float* out) {
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  constexpr int SW = (NS * 16 + 63) / 64;  // index words per lane that feed the scalar gathers
  constexpr int PER = 64 * NV + 64 * SW;   // indices consumed per step (the scalar part may use only NS*16 of its 64*SW)
  float acc = 0.f, sacc = 0.f;
  for (size_t base = wave * PER; base + PER <= n; base += nwaves * PER) {
    uint32_t c[NV > 0 ? NV : 1], s[SW > 0 ? SW : 1];
#pragma unroll
    for (int j = 0; j < NV; j++) c[j] = __builtin_nontemporal_load(&idx[base + j * 64 + lane]);
#pragma unroll
    for (int j = 0; j < SW; j++) s[j] = __builtin_nontemporal_load(&idx[base + (NV + j) * 64 + lane]);
    float g[NV > 0 ? NV : 1];
#pragma unroll
    for (int j = 0; j < NV; j++) g[j] = *(const float*)((const char*)x + c[j]);  // in flight while the scalar loads run
#pragma unroll
    for (int k = 0; k < NS; k++) sacc += sgather16(x, s[k / 4], (k % 4) * 16);
#pragma unroll
    for (int j = 0; j < NV; j++) acc += g[j];
  }
  acc += sacc;
  if (acc == 12345.678f) out[0] = acc;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; OK(hipEventSynchronize(b)); OK(hipEventElapsedTime(&ms, a, b)); return ms; }

template <int NV, int NS>
static void run(const char* name, const uint32_t* idx, const float* x, size_t n, float* out, int per_cu, double table_mb, hipEvent_t e0, hipEvent_t e1) {
  constexpr int SW = (NS * 16 + 63) / 64;
  constexpr int PER = 64 * NV + 64 * SW;
  const size_t steps = n / PER;
  const double gathers = (double)steps * (64 * NV + 16 * NS);
  float best = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    OK(hipEventRecord(e0));
    k_mix<NV, NS><<<256 * per_cu, 256>>>(idx, x, n, out);
    OK(hipEventRecord(e1));
    const float ms = time_ms(e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("table %7.1f MB  %-28s %d wg/CU: %8.3f ms  %7.1f G gathers/s  (vector %6.1f, scalar %6.1f G/s)\n", table_mb, name, per_cu, best, gathers / best * 1e-6,
         (double)steps * 64 * NV / best * 1e-6, (double)steps * 16 * NS / best * 1e-6);
}

int main(int argc, char** argv) {
  const size_t n = (size_t)1 << 28;
  uint32_t* idx; float *x, *out;
  OK(hipMalloc(&idx, n * 4)); OK(hipMalloc(&out, 64));
  const size_t maxtab = (size_t)1 << 26;
  OK(hipMalloc(&x, maxtab * 4));
  k_fill_x<<<(unsigned)(maxtab / 256), 256>>>(x, maxtab);
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  for (int lg : {14, 20, 23, 26}) {
    const uint32_t table = 1u << lg;
    const double mb = table * 4e-6;
    k_fill_idx<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, table);
    OK(hipDeviceSynchronize());
    for (int per_cu : {4, 8}) {
      run<8, 0>("vector only (8 per lane)", idx, x, n, out, per_cu, mb, e0, e1);
      run<0, 4>("scalar only (64 per wave)", idx, x, n / 4, out, per_cu, mb, e0, e1);
      run<8, 1>("8 x 64 vector + 16 scalar", idx, x, n, out, per_cu, mb, e0, e1);
      run<8, 2>("8 x 64 vector + 32 scalar", idx, x, n, out, per_cu, mb, e0, e1);
      run<8, 4>("8 x 64 vector + 64 scalar", idx, x, n, out, per_cu, mb, e0, e1);
      run<4, 4>("4 x 64 vector + 64 scalar", idx, x, n, out, per_cu, mb, e0, e1);
    }
  }
  return 0;
}
