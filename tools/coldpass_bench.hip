// tools/coldpass_bench.hip -- microbenchmark behind the hot/cold split of the multiply.
//
// The messages of COLD columns (the many low-degree sources whose x entries miss the L2s when gathered at random)
// can reach the rows another way: a first pass sweeps the cold part of x in source order and writes every cold
// edge's message into a stream laid out by (bin of destination rows, chunk of sources); the pull kernels then read
// their bin's part of the stream coalesced, scatter it into LDS by a static 16-bit slot and fold from there.
// This tool times the two new pieces on synthetic structure of the real size:
//   V1  pass 1 with the chunk's x values in LDS (no gathers at all), runs of R consecutive floats per (bin, chunk)
//   V2  pass 1 with L2-resident super-chunks (one L2-hit gather per edge), long sequential output
//   RD  the bin read of pass 2: stream + slot16 coalesced, ds_write scatter, barrier, LDS reads
// hipcc --offload-arch=gfx950 -O3 tools/coldpass_bench.hip -o build/coldpass_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}

// chunk k's edge list: edge i belongs to run i / R (= bin), position i % R inside it
__global__ void k_init_v1(uint16_t* col16, uint32_t* pos, int64_t per_chunk, int R, int C, int chunk_src) {
  const int k = blockIdx.y;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per_chunk; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = (int64_t)k * per_chunk + i;
    col16[e] = (uint16_t)(mix((uint32_t)e * 2654435761u + 17u) % (uint32_t)chunk_src);
    const int64_t b = i / R;
    pos[e] = (uint32_t)((b * C + k) * (int64_t)R + (i % R));
  }
}

template <int CHUNK>
__global__ void __launch_bounds__(1024) k_pass1_v1(const float* __restrict__ x, const uint16_t* __restrict__ col16,
                                                   const uint32_t* __restrict__ pos, float* __restrict__ S, int64_t per_chunk, int split) {
  __shared__ float s_x[CHUNK];
  const int k = blockIdx.x / split, part = blockIdx.x % split;
  for (int i = threadIdx.x; i < CHUNK; i += 1024) s_x[i] = x[(int64_t)k * CHUNK + i];
  __syncthreads();
  const int64_t lo = per_chunk * part / split, hi = per_chunk * (part + 1) / split;
  const int64_t base = (int64_t)k * per_chunk;
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 1024 * 4) {
    uint16_t c[4];
    uint32_t p[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t i = i0 + j * 1024;
      if (i < hi) { c[j] = __builtin_nontemporal_load(&col16[base + i]); p[j] = __builtin_nontemporal_load(&pos[base + i]); }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t i = i0 + j * 1024;
      if (i < hi) S[p[j]] = s_x[c[j]];
    }
  }
}

__global__ void k_init_v2(int32_t* col32, int64_t n, int64_t per_super, int super_src) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = e / per_super;
    col32[e] = (int32_t)(s * super_src + mix((uint32_t)e * 2246822519u + 3u) % (uint32_t)super_src);
  }
}
__global__ void __launch_bounds__(256) k_pass1_v2(const float* __restrict__ x, const int32_t* __restrict__ col32, float* __restrict__ S, int64_t n) {
  const int64_t b0 = (int64_t)blockIdx.x * 2048;
  int c[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { const int64_t e = b0 + threadIdx.x + j * 256; c[j] = e < n ? __builtin_nontemporal_load(&col32[e]) : -1; }
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; j++) m[j] = c[j] >= 0 ? x[c[j]] : 0.f;
#pragma unroll
  for (int j = 0; j < 8; j++) { const int64_t e = b0 + threadIdx.x + j * 256; if (e < n) __builtin_nontemporal_store(m[j], &S[e]); }
}

// pass-2 side: a bin's entries into LDS by slot, then read back in order
template <int BIN, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_read_bins(const float* __restrict__ S, const uint16_t* __restrict__ slot, float* __restrict__ out, int nbins) {
  __shared__ float s_c[BIN];
  float acc = 0.f;
  for (int b = blockIdx.x; b < nbins; b += gridDim.x) {
    const int64_t base = (int64_t)b * BIN;
    for (int i = threadIdx.x; i < BIN; i += BLOCK) s_c[__builtin_nontemporal_load(&slot[base + i])] = __builtin_nontemporal_load(&S[base + i]);
    __syncthreads();
    for (int i = threadIdx.x; i < BIN; i += BLOCK) acc += s_c[i];
    __syncthreads();
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_init_slot(uint16_t* slot, int64_t n, int BIN) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    slot[e] = (uint16_t)((mix((uint32_t)(e / BIN)) + (uint32_t)(e % BIN) * 7919u) % (uint32_t)BIN);  // a permutation of the bin when BIN is coprime to 7919
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; OK(hipEventSynchronize(b)); OK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv) {
  const int64_t E = argc > 1 ? atoll(argv[1]) : 290000000ll;  // cold edges
  constexpr int CHUNK = 32768;
  const int C = 1000;  // chunks of 32 K sources
  hipEvent_t e0, e1;
  OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  float *x, *S, *out;
  OK(hipMalloc(&x, (size_t)C * CHUNK * 4 + (64 << 20)));
  OK(hipMemset(x, 0, (size_t)C * CHUNK * 4));
  OK(hipMalloc(&S, (size_t)(E + (1 << 20)) * 4 + (64ll << 20)));
  OK(hipMalloc(&out, 64));
  {
    uint16_t* col16; uint32_t* pos;
    OK(hipMalloc(&col16, (size_t)(E + (1 << 20)) * 2));
    OK(hipMalloc(&pos, (size_t)(E + (1 << 20)) * 4));
    for (int R : {4, 8, 13, 16, 32, 64, 256}) {
      int64_t per_chunk = E / C / R * R;
      k_init_v1<<<dim3(256, C), 256>>>(col16, pos, per_chunk, R, C, CHUNK);
      OK(hipDeviceSynchronize());
      for (int split : {1, 2}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
          OK(hipEventRecord(e0));
          k_pass1_v1<CHUNK><<<C * split, 1024>>>(x, col16, pos, S, per_chunk, split);
          OK(hipEventRecord(e1));
          const float ms = time_ms(e0, e1);
          if (rep && ms < best) best = ms;
        }
        printf("V1 (x chunk in LDS) run %3d floats split %d: %8.3f ms for %lld edges = %6.1f G edges/s (%.0f GB/s of 10 B/edge)\n", R, split, best,
               (long long)(per_chunk * C), per_chunk * C / best * 1e-6, per_chunk * C * 10.0 / best * 1e-6);
      }
    }
    OK(hipFree(col16)); OK(hipFree(pos));
  }
  {
    int32_t* col32;
    OK(hipMalloc(&col32, (size_t)E * 4));
    for (int super_src : {262144, 524288, 1048576, 2097152}) {
      const int nsuper = (int)(((int64_t)C * CHUNK) / super_src);
      const int64_t per_super = E / nsuper;
      const int64_t n = per_super * nsuper;
      k_init_v2<<<4096, 256>>>(col32, n, per_super, super_src);
      OK(hipDeviceSynchronize());
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        OK(hipEventRecord(e0));
        k_pass1_v2<<<(unsigned)((n + 2047) / 2048), 256>>>(x, col32, S, n);
        OK(hipEventRecord(e1));
        const float ms = time_ms(e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("V2 (L2 super-chunks of %7d sources = %4.1f MB): %8.3f ms for %lld edges = %6.1f G edges/s\n", super_src, super_src * 4e-6, best, (long long)n, n / best * 1e-6);
    }
    OK(hipFree(col32));
  }
  {
    uint16_t* slot;
    OK(hipMalloc(&slot, (size_t)(E + (1 << 20)) * 2));
    auto run = [&](auto bin_c, auto block_c, int per_cu) {
      constexpr int BIN = decltype(bin_c)::value, BLOCK = decltype(block_c)::value;
      const int nbins = (int)(E / BIN);
      k_init_slot<<<4096, 256>>>(slot, (int64_t)nbins * BIN, BIN);
      OK(hipDeviceSynchronize());
      float best = 1e9f;
      for (int rep = 0; rep < 4; rep++) {
        OK(hipEventRecord(e0));
        k_read_bins<BIN, BLOCK><<<256 * per_cu, BLOCK>>>(S, slot, out, nbins);
        OK(hipEventRecord(e1));
        const float ms = time_ms(e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("RD bins of %5d entries, %4d threads, %d per CU: %8.3f ms for %lld entries = %6.1f G entries/s (%.0f GB/s of 6 B/entry)\n", BIN, BLOCK, per_cu, best,
             (long long)nbins * BIN, (double)nbins * BIN / best * 1e-6, (double)nbins * BIN * 6.0 / best * 1e-6);
    };
    run(std::integral_constant<int, 4099>(), std::integral_constant<int, 256>(), 8);
    run(std::integral_constant<int, 8209>(), std::integral_constant<int, 256>(), 4);
    run(std::integral_constant<int, 16411>(), std::integral_constant<int, 1024>(), 1);
    run(std::integral_constant<int, 16411>(), std::integral_constant<int, 1024>(), 2);
    OK(hipFree(slot));
  }
  return 0;
}
