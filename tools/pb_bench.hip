// pb_bench.hip -- prototype: propagation blocking for the PageRank multiply.
//
// Question: how fast is y = A (x) x on RMAT-<scale> when no edge costs an L1-missing gather?
//   pass 1  one workgroup per chunk of CH consecutive NATIVE source ids: the chunk's x values sit in LDS; every edge of
//           the chunk writes its product to a precomputed slot of its destination bin's stream -- the chunk's edges are
//           stored grouped by bin, so the writes are runs of consecutive floats;
//   pass 2  one workgroup per bin of BIN consecutive NATIVE destination ids: accumulators in LDS, the bin's stream is
//           read once, sequentially, and added with LDS float atomics.
// Per edge: 2 B (source index inside the chunk) + 4 B (slot) + 4 B (product written) + 4 B + 2 B (product and
// destination index read back) = 16 B of streaming instead of one 64-byte random fetch at the L2's request rate.
// The sums are NOT in the reference's order (LDS atomics): this variant answers to the 1e-6 relative tolerance of the
// north star, not to bit-exactness.  Rows above GIANT in-edges are left out (they would keep the giant-row replay).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pb_bench.hip -Iinclude -Lgraphmat_amd -lgraphmat_hip \
//         -Wl,-rpath,$PWD/graphmat_amd -o build/pb_bench
//   build/pb_bench <scale> [chunk_log2=15] [bin_log2=13] [giant=32768]
#include <string.h>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "graphmat_hip.h"

#define CK(e)                                                                       \
  do {                                                                              \
    hipError_t e_ = (e);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

constexpr int kT = 256;
static inline int gridf(int64_t n) { return (int)((n + kT - 1) / kT); }

__host__ __device__ inline int to_native0(int vertex1, int nparts, int len) {
  int v = vertex1 - 1;
  int height = len / nparts;
  int vmax = height * nparts;
  if (v >= vmax) return v;
  return (v / nparts) + (v % nparts) * height;
}

template <class T>
T* dalloc(size_t n) {
  T* p = nullptr;
  CK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
  return p;
}

__global__ void k_native(int* src, int* dst, int64_t nnz, int nparts, int nv, uint32_t* indeg) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= nnz) return;
  const int s = to_native0(src[e], nparts, nv), d = to_native0(dst[e], nparts, nv);
  src[e] = s;
  dst[e] = d;
  atomicAdd(&indeg[d], 1u);
}
// Bins: coarse bins of 2^bin_lg consecutive destinations, each cut into sub-bins of about `limit` entries by the running
// count of kept in-edges (a destination's entries all go to the sub-bin its first entry falls into), so that no
// workgroup of pass 2 gets a hub region's millions of entries.
__global__ void k_kept_deg(const uint32_t* indeg, int nv, uint32_t giant, uint32_t* kd) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i < nv) kd[i] = indeg[i] <= giant ? indeg[i] : 0u;
}
__device__ __forceinline__ uint32_t bin_of(int d, const uint32_t* pre, const uint32_t* subbase, const uint32_t* nsub, int bin_lg, uint32_t limit) {
  const int cb = d >> bin_lg;
  uint32_t j = (pre[d] - pre[cb << bin_lg]) / limit;
  if (j >= nsub[cb]) j = nsub[cb] - 1;
  return subbase[cb] + j;
}
__global__ void k_bin_ranges(int nv, const uint32_t* pre, const uint32_t* subbase, const uint32_t* nsub, int bin_lg, uint32_t limit, int* lo, int* hi,
                             int* coarse) {
  int d = blockIdx.x * kT + threadIdx.x;
  if (d >= nv) return;
  const uint32_t b = bin_of(d, pre, subbase, nsub, bin_lg, limit);
  const int mask = (1 << bin_lg) - 1;
  const bool first = (d & mask) == 0 || bin_of(d - 1, pre, subbase, nsub, bin_lg, limit) != b;
  const bool last = (d & mask) == mask || bin_of(d + 1, pre, subbase, nsub, bin_lg, limit) != b;
  if (first) { lo[b] = d & mask; coarse[b] = d >> bin_lg; }
  if (last) hi[b] = (d & mask) + 1;
}
// keys of the two orders; edges into giant rows get the last key (sorted to the end, dropped)
__global__ void k_keys(const int* src, const int* dst, int64_t nnz, const uint32_t* indeg, uint32_t giant, int chunk_lg, int bin_lg,
                       int binbits, int nchunks_lg, const uint32_t* pre, const uint32_t* subbase, const uint32_t* nsub, uint32_t limit,
                       uint32_t* key_q, uint32_t* key_r, uint32_t* id, unsigned long long* kept) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  bool keep = false;
  if (e < nnz) {
    const int s = src[e], d = dst[e];
    keep = indeg[d] <= giant;
    const uint32_t c = (uint32_t)s >> chunk_lg, b = bin_of(d, pre, subbase, nsub, bin_lg, limit);
    key_q[e] = keep ? ((b << nchunks_lg) | c) : 0xffffffffu;
    key_r[e] = keep ? ((c << binbits) | b) : 0xffffffffu;
    id[e] = (uint32_t)e;
  }
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(kept, (unsigned long long)__popcll(m));
}
__global__ void k_invert(const uint32_t* eq, int64_t n, uint32_t* qpos) {
  int64_t q = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (q < n) qpos[eq[q]] = (uint32_t)q;
}
__global__ void k_pass1_arrays(const uint32_t* er, int64_t n, const int* src, const uint32_t* qpos, int chunk_mask, uint16_t* srcl, uint32_t* slot) {
  int64_t r = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (r >= n) return;
  const uint32_t e = er[r];
  srcl[r] = (uint16_t)(src[e] & chunk_mask);
  slot[r] = qpos[e];
}
__global__ void k_pass2_arrays(const uint32_t* eq, int64_t n, const int* dst, int bin_mask, uint16_t* dstl) {
  int64_t q = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (q < n) dstl[q] = (uint16_t)(dst[eq[q]] & bin_mask);
}
// bounds[k] = first position whose (key >> shift) >= k
__global__ void k_bounds(const uint32_t* keys, int64_t n, int shift, int nk, int64_t* bounds) {
  const int k = blockIdx.x * kT + threadIdx.x;
  if (k > nk) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((keys[mid] >> shift) < (uint32_t)k) lo = mid + 1; else hi = mid;
  }
  bounds[k] = lo;
}
__global__ void k_fill_x(float* x, int n) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) x[i] = 0.25f + (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f);
}
__global__ void k_ref(const int* src, const int* dst, int64_t nnz, const uint32_t* indeg, uint32_t giant, const float* x, float* y) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e < nnz && indeg[dst[e]] <= giant) atomicAdd(&y[dst[e]], x[src[e]]);
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------
constexpr int kP1 = 1024;
struct Piece { int chunk; int pad; int64_t r0, r1; };
template <int CHUNK_LG>
__global__ void __launch_bounds__(kP1)
k_pass1(const float* __restrict__ x, const Piece* __restrict__ pieces, const uint16_t* __restrict__ srcl, const uint32_t* __restrict__ slot,
        float* __restrict__ vals) {
  extern __shared__ float s_x[];
  const Piece pc = pieces[blockIdx.x];
  const int64_t r0 = pc.r0, r1 = pc.r1;
  const float* xc = x + ((size_t)pc.chunk << CHUNK_LG);
  for (int i = threadIdx.x; i < (1 << CHUNK_LG); i += kP1) s_x[i] = xc[i];
  __syncthreads();
  constexpr int U = 4;
  int64_t r = r0 + threadIdx.x;
  for (; r + (int64_t)(U - 1) * kP1 < r1; r += (int64_t)U * kP1) {
    uint16_t sl[U];
    uint32_t q[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      sl[u] = __builtin_nontemporal_load(srcl + r + (int64_t)u * kP1);
      q[u] = __builtin_nontemporal_load(slot + r + (int64_t)u * kP1);
    }
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(s_x[sl[u]], vals + q[u]);
  }
  for (; r < r1; r += kP1) vals[slot[r]] = s_x[srcl[r]];
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------
constexpr int kP2 = 256;
template <int BIN_LG>
__global__ void __launch_bounds__(kP2)
k_pass2(const int64_t* __restrict__ qstart, const int* __restrict__ lo, const int* __restrict__ hi, const int* __restrict__ coarse,
        const uint16_t* __restrict__ dstl, const float* __restrict__ vals, float* __restrict__ y) {
  __shared__ float s_acc[1 << BIN_LG];
  const int b = blockIdx.x;
  const int d0 = lo[b], d1 = hi[b];
  for (int i = d0 + threadIdx.x; i < d1; i += kP2) s_acc[i] = 0.f;
  __syncthreads();
  const int64_t q0 = qstart[b], q1 = qstart[b + 1];
  constexpr int U = 8;
  int64_t q = q0 + threadIdx.x;
  for (; q + (int64_t)(U - 1) * kP2 < q1; q += (int64_t)U * kP2) {
    uint16_t d[U];
    float v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      d[u] = __builtin_nontemporal_load(dstl + q + (int64_t)u * kP2);
      v[u] = __builtin_nontemporal_load(vals + q + (int64_t)u * kP2);
    }
#pragma unroll
    for (int u = 0; u < U; u++) atomicAdd(&s_acc[d[u]], v[u]);
  }
  for (; q < q1; q += kP2) atomicAdd(&s_acc[dstl[q]], vals[q]);
  __syncthreads();
  float* yb = y + ((size_t)coarse[b] << BIN_LG);
  for (int i = d0 + threadIdx.x; i < d1; i += kP2) yb[i] = s_acc[i];
}

template <class K>
static void sort_pairs(K* kin, K* kout, uint32_t* vin, uint32_t* vout, int64_t n, unsigned bits) {
  size_t tb = 0;
  void* tmp = nullptr;
  CK(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, (size_t)n, 0u, bits));
  CK(hipMalloc(&tmp, tb));
  CK(rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, (size_t)n, 0u, bits));
  CK(hipDeviceSynchronize());
  CK(hipFree(tmp));
}

template <int CHUNK_LG, int BIN_LG>
static void run(int scale, uint32_t giant) {
  const int nparts = 16;
  const int nv = 1 << scale;
  const int64_t nnz = (int64_t)16 << scale;
  const int nchunks_lg = scale - CHUNK_LG, nbins_lg = scale - BIN_LG;
  const int nchunks = 1 << nchunks_lg, nbins = 1 << nbins_lg;
  if (nchunks_lg + nbins_lg > 31) { printf("keys do not fit 32 bits\n"); return; }
  printf("RMAT-%d V=%d E=%lld: %d chunks of %d sources, %d bins of %d destinations, rows above %u in-edges left out\n", scale, nv,
         (long long)nnz, nchunks, 1 << CHUNK_LG, nbins, 1 << BIN_LG, giant);
  int *src = dalloc<int>(nnz), *dst = dalloc<int>(nnz);
  if (gm_rmat_generate(scale, 1, 0, nnz, src, dst, nullptr, 0, nullptr) != 0) { printf("rmat: %s\n", gm_last_error()); exit(1); }
  uint32_t* indeg = dalloc<uint32_t>(nv);
  CK(hipMemset(indeg, 0, (size_t)nv * 4));
  hipLaunchKernelGGL(k_native, dim3(gridf(nnz)), dim3(kT), 0, 0, src, dst, nnz, nparts, nv, indeg);
  // variable bins
  const uint32_t limit = 131072;  // entries per sub-bin (plus at most one row's)
  uint32_t *kd = dalloc<uint32_t>(nv + 1), *pre = dalloc<uint32_t>(nv + 1);
  CK(hipMemset(kd, 0, (size_t)(nv + 1) * 4));
  hipLaunchKernelGGL(k_kept_deg, dim3(gridf(nv)), dim3(kT), 0, 0, indeg, nv, giant, kd);
  {
    size_t tb = 0;
    void* tmp = nullptr;
    CK(rocprim::exclusive_scan(nullptr, tb, kd, pre, 0u, (size_t)nv + 1, rocprim::plus<uint32_t>()));
    CK(hipMalloc(&tmp, tb));
    CK(rocprim::exclusive_scan(tmp, tb, kd, pre, 0u, (size_t)nv + 1, rocprim::plus<uint32_t>()));
    CK(hipDeviceSynchronize());
    CK(hipFree(tmp));
  }
  std::vector<uint32_t> hpre((size_t)nbins + 1), h_nsub(nbins), h_subbase(nbins);
  for (int cb = 0; cb <= nbins; cb++) CK(hipMemcpy(&hpre[cb], pre + ((size_t)cb << BIN_LG), 4, hipMemcpyDeviceToHost));
  uint32_t NB = 0;
  for (int cb = 0; cb < nbins; cb++) {
    const uint32_t ent = hpre[cb + 1] - hpre[cb];
    h_nsub[cb] = std::max(1u, (ent + limit - 1) / limit);
    h_subbase[cb] = NB;
    NB += h_nsub[cb];
  }
  int binbits = 1;
  while ((1u << binbits) < NB) binbits++;
  printf("%u bins after cutting the coarse ones at %u entries\n", NB, limit);
  if (binbits + nchunks_lg > 31) { printf("keys do not fit\n"); return; }
  uint32_t *d_nsub = dalloc<uint32_t>(nbins), *d_subbase = dalloc<uint32_t>(nbins);
  CK(hipMemcpy(d_nsub, h_nsub.data(), (size_t)nbins * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_subbase, h_subbase.data(), (size_t)nbins * 4, hipMemcpyHostToDevice));
  int *blo = dalloc<int>(NB), *bhi = dalloc<int>(NB), *bcoarse = dalloc<int>(NB);
  CK(hipMemset(blo, 0, (size_t)NB * 4)); CK(hipMemset(bhi, 0, (size_t)NB * 4)); CK(hipMemset(bcoarse, 0, (size_t)NB * 4));
  hipLaunchKernelGGL(k_bin_ranges, dim3(gridf(nv)), dim3(kT), 0, 0, nv, pre, d_subbase, d_nsub, BIN_LG, limit, blo, bhi, bcoarse);
  uint32_t *kq = dalloc<uint32_t>(nnz), *kr = dalloc<uint32_t>(nnz), *ko = dalloc<uint32_t>(nnz), *id = dalloc<uint32_t>(nnz),
           *eq = dalloc<uint32_t>(nnz), *er = dalloc<uint32_t>(nnz);
  unsigned long long* d_kept = dalloc<unsigned long long>(1);
  CK(hipMemset(d_kept, 0, 8));
  hipLaunchKernelGGL(k_keys, dim3(gridf(nnz)), dim3(kT), 0, 0, src, dst, nnz, indeg, giant, CHUNK_LG, BIN_LG, binbits, nchunks_lg, pre, d_subbase, d_nsub,
                     limit, kq, kr, id, d_kept);
  unsigned long long kept = 0;
  CK(hipMemcpy(&kept, d_kept, 8, hipMemcpyDeviceToHost));
  printf("edges on the blocked path: %llu (%.1f %%)\n", kept, 100.0 * (double)kept / (double)nnz);
  // q order (bin, chunk) and r order (chunk, bin); both stable from the input order
  sort_pairs(kq, ko, id, eq, nnz, (unsigned)(nchunks_lg + binbits + 1));
  int64_t *qstart = dalloc<int64_t>(NB + 1), *rstart = dalloc<int64_t>(nchunks + 1);
  hipLaunchKernelGGL(k_bounds, dim3(gridf(NB + 1)), dim3(kT), 0, 0, ko, (int64_t)kept, nchunks_lg, (int)NB, qstart);
  CK(hipDeviceSynchronize());
  sort_pairs(kr, ko, id, er, nnz, (unsigned)(nchunks_lg + binbits + 1));
  hipLaunchKernelGGL(k_bounds, dim3(gridf(nchunks + 1)), dim3(kT), 0, 0, ko, (int64_t)kept, binbits, nchunks, rstart);
  CK(hipDeviceSynchronize());
  CK(hipFree(kq)); CK(hipFree(kr)); CK(hipFree(ko)); CK(hipFree(id));
  // pieces of pass 1: a chunk's entries in slices of at most kPiece (a hub source's chunk has 100x the mean)
  const int64_t kPiece = 262144;
  std::vector<Piece> hp;
  {
    std::vector<int64_t> hq(NB + 1), hr(nchunks + 1);
    CK(hipMemcpy(hq.data(), qstart, (size_t)(NB + 1) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), rstart, (size_t)(nchunks + 1) * 8, hipMemcpyDeviceToHost));
    int64_t mq = 0, mr = 0;
    for (uint32_t b = 0; b < NB; b++) mq = std::max(mq, hq[b + 1] - hq[b]);
    for (int c = 0; c < nchunks; c++) {
      mr = std::max(mr, hr[c + 1] - hr[c]);
      for (int64_t a = hr[c]; a < hr[c + 1]; a += kPiece) hp.push_back(Piece{c, 0, a, std::min(a + kPiece, hr[c + 1])});
    }
    printf("largest chunk %lld entries (mean %.0f) -> %zu pieces; largest bin %lld entries (mean %.0f)\n", (long long)mr, (double)kept / nchunks,
           hp.size(), (long long)mq, (double)kept / NB);
  }
  Piece* d_pieces = dalloc<Piece>(hp.size());
  CK(hipMemcpy(d_pieces, hp.data(), hp.size() * sizeof(Piece), hipMemcpyHostToDevice));
  uint32_t* qpos = dalloc<uint32_t>(nnz);
  hipLaunchKernelGGL(k_invert, dim3(gridf((int64_t)kept)), dim3(kT), 0, 0, eq, (int64_t)kept, qpos);
  uint16_t *srcl = dalloc<uint16_t>(kept), *dstl = dalloc<uint16_t>(kept);
  uint32_t* slot = dalloc<uint32_t>(kept);
  hipLaunchKernelGGL(k_pass1_arrays, dim3(gridf((int64_t)kept)), dim3(kT), 0, 0, er, (int64_t)kept, src, qpos, (1 << CHUNK_LG) - 1, srcl, slot);
  hipLaunchKernelGGL(k_pass2_arrays, dim3(gridf((int64_t)kept)), dim3(kT), 0, 0, eq, (int64_t)kept, dst, (1 << BIN_LG) - 1, dstl);
  CK(hipDeviceSynchronize());
  CK(hipFree(qpos)); CK(hipFree(eq)); CK(hipFree(er));
  float *x = dalloc<float>(nv), *y = dalloc<float>(nv), *yref = dalloc<float>(nv), *vals = dalloc<float>(kept);
  hipLaunchKernelGGL(k_fill_x, dim3(gridf(nv)), dim3(kT), 0, 0, x, nv);
  CK(hipMemset(yref, 0, (size_t)nv * 4));
  CK(hipMemset(y, 0, (size_t)nv * 4));
  hipLaunchKernelGGL(k_ref, dim3(gridf(nnz)), dim3(kT), 0, 0, src, dst, nnz, indeg, giant, x, yref);
  CK(hipDeviceSynchronize());
  const size_t lds1 = (size_t)4 << CHUNK_LG;
  CK(hipFuncSetAttribute((const void*)k_pass1<CHUNK_LG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float best1 = 1e9f, best2 = 1e9f, sum = 0.f;
  const int reps = 8;
  for (int it = 0; it < reps + 2; it++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_pass1<CHUNK_LG>), dim3((unsigned)hp.size()), dim3(kP1), lds1, 0, x, d_pieces, srcl, slot, vals);
    CK(hipEventRecord(e1, 0));
    hipLaunchKernelGGL((k_pass2<BIN_LG>), dim3(NB), dim3(kP2), 0, 0, qstart, blo, bhi, bcoarse, dstl, vals, y);
    CK(hipEventRecord(e2, 0));
    CK(hipEventSynchronize(e2));
    CK(hipGetLastError());
    float t1 = 0, t2 = 0;
    CK(hipEventElapsedTime(&t1, e0, e1));
    CK(hipEventElapsedTime(&t2, e1, e2));
    if (it >= 2) { best1 = std::min(best1, t1); best2 = std::min(best2, t2); sum += t1 + t2; }
  }
  // check against the atomic reference (both are unordered fp32 sums: relative 1e-5 of the row sum)
  std::vector<float> hy(nv), hr(nv);
  CK(hipMemcpy(hy.data(), y, (size_t)nv * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hr.data(), yref, (size_t)nv * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  long long bad = 0;
  for (int i = 0; i < nv; i++) {
    const double d = fabs((double)hy[i] - (double)hr[i]), m = fabs((double)hr[i]);
    const double rel = m > 0 ? d / m : d;
    worst = std::max(worst, rel);
    if (rel > 1e-4) bad++;
  }
  const double bytes1 = (double)kept * 10.0, bytes2 = (double)kept * 6.0;
  printf("pass 1 %.3f ms (%.0f GB/s of 10 B/edge), pass 2 %.3f ms (%.0f GB/s of 6 B/edge); best sum %.3f ms, mean %.3f ms => %.1f G edges/s; "
         "worst relative difference to an atomic reference %.2e (%lld rows above 1e-4)\n",
         best1, bytes1 / best1 / 1e6, best2, bytes2 / best2 / 1e6, best1 + best2, sum / reps, (double)kept / (best1 + best2) / 1e6, worst, bad);
}

int main(int argc, char** argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 22;
  const int chunk_lg = argc > 2 ? atoi(argv[2]) : 15;
  const int bin_lg = argc > 3 ? atoi(argv[3]) : 13;
  const uint32_t giant = argc > 4 ? (uint32_t)atoi(argv[4]) : 32768u;
  if (chunk_lg == 15 && bin_lg == 13) run<15, 13>(scale, giant);
  else if (chunk_lg == 15 && bin_lg == 12) run<15, 12>(scale, giant);
  else if (chunk_lg == 15 && bin_lg == 14) run<15, 14>(scale, giant);
  else if (chunk_lg == 14 && bin_lg == 13) run<14, 13>(scale, giant);
  else if (chunk_lg == 14 && bin_lg == 12) run<14, 12>(scale, giant);
  else { printf("unsupported chunk/bin combination\n"); return 1; }
  return 0;
}
