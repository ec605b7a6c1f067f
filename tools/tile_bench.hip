// tile_bench.hip -- prototype of the column-blocked ordered multiply.
//
// Question it answers: how fast is y = A (x) x on RMAT-26 when the columns are cut into B contiguous
// NATIVE ranges ("tiles"), each tile's slice of x is degree-ranked (hot entries first: L2 / LDS
// resident) and the tiles are multiplied one after the other, every row carrying its running value
// through y?  Because the tiles are native ranges, a row's ascending-native-column fold is exactly the
// concatenation of its per-tile segments, so the result is bit-identical to the unblocked ordered fold
// (checked below against a one-thread-per-row reference).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/tile_bench.hip -Iinclude \
//         -Igraphmat_amd/csrc -Lgraphmat_amd -lgraphmat_hip -Wl,-rpath,$PWD/graphmat_amd -o build/tile_bench
//   build/tile_bench <scale> <slice_entries_log2> [check]
#include <string.h>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "graphmat_hip.h"

#define CK(e)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (e);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);        \
      exit(1);                                                                             \
    }                                                                                      \
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

__global__ void k_degree(const int* src, const int* dst, int64_t nnz, int nparts, int nv, uint32_t* deg, uint32_t* indeg) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= nnz) return;
  int sn = to_native0(src[e], nparts, nv), dn = to_native0(dst[e], nparts, nv);
  atomicAdd(&deg[sn], 1u);
  atomicAdd(&deg[dn], 1u);
  atomicAdd(&indeg[dn], 1u);
}
__global__ void k_live(const uint32_t* deg, int nv, uint32_t* live) {
  int v = blockIdx.x * kT + threadIdx.x;
  if (v < nv) live[v] = deg[v] ? 1u : 0u;
}
// block of native vertex v = (live vertices before v) / slice;  dead vertices get block 255
__global__ void k_rank_keys(const uint32_t* deg, const uint32_t* live_prefix, int nv, int slice, uint32_t* keys, int* ids, uint8_t* blk) {
  int v = blockIdx.x * kT + threadIdx.x;
  if (v >= nv) return;
  keys[v] = 0xffffffffu - deg[v];
  ids[v] = v;
  blk[v] = deg[v] ? (uint8_t)(live_prefix[v] / (uint32_t)slice) : (uint8_t)255;
}
__global__ void k_gather_u8(const uint8_t* blk, const int* order, int nv, uint8_t* out) {
  int k = blockIdx.x * kT + threadIdx.x;
  if (k < nv) out[k] = blk[order[k]];
}
__global__ void k_assign(const int* order2, int nv, int* dev_of_native, int* native_of_dev) {
  int k = blockIdx.x * kT + threadIdx.x;
  if (k >= nv) return;
  dev_of_native[order2[k]] = k;
  native_of_dev[k] = order2[k];
}
// first device id of every block (block of device id k = blk[native_of_dev[k]])
__global__ void k_block_base(const uint8_t* blk, const int* native_of_dev, int nv, int* base /* [256], init nv */) {
  int k = blockIdx.x * kT + threadIdx.x;
  if (k >= nv) return;
  atomicMin(&base[blk[native_of_dev[k]]], k);
}
// keys: tile order  blk(col) << 52 | devrow << 26 | native col ; plain order devrow << 26 | native col
__global__ void k_keys(const int* src, const int* dst, int64_t nnz, int nparts, int nv, const int* dev_of_native, const uint8_t* blk,
                       const uint32_t* indeg, uint32_t giant, uint64_t* ktile, uint64_t* kplain, int* minblk) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= nnz) return;
  int sn = to_native0(src[e], nparts, nv), dn = to_native0(dst[e], nparts, nv);
  uint64_t r = (uint64_t)dev_of_native[dn];
  if (indeg[dn] > giant) {  // giant rows stay with the two-pass giant path: not part of the tiles
    ktile[e] = ~0ull;
    kplain[e] = ~0ull;
    return;
  }
  uint64_t b = blk[sn];
  ktile[e] = (b << 52) | (r << 26) | (uint64_t)sn;
  kplain[e] = (r << 26) | (uint64_t)sn;
  atomicMin(&minblk[r], (int)b);
}
__global__ void k_seg_flags(const uint64_t* ktile, int64_t n, uint8_t* flag) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= n) return;
  flag[e] = (e == 0 || (ktile[e] >> 26) != (ktile[e - 1] >> 26)) ? 1 : 0;
}
__global__ void k_seg_fill(const uint64_t* ktile, const uint32_t* seg_ptr, int nseg, const int* minblk, int* seg_row, uint8_t* seg_blk) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nseg) return;
  uint64_t k = ktile[seg_ptr[i]];
  int row = (int)((k >> 26) & 0x3ffffffu), b = (int)(k >> 52);
  seg_row[i] = row | (minblk[row] == b ? (int)0x80000000 : 0);
  seg_blk[i] = (uint8_t)b;
}
__global__ void k_cols(const uint64_t* ktile, int64_t n, const int* dev_of_native, const int* base, int* col) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= n) return;
  uint64_t k = ktile[e];
  col[e] = dev_of_native[(int)(k & 0x3ffffffu)] - base[(int)(k >> 52)];
}
// chunk starts: tile change, every 256th segment, long segments alone, or the edge position crosses a multiple of chn
__global__ void k_chunk_flags(const uint32_t* seg_ptr, const uint8_t* seg_blk, int nseg, int chn, uint8_t* flag) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nseg) return;
  bool st = i == 0 || (i & 255) == 0 || seg_blk[i] != seg_blk[i - 1];
  uint32_t a = seg_ptr[i], b = seg_ptr[i + 1];
  if (b - a > (uint32_t)chn) st = true;
  if (!st) {
    uint32_t pa = seg_ptr[i - 1];
    st = (a - pa > (uint32_t)chn) || (a / (uint32_t)chn != pa / (uint32_t)chn);
  }
  flag[i] = st ? 1 : 0;
}
__global__ void k_first_chunk_of_tile(const int* chunk_seg, int nchunk, const uint8_t* seg_blk, int* tile_chunk /* [257] init nchunk */) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nchunk) return;
  atomicMin(&tile_chunk[seg_blk[chunk_seg[i]]], i);
}
__global__ void k_fill_x(float* x, int n) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (uint32_t)i * 2654435761u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  x[i] = (float)(h >> 8) * (1.0f / 16777216.0f) * (1.0f / (float)(1 + (h & 63)));
}

// reference: one thread per row folds its edges in (row, native col) order
__global__ void k_ref_rowptr(const uint64_t* kplain, int64_t n, int nrows, int64_t* rowptr) {
  int r = blockIdx.x * kT + threadIdx.x;
  if (r > nrows) return;
  uint64_t target = (uint64_t)r << 26;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (kplain[mid] < target) lo = mid + 1; else hi = mid;
  }
  rowptr[r] = lo;
}
__global__ void k_ref_fold(const uint64_t* kplain, const int64_t* rowptr, int nrows, const int* dev_of_native, const float* x, float* y) {
  int r = blockIdx.x * kT + threadIdx.x;
  if (r >= nrows) return;
  int64_t a = rowptr[r], b = rowptr[r + 1];
  if (a == b) return;
  float acc = x[dev_of_native[(int)(kplain[a] & 0x3ffffffu)]];
  for (int64_t k = a + 1; k < b; k++) acc += x[dev_of_native[(int)(kplain[k] & 0x3ffffffu)]];
  y[r] = acc;
}

// ---- the tile kernel ---------------------------------------------------------------------------
// A workgroup walks chunks of one tile.  Chunk = consecutive segments (row pieces) holding < 2*CHN
// edges and at most 256 segments, or one long segment alone.  Phase 1: coalesced column ids, parallel
// gathers (LDS hot set or L2-resident slice), products to LDS in edge order.  Phase 2: one lane per
// segment folds in stored order starting from the row's running value in y.
template <int HOT, int CHN>
__global__ void __launch_bounds__(256)
k_tile(const int* __restrict__ col, const uint32_t* __restrict__ seg_ptr, const int* __restrict__ seg_row,
       const int* __restrict__ chunk_seg, int chunk0, int nchunk, const float* __restrict__ xs, int slice_len,
       float* __restrict__ y) {
  constexpr int CAP = 2 * CHN;
  constexpr int PER = CAP / 256;
  constexpr int PAD = CAP + CAP / 32;
  __shared__ float s_hot[HOT > 0 ? HOT : 1];
  __shared__ float s_msg[PAD];
#define SLOT(k) ((k) + ((k) >> 5))
  const int tid = threadIdx.x;
  const int nhot = HOT < slice_len ? HOT : slice_len;
  for (int i = tid; i < nhot; i += 256) s_hot[i] = xs[i];
  __syncthreads();
  for (int ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    const int s0 = chunk_seg[chunk0 + ch], s1 = chunk_seg[chunk0 + ch + 1];
    const uint32_t e0 = seg_ptr[s0], e1 = seg_ptr[s1];
    const int n = (int)(e1 - e0);
    if (n <= CAP) {
      int c[PER];
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const int k = tid + j * 256;
        c[j] = k < n ? __builtin_nontemporal_load(&col[e0 + k]) : -1;
      }
      float m[PER];
#pragma unroll
      for (int j = 0; j < PER; j++) {
        m[j] = 0.f;
        if (c[j] >= 0) m[j] = (HOT > 0 && c[j] < nhot) ? s_hot[c[j]] : xs[c[j]];
      }
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const int k = tid + j * 256;
        if (k < n) s_msg[SLOT(k)] = m[j];
      }
      __syncthreads();
      const int sg = s0 + tid;
      if (sg < s1) {
        const int rw = seg_row[sg];
        const int row = rw & 0x7fffffff;
        int kb = (int)(seg_ptr[sg] - e0);
        const int ke = (int)(seg_ptr[sg + 1] - e0);
        float acc;
        if (rw < 0) { acc = s_msg[SLOT(kb)]; kb++; } else acc = y[row];
        for (; kb + 4 <= ke; kb += 4) {
          float t0 = s_msg[SLOT(kb)], t1 = s_msg[SLOT(kb + 1)], t2 = s_msg[SLOT(kb + 2)], t3 = s_msg[SLOT(kb + 3)];
          acc += t0; acc += t1; acc += t2; acc += t3;
        }
        for (; kb < ke; kb++) acc += s_msg[SLOT(kb)];
        y[row] = acc;
      }
      __syncthreads();
    } else {
      // one long segment: sub-chunks of CAP edges, wave 0 folds each out of LDS with lane broadcasts
      const int rw = seg_row[s0];
      const int row = rw & 0x7fffffff;
      bool has = rw >= 0;
      float acc = 0.f;
      if (has && tid < 64) acc = y[row];
      for (uint32_t b0 = e0; b0 < e1; b0 += CAP) {
        const int nn = (int)((e1 - b0) < (uint32_t)CAP ? (e1 - b0) : (uint32_t)CAP);
#pragma unroll
        for (int j = 0; j < PER; j++) {
          const int k = tid + j * 256;
          if (k < nn) {
            const int cc = __builtin_nontemporal_load(&col[b0 + k]);
            s_msg[k] = (HOT > 0 && cc < nhot) ? s_hot[cc] : xs[cc];
          }
        }
        __syncthreads();
        if (tid < 64) {
          for (int kb = 0; kb < nn; kb += 64) {
            const int k = kb + tid;
            const float t = k < nn ? s_msg[k] : 0.f;
            const int cnt = (nn - kb) < 64 ? (nn - kb) : 64;
            int i = 0;
            if (!has) { acc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 0)); has = true; i = 1; }
            for (; i < cnt; i++) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), i));
          }
        }
        __syncthreads();
      }
      if (tid == 0) y[row] = acc;
    }
  }
#undef SLOT
}

struct Tiles {
  int B;
  std::vector<int> base, chunk0;  // per tile: first device id of its slice, first chunk
  int nchunk, nseg;
  int64_t ne;
  int *col, *seg_row, *chunk_seg;
  uint32_t* seg_ptr;
};

template <int HOT, int CHN>
float run_passes(const Tiles& t, const float* x, float* y, int wgs_per_cu, int reps, std::vector<float>* per_tile) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e9f;
  std::vector<hipEvent_t> ev(t.B + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int rep = 0; rep < reps; rep++) {
    CK(hipEventRecord(a));
    for (int bb = 0; bb < t.B; bb++) {
      CK(hipEventRecord(ev[bb]));
      const int nch = t.chunk0[bb + 1] - t.chunk0[bb];
      if (nch <= 0) continue;
      const int grid = std::min(nch, 256 * wgs_per_cu);
      const int slice_len = t.base[bb + 1] - t.base[bb];
      hipLaunchKernelGGL((k_tile<HOT, CHN>), dim3(grid), dim3(256), 0, 0, t.col, t.seg_ptr, t.seg_row, t.chunk_seg, t.chunk0[bb], nch,
                         x + t.base[bb], slice_len, y);
    }
    CK(hipEventRecord(ev[t.B]));
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) {
      best = ms;
      if (per_tile) {
        per_tile->resize(t.B);
        for (int bb = 0; bb < t.B; bb++) CK(hipEventElapsedTime(&(*per_tile)[bb], ev[bb], ev[bb + 1]));
      }
    }
  }
  return best;
}

int main(int argc, char** argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 22;
  const int slice_lg = argc > 2 ? atoi(argv[2]) : 21;
  const bool check = argc > 3 ? atoi(argv[3]) != 0 : true;
  const int nparts = 16;
  const int nv = 1 << scale;
  const int64_t nnz = (int64_t)16 << scale;
  const uint32_t giant = 32768;
  if (scale > 26) { printf("prototype keys hold 26-bit ids\n"); return 1; }
  printf("RMAT-%d V=%d E=%lld slice=2^%d entries\n", scale, nv, (long long)nnz, slice_lg);

  int *src = dalloc<int>(nnz), *dst = dalloc<int>(nnz);
  if (gm_rmat_generate(scale, 1, 0, nnz, src, dst, nullptr, 0, nullptr) != 0) { printf("rmat: %s\n", gm_last_error()); return 1; }
  uint32_t *deg = dalloc<uint32_t>(nv), *indeg = dalloc<uint32_t>(nv), *live = dalloc<uint32_t>(nv), *lpre = dalloc<uint32_t>(nv);
  CK(hipMemset(deg, 0, (size_t)nv * 4));
  CK(hipMemset(indeg, 0, (size_t)nv * 4));
  hipLaunchKernelGGL(k_degree, dim3(gridf(nnz)), dim3(kT), 0, 0, src, dst, nnz, nparts, nv, deg, indeg);
  hipLaunchKernelGGL(k_live, dim3(gridf(nv)), dim3(kT), 0, 0, deg, nv, live);
  size_t tb = 0;
  void* tmp = nullptr;
  CK(rocprim::exclusive_scan(nullptr, tb, live, lpre, 0u, (size_t)nv, rocprim::plus<uint32_t>()));
  CK(hipMalloc(&tmp, tb));
  CK(rocprim::exclusive_scan(tmp, tb, live, lpre, 0u, (size_t)nv, rocprim::plus<uint32_t>()));
  CK(hipFree(tmp));
  uint32_t last_pre = 0, last_live = 0;
  CK(hipMemcpy(&last_pre, lpre + nv - 1, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&last_live, live + nv - 1, 4, hipMemcpyDeviceToHost));
  const int nlive = (int)(last_pre + last_live);
  const int slice = 1 << slice_lg;
  const int B = (nlive + slice - 1) / slice;
  printf("live vertices %d -> %d tiles\n", nlive, B);
  if (B > 250) { printf("too many tiles\n"); return 1; }

  // device order: (block, degree descending, native id)
  uint32_t *k1 = dalloc<uint32_t>(nv), *k1o = dalloc<uint32_t>(nv);
  int *ids = dalloc<int>(nv), *order = dalloc<int>(nv), *order2 = dalloc<int>(nv);
  uint8_t *blk = dalloc<uint8_t>(nv), *bk = dalloc<uint8_t>(nv), *bko = dalloc<uint8_t>(nv);
  hipLaunchKernelGGL(k_rank_keys, dim3(gridf(nv)), dim3(kT), 0, 0, deg, lpre, nv, slice, k1, ids, blk);
  tb = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tb, k1, k1o, ids, order, (size_t)nv, 0u, 32u));
  CK(hipMalloc(&tmp, tb));
  CK(rocprim::radix_sort_pairs(tmp, tb, k1, k1o, ids, order, (size_t)nv, 0u, 32u));
  CK(hipFree(tmp));
  hipLaunchKernelGGL(k_gather_u8, dim3(gridf(nv)), dim3(kT), 0, 0, blk, order, nv, bk);
  tb = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tb, bk, bko, order, order2, (size_t)nv, 0u, 8u));
  CK(hipMalloc(&tmp, tb));
  CK(rocprim::radix_sort_pairs(tmp, tb, bk, bko, order, order2, (size_t)nv, 0u, 8u));
  CK(hipFree(tmp));
  int *don = dalloc<int>(nv), *nod = dalloc<int>(nv), *d_base = dalloc<int>(257);
  hipLaunchKernelGGL(k_assign, dim3(gridf(nv)), dim3(kT), 0, 0, order2, nv, don, nod);
  std::vector<int> h_base(257, nv);
  CK(hipMemcpy(d_base, h_base.data(), 257 * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_block_base, dim3(gridf(nv)), dim3(kT), 0, 0, blk, nod, nv, d_base);
  CK(hipMemcpy(h_base.data(), d_base, 257 * 4, hipMemcpyDeviceToHost));
  Tiles t;
  t.B = B;
  t.base.assign(h_base.begin(), h_base.begin() + B);
  t.base.push_back(nlive);
  for (int b = 0; b < B; b++) if (t.base[b] > t.base[b + 1]) { printf("bad tile base %d\n", b); return 1; }

  // edge keys, both orders
  uint64_t *kt = dalloc<uint64_t>(nnz), *kto = dalloc<uint64_t>(nnz), *kp = dalloc<uint64_t>(nnz);
  int* minblk = dalloc<int>(nv);
  CK(hipMemset(minblk, 0x7f, (size_t)nv * 4));
  hipLaunchKernelGGL(k_keys, dim3(gridf(nnz)), dim3(kT), 0, 0, src, dst, nnz, nparts, nv, don, blk, indeg, giant, kt, kp, minblk);
  CK(hipDeviceSynchronize());
  CK(hipFree(src));
  CK(hipFree(dst));
  tb = 0;
  CK(rocprim::radix_sort_keys(nullptr, tb, kt, kto, (size_t)nnz, 0u, 60u));
  CK(hipMalloc(&tmp, tb));
  CK(rocprim::radix_sort_keys(tmp, tb, kt, kto, (size_t)nnz, 0u, 60u));
  CK(hipDeviceSynchronize());
  // number of tile edges = first key with all ones
  int64_t ne;
  {
    std::vector<uint64_t> probe(1);
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
      int64_t mid = (lo + hi) / 2;
      CK(hipMemcpy(probe.data(), kto + mid, 8, hipMemcpyDeviceToHost));
      if (probe[0] == ~0ull) hi = mid; else lo = mid + 1;
    }
    ne = lo;
  }
  t.ne = ne;
  printf("tile edges %lld (giant rows keep %lld)\n", (long long)ne, (long long)(nnz - ne));
  // segments
  uint8_t* flag = dalloc<uint8_t>(ne);
  hipLaunchKernelGGL(k_seg_flags, dim3(gridf(ne)), dim3(kT), 0, 0, kto, ne, flag);
  uint32_t* seg_ptr = dalloc<uint32_t>(ne + 2);
  unsigned int* d_cnt = dalloc<unsigned int>(4);
  rocprim::counting_iterator<uint32_t> cit(0);
  size_t tb2 = 0;
  CK(rocprim::select(nullptr, tb2, cit, flag, seg_ptr, d_cnt, (size_t)ne));
  void* tmp2;
  CK(hipMalloc(&tmp2, tb2));
  CK(rocprim::select(tmp2, tb2, cit, flag, seg_ptr, d_cnt, (size_t)ne));
  unsigned int nseg = 0;
  CK(hipMemcpy(&nseg, d_cnt, 4, hipMemcpyDeviceToHost));
  uint32_t ne32 = (uint32_t)ne;
  CK(hipMemcpy(seg_ptr + nseg, &ne32, 4, hipMemcpyHostToDevice));
  t.nseg = (int)nseg;
  printf("segments %u (%.2f edges each)\n", nseg, (double)ne / nseg);
  int* seg_row = dalloc<int>(nseg);
  uint8_t* seg_blk = dalloc<uint8_t>(nseg);
  hipLaunchKernelGGL(k_seg_fill, dim3(gridf(nseg)), dim3(kT), 0, 0, kto, seg_ptr, (int)nseg, minblk, seg_row, seg_blk);
  int* col = dalloc<int>(ne);
  hipLaunchKernelGGL(k_cols, dim3(gridf(ne)), dim3(kT), 0, 0, kto, ne, don, d_base, col);
  CK(hipDeviceSynchronize());
  CK(hipFree(kt));
  CK(hipFree(kto));
  CK(hipFree(tmp));
  CK(hipFree(flag));
  t.col = col;
  t.seg_ptr = seg_ptr;
  t.seg_row = seg_row;

  float *x = dalloc<float>(nv), *y = dalloc<float>(nv), *yref = dalloc<float>(nv);
  hipLaunchKernelGGL(k_fill_x, dim3(gridf(nv)), dim3(kT), 0, 0, x, nv);
  CK(hipMemset(y, 0, (size_t)nv * 4));
  CK(hipMemset(yref, 0, (size_t)nv * 4));

  if (check) {
    uint64_t* kpo = dalloc<uint64_t>(nnz);
    tb = 0;
    CK(rocprim::radix_sort_keys(nullptr, tb, kp, kpo, (size_t)nnz, 0u, 64u));
    CK(hipMalloc(&tmp, tb));
    CK(rocprim::radix_sort_keys(tmp, tb, kp, kpo, (size_t)nnz, 0u, 64u));
    int64_t* rowptr = dalloc<int64_t>(nv + 1);
    hipLaunchKernelGGL(k_ref_rowptr, dim3(gridf(nv + 1)), dim3(kT), 0, 0, kpo, ne, nv, rowptr);
    hipLaunchKernelGGL(k_ref_fold, dim3(gridf(nv)), dim3(kT), 0, 0, kpo, rowptr, nv, don, x, yref);
    CK(hipDeviceSynchronize());
    CK(hipFree(kpo));
    CK(hipFree(tmp));
    CK(hipFree(rowptr));
  }
  CK(hipFree(kp));

  uint8_t* cflag = dalloc<uint8_t>(nseg);
  int* chunk_seg = dalloc<int>(nseg + 2);
  int* d_tc = dalloc<int>(257);
  std::vector<float> h_y(nv), h_ref(nv);
  if (check) CK(hipMemcpy(h_ref.data(), yref, (size_t)nv * 4, hipMemcpyDeviceToHost));

  auto build_chunks = [&](int chn) {
    hipLaunchKernelGGL(k_chunk_flags, dim3(gridf(nseg)), dim3(kT), 0, 0, seg_ptr, seg_blk, (int)nseg, chn, cflag);
    rocprim::counting_iterator<int> cit2(0);
    size_t tb3 = 0;
    CK(rocprim::select(nullptr, tb3, cit2, cflag, chunk_seg, d_cnt, (size_t)nseg));
    void* tmp3;
    CK(hipMalloc(&tmp3, tb3));
    CK(rocprim::select(tmp3, tb3, cit2, cflag, chunk_seg, d_cnt, (size_t)nseg));
    unsigned int nchunk = 0;
    CK(hipMemcpy(&nchunk, d_cnt, 4, hipMemcpyDeviceToHost));
    int ns = (int)nseg;
    CK(hipMemcpy(chunk_seg + nchunk, &ns, 4, hipMemcpyHostToDevice));
    std::vector<int> tc(257, (int)nchunk);
    CK(hipMemcpy(d_tc, tc.data(), 257 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_first_chunk_of_tile, dim3(gridf(nchunk)), dim3(kT), 0, 0, chunk_seg, (int)nchunk, seg_blk, d_tc);
    CK(hipMemcpy(tc.data(), d_tc, 257 * 4, hipMemcpyDeviceToHost));
    CK(hipFree(tmp3));
    t.nchunk = (int)nchunk;
    t.chunk_seg = chunk_seg;
    t.chunk0.assign(B + 1, (int)nchunk);
    for (int b = B - 1; b >= 0; b--) t.chunk0[b] = std::min(tc[b], t.chunk0[b + 1]);
    t.chunk0[0] = 0;
    printf("  chn=%d: %u chunks (%.0f edges, %.1f segments each)\n", chn, nchunk, (double)ne / nchunk, (double)nseg / nchunk);
  };

  auto report = [&](const char* name, float ms, const std::vector<float>& pt) {
    printf("%-28s %7.3f ms  %6.1f G edges/s   per tile:", name, ms, ne / ms / 1e6);
    for (size_t i = 0; i < pt.size() && i < 40; i++) printf(" %.3f", pt[i]);
    printf("\n");
    if (check) {
      CK(hipMemcpy(h_y.data(), y, (size_t)nv * 4, hipMemcpyDeviceToHost));
      int64_t bad = 0;
      for (int i = 0; i < nv; i++)
        if (memcmp(&h_y[i], &h_ref[i], 4) != 0) { if (bad < 5) printf("    MISMATCH row %d: %.9g vs %.9g\n", i, h_y[i], h_ref[i]); bad++; }
      printf("    %s (%lld mismatching rows)\n", bad ? "WRONG" : "bit-exact vs per-row ordered fold", (long long)bad);
    }
    fflush(stdout);
  };
  std::vector<float> pt;
#define RUN(H, C, W)                                                     \
  {                                                                      \
    CK(hipMemset(y, 0, (size_t)nv * 4));                                 \
    float ms = run_passes<H, C>(t, x, y, W, 5, &pt);                     \
    char nm[64];                                                         \
    snprintf(nm, 64, "hot=%d chn=%d wg/cu=%d", H, C, W);                 \
    report(nm, ms, pt);                                                  \
  }
  build_chunks(1024);
  RUN(0, 1024, 8)
  RUN(8192, 1024, 4)
  RUN(8192, 1024, 3)
  RUN(16384, 1024, 2)
  RUN(32768, 1024, 1)
  build_chunks(2048);
  RUN(0, 2048, 4)
  RUN(8192, 2048, 3)
  RUN(16384, 2048, 2)
  return 0;
}
