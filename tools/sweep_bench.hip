// tools/sweep_bench.hip -- prototype of a ROW-STATIONARY, SLICE-SWEEPING multiply for the medium rows.
//
// Where the RMAT-26 PageRank iteration stands (profiles/r04_instruction_diet.md): it runs at the chip's rate for its
// mix of gathers -- 754 M L1->L2 requests, 27 % of them L2 misses, 130 G requests/s.  The rows of 65..4096 edges (2.5 M
// rows, 680 M of the 1074 M edges) go through 8 column tiles today (16 MB of x per tile against 4 MB of L2 per XCD: 96 M
// L2 misses) and more tiles cost more than they save: every tile pass is two launches, reloads the hot sets, and carries
// each row's running value through y.  This prototype measures the alternative:
//   * ONE persistent launch; workgroup w (one per CU) OWNS a fixed set of medium rows and keeps their running values in
//     LDS for the whole multiply (2.5 M rows / 256 CUs x 4 B = 40 KB);
//   * the native column range is cut into T slices that serve equally many gathers (T = 8 .. 64; 64 slices = 2 MB of x
//     each: L2-resident); all workgroups sweep the slices in the same order, so at any time the chip gathers from one
//     or two slices only; per slice a workgroup loads the slice's HOT busiest entries into LDS, then its waves take the
//     pieces (row, slice) of its rows 64 at a time, one lane per piece -- coalesced column ids, gathers, messages staged in
//     the wave's LDS strip, every lane folds its piece in stored order onto the row's running value in LDS;
//   * pieces are stored [workgroup][slice][row][native column]: a row's edges are folded in ascending native column
//     order exactly as today (slices are native ranges), so the sums are the same bits as a serial fold (checked below).
// Everything here is built by the tool itself from the library's RMAT generator; nothing of it is in the library.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude tools/sweep_bench.hip -Lgraphmat_amd -lgraphmat_hip -o build/sweep_bench
//   LD_LIBRARY_PATH=graphmat_amd build/sweep_bench [scale 26] [slices 32] [reps 5]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <vector>
#include "graphmat_hip.h"

#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); exit(1); } } while (0)

// (closing session of round 4: -DKWG=4096 -DKBLOCK=64 -DKACC=1024 makes every WAVE its own "workgroup" with private rows -- the per-slice
// barrier then costs nothing and a wave never waits for another; only sensible without the LDS hot set)
#ifndef KWG
#define KWG 256
#endif
#ifndef KBLOCK
#define KBLOCK 1024
#endif
#ifndef KACC
#define KACC 10240
#endif
constexpr int kWG = KWG;        // workgroups (256 = CUs)
constexpr int kBlock = KBLOCK;  // threads per workgroup
constexpr int kMaxT = 64;
constexpr int kRowLo = 65, kRowHi = 4096;

__global__ void k_deg_in(const int32_t* __restrict__ dst, int64_t ne, uint32_t* __restrict__ deg) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&deg[dst[i] - 1], 1u);
}
__global__ void k_flag_medium(const uint32_t* __restrict__ deg, int nv, unsigned char* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) flag[i] = (deg[i] >= (uint32_t)kRowLo && deg[i] <= (uint32_t)kRowHi) ? 1 : 0;
}
__global__ void k_iota(int32_t* __restrict__ a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ void k_gather_deg(const int32_t* __restrict__ rows, int n, const uint32_t* __restrict__ deg, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = deg[rows[i]];
}
__global__ void k_every(const int32_t* __restrict__ in, int n, int step, int first, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(size_t)i * step + first];
}
__global__ void k_rank_of(const int32_t* __restrict__ rows_sorted, int n, int32_t* __restrict__ rank_of) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rank_of[rows_sorted[i]] = i;
}
// weight of a column = how often the medium rows gather it
__global__ void k_col_weight(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t ne, const int32_t* __restrict__ rank_of, uint32_t* __restrict__ w) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x)
    if (rank_of[dst[i] - 1] >= 0) atomicAdd(&w[src[i] - 1], 1u);
}
__global__ void k_widen(const uint32_t* __restrict__ w, int n, unsigned long long* __restrict__ o) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = w[i];
}
// bound[k] = first column whose inclusive prefix weight reaches k * total / T   (bound[0] = 0, bound[T] = nv)
__global__ void k_bounds(const unsigned long long* __restrict__ pre, int nv, int T, int32_t* __restrict__ bound) {
  const int k = threadIdx.x;
  if (k > T) return;
  if (k == 0) { bound[0] = 0; return; }
  if (k == T) { bound[T] = nv; return; }
  const unsigned long long total = pre[nv - 1], want = total / (unsigned)T * (unsigned)k;
  int lo = 0, hi = nv;
  while (lo < hi) { const int mid = (lo + hi) / 2; if (pre[mid] >= want) hi = mid; else lo = mid + 1; }
  bound[k] = lo;
}
__device__ __forceinline__ int slice_of(const int32_t* __restrict__ bound, int T, int c) {
  int lo = 0, hi = T;  // largest s with bound[s] <= c
  while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (bound[mid] <= c) lo = mid; else hi = mid; }
  return lo;
}
__global__ void k_col_keys(const uint32_t* __restrict__ w, int nv, const int32_t* __restrict__ bound, int T, unsigned long long* __restrict__ key) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nv) key[c] = ((unsigned long long)slice_of(bound, T, c) << 32) | (unsigned long long)(0xffffffffu - w[c]);
}
__global__ void k_col_map(const int32_t* __restrict__ cols_sorted, const unsigned long long* __restrict__ keys_sorted, int nv, int32_t* __restrict__ dev_of, int32_t* __restrict__ slice_base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  dev_of[cols_sorted[i]] = i;
  const int s = (int)(keys_sorted[i] >> 32);
  if (i == 0 || (int)(keys_sorted[i - 1] >> 32) != s) slice_base[s] = i;
}
// key = wg(8) | slice(6) | local row(16) | native column(26..27) ; value = device column
__global__ void k_edge_keys(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t ne, const int32_t* __restrict__ rank_of,
                            const int32_t* __restrict__ bound, int T, const int32_t* __restrict__ dev_of, int cbits, unsigned long long* __restrict__ key,
                            int32_t* __restrict__ val, unsigned long long* __restrict__ key2) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = rank_of[dst[i] - 1], c = src[i] - 1;
    if (r < 0) { key[i] = ~0ull; key2[i] = ~0ull; val[i] = 0; continue; }
    const unsigned long long wg = (unsigned)r % kWG, local = (unsigned)r / kWG;
    key[i] = (((wg << 6 | (unsigned long long)slice_of(bound, T, c)) << 16 | local) << cbits) | (unsigned long long)c;
    key2[i] = ((unsigned long long)r << cbits) | (unsigned long long)c;  // the reference's order: row, then native column
    val[i] = dev_of[c];
  }
}
__global__ void k_heads(const unsigned long long* __restrict__ key, int64_t n, int cbits, uint32_t* __restrict__ head) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    head[i] = (i == 0 || (key[i] >> cbits) != (key[i - 1] >> cbits)) ? 1u : 0u;
}
__global__ void k_pieces(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ head, const uint32_t* __restrict__ pidx_incl, int64_t n, int cbits,
                         uint32_t* __restrict__ piece_start, uint16_t* __restrict__ piece_row, int32_t* __restrict__ blk_first, int T) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!head[i]) continue;
    const uint32_t p = pidx_incl[i] - 1;
    const unsigned long long k = key[i] >> cbits;
    piece_start[p] = (uint32_t)i;
    piece_row[p] = (uint16_t)(k & 0xffff);
    const int blk = (int)(k >> 16);  // wg << 6 | slice
    if (i == 0 || (int)((key[i - 1] >> cbits) >> 16) != blk) blk_first[(blk >> 6) * T + (blk & 63)] = (int32_t)p;
  }
}
__global__ void k_row_starts(const unsigned long long* __restrict__ key2, int64_t n, int cbits, uint32_t* __restrict__ row_start) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (i == 0 || (key2[i] >> cbits) != (key2[i - 1] >> cbits)) row_start[key2[i] >> cbits] = (uint32_t)i;
}
// reference: one lane per medium row, serial fold in ascending native column order
__global__ void k_reference(const uint32_t* __restrict__ row_start, int nmed, int64_t nedges, const int32_t* __restrict__ col, const float* __restrict__ x, float* __restrict__ y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nmed) return;
  const int64_t e0 = row_start[r], e1 = r + 1 < nmed ? row_start[r + 1] : nedges;
  float acc = x[col[e0]];
  for (int64_t k = e0 + 1; k < e1; k++) acc += x[col[k]];
  y[r] = acc;
}
__global__ void k_fill_x(float* __restrict__ x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = (float)(h >> 8) * (1.0f / 16777216.0f) + 1e-3f; }
}

// ---- the sweep --------------------------------------------------------------------------------------------------------
template <int HOT, int ACC>
__global__ void __launch_bounds__(kBlock)
k_sweep(const int32_t* __restrict__ col, const uint32_t* __restrict__ piece_start, const uint16_t* __restrict__ piece_row, const int32_t* __restrict__ blk_first,
        const int32_t* __restrict__ slice_base, const int32_t* __restrict__ slice_len, int T, const float* __restrict__ x, float* __restrict__ y_by_rank, int nmed) {
  constexpr int W = kBlock / 64;
  constexpr int CH = 512, PER = CH / 64;
  constexpr int kPadw = CH + CH / 32;
  __shared__ float s_hot[HOT];
  __shared__ float s_acc[ACC];
  __shared__ float s_msg[W][kPadw];
#define SLOT(k) ((k) + ((k) >> 5))
  const int wg = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* sm = s_msg[wv];
  for (int i = threadIdx.x; i < ACC; i += kBlock) s_acc[i] = 0.f;
  for (int s = 0; s < T; s++) {
    const int base = slice_base[s];
    const int nhot = slice_len[s] < HOT ? slice_len[s] : HOT;
    __syncthreads();  // (the previous slice's folds are done: its hot set may go, its running values are in s_acc)
    for (int i = threadIdx.x; i < nhot; i += kBlock) s_hot[i] = x[base + i];
    __syncthreads();
    const int pb = blk_first[wg * T + s], pe = blk_first[wg * T + s + 1];
    for (int p0 = pb + wv * 64; p0 < pe; p0 += W * 64) {
      const int p = p0 + lane;
      uint32_t e0 = 0, e1 = 0;
      int rl = 0;
      if (p < pe) { e0 = piece_start[p]; e1 = piece_start[p + 1]; rl = piece_row[p]; }
      const int lastl = (pe - p0 - 1) < 63 ? (pe - p0 - 1) : 63;
      const uint32_t g0 = __builtin_amdgcn_readlane(e0, 0), g1 = __builtin_amdgcn_readlane(e1, lastl);
      float acc = s_acc[rl];
      for (uint32_t c0 = g0; c0 < g1; c0 += CH) {
        const int n = (int)((g1 - c0) < (uint32_t)CH ? (g1 - c0) : (uint32_t)CH);
        int c[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
          const int k = lane + 64 * j;
          c[j] = __builtin_nontemporal_load(&col[c0 + (k < n ? k : n - 1)]);
        }
        float m[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
          const unsigned rel = (unsigned)(c[j] - base);
          const bool h = rel < (unsigned)nhot;
          const float mh = s_hot[h ? rel : 0u];
          const float mg = x[h ? base : c[j]];
          m[j] = h ? mh : mg;
        }
#pragma unroll
        for (int j = 0; j < PER; j++) {
          const int k = lane + 64 * j;
          if (k < n) sm[SLOT(k)] = m[j];
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t ka = e0 > c0 ? e0 : c0, kb = e1 < c0 + n ? e1 : c0 + n;
        if (ka < kb) {
          int k = (int)(ka - c0);
          const int ke = (int)(kb - c0);
          for (; k + 4 <= ke; k += 4) {
            float r[4];
#pragma unroll
            for (int u = 0; u < 4; u++) r[u] = sm[SLOT(k + u)];
#pragma unroll
            for (int u = 0; u < 4; u++) acc += r[u];
          }
          for (; k < ke; k++) acc += sm[SLOT(k)];
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (p < pe) s_acc[rl] = acc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ACC; i += kBlock) {
    const long long r = (long long)i * kWG + wg;
    if (r < nmed) y_by_rank[r] = s_acc[i];
  }
#undef SLOT
}

template <class K, class V>
static void sort_pairs(K* kin, K* kout, V* vin, V* vout, size_t n, int bits) {
  size_t tb = 0;
  OK(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, 0, bits, (hipStream_t)0));
  void* tmp; OK(hipMalloc(&tmp, tb + 256));
  OK(rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, 0, bits, (hipStream_t)0));
  OK(hipDeviceSynchronize());
  OK(hipFree(tmp));
}

int main(int argc, char** argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 26;
  const int T = argc > 2 ? atoi(argv[2]) : 32;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  // shard emulation (round 4, closing session): only the medium rows that the degree-ranked deal gives shard `shard` of
  // `nshards` (every nshards-th row of the length ranking) are swept, against the WHOLE message vector -- what a row shard
  // of a multi-GPU run would sweep if its device order were (slice, degree rank) dealt over the shards
  const int nshards = argc > 4 ? atoi(argv[4]) : 1;
  const int shard = argc > 5 ? atoi(argv[5]) : 0;
  if (nshards < 1 || shard < 0 || shard >= nshards) { printf("shard: 0..nshards-1\n"); return 1; }
  if (T < 1 || T > kMaxT) { printf("slices: 1..%d\n", kMaxT); return 1; }
  const int nv = 1 << scale;
  const int64_t ne = 16ll * nv;
  const int cbits = scale;  // native column id bits
  if (13 + 6 + 16 + cbits > 64) { printf("scale too large\n"); return 1; }
  const int G = 4096;
  int32_t *src, *dst;
  OK(hipMalloc(&src, ne * 4)); OK(hipMalloc(&dst, ne * 4));
  if (gm_rmat_generate(scale, 1, 0, ne, src, dst, nullptr, 0, nullptr) != 0) { printf("gm_rmat_generate: %s\n", gm_last_error()); return 1; }
  OK(hipDeviceSynchronize());
  // medium rows, ranked by length (descending) and dealt round robin over the workgroups
  uint32_t* deg; OK(hipMalloc(&deg, (size_t)nv * 4)); OK(hipMemset(deg, 0, (size_t)nv * 4));
  k_deg_in<<<G, 256>>>(dst, ne, deg);
  unsigned char* flag; OK(hipMalloc(&flag, nv));
  k_flag_medium<<<(nv + 255) / 256, 256>>>(deg, nv, flag);
  int32_t *iota, *rows; uint32_t* d_cnt;
  OK(hipMalloc(&iota, (size_t)nv * 4)); OK(hipMalloc(&rows, (size_t)nv * 4)); OK(hipMalloc(&d_cnt, 16));
  k_iota<<<(nv + 255) / 256, 256>>>(iota, nv);
  {
    size_t tb = 0;
    OK(rocprim::select(nullptr, tb, iota, flag, rows, d_cnt, (size_t)nv, (hipStream_t)0));
    void* tmp; OK(hipMalloc(&tmp, tb + 256));
    OK(rocprim::select(tmp, tb, iota, flag, rows, d_cnt, (size_t)nv, (hipStream_t)0));
    OK(hipDeviceSynchronize()); OK(hipFree(tmp));
  }
  uint32_t nmed_u = 0; OK(hipMemcpy(&nmed_u, d_cnt, 4, hipMemcpyDeviceToHost));
  int nmed = (int)nmed_u;
  uint32_t *rdeg, *rdeg2; int32_t* rows_sorted;
  OK(hipMalloc(&rdeg, (size_t)nmed * 4)); OK(hipMalloc(&rdeg2, (size_t)nmed * 4)); OK(hipMalloc(&rows_sorted, (size_t)nmed * 4));
  k_gather_deg<<<(nmed + 255) / 256, 256>>>(rows, nmed, deg, rdeg);
  {
    size_t tb = 0;
    OK(rocprim::radix_sort_pairs_desc(nullptr, tb, rdeg, rdeg2, rows, rows_sorted, (size_t)nmed, 0, 32, (hipStream_t)0));
    void* tmp; OK(hipMalloc(&tmp, tb + 256));
    OK(rocprim::radix_sort_pairs_desc(tmp, tb, rdeg, rdeg2, rows, rows_sorted, (size_t)nmed, 0, 32, (hipStream_t)0));
    OK(hipDeviceSynchronize()); OK(hipFree(tmp));
  }
  if (nshards > 1) {  // keep ranks shard, shard + nshards, ...
    const int keep = (nmed - shard + nshards - 1) / nshards;
    int32_t* sub; OK(hipMalloc(&sub, (size_t)keep * 4));
    k_every<<<(keep + 255) / 256, 256>>>(rows_sorted, keep, nshards, shard, sub);
    OK(hipDeviceSynchronize());
    rows_sorted = sub;
    nmed = keep;
  }
  int32_t* rank_of; OK(hipMalloc(&rank_of, (size_t)nv * 4)); OK(hipMemset(rank_of, 0xff, (size_t)nv * 4));
  k_rank_of<<<(nmed + 255) / 256, 256>>>(rows_sorted, nmed, rank_of);
  const int rows_per_wg = (nmed + kWG - 1) / kWG;
  // column slices (equal gather weight), degree rank inside a slice
  uint32_t* w; OK(hipMalloc(&w, (size_t)nv * 4)); OK(hipMemset(w, 0, (size_t)nv * 4));
  k_col_weight<<<G, 256>>>(src, dst, ne, rank_of, w);
  unsigned long long *w64, *pre;
  OK(hipMalloc(&w64, (size_t)nv * 8)); OK(hipMalloc(&pre, (size_t)nv * 8));
  k_widen<<<(nv + 255) / 256, 256>>>(w, nv, w64);
  {
    size_t tb = 0;
    OK(rocprim::inclusive_scan(nullptr, tb, w64, pre, (size_t)nv, rocprim::plus<unsigned long long>(), (hipStream_t)0));
    void* tmp; OK(hipMalloc(&tmp, tb + 256));
    OK(rocprim::inclusive_scan(tmp, tb, w64, pre, (size_t)nv, rocprim::plus<unsigned long long>(), (hipStream_t)0));
    OK(hipDeviceSynchronize()); OK(hipFree(tmp));
  }
  int32_t* bound; OK(hipMalloc(&bound, (kMaxT + 2) * 4));
  k_bounds<<<1, 128>>>(pre, nv, T, bound);
  unsigned long long total_w = 0; OK(hipMemcpy(&total_w, pre + (nv - 1), 8, hipMemcpyDeviceToHost));
  unsigned long long *ckey = w64, *ckey2 = pre;  // (reused)
  int32_t *cols_sorted, *dev_of, *slice_base;
  OK(hipMalloc(&cols_sorted, (size_t)nv * 4)); OK(hipMalloc(&dev_of, (size_t)nv * 4)); OK(hipMalloc(&slice_base, (kMaxT + 2) * 4));
  OK(hipMemset(slice_base, 0, (kMaxT + 2) * 4));
  k_col_keys<<<(nv + 255) / 256, 256>>>(w, nv, bound, T, ckey);
  sort_pairs(ckey, ckey2, iota, cols_sorted, (size_t)nv, 40);
  k_col_map<<<(nv + 255) / 256, 256>>>(cols_sorted, ckey2, nv, dev_of, slice_base);
  OK(hipDeviceSynchronize());
  std::vector<int32_t> h_bound(T + 1), h_base(T + 1), h_len(T);
  OK(hipMemcpy(h_bound.data(), bound, (T + 1) * 4, hipMemcpyDeviceToHost));
  OK(hipMemcpy(h_base.data(), slice_base, T * 4, hipMemcpyDeviceToHost));
  h_base[T] = nv;
  for (int s = 0; s < T; s++) h_len[s] = h_base[s + 1] - h_base[s];
  int32_t* slice_len; OK(hipMalloc(&slice_len, T * 4)); OK(hipMemcpy(slice_len, h_len.data(), T * 4, hipMemcpyHostToDevice));
  OK(hipFree(w64)); OK(hipFree(pre));
  // edges of the medium rows in the sweep's order and in the reference's order
  unsigned long long *k1, *k1s, *k2, *k2s; int32_t *v, *v1s, *v2s;
  OK(hipMalloc(&k1, ne * 8)); OK(hipMalloc(&k1s, ne * 8)); OK(hipMalloc(&k2, ne * 8)); OK(hipMalloc(&k2s, ne * 8));
  OK(hipMalloc(&v, ne * 4)); OK(hipMalloc(&v1s, ne * 4)); OK(hipMalloc(&v2s, ne * 4));
  k_edge_keys<<<G, 256>>>(src, dst, ne, rank_of, bound, T, dev_of, cbits, k1, v, k2);
  OK(hipDeviceSynchronize());
  OK(hipFree(src)); OK(hipFree(dst));
  sort_pairs(k1, k1s, v, v1s, (size_t)ne, 64);
  sort_pairs(k2, k2s, v, v2s, (size_t)ne, 64);
  OK(hipFree(k1)); OK(hipFree(k2)); OK(hipFree(v));
  const int64_t nedges = (int64_t)total_w;  // medium edges (the rest sorted to the end with key ~0)
  // pieces
  uint32_t *head, *pidx;
  OK(hipMalloc(&head, nedges * 4)); OK(hipMalloc(&pidx, nedges * 4));
  k_heads<<<G, 256>>>(k1s, nedges, cbits, head);
  {
    size_t tb = 0;
    OK(rocprim::inclusive_scan(nullptr, tb, head, pidx, (size_t)nedges, rocprim::plus<uint32_t>(), (hipStream_t)0));
    void* tmp; OK(hipMalloc(&tmp, tb + 256));
    OK(rocprim::inclusive_scan(tmp, tb, head, pidx, (size_t)nedges, rocprim::plus<uint32_t>(), (hipStream_t)0));
    OK(hipDeviceSynchronize()); OK(hipFree(tmp));
  }
  uint32_t npieces = 0; OK(hipMemcpy(&npieces, pidx + (nedges - 1), 4, hipMemcpyDeviceToHost));
  uint32_t* piece_start; uint16_t* piece_row; int32_t* blk_first;
  OK(hipMalloc(&piece_start, ((size_t)npieces + 1) * 4)); OK(hipMalloc(&piece_row, ((size_t)npieces + 1) * 2)); OK(hipMalloc(&blk_first, ((size_t)kWG * T + 1) * 4));
  OK(hipMemset(blk_first, 0xff, ((size_t)kWG * T + 1) * 4));
  k_pieces<<<G, 256>>>(k1s, head, pidx, nedges, cbits, piece_start, piece_row, blk_first, T);
  const uint32_t ne32 = (uint32_t)nedges;
  OK(hipMemcpy(piece_start + npieces, &ne32, 4, hipMemcpyHostToDevice));
  {
    std::vector<int32_t> h((size_t)kWG * T + 1);
    OK(hipMemcpy(h.data(), blk_first, h.size() * 4, hipMemcpyDeviceToHost));
    h[(size_t)kWG * T] = (int32_t)npieces;
    for (int64_t b = (int64_t)kWG * T - 1; b >= 0; b--) if (h[b] < 0) h[b] = h[b + 1];
    OK(hipMemcpy(blk_first, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  OK(hipFree(head)); OK(hipFree(pidx));
  // reference structure
  uint32_t* row_start; OK(hipMalloc(&row_start, ((size_t)nmed + 1) * 4));
  k_row_starts<<<G, 256>>>(k2s, nedges, cbits, row_start);
  float *x, *y, *yref;
  OK(hipMalloc(&x, (size_t)nv * 4)); OK(hipMalloc(&y, (size_t)nmed * 4)); OK(hipMalloc(&yref, (size_t)nmed * 4));
  k_fill_x<<<(nv + 255) / 256, 256>>>(x, nv);
  OK(hipMemset(y, 0, (size_t)nmed * 4));
  OK(hipDeviceSynchronize());
  if (nshards > 1) printf("shard %d of %d: ", shard, nshards);
  printf("[%d workgroups x %d threads] RMAT-%d: %d rows of %d..%d edges, %lld edges, %u pieces (%.1f edges each), %d slices, %d rows per workgroup\n", kWG, kBlock, scale, nmed, kRowLo, kRowHi, (long long)nedges, npieces,
         (double)nedges / npieces, T, rows_per_wg);
  if (nedges >= (1ll << 32)) { printf("too many edges for 32-bit positions\n"); return 1; }
  constexpr int ACC = KACC;
  if (rows_per_wg > ACC) { printf("rows per workgroup %d > %d\n", rows_per_wg, ACC); return 1; }
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  auto time_it = [&](auto launch, const char* name) {
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < reps + 1; r++) {
      OK(hipEventRecord(e0));
      launch();
      OK(hipEventRecord(e1));
      OK(hipEventSynchronize(e1));
      float ms; OK(hipEventElapsedTime(&ms, e0, e1));
      if (r) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%-44s best %.3f ms, mean %.3f ms  = %.2f ps per edge, %.1f G edges/s\n", name, best, sum / reps, best * 1e9 / nedges, nedges / best * 1e-6);
  };
  constexpr bool kBig = KBLOCK >= 1024;  // (small workgroups: no room for hot sets; the first, verified launch is the one without)
  if constexpr (kBig) time_it([&]() { k_sweep<18432, ACC><<<kWG, kBlock>>>(v1s, piece_start, piece_row, blk_first, slice_base, slice_len, T, x, y, nmed); }, "sweep, 18432 hot entries per slice");
  else time_it([&]() { k_sweep<1, ACC><<<kWG, kBlock>>>(v1s, piece_start, piece_row, blk_first, slice_base, slice_len, T, x, y, nmed); }, "sweep, no hot set (small workgroups)");
  OK(hipGetLastError());
  k_reference<<<(nmed + 255) / 256, 256>>>(row_start, nmed, nedges, v2s, x, yref);
  OK(hipDeviceSynchronize());
  {
    std::vector<float> a(nmed), b(nmed);
    OK(hipMemcpy(a.data(), y, (size_t)nmed * 4, hipMemcpyDeviceToHost));
    OK(hipMemcpy(b.data(), yref, (size_t)nmed * 4, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (int i = 0; i < nmed; i++) bad += memcmp(&a[i], &b[i], 4) != 0;
    printf("against the serial fold in ascending native column order: %lld of %d rows differ (bit compare)\n", (long long)bad, nmed);
  }
  if constexpr (kBig) time_it([&]() { k_sweep<8192, ACC><<<kWG, kBlock>>>(v1s, piece_start, piece_row, blk_first, slice_base, slice_len, T, x, y, nmed); }, "sweep, 8192 hot entries per slice");
  if constexpr (kBig) time_it([&]() { k_sweep<4096, ACC><<<kWG, kBlock>>>(v1s, piece_start, piece_row, blk_first, slice_base, slice_len, T, x, y, nmed); }, "sweep, 4096 hot entries per slice");
  if constexpr (KBLOCK >= 256) time_it([&]() { k_sweep<2048, ACC><<<kWG, kBlock>>>(v1s, piece_start, piece_row, blk_first, slice_base, slice_len, T, x, y, nmed); }, "sweep, 2048 hot entries per slice");
  time_it([&]() { k_sweep<1, ACC><<<kWG, kBlock>>>(v1s, piece_start, piece_row, blk_first, slice_base, slice_len, T, x, y, nmed); }, "sweep, no hot set");
  return 0;
}
