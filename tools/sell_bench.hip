// tools/sell_bench.hip -- prototype (round 5) of the row-stationary sweep in a SLICED-ELLPACK layout, and of the short rows
// in the same layout without slices.
//
// Where round 4 left the medium rows: k_spmv_sweep, one lane per (row, slice) piece, the pieces of a wave's 64 lanes taken
// out of a CSR-ordered column stream in 512-edge steps, staged through LDS and folded lane by lane: ~50 wave instructions
// per 64 edges, a fold whose length is the LONGEST part of a piece inside the step, 98 G L2 requests/s against the 230 G/s the
// chip sustains -- issue- and latency-bound.  This prototype keeps everything that makes the sweep exact (workgroup w owns
// the rows of length rank r % 256 == w, running values in LDS, slices = ascending native column ranges, a piece's edges in
// ascending native column order) and changes the storage of a (workgroup, slice) block:
//   * the block's pieces are sorted by length (descending) and cut into groups of 64: lane = piece;
//   * a group's column ids are stored TRANSPOSED, [k][lane], padded to the group's longest piece (the sort makes the
//     padding small): row k of a group is ONE coalesced 256-byte load that hands every lane the k-th edge of ITS piece;
//   * no LDS staging, no wave barrier, no per-lane bounds: per 64 edges one column load, one gather (LDS hot set or L2),
//     one add under the "lane still has an edge" mask (padding entries carry a flag bit);
//   * the waves of a workgroup take CONTIGUOUS ranges of the block's groups, balanced by rows + groups at build time, so a
//     wave's column ids of a slice are one contiguous stream that it prefetches a batch ahead whatever the group borders;
//   * long pieces (rows of 4097 .. 32768 edges) need no kernel of their own: 64 of them side by side are a full-width group.
// Column entries are BYTE offsets (column << 2) so that neither the LDS nor the global address needs a shift.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude tools/sell_bench.hip -Lgraphmat_amd -lgraphmat_hip -o build/sell_bench
//   LD_LIBRARY_PATH=graphmat_amd build/sell_bench [scale 26] [slices 64] [reps 5] [row_hi 4096]
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

constexpr int kWG = 256;
constexpr int kBlock = 1024;
constexpr int kW = kBlock / 64;
constexpr int kMaxT = 128;
constexpr int kRowLo = 65;
constexpr uint32_t kPad = 0x80000000u;  // flag bit of a padding entry (the rest of it: the slice's first entry)

// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_deg_in(const int32_t* __restrict__ dst, int64_t ne, uint32_t* __restrict__ deg) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&deg[dst[i] - 1], 1u);
}
__global__ void k_flag_range(const uint32_t* __restrict__ deg, int nv, uint32_t lo, uint32_t hi, unsigned char* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) flag[i] = (deg[i] >= lo && deg[i] <= hi) ? 1 : 0;
}
__global__ void k_iota(int32_t* __restrict__ a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ void k_gather_deg(const int32_t* __restrict__ rows, int n, const uint32_t* __restrict__ deg, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = deg[rows[i]];
}
__global__ void k_rank_of(const int32_t* __restrict__ rows_sorted, int n, int32_t* __restrict__ rank_of) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rank_of[rows_sorted[i]] = i;
}
// weight of a column = how often it is gathered (all rows)
__global__ void k_col_weight(const int32_t* __restrict__ src, int64_t ne, uint32_t* __restrict__ w) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&w[src[i] - 1], 1u);
}
__global__ void k_widen(const uint32_t* __restrict__ w, int n, unsigned long long* __restrict__ o) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = w[i];
}
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
  int lo = 0, hi = T;
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
// key = wg(8) | slice(7) | local row(16) | native column ; value = device column ; key2 = rank | native column
__global__ void k_edge_keys(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t ne, const int32_t* __restrict__ rank_of,
                            const int32_t* __restrict__ bound, int T, const int32_t* __restrict__ dev_of, int cbits, unsigned long long* __restrict__ key,
                            int32_t* __restrict__ val, unsigned long long* __restrict__ key2) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = rank_of[dst[i] - 1], c = src[i] - 1;
    if (r < 0) { key[i] = ~0ull; key2[i] = ~0ull; val[i] = 0; continue; }
    const unsigned long long wg = (unsigned)r % kWG, local = (unsigned)r / kWG;
    key[i] = (((wg << 7 | (unsigned long long)slice_of(bound, T, c)) << 16 | local) << cbits) | (unsigned long long)c;
    key2[i] = ((unsigned long long)r << cbits) | (unsigned long long)c;
    val[i] = dev_of[c];
  }
}
__global__ void k_count_valid(const unsigned long long* __restrict__ key, int64_t n, unsigned long long* __restrict__ cnt) {
  unsigned long long c = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += key[i] != ~0ull;
  atomicAdd(cnt, c);
}
__global__ void k_heads(const unsigned long long* __restrict__ key, int64_t n, int cbits, uint32_t* __restrict__ head) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    head[i] = (i == 0 || (key[i] >> cbits) != (key[i - 1] >> cbits)) ? 1u : 0u;
}
// pieces in (workgroup, slice, row) order: start, slot, block
__global__ void k_pieces(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ head, const uint32_t* __restrict__ pidx_incl, int64_t n, int cbits,
                         uint32_t* __restrict__ piece_start, uint16_t* __restrict__ piece_row, int32_t* __restrict__ blk_first, int T) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!head[i]) continue;
    const uint32_t p = pidx_incl[i] - 1;
    const unsigned long long k = key[i] >> cbits;
    piece_start[p] = (uint32_t)i;
    piece_row[p] = (uint16_t)(k & 0xffff);
    const int blk = (int)(k >> 16);  // wg << 7 | slice
    if (i == 0 || (int)((key[i - 1] >> cbits) >> 16) != blk) blk_first[(blk >> 7) * T + (blk & 127)] = (int32_t)p;
  }
}
// sort key of a piece inside its block: longest first
__global__ void k_piece_keys(const uint32_t* __restrict__ piece_start, uint32_t np, const int32_t* __restrict__ blk_first, int nblk, uint32_t* __restrict__ pkey, uint32_t* __restrict__ pid,
                             const uint16_t* __restrict__ piece_row, int T, int* __restrict__ rowmin) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  int lo = 0, hi = nblk;  // largest b with blk_first[b] <= p (empty blocks share a start: take the last of them)
  while (hi - lo > 1) { const int mid = (lo + hi) / 2; if ((uint32_t)blk_first[mid] <= p) lo = mid; else hi = mid; }
  const uint32_t len = piece_start[p + 1] - piece_start[p];
  pkey[p] = ((uint32_t)lo << 16) | (0xffffu - (len > 0xffffu ? 0xffffu : len));
  pid[p] = p;
  atomicMin(&rowmin[(int)piece_row[p] * kWG + lo / T], lo % T);  // the first slice in which the row (rank = slot * 256 + workgroup) has an edge
}
__global__ void k_block_groups(const int32_t* __restrict__ blk_first, int nblk, uint32_t* __restrict__ ng) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nblk) ng[b] = (uint32_t)(blk_first[b + 1] - blk_first[b] + 63) / 64;
}
// group g of block b: pieces q0 .. q0+63 of the sorted order; width = the first (longest) piece
__global__ void k_group_sizes(const int32_t* __restrict__ blk_first, const uint32_t* __restrict__ grp_first, int nblk, const uint32_t* __restrict__ sp,
                              const uint32_t* __restrict__ piece_start, uint32_t* __restrict__ gsize, uint32_t* __restrict__ gq0, uint32_t* __restrict__ gblk) {
  const int b = blockIdx.x;
  const uint32_t g0 = grp_first[b], g1 = grp_first[b + 1];
  for (uint32_t g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
    const uint32_t q0 = (uint32_t)blk_first[b] + (g - g0) * 64;
    const uint32_t p = sp[q0];
    gsize[g] = (piece_start[p + 1] - piece_start[p]) * 64;
    gq0[g] = q0;
    gblk[g] = (uint32_t)b;
  }
}
// fill a group: scol[gbase + k * 64 + lane] = byte offset of the k-th column of the lane's piece, or the padding entry
__global__ void k_fill_groups(uint32_t ngroups, const uint32_t* __restrict__ gbase, const uint32_t* __restrict__ gq0, const uint32_t* __restrict__ gblk,
                              const int32_t* __restrict__ blk_first, const uint32_t* __restrict__ sp, const uint32_t* __restrict__ piece_start,
                              const uint16_t* __restrict__ piece_row, const int32_t* __restrict__ col_sorted, const int32_t* __restrict__ slice_base, int T,
                              uint32_t* __restrict__ scol, uint16_t* __restrict__ pslot, const int* __restrict__ rowmin) {
  for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const uint32_t b = gblk[g], q0 = gq0[g], qe = (uint32_t)blk_first[b + 1];
    const uint32_t base = gbase[g], n = gbase[g + 1] - base;
    const uint32_t pad = kPad | ((uint32_t)slice_base[b % T] << 2);
    const int lane = threadIdx.x & 63;
    const uint32_t q = q0 + lane;
    uint32_t ps = 0, len = 0;
    if (q < qe) {
      const uint32_t p = sp[q]; ps = piece_start[p]; len = piece_start[p + 1] - ps;
      const int first = rowmin[(int)piece_row[p] * kWG + (int)(b / T)] == (int)(b % T);
      if (threadIdx.x < 64) pslot[(size_t)g * 64 + lane] = (uint16_t)(piece_row[p] | (first ? 0x8000 : 0));
    }
    else if (threadIdx.x < 64) pslot[(size_t)g * 64 + lane] = 0xffff;
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
      const uint32_t k = j >> 6;
      scol[base + j] = k < len ? ((uint32_t)col_sorted[ps + k] << 2) : pad;
    }
  }
}
// contiguous ranges of a block's groups for the W waves, balanced by rows + 2 per group; the last fold_waves waves (they fold
// the long rows first) get fold_share percent of an equal share
__global__ void k_wave_ranges(const uint32_t* __restrict__ grp_first, int nblk, const uint32_t* __restrict__ gbase, uint32_t* __restrict__ wfirst, int fold_waves, int fold_share) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const uint32_t g0 = grp_first[b], g1 = grp_first[b + 1];
  unsigned long long total = 0;
  for (uint32_t g = g0; g < g1; g++) total += (gbase[g + 1] - gbase[g]) / 64 + 2;
  const unsigned long long units = (unsigned long long)(kW - fold_waves) * 100 + (unsigned long long)fold_waves * fold_share;
  uint32_t g = g0;
  unsigned long long acc = 0, share = 0;
  for (int w = 0; w < kW; w++) {
    wfirst[(size_t)b * (kW + 1) + w] = g;
    share += w >= kW - fold_waves ? fold_share : 100;
    const unsigned long long want = total * share / units;
    while (g < g1 && acc + ((gbase[g + 1] - gbase[g]) / 64 + 2 + 1) / 2 <= want) { acc += (gbase[g + 1] - gbase[g]) / 64 + 2; g++; }
  }
  wfirst[(size_t)b * (kW + 1) + kW] = g1;
}
__global__ void k_row_starts(const unsigned long long* __restrict__ key2, int64_t n, int cbits, uint32_t* __restrict__ row_start) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (i == 0 || (key2[i] >> cbits) != (key2[i - 1] >> cbits)) row_start[key2[i] >> cbits] = (uint32_t)i;
}
__global__ void k_reference(const uint32_t* __restrict__ row_start, int nrows, int64_t nedges, const int32_t* __restrict__ col, const float* __restrict__ x, float* __restrict__ y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int64_t e0 = row_start[r], e1 = r + 1 < nrows ? row_start[r + 1] : nedges;
  float acc = x[col[e0]];
  for (int64_t k = e0 + 1; k < e1; k++) acc += x[col[k]];
  y[r] = acc;
}
__global__ void k_fill_x(float* __restrict__ x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = (float)(h >> 8) * (1.0f / 16777216.0f) + 1e-3f; }
}

// ---- the sweep over SELL groups ---------------------------------------------------------------------------------------
// U rows of 64 entries per batch; the next batch's column entries are requested before this batch's messages are waited for.
// Medium rows: SELL groups (pslot: bits 0-14 accumulator slot, bit 15 = the row's first piece: its first message is assigned).
// Long rows (optional, NLP > 0: at most NLP per workgroup, slots ACC - NLP ..): their pieces are too few and too uneven for
// 64-wide groups, so per slice ALL waves gather the block's long-row messages into an LDS stage (coalesced column stream in
// (slot, native column) order, no padding), and the last NLP threads of the workgroup then fold one piece each out of LDS.
template <int HOT, int ACC, int U, int STG, int NLP, bool ASYNC = false>
__global__ void __launch_bounds__(kBlock)
k_sell_sweep(const uint32_t* __restrict__ scol, const uint32_t* __restrict__ gbase, const uint16_t* __restrict__ pslot, const uint32_t* __restrict__ wfirst,
             const uint32_t* __restrict__ lcol, const uint32_t* __restrict__ lps,
             const int32_t* __restrict__ slice_base, const int32_t* __restrict__ slice_len, int T, const float* __restrict__ x, float* __restrict__ y_by_rank, int nrows,
             float* __restrict__ yl_by_rank, int nlong) {
  __shared__ float s_hot[HOT];
  __shared__ float s_acc[ACC];
  __shared__ float s_stage[STG > 0 ? STG : 1];
  __shared__ int s_cnt;
  const int wg = blockIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const char* __restrict__ xb = (const char*)x;
  constexpr int LB = ACC - NLP;
  const int lj = (int)threadIdx.x - (kBlock - NLP);  // long-row slot of this thread (the last NLP threads fold)
  bool lhas = false;
  float lacc = 0.f;
  for (int s = 0; s < T; s++) {
    const uint32_t base4 = (uint32_t)slice_base[s] << 2;
    const int nhot = slice_len[s] < HOT ? slice_len[s] : HOT;
    const uint32_t nhot4 = (uint32_t)nhot << 2;
    __syncthreads();
    for (int i = threadIdx.x; i < nhot; i += kBlock) s_hot[i] = x[slice_base[s] + i];
    if (ASYNC && threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if constexpr (NLP > 0 && ASYNC) {
      // every wave stages its share of the block's long-row messages and signs off on an LDS counter; only the folding
      // waves wait for the others (the block fits the stage: checked by the host)
      const size_t eb = (size_t)(wg * T + s) * NLP;
      const uint32_t l0 = lps[eb], l1 = lps[eb + NLP];
      const uint32_t n = l1 - l0, nr = (n + 63) >> 6;
      const uint32_t ra = nr * (uint32_t)wv / kW, rb = nr * (uint32_t)(wv + 1) / kW;
      uint32_t ps = 0, pe = 0;
      if (lj >= 0) { ps = lps[eb + lj]; pe = lps[eb + lj + 1]; }
      for (uint32_t r0 = ra; r0 < rb; r0 += 4) {
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t i = (r0 + j) * 64 + lane; c[j] = __builtin_nontemporal_load(&lcol[l0 + (i < n ? i : n - 1)]); }
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t rel4 = c[j] - base4;
          const bool h = rel4 < nhot4;
          const float mh = *(const float*)((const char*)s_hot + (h ? rel4 : 0u));
          const float mg = *(const float*)(xb + (h ? base4 : c[j]));
          m[j] = h ? mh : mg;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint32_t i = (r0 + j) * 64 + lane; if (r0 + j < rb && i < n) s_stage[i] = m[j]; }
      }
      if (lane == 0) __hip_atomic_fetch_add(&s_cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lj >= 0) {
        while (__hip_atomic_load(&s_cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kW) __builtin_amdgcn_s_sleep(1);
        uint32_t k = ps;
        if (k < pe && !lhas) { lacc = s_stage[k - l0]; lhas = true; k++; }
        for (; k + 4 <= pe; k += 4) {
          float r[4];
#pragma unroll
          for (int u = 0; u < 4; u++) r[u] = s_stage[k - l0 + u];
#pragma unroll
          for (int u = 0; u < 4; u++) lacc += r[u];
        }
        for (; k < pe; k++) lacc += s_stage[k - l0];
      }
    } else if constexpr (NLP > 0) {
      const size_t eb = (size_t)(wg * T + s) * NLP;
      const uint32_t l0 = lps[eb], l1 = lps[eb + NLP];
      uint32_t ps = 0, pe = 0;
      if (lj >= 0) { ps = lps[eb + lj]; pe = lps[eb + lj + 1]; }
      for (uint32_t c0 = l0; c0 < l1; c0 += STG) {
        const uint32_t n = l1 - c0 < (uint32_t)STG ? l1 - c0 : (uint32_t)STG;
        if (c0 != l0) __syncthreads();  // the previous chunk is folded: the stage may be overwritten
        for (uint32_t i0 = 0; i0 < n; i0 += kBlock * 4) {
          uint32_t c[4];
#pragma unroll
          for (int j = 0; j < 4; j++) { const uint32_t i = i0 + j * kBlock + threadIdx.x; c[j] = __builtin_nontemporal_load(&lcol[c0 + (i < n ? i : n - 1)]); }
          float m[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t rel4 = c[j] - base4;
            const bool h = rel4 < nhot4;
            const float mh = *(const float*)((const char*)s_hot + (h ? rel4 : 0u));
            const float mg = *(const float*)(xb + (h ? base4 : c[j]));
            m[j] = h ? mh : mg;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) { const uint32_t i = i0 + j * kBlock + threadIdx.x; if (i < n) s_stage[i] = m[j]; }
        }
        __syncthreads();
        if (lj >= 0) {
          uint32_t k = ps > c0 ? ps : c0;
          const uint32_t ke = pe < c0 + n ? pe : c0 + n;
          if (k < ke && !lhas) { lacc = s_stage[k - c0]; lhas = true; k++; }
          for (; k + 4 <= ke; k += 4) {
            float r[4];
#pragma unroll
            for (int u = 0; u < 4; u++) r[u] = s_stage[k - c0 + u];
#pragma unroll
            for (int u = 0; u < 4; u++) lacc += r[u];
          }
          for (; k < ke; k++) lacc += s_stage[k - c0];
        }
      }
    }
    const uint32_t* __restrict__ wf = wfirst + (size_t)(wg * T + s) * (kW + 1);
    uint32_t g = __builtin_amdgcn_readfirstlane(wf[wv]);
    const uint32_t gl = __builtin_amdgcn_readfirstlane(wf[wv + 1]);
    if (g >= gl) continue;
    uint32_t r = __builtin_amdgcn_readfirstlane(gbase[g] >> 6);            // current row of the stream (in units of 64 entries)
    const uint32_t rend = __builtin_amdgcn_readfirstlane(gbase[gl] >> 6);  // end of the wave's stream
    uint32_t gend = __builtin_amdgcn_readfirstlane(gbase[g + 1] >> 6);     // end of the current group
    int slot = pslot[(size_t)g * 64 + lane];
    int nslot = g + 1 < gl ? pslot[(size_t)(g + 1) * 64 + lane] : 0xffff;
    float acc = slot != 0xffff ? s_acc[slot & 0x7fff] : 0.f;
    bool has = !(slot & 0x8000);
    uint32_t c[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
      const uint32_t rr = r + j < rend ? r + j : rend - 1;
      c[j] = __builtin_nontemporal_load(&scol[(size_t)rr * 64 + lane]);
    }
    while (r < rend) {
      float m[U];
      bool valid[U];
#pragma unroll
      for (int j = 0; j < U; j++) {
        valid[j] = (int32_t)c[j] >= 0;
        const uint32_t c4 = c[j] & 0x7fffffffu;
        const uint32_t rel4 = c4 - base4;
        const bool h = rel4 < nhot4;
        const float mh = *(const float*)((const char*)s_hot + (h ? rel4 : 0u));
        const float mg = *(const float*)(xb + (h ? base4 : c4));
        m[j] = h ? mh : mg;
      }
      const uint32_t r0 = r;
      r += U;
      if (r < rend) {
#pragma unroll
        for (int j = 0; j < U; j++) {
          const uint32_t rr = r + j < rend ? r + j : rend - 1;
          c[j] = __builtin_nontemporal_load(&scol[(size_t)rr * 64 + lane]);
        }
      }
#pragma unroll
      for (int j = 0; j < U; j++) {
        if (r0 + j < rend) {
          if (r0 + j == gend) {  // the group is done: its running values go back, the next group's come out
            if (slot != 0xffff) s_acc[slot & 0x7fff] = acc;
            g++;
            gend = __builtin_amdgcn_readfirstlane(gbase[g + 1] >> 6);
            slot = nslot;
            nslot = g + 1 < gl ? pslot[(size_t)(g + 1) * 64 + lane] : 0xffff;
            acc = slot != 0xffff ? s_acc[slot & 0x7fff] : 0.f;
            has = !(slot & 0x8000);
          }
          if (valid[j]) { acc = has ? acc + m[j] : m[j]; has = true; }
        }
      }
    }
    if (slot != 0xffff) s_acc[slot & 0x7fff] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LB; i += kBlock) {
    const long long rr = (long long)i * kWG + wg;
    if (rr < nrows) y_by_rank[rr] = s_acc[i];
  }
  if constexpr (NLP > 0) {
    if (lj >= 0) { const long long rr = (long long)lj * kWG + wg; if (rr < nlong && lhas) yl_by_rank[rr] = lacc; }
  }
}

// long rows: entry e = (workgroup * T + slice) * NLP + slot -> first position in the sorted key stream
__global__ void k_long_starts(const unsigned long long* __restrict__ key, int64_t n, int cbits, int T, int NLP, size_t nent, uint32_t* __restrict__ lps) {
  const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (e > nent) return;
  if (e == nent) { lps[e] = (uint32_t)n; return; }
  const size_t b = e / NLP;
  const unsigned long long wg = b / T, sl = b % T, j = e % NLP;
  const unsigned long long want = (((wg << 7 | sl) << 16) | j) << cbits;
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (key[mid] >= want) hi = mid; else lo = mid + 1; }
  lps[e] = (uint32_t)lo;
}
__global__ void k_shift2(const int32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)in[i] << 2;
}

// ---- short rows: lane = row, whole rows, groups of 64 rows of (nearly) equal length ------------------------------------
// plain form: one wave per group
template <int U>
__global__ void __launch_bounds__(256)
k_sell_short(const uint32_t* __restrict__ scol, const uint32_t* __restrict__ gbase, uint32_t ngroups, const float* __restrict__ x, float* __restrict__ y_by_rank, int nrows) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= ngroups) return;
  const uint32_t b = gbase[g], w = (gbase[g + 1] - b) >> 6;
  const char* __restrict__ xb = (const char*)x;
  float acc = 0.f;
  bool has = false;
  for (uint32_t k = 0; k < w; k += U) {
    uint32_t c[U];
#pragma unroll
    for (int j = 0; j < U; j++) c[j] = __builtin_nontemporal_load(&scol[(size_t)b + (size_t)(k + j < w ? k + j : w - 1) * 64 + lane]);
    float m[U];
#pragma unroll
    for (int j = 0; j < U; j++) m[j] = *(const float*)(xb + (c[j] & 0x7fffffffu));
#pragma unroll
    for (int j = 0; j < U; j++)
      if (k + j < w && (int32_t)c[j] >= 0) { acc = has ? acc + m[j] : m[j]; has = true; }
  }
  const long long r = (long long)g * 64 + lane;
  if (r < nrows && has) y_by_rank[r] = acc;
}
// persistent form with the hottest columns in LDS: hot columns are entries [slice_base[s], slice_base[s] + HS) of every slice;
// the column entry of a hot column is kHotBit | (LDS byte offset)
constexpr uint32_t kHotBit = 0x40000000u;
template <int HOTN, int U>
__global__ void __launch_bounds__(kBlock)
k_sell_short_hot(const uint32_t* __restrict__ scol, const uint32_t* __restrict__ gbase, uint32_t ngroups, const int32_t* __restrict__ slice_base, int T, int HS,
                 const float* __restrict__ x, float* __restrict__ y_by_rank, int nrows) {
  __shared__ float s_hot[HOTN];
  for (int i = threadIdx.x; i < T * HS; i += kBlock) s_hot[i] = x[slice_base[i / HS] + i % HS];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const char* __restrict__ xb = (const char*)x;
  const uint32_t nw = gridDim.x * kW, w0 = blockIdx.x * kW + (threadIdx.x >> 6);
  // contiguous ranges of groups per wave, equal rows
  const uint32_t rows_total = gbase[ngroups] >> 6;
  (void)rows_total;
  for (uint32_t g = w0; g < ngroups; g += nw) {
    const uint32_t b = gbase[g], w = (gbase[g + 1] - b) >> 6;
    float acc = 0.f;
    bool has = false;
    for (uint32_t k = 0; k < w; k += U) {
      uint32_t c[U];
#pragma unroll
      for (int j = 0; j < U; j++) c[j] = __builtin_nontemporal_load(&scol[(size_t)b + (size_t)(k + j < w ? k + j : w - 1) * 64 + lane]);
      float m[U];
#pragma unroll
      for (int j = 0; j < U; j++) {
        const bool h = (c[j] & kHotBit) != 0;
        const uint32_t off = c[j] & 0x3fffffffu;
        const float mh = *(const float*)((const char*)s_hot + (h ? off : 0u));
        const float mg = *(const float*)(xb + (h ? 0u : off));
        m[j] = h ? mh : mg;
      }
#pragma unroll
      for (int j = 0; j < U; j++)
        if (k + j < w && (int32_t)c[j] >= 0) { acc = has ? acc + m[j] : m[j]; has = true; }
    }
    const long long r = (long long)g * 64 + lane;
    if (r < nrows && has) y_by_rank[r] = acc;
  }
}
// short rows: key = rank | native column
__global__ void k_short_keys(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t ne, const int32_t* __restrict__ rank_of,
                             const int32_t* __restrict__ dev_of, int cbits, unsigned long long* __restrict__ key, int32_t* __restrict__ val) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = rank_of[dst[i] - 1], c = src[i] - 1;
    if (r < 0) { key[i] = ~0ull; val[i] = 0; continue; }
    key[i] = ((unsigned long long)r << cbits) | (unsigned long long)c;
    val[i] = dev_of[c];
  }
}
__global__ void k_short_gsize(const uint32_t* __restrict__ row_start, int nrows, int64_t nedges, uint32_t ngroups, uint32_t* __restrict__ gsize) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  const int r = (int)g * 64;  // the longest row of the group (rows are ranked by length, descending)
  const int64_t e0 = row_start[r], e1 = r + 1 < nrows ? row_start[r + 1] : nedges;
  gsize[g] = (uint32_t)(e1 - e0) * 64;
}
__global__ void k_short_fill(const uint32_t* __restrict__ row_start, int nrows, int64_t nedges, uint32_t ngroups, const uint32_t* __restrict__ gbase,
                             const int32_t* __restrict__ col_sorted, const int32_t* __restrict__ slice_base, const int32_t* __restrict__ bound_dev, int T, int HS,
                             uint32_t* __restrict__ scol_plain, uint32_t* __restrict__ scol_hot) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= ngroups) return;
  const int r = (int)g * 64 + lane;
  int64_t e0 = 0, e1 = 0;
  if (r < nrows) { e0 = row_start[r]; e1 = r + 1 < nrows ? row_start[r + 1] : nedges; }
  const uint32_t b = gbase[g], w = (gbase[g + 1] - b) >> 6;
  for (uint32_t k = 0; k < w; k++) {
    uint32_t vp = kPad, vh = kPad;
    if ((int64_t)k < e1 - e0) {
      const int c = col_sorted[e0 + k];
      vp = (uint32_t)c << 2;
      int lo = 0, hi = T;  // slice of the device column
      while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (slice_base[mid] <= c) lo = mid; else hi = mid; }
      const int rel = c - slice_base[lo];
      vh = rel < HS ? (kHotBit | (uint32_t)((lo * HS + rel) << 2)) : vp;
    }
    scol_plain[(size_t)b + (size_t)k * 64 + lane] = vp;
    scol_hot[(size_t)b + (size_t)k * 64 + lane] = vh;
  }
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
template <class T>
static void excl_scan(T* in, T* out, size_t n) {
  size_t tb = 0;
  OK(rocprim::exclusive_scan(nullptr, tb, in, out, T(0), n, rocprim::plus<T>(), (hipStream_t)0));
  void* tmp; OK(hipMalloc(&tmp, tb + 256));
  OK(rocprim::exclusive_scan(tmp, tb, in, out, T(0), n, rocprim::plus<T>(), (hipStream_t)0));
  OK(hipDeviceSynchronize());
  OK(hipFree(tmp));
}
template <class T>
static void incl_scan(T* in, T* out, size_t n) {
  size_t tb = 0;
  OK(rocprim::inclusive_scan(nullptr, tb, in, out, n, rocprim::plus<T>(), (hipStream_t)0));
  void* tmp; OK(hipMalloc(&tmp, tb + 256));
  OK(rocprim::inclusive_scan(tmp, tb, in, out, n, rocprim::plus<T>(), (hipStream_t)0));
  OK(hipDeviceSynchronize());
  OK(hipFree(tmp));
}

// rows with lo <= in-degree <= hi, ranked by length (descending, ties by id): rank_of[v] (-1 elsewhere), returns the count
static int rank_rows(const uint32_t* deg, int nv, uint32_t lo, uint32_t hi, int32_t* iota, int32_t* rank_of) {
  unsigned char* flag; OK(hipMalloc(&flag, nv));
  k_flag_range<<<(nv + 255) / 256, 256>>>(deg, nv, lo, hi, flag);
  int32_t* rows; uint32_t* d_cnt;
  OK(hipMalloc(&rows, (size_t)nv * 4)); OK(hipMalloc(&d_cnt, 16));
  {
    size_t tb = 0;
    OK(rocprim::select(nullptr, tb, iota, flag, rows, d_cnt, (size_t)nv, (hipStream_t)0));
    void* tmp; OK(hipMalloc(&tmp, tb + 256));
    OK(rocprim::select(tmp, tb, iota, flag, rows, d_cnt, (size_t)nv, (hipStream_t)0));
    OK(hipDeviceSynchronize()); OK(hipFree(tmp));
  }
  uint32_t n_u = 0; OK(hipMemcpy(&n_u, d_cnt, 4, hipMemcpyDeviceToHost));
  const int n = (int)n_u;
  uint32_t *rdeg, *rdeg2; int32_t* rows_sorted;
  OK(hipMalloc(&rdeg, (size_t)n * 4 + 4)); OK(hipMalloc(&rdeg2, (size_t)n * 4 + 4)); OK(hipMalloc(&rows_sorted, (size_t)n * 4 + 4));
  k_gather_deg<<<(n + 255) / 256, 256>>>(rows, n, deg, rdeg);
  {
    size_t tb = 0;
    OK(rocprim::radix_sort_pairs_desc(nullptr, tb, rdeg, rdeg2, rows, rows_sorted, (size_t)n, 0, 32, (hipStream_t)0));
    void* tmp; OK(hipMalloc(&tmp, tb + 256));
    OK(rocprim::radix_sort_pairs_desc(tmp, tb, rdeg, rdeg2, rows, rows_sorted, (size_t)n, 0, 32, (hipStream_t)0));
    OK(hipDeviceSynchronize()); OK(hipFree(tmp));
  }
  OK(hipMemset(rank_of, 0xff, (size_t)nv * 4));
  k_rank_of<<<(n + 255) / 256, 256>>>(rows_sorted, n, rank_of);
  OK(hipDeviceSynchronize());
  OK(hipFree(flag)); OK(hipFree(rows)); OK(hipFree(d_cnt)); OK(hipFree(rdeg)); OK(hipFree(rdeg2)); OK(hipFree(rows_sorted));
  return n;
}

static hipEvent_t ev0, ev1;
template <class F>
static float time_it(F launch, const char* name, int reps, int64_t nedges) {
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < reps + 1; r++) {
    OK(hipEventRecord(ev0));
    launch();
    OK(hipEventRecord(ev1));
    OK(hipEventSynchronize(ev1));
    float ms; OK(hipEventElapsedTime(&ms, ev0, ev1));
    if (r) { best = ms < best ? ms : best; sum += ms; }
  }
  OK(hipGetLastError());
  printf("%-52s best %.3f ms, mean %.3f ms  = %.2f ps per edge, %.1f G edges/s\n", name, best, sum / reps, best * 1e9 / nedges, nedges / best * 1e-6);
  fflush(stdout);
  return best;
}
static void compare(const float* y, const float* yref, int n, const char* what) {
  std::vector<float> a(n), b(n);
  OK(hipMemcpy(a.data(), y, (size_t)n * 4, hipMemcpyDeviceToHost));
  OK(hipMemcpy(b.data(), yref, (size_t)n * 4, hipMemcpyDeviceToHost));
  int64_t bad = 0;
  for (int i = 0; i < n; i++) bad += memcmp(&a[i], &b[i], 4) != 0;
  printf("%s against the serial fold in ascending native column order: %lld of %d rows differ (bit compare)\n", what, (long long)bad, n);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 26;
  const int T = argc > 2 ? atoi(argv[2]) : 64;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const int row_hi = argc > 4 ? atoi(argv[4]) : 32768;   // long rows up to here are staged through LDS
  const int do_short = argc > 5 ? atoi(argv[5]) : 1;
  const int row_mid = argc > 6 ? atoi(argv[6]) : 4096;  // medium rows (SELL groups) up to here
  const int fold_waves = argc > 7 ? atoi(argv[7]) : 2;  // the last waves fold the long rows: they get fold_share percent of an equal share of the medium groups
  const int fold_share = argc > 8 ? atoi(argv[8]) : 100;
  if (T < 1 || T > kMaxT) { printf("slices: 1..%d\n", kMaxT); return 1; }
  const int nv = 1 << scale;
  const int64_t ne = 16ll * nv;
  const int cbits = scale;
  const int G = 4096;
  OK(hipEventCreate(&ev0)); OK(hipEventCreate(&ev1));
  int32_t *src, *dst;
  OK(hipMalloc(&src, ne * 4)); OK(hipMalloc(&dst, ne * 4));
  if (gm_rmat_generate(scale, 1, 0, ne, src, dst, nullptr, 0, nullptr) != 0) { printf("gm_rmat_generate: %s\n", gm_last_error()); return 1; }
  OK(hipDeviceSynchronize());
  uint32_t* deg; OK(hipMalloc(&deg, (size_t)nv * 4)); OK(hipMemset(deg, 0, (size_t)nv * 4));
  k_deg_in<<<G, 256>>>(dst, ne, deg);
  int32_t* iota; OK(hipMalloc(&iota, (size_t)nv * 4));
  k_iota<<<(nv + 255) / 256, 256>>>(iota, nv);
  int32_t* rank_of; OK(hipMalloc(&rank_of, (size_t)nv * 4));
  // column slices (equal gather weight over ALL edges), weight rank inside a slice
  uint32_t* w; OK(hipMalloc(&w, (size_t)nv * 4)); OK(hipMemset(w, 0, (size_t)nv * 4));
  k_col_weight<<<G, 256>>>(src, ne, w);
  unsigned long long *w64, *pre;
  OK(hipMalloc(&w64, (size_t)nv * 8)); OK(hipMalloc(&pre, (size_t)nv * 8));
  k_widen<<<(nv + 255) / 256, 256>>>(w, nv, w64);
  incl_scan(w64, pre, (size_t)nv);
  int32_t* bound; OK(hipMalloc(&bound, (kMaxT + 2) * 4));
  k_bounds<<<1, 256>>>(pre, nv, T, bound);
  unsigned long long *ckey = w64, *ckey2 = pre;
  int32_t *cols_sorted, *dev_of, *slice_base;
  OK(hipMalloc(&cols_sorted, (size_t)nv * 4)); OK(hipMalloc(&dev_of, (size_t)nv * 4)); OK(hipMalloc(&slice_base, (kMaxT + 2) * 4));
  OK(hipMemset(slice_base, 0, (kMaxT + 2) * 4));
  k_col_keys<<<(nv + 255) / 256, 256>>>(w, nv, bound, T, ckey);
  sort_pairs(ckey, ckey2, iota, cols_sorted, (size_t)nv, 40);
  k_col_map<<<(nv + 255) / 256, 256>>>(cols_sorted, ckey2, nv, dev_of, slice_base);
  OK(hipDeviceSynchronize());
  std::vector<int32_t> h_base(T + 1), h_len(T);
  OK(hipMemcpy(h_base.data(), slice_base, T * 4, hipMemcpyDeviceToHost));
  h_base[T] = nv;
  OK(hipMemcpy(slice_base, h_base.data(), (T + 1) * 4, hipMemcpyHostToDevice));
  for (int s = 0; s < T; s++) h_len[s] = h_base[s + 1] - h_base[s];
  int32_t* slice_len; OK(hipMalloc(&slice_len, T * 4)); OK(hipMemcpy(slice_len, h_len.data(), T * 4, hipMemcpyHostToDevice));
  OK(hipFree(w64)); OK(hipFree(pre));
  float* x; OK(hipMalloc(&x, (size_t)nv * 4));
  k_fill_x<<<(nv + 255) / 256, 256>>>(x, nv);

  // ================= the swept rows =================
  {
    // ---- long rows (row_mid+1 .. row_hi): staged through LDS ----
    constexpr int NLP = 128;
    uint32_t *lcol = nullptr, *lps = nullptr;
    uint32_t maxblk = 0;
    float *yl = nullptr, *ylref = nullptr;
    int nlong = 0;
    int64_t ledges = 0;
    if (row_hi > row_mid) {
      nlong = rank_rows(deg, nv, (uint32_t)row_mid + 1, (uint32_t)row_hi, iota, rank_of);
      if ((nlong + kWG - 1) / kWG > NLP) { printf("long rows per workgroup %d > %d\n", (nlong + kWG - 1) / kWG, NLP); return 1; }
      unsigned long long *k1, *k1s, *k2, *k2s; int32_t *v, *v1s, *v2s;
      OK(hipMalloc(&k1, ne * 8)); OK(hipMalloc(&k1s, ne * 8)); OK(hipMalloc(&k2, ne * 8)); OK(hipMalloc(&k2s, ne * 8));
      OK(hipMalloc(&v, ne * 4)); OK(hipMalloc(&v1s, ne * 4)); OK(hipMalloc(&v2s, ne * 4));
      k_edge_keys<<<G, 256>>>(src, dst, ne, rank_of, bound, T, dev_of, cbits, k1, v, k2);
      unsigned long long* d_cnt; OK(hipMalloc(&d_cnt, 8)); OK(hipMemset(d_cnt, 0, 8));
      k_count_valid<<<G, 256>>>(k1, ne, d_cnt);
      unsigned long long nedges_u = 0; OK(hipMemcpy(&nedges_u, d_cnt, 8, hipMemcpyDeviceToHost));
      ledges = (int64_t)nedges_u;
      sort_pairs(k1, k1s, v, v1s, (size_t)ne, 64);
      sort_pairs(k2, k2s, v, v2s, (size_t)ne, 64);
      OK(hipFree(k1)); OK(hipFree(k2)); OK(hipFree(v));
      const size_t nent = (size_t)kWG * T * NLP;
      OK(hipMalloc(&lps, (nent + 1) * 4)); OK(hipMalloc(&lcol, ((size_t)ledges + 64) * 4));
      k_long_starts<<<(unsigned)((nent + 1 + 255) / 256), 256>>>(k1s, ledges, cbits, T, NLP, nent, lps);
      k_shift2<<<G, 256>>>(v1s, ledges, lcol);
      uint32_t* row_start; OK(hipMalloc(&row_start, ((size_t)nlong + 1) * 4));
      k_row_starts<<<G, 256>>>(k2s, ledges, cbits, row_start);
      OK(hipMalloc(&yl, (size_t)nlong * 4)); OK(hipMalloc(&ylref, (size_t)nlong * 4));
      k_reference<<<(nlong + 255) / 256, 256>>>(row_start, nlong, ledges, v2s, x, ylref);
      OK(hipDeviceSynchronize());
      OK(hipFree(k1s)); OK(hipFree(k2s)); OK(hipFree(v1s)); OK(hipFree(v2s)); OK(hipFree(row_start));
      // the largest block (edges of one workgroup's long rows inside one slice)
      std::vector<uint32_t> h(nent + 1);
      OK(hipMemcpy(h.data(), lps, (nent + 1) * 4, hipMemcpyDeviceToHost));
      uint32_t mx = 0, mxp = 0;
      for (size_t bI = 0; bI < (size_t)kWG * T; bI++) { const uint32_t d = h[(bI + 1) * NLP] - h[bI * NLP]; mx = d > mx ? d : mx; }
      for (size_t e = 0; e < nent; e++) { const uint32_t d = h[e + 1] - h[e]; mxp = d > mxp ? d : mxp; }
      maxblk = mx;
      printf("long rows %d..%d: %d rows, %lld edges; per (workgroup, slice) block %.0f edges on average, %u at most; longest piece %u\n", row_mid + 1, row_hi, nlong, (long long)ledges,
             (double)ledges / (kWG * T), mx, mxp);
    }
    // ---- medium rows (kRowLo .. row_mid): SELL groups ----
    const int nrows = rank_rows(deg, nv, kRowLo, (uint32_t)row_mid, iota, rank_of);
    const int rows_per_wg = (nrows + kWG - 1) / kWG;
    unsigned long long *k1, *k1s, *k2, *k2s; int32_t *v, *v1s, *v2s;
    OK(hipMalloc(&k1, ne * 8)); OK(hipMalloc(&k1s, ne * 8)); OK(hipMalloc(&k2, ne * 8)); OK(hipMalloc(&k2s, ne * 8));
    OK(hipMalloc(&v, ne * 4)); OK(hipMalloc(&v1s, ne * 4)); OK(hipMalloc(&v2s, ne * 4));
    k_edge_keys<<<G, 256>>>(src, dst, ne, rank_of, bound, T, dev_of, cbits, k1, v, k2);
    unsigned long long* d_cnt; OK(hipMalloc(&d_cnt, 8)); OK(hipMemset(d_cnt, 0, 8));
    k_count_valid<<<G, 256>>>(k1, ne, d_cnt);
    unsigned long long nedges_u = 0; OK(hipMemcpy(&nedges_u, d_cnt, 8, hipMemcpyDeviceToHost));
    const int64_t nedges = (int64_t)nedges_u;
    sort_pairs(k1, k1s, v, v1s, (size_t)ne, 64);
    sort_pairs(k2, k2s, v, v2s, (size_t)ne, 64);
    OK(hipFree(k1)); OK(hipFree(k2)); OK(hipFree(v));
    uint32_t *head, *pidx;
    OK(hipMalloc(&head, nedges * 4)); OK(hipMalloc(&pidx, nedges * 4));
    k_heads<<<G, 256>>>(k1s, nedges, cbits, head);
    incl_scan(head, pidx, (size_t)nedges);
    uint32_t npieces = 0; OK(hipMemcpy(&npieces, pidx + (nedges - 1), 4, hipMemcpyDeviceToHost));
    const int nblk = kWG * T;
    uint32_t* piece_start; uint16_t* piece_row; int32_t* blk_first;
    OK(hipMalloc(&piece_start, ((size_t)npieces + 1) * 4)); OK(hipMalloc(&piece_row, ((size_t)npieces + 1) * 2)); OK(hipMalloc(&blk_first, ((size_t)nblk + 1) * 4));
    OK(hipMemset(blk_first, 0xff, ((size_t)nblk + 1) * 4));
    k_pieces<<<G, 256>>>(k1s, head, pidx, nedges, cbits, piece_start, piece_row, blk_first, T);
    const uint32_t ne32 = (uint32_t)nedges;
    OK(hipMemcpy(piece_start + npieces, &ne32, 4, hipMemcpyHostToDevice));
    {
      std::vector<int32_t> h((size_t)nblk + 1);
      OK(hipMemcpy(h.data(), blk_first, h.size() * 4, hipMemcpyDeviceToHost));
      h[(size_t)nblk] = (int32_t)npieces;
      for (int64_t b = (int64_t)nblk - 1; b >= 0; b--) if (h[b] < 0) h[b] = h[b + 1];
      OK(hipMemcpy(blk_first, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    OK(hipFree(head)); OK(hipFree(pidx)); OK(hipFree(k1s));
    if (nedges >= (1ll << 32)) { printf("too many edges for 32-bit positions\n"); return 1; }
    // pieces sorted by (block, length descending)
    uint32_t *pkey, *pkey2, *pid, *sp;
    int* rowmin; OK(hipMalloc(&rowmin, ((size_t)rows_per_wg * kWG + 1) * 4)); OK(hipMemset(rowmin, 0x7f, ((size_t)rows_per_wg * kWG + 1) * 4));
    OK(hipMalloc(&pkey, (size_t)npieces * 4)); OK(hipMalloc(&pkey2, (size_t)npieces * 4)); OK(hipMalloc(&pid, (size_t)npieces * 4)); OK(hipMalloc(&sp, (size_t)npieces * 4));
    k_piece_keys<<<(npieces + 255) / 256, 256>>>(piece_start, npieces, blk_first, nblk, pkey, pid, piece_row, T, rowmin);
    sort_pairs(pkey, pkey2, pid, sp, (size_t)npieces, 32);
    OK(hipFree(pkey)); OK(hipFree(pkey2)); OK(hipFree(pid));
    uint32_t *ng, *grp_first;
    OK(hipMalloc(&ng, ((size_t)nblk + 1) * 4)); OK(hipMalloc(&grp_first, ((size_t)nblk + 1) * 4));
    OK(hipMemset(ng, 0, ((size_t)nblk + 1) * 4));
    k_block_groups<<<(nblk + 255) / 256, 256>>>(blk_first, nblk, ng);
    excl_scan(ng, grp_first, (size_t)nblk + 1);
    uint32_t ngroups = 0; OK(hipMemcpy(&ngroups, grp_first + nblk, 4, hipMemcpyDeviceToHost));
    uint32_t *gsize, *gbase, *gq0, *gblk;
    OK(hipMalloc(&gsize, ((size_t)ngroups + 1) * 4)); OK(hipMalloc(&gbase, ((size_t)ngroups + 1) * 4)); OK(hipMalloc(&gq0, ((size_t)ngroups + 1) * 4)); OK(hipMalloc(&gblk, ((size_t)ngroups + 1) * 4));
    OK(hipMemset(gsize, 0, ((size_t)ngroups + 1) * 4));
    k_group_sizes<<<nblk, 64>>>(blk_first, grp_first, nblk, sp, piece_start, gsize, gq0, gblk);
    {
      std::vector<uint32_t> hs((size_t)ngroups);
      OK(hipMemcpy(hs.data(), gsize, (size_t)ngroups * 4, hipMemcpyDeviceToHost));
      unsigned long long tot = 0; for (uint32_t gI = 0; gI < ngroups; gI++) tot += hs[gI];
      printf("medium rows %d..%d: %d rows, %lld edges, %u pieces (%.2f edges each), %u groups, padded entries %llu (+%.2f %%), %d slices, %d rows per workgroup\n", kRowLo, row_mid, nrows,
             (long long)nedges, npieces, (double)nedges / npieces, ngroups, tot, 100.0 * ((double)tot / nedges - 1.0), T, rows_per_wg);
      if (tot >= (1ull << 32)) { printf("too many padded entries for 32-bit positions\n"); return 1; }
    }
    excl_scan(gsize, gbase, (size_t)ngroups + 1);
    uint32_t total_entries = 0; OK(hipMemcpy(&total_entries, gbase + ngroups, 4, hipMemcpyDeviceToHost));
    uint32_t* scol; uint16_t* pslot;
    OK(hipMalloc(&scol, ((size_t)total_entries + 64 * 64) * 4)); OK(hipMalloc(&pslot, ((size_t)ngroups + 1) * 64 * 2));
    k_fill_groups<<<65536, 256>>>(ngroups, gbase, gq0, gblk, blk_first, sp, piece_start, piece_row, v1s, slice_base, T, scol, pslot, rowmin);
    uint32_t* wfirst; OK(hipMalloc(&wfirst, (size_t)nblk * (kW + 1) * 4));
    k_wave_ranges<<<(nblk + 63) / 64, 64>>>(grp_first, nblk, gbase, wfirst, fold_waves, fold_share);
    OK(hipDeviceSynchronize());
    OK(hipFree(v1s)); OK(hipFree(sp)); OK(hipFree(piece_start)); OK(hipFree(piece_row));
    uint32_t* row_start; OK(hipMalloc(&row_start, ((size_t)nrows + 1) * 4));
    k_row_starts<<<G, 256>>>(k2s, nedges, cbits, row_start);
    float *y, *yref;
    OK(hipMalloc(&y, (size_t)nrows * 4)); OK(hipMalloc(&yref, (size_t)nrows * 4));
    k_reference<<<(nrows + 255) / 256, 256>>>(row_start, nrows, nedges, v2s, x, yref);
    OK(hipDeviceSynchronize());
    OK(hipFree(k2s)); OK(hipFree(v2s));
    constexpr int ACC = 10176;
    if (rows_per_wg > ACC - NLP) { printf("rows per workgroup %d > %d\n", rows_per_wg, ACC - NLP); return 1; }
    const int64_t alledges = nedges + ledges;
#define RUN(HOT, U, STG, NL, AS) do { if ((AS) && maxblk > (uint32_t)(STG)) { printf("(block of %u edges does not fit a stage of %d)\n", maxblk, (int)(STG)); break; } OK(hipMemset(y, 0, (size_t)nrows * 4)); if (nlong) OK(hipMemset(yl, 0, (size_t)nlong * 4)); \
      if ((NL) == 0 || nlong > 0) { \
      time_it([&]() { k_sell_sweep<HOT, ACC, U, STG, NL, AS><<<kWG, kBlock>>>(scol, gbase, pslot, wfirst, lcol, lps, slice_base, slice_len, T, x, y, nrows, yl, nlong); }, \
              "SELL sweep, hot " #HOT ", batch " #U ", stage " #STG ", long slots " #NL ", async " #AS, reps, (NL) ? alledges : nedges); \
      compare(y, yref, nrows, "   medium rows"); if (NL) compare(yl, ylref, nlong, "   long rows"); } } while (0)
    RUN(26624, 8, 0, 0, false);
    RUN(18432, 8, 0, 0, false);
    RUN(22528, 8, 8192, 128, false);
    RUN(22528, 8, 8192, 128, true);
    RUN(21504, 8, 9216, 128, false);
    RUN(21504, 8, 9216, 128, true);
    RUN(18432, 8, 12288, 128, false);
    RUN(18432, 8, 12288, 128, true);
    RUN(16384, 8, 14336, 128, false);
    RUN(16384, 8, 14336, 128, true);
#undef RUN
    OK(hipFree(scol)); OK(hipFree(pslot)); OK(hipFree(wfirst)); OK(hipFree(gbase)); OK(hipFree(gsize)); OK(hipFree(gq0)); OK(hipFree(gblk)); OK(hipFree(y)); OK(hipFree(yref));
    OK(hipFree(row_start)); OK(hipFree(blk_first)); OK(hipFree(ng)); OK(hipFree(grp_first));
  }

  // ================= the short rows =================
  if (do_short) {
    const int nrows = rank_rows(deg, nv, 1, kRowLo - 1, iota, rank_of);
    unsigned long long *k1, *k1s; int32_t *v, *v1s;
    OK(hipMalloc(&k1, ne * 8)); OK(hipMalloc(&k1s, ne * 8)); OK(hipMalloc(&v, ne * 4)); OK(hipMalloc(&v1s, ne * 4));
    k_short_keys<<<G, 256>>>(src, dst, ne, rank_of, dev_of, cbits, k1, v);
    unsigned long long* d_cnt; OK(hipMalloc(&d_cnt, 8)); OK(hipMemset(d_cnt, 0, 8));
    k_count_valid<<<G, 256>>>(k1, ne, d_cnt);
    unsigned long long nedges_u = 0; OK(hipMemcpy(&nedges_u, d_cnt, 8, hipMemcpyDeviceToHost));
    const int64_t nedges = (int64_t)nedges_u;
    sort_pairs(k1, k1s, v, v1s, (size_t)ne, 64);
    OK(hipFree(k1)); OK(hipFree(v));
    uint32_t* row_start; OK(hipMalloc(&row_start, ((size_t)nrows + 1) * 4));
    k_row_starts<<<G, 256>>>(k1s, nedges, cbits, row_start);
    OK(hipDeviceSynchronize());
    OK(hipFree(k1s));
    const uint32_t ngroups = (uint32_t)((nrows + 63) / 64);
    uint32_t *gsize, *gbase;
    OK(hipMalloc(&gsize, ((size_t)ngroups + 1) * 4)); OK(hipMalloc(&gbase, ((size_t)ngroups + 1) * 4));
    OK(hipMemset(gsize, 0, ((size_t)ngroups + 1) * 4));
    k_short_gsize<<<(ngroups + 255) / 256, 256>>>(row_start, nrows, nedges, ngroups, gsize);
    excl_scan(gsize, gbase, (size_t)ngroups + 1);
    uint32_t total_entries = 0; OK(hipMemcpy(&total_entries, gbase + ngroups, 4, hipMemcpyDeviceToHost));
    printf("short rows 1..%d: %d rows, %lld edges, %u groups, padded entries %u (+%.2f %%)\n", kRowLo - 1, nrows, (long long)nedges, ngroups, total_entries, 100.0 * ((double)total_entries / nedges - 1.0));
    constexpr int HOTN = 36864;
    const int HS = HOTN / T;
    uint32_t *scol_plain, *scol_hot;
    OK(hipMalloc(&scol_plain, ((size_t)total_entries + 64) * 4)); OK(hipMalloc(&scol_hot, ((size_t)total_entries + 64) * 4));
    k_short_fill<<<(ngroups + 3) / 4, 256>>>(row_start, nrows, nedges, ngroups, gbase, v1s, slice_base, bound, T, HS, scol_plain, scol_hot);
    float *y, *yref;
    OK(hipMalloc(&y, (size_t)nrows * 4)); OK(hipMalloc(&yref, (size_t)nrows * 4));
    k_reference<<<(nrows + 255) / 256, 256>>>(row_start, nrows, nedges, v1s, x, yref);
    OK(hipDeviceSynchronize());
    OK(hipMemset(y, 0, (size_t)nrows * 4));
    time_it([&]() { k_sell_short<4><<<(ngroups + 3) / 4, 256>>>(scol_plain, gbase, ngroups, x, y, nrows); }, "SELL short rows, one wave per group, batch 4", reps, nedges);
    compare(y, yref, nrows, "  ");
    OK(hipMemset(y, 0, (size_t)nrows * 4));
    time_it([&]() { k_sell_short<8><<<(ngroups + 3) / 4, 256>>>(scol_plain, gbase, ngroups, x, y, nrows); }, "SELL short rows, one wave per group, batch 8", reps, nedges);
    compare(y, yref, nrows, "  ");
    OK(hipMemset(y, 0, (size_t)nrows * 4));
    time_it([&]() { k_sell_short<2><<<(ngroups + 3) / 4, 256>>>(scol_plain, gbase, ngroups, x, y, nrows); }, "SELL short rows, one wave per group, batch 2", reps, nedges);
    compare(y, yref, nrows, "  ");
    OK(hipMemset(y, 0, (size_t)nrows * 4));
    time_it([&]() { k_sell_short_hot<HOTN, 4><<<kWG, kBlock>>>(scol_hot, gbase, ngroups, slice_base, T, HS, x, y, nrows); }, "SELL short rows, persistent, 36864 hot in LDS, batch 4", reps, nedges);
    compare(y, yref, nrows, "  ");
    OK(hipMemset(y, 0, (size_t)nrows * 4));
    time_it([&]() { k_sell_short_hot<HOTN, 8><<<kWG, kBlock>>>(scol_hot, gbase, ngroups, slice_base, T, HS, x, y, nrows); }, "SELL short rows, persistent, 36864 hot in LDS, batch 8", reps, nedges);
    compare(y, yref, nrows, "  ");
  }
  return 0;
}
