// tools/blocked_bench.hip -- prototype (round 5, closing session): the SHORT rows (1 .. row_hi edges) as a column-blocked,
// wave-stationary stream -- measured before anything is built, like tools/sell_bench.hip was for the sweep.
//
// Where the short rows stand: k_spmv_rowblock, 162 M edges of RMAT-26 in 1.27-1.30 ms (7.8 ps per edge; 1.02 L1->L2 requests and
// 0.43 L2 misses per edge: the vector-memory path of every CU is full of misses), and 21.4 ms for the 1.07 G edges of a graph
// without skew (uniform 16-out-regular, every gather a miss: 52 G gathers/s, the 268 MB-table rate of tools/gather_bench.hip).
// A gather that hits in L2 costs a quarter of one that misses (200 G/s chip-wide against 54), so the question is what an
// order-preserving form costs in which every gather goes to an L2-resident slice of x:
//   * the selected rows, in native order, are cut into WAVE blocks of RB rows; a wave owns a block for ALL of its columns, with
//     the rows' running values in its own RB x 4 bytes of LDS: no workgroup barrier anywhere, no atomics;
//   * the columns are cut into S slices (contiguous NATIVE ranges of equal gather weight: a row's fold in ascending native
//     column order is then the concatenation of its per-slice parts) and a block's entries are stored slice after slice,
//     inside a slice row after row, inside a row in ascending native column order -- ONE contiguous stream per block, 6 bytes
//     per edge (column 4, local row 2 with a "first message of the row: assign" flag);
//   * a wave reads its stream in chunks of 64 entries (coalesced), gathers x (all waves of the chip walk the slices at about the
//     same pace, so the slice everybody gathers from is L2 resident), and folds in order: lanes holding the same row form a
//     run whose head lane reads the running value from LDS, adds the run's messages one after the other and writes it back.
// Every result is compared bit for bit with a serial fold of the rows' edges in ascending native column order.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude tools/blocked_bench.hip -Lgraphmat_amd -lgraphmat_hip -o build/blocked_bench
//   LD_LIBRARY_PATH=graphmat_amd build/blocked_bench [scale 26] [graph 0 = RMAT, 1 = uniform] [slices 128] [reps 5] [row_hi 64] [forms 3]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <vector>
#include "graphmat_hip.h"

#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kMaxS = 512;
static hipEvent_t ev0, ev1;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// every vertex draws 16 out-neighbours uniformly at random (the shape of the reference's test/generator.h:73-105)
__global__ void k_uniform_edges(int nv, int32_t* __restrict__ src, int32_t* __restrict__ dst) {
  const int64_t ne = 16ll * nv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
    src[i] = (int32_t)(i >> 4) + 1;
    dst[i] = (int32_t)(mix64((uint64_t)i * 0x2545F4914F6CDD1Dull + 12345) % (uint64_t)nv) + 1;
  }
}
__global__ void k_deg_in(const int32_t* __restrict__ dst, int64_t ne, uint32_t* __restrict__ deg) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&deg[dst[i] - 1], 1u);
}
__global__ void k_flag(const uint32_t* __restrict__ deg, int nv, uint32_t hi, uint32_t* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) flag[i] = (deg[i] >= 1 && deg[i] <= hi) ? 1u : 0u;
}
__global__ void k_rank(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ excl, int nv, int32_t* __restrict__ rank_of) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) rank_of[i] = flag[i] ? (int32_t)excl[i] : -1;
}
// gather weight of a column: how often the SELECTED rows read it; and the smallest column of every selected row
__global__ void k_col_weight(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t ne, const int32_t* __restrict__ rank_of,
                             unsigned long long* __restrict__ w, uint32_t* __restrict__ rowmin) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = rank_of[dst[i] - 1];
    if (r < 0) continue;
    atomicAdd(&w[src[i] - 1], 1ull);
    atomicMin(&rowmin[r], (uint32_t)(src[i] - 1));
  }
}
__global__ void k_bounds(const unsigned long long* __restrict__ pre, int nv, int S, int32_t* __restrict__ bound) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > S) return;
  if (k == 0) { bound[0] = 0; return; }
  if (k == S) { bound[S] = nv; return; }
  const unsigned long long total = pre[nv - 1], want = total / (unsigned)S * (unsigned)k;
  int lo = 0, hi = nv;
  while (lo < hi) { const int mid = (lo + hi) / 2; if (pre[mid] >= want) hi = mid; else lo = mid + 1; }
  bound[k] = lo;
}
__device__ __forceinline__ int slice_of(const int32_t* __restrict__ bound, int S, int c) {
  int lo = 0, hi = S;
  while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (bound[mid] <= c) lo = mid; else hi = mid; }
  return lo;
}
// key = block | slice(9) | local row(RBITS) | native column ; key2 = row | native column (the serial fold's order)
__global__ void k_edge_keys(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t ne, const int32_t* __restrict__ rank_of,
                            const int32_t* __restrict__ bound, int S, int rbits, int cbits, unsigned long long* __restrict__ key, unsigned long long* __restrict__ key2) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = rank_of[dst[i] - 1], c = src[i] - 1;
    if (r < 0) { key[i] = ~0ull; key2[i] = ~0ull; continue; }
    const unsigned long long blk = (unsigned)r >> rbits, local = (unsigned)r & ((1u << rbits) - 1);
    key[i] = ((((blk << 9) | (unsigned long long)slice_of(bound, S, c)) << rbits | local) << cbits) | (unsigned long long)c;
    key2[i] = ((unsigned long long)r << cbits) | (unsigned long long)c;
  }
}
// the stream: column, local row | first flag; where every block starts
__global__ void k_entries(const unsigned long long* __restrict__ key, int64_t n, int rbits, int cbits, const uint32_t* __restrict__ rowmin,
                          uint32_t* __restrict__ ecol, uint16_t* __restrict__ erow, uint32_t* __restrict__ boff) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long k = key[i];
    const uint32_t c = (uint32_t)(k & ((1ull << cbits) - 1));
    const uint32_t local = (uint32_t)(k >> cbits) & ((1u << rbits) - 1);
    const uint32_t blk = (uint32_t)(k >> (cbits + rbits + 9));
    const uint32_t r = (blk << rbits) | local;
    const bool first = c == rowmin[r] && (i == 0 || key[i - 1] != k);  // (duplicate edges: the first of them)
    const bool segstart = i == 0 || (key[i - 1] >> (cbits + rbits)) != (k >> (cbits + rbits));  // first entry of a (block, slice) segment
    ecol[i] = c;
    erow[i] = (uint16_t)(local | (first ? 0x8000u : 0u) | (segstart ? 0x4000u : 0u));
    if (i == 0 || (uint32_t)(key[i - 1] >> (cbits + rbits + 9)) != blk) boff[blk] = (uint32_t)i;
  }
}
__global__ void k_heads2(const unsigned long long* __restrict__ key2, int64_t n, int cbits, uint32_t* __restrict__ rowptr) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = (uint32_t)(key2[i] >> cbits);
    if (i == 0 || (uint32_t)(key2[i - 1] >> cbits) != r) rowptr[r] = (uint32_t)i;
  }
}
__global__ void k_serial(const unsigned long long* __restrict__ key2, const uint32_t* __restrict__ rowptr, int nrows, int cbits, const float* __restrict__ x, float* __restrict__ y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const uint32_t a = rowptr[r], e = rowptr[r + 1];
  float v = x[(uint32_t)(key2[a] & ((1ull << cbits) - 1))];
  for (uint32_t i = a + 1; i < e; i++) v += x[(uint32_t)(key2[i] & ((1ull << cbits) - 1))];
  y[r] = v;
}
__global__ void k_fill_x(float* __restrict__ x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = (float)(h >> 8) * (1.0f / 16777216.0f) + 1e-3f; }
}
__global__ void k_diff(const float* __restrict__ a, const float* __restrict__ b, int n, unsigned long long* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && __float_as_uint(a[i]) != __float_as_uint(b[i])) atomicAdd(cnt, 1ull);
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// One wave per block of RB rows, persistent; U chunks of 64 entries per step, the next step's entries requested before this
// step's messages are waited for.  MODE 0: as described; 1: no gathers (the messages are the column ids: the stream's floor);
// 2: gathers but no fold (what the fold costs).
template <int RB, int U, int MODE>
__global__ void __launch_bounds__(1024) k_blocked(const uint32_t* __restrict__ ecol, const uint16_t* __restrict__ erow, const uint32_t* __restrict__ boff, int nblk,
                                                  const float* __restrict__ x, float* __restrict__ y, int nrows, int wg_threads) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = wg_threads >> 6;
  float* acc = lds + wave * RB;
  for (int blk = blockIdx.x * nw + wave; blk < nblk; blk += gridDim.x * nw) {
    const uint32_t a = boff[blk], e = boff[blk + 1];
    uint32_t c[U], rr[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const uint32_t p = a + u * 64 + lane; c[u] = ecol[p]; rr[u] = erow[p]; }  // (both arrays are padded behind the last entry)
    float sink = 0.f;
    for (uint32_t p0 = a; p0 < e; p0 += 64 * U) {
      float m[U];
#pragma unroll
      for (int u = 0; u < U; u++) m[u] = MODE == 1 ? __uint_as_float(c[u]) : x[c[u]];
      uint32_t r[U];
#pragma unroll
      for (int u = 0; u < U; u++) r[u] = rr[u];
      // the next step's entries
#pragma unroll
      for (int u = 0; u < U; u++) { const uint32_t p = p0 + 64 * U + u * 64 + lane; c[u] = ecol[p]; rr[u] = erow[p]; }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t p = p0 + u * 64 + lane;
        const bool valid = p < e;
        if (MODE == 2) { sink += valid ? m[u] : 0.f; continue; }
        const uint32_t id = valid ? (r[u] & 0x3fffu) : (0x10000u + lane);
        const uint32_t prev = __shfl_up(id, 1);
        const bool sstart = valid && (r[u] & 0x4000u);
        const bool head = valid && (lane == 0 || id != prev || sstart);
        const unsigned long long H = __ballot(head) | (__ballot(!valid));  // (an invalid lane ends a run as a head would)
        const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1));
        const int runlen = above ? (__builtin_ctzll(above) + 1) : (64 - lane);
        // a chunk that spans several slices may hold the same row once per slice: the slices' runs are folded one slice after the other
        const unsigned long long SS = __ballot(sstart);
        const int seg = __popcll(SS & ((lane == 63 ? 0ull : (2ull << lane)) - 1ull));
        const int nseg = __popcll(SS) + 1;
        for (int sg = 0; sg < nseg; sg++) {
          const bool h = head && seg == sg;
          float v = 0.f;
          if (h) v = (r[u] & 0x8000u) ? m[u] : acc[id] + m[u];
          for (int k = 1; __ballot(h && k < runlen); k++) {
            const float t = __shfl(m[u], (lane + k) & 63);
            if (h && k < runlen) v += t;
          }
          if (h) acc[id] = v;
        }
      }
    }
    if (MODE == 2) { if (sink == 123.456f) y[0] = sink; continue; }
    // the block's results (every selected row has at least one edge: its value was assigned)
    const int r0 = blk * RB;
    for (int i = lane; i < RB; i += 64) if (r0 + i < nrows) y[r0 + i] = acc[i];
  }
}

template <class K>
static void sort_keys(K* kin, K* kout, size_t n, int bits) {
  size_t tb = 0;
  OK(rocprim::radix_sort_keys(nullptr, tb, kin, kout, n, 0, bits, (hipStream_t)0));
  void* tmp; OK(hipMalloc(&tmp, tb + 256));
  OK(rocprim::radix_sort_keys(tmp, tb, kin, kout, n, 0, bits, (hipStream_t)0));
  OK(hipDeviceSynchronize());
  OK(hipFree(tmp));
}
template <class T>
static void scan(T* in, T* out, size_t n, bool inclusive) {
  size_t tb = 0;
  if (inclusive) OK(rocprim::inclusive_scan(nullptr, tb, in, out, n, rocprim::plus<T>(), (hipStream_t)0));
  else OK(rocprim::exclusive_scan(nullptr, tb, in, out, T(0), n, rocprim::plus<T>(), (hipStream_t)0));
  void* tmp; OK(hipMalloc(&tmp, tb + 256));
  if (inclusive) OK(rocprim::inclusive_scan(tmp, tb, in, out, n, rocprim::plus<T>(), (hipStream_t)0));
  else OK(rocprim::exclusive_scan(tmp, tb, in, out, T(0), n, rocprim::plus<T>(), (hipStream_t)0));
  OK(hipDeviceSynchronize());
  OK(hipFree(tmp));
}

struct Stream { uint32_t* ecol; uint16_t* erow; uint32_t* boff; int nblk; int64_t n; };

template <int RB, int U, int MODE>
static void run_one(const char* what, const Stream& st, const float* x, float* y, const float* yref, int nrows, int reps, int wg_threads, int wgs_per_cu) {
  const size_t lds = (size_t)(wg_threads / 64) * RB * 4;
  OK(hipFuncSetAttribute((const void*)k_blocked<RB, U, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = 256 * wgs_per_cu;
  float best = 1e9f, sum = 0.f;
  for (int it = 0; it < reps + 1; it++) {
    OK(hipEventRecord(ev0, 0));
    hipLaunchKernelGGL((k_blocked<RB, U, MODE>), dim3(grid), dim3(wg_threads), lds, 0, st.ecol, st.erow, st.boff, st.nblk, x, y, nrows, wg_threads);
    OK(hipEventRecord(ev1, 0));
    OK(hipEventSynchronize(ev1));
    float ms; OK(hipEventElapsedTime(&ms, ev0, ev1));
    if (it == 0) continue;
    best = ms < best ? ms : best; sum += ms;
  }
  OK(hipGetLastError());
  printf("%-58s RB %4d U %d, %4d threads x %d per CU: best %.3f ms, mean %.3f ms = %.2f ps per edge, %.1f G edges/s", what, RB, U, wg_threads, wgs_per_cu, best, sum / reps,
         best * 1e9 / (double)st.n, (double)st.n / best / 1e6);
  if (MODE == 0) {
    unsigned long long* d; OK(hipMalloc(&d, 8)); OK(hipMemset(d, 0, 8));
    k_diff<<<(nrows + 255) / 256, 256>>>(y, yref, nrows, d);
    unsigned long long h = 0; OK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); OK(hipFree(d));
    printf("; %llu of %d rows differ from the serial fold", h, nrows);
  }
  printf("\n");
  fflush(stdout);
}

template <int RB>
static void build_and_run(const int32_t* src, const int32_t* dst, int64_t ne, int nv, int scale, const int32_t* rank_of, int nrows, const int32_t* bound, int S,
                          const uint32_t* rowmin, const float* x, int reps, unsigned long long* k1, unsigned long long* k1s, unsigned long long* k2s, int64_t nsel,
                          const float* yref, float* y) {
  int rbits = 0; while ((1 << rbits) < RB) rbits++;
  const int cbits = scale, G = 4096;
  k_edge_keys<<<G, 256>>>(src, dst, ne, rank_of, bound, S, rbits, cbits, k1, k2s);  // (k2s is scratch here)
  sort_keys(k1, k1s, (size_t)ne, 64);
  const int nblk = (nrows + RB - 1) / RB;
  Stream st;
  st.nblk = nblk; st.n = nsel;
  OK(hipMalloc(&st.ecol, ((size_t)nsel + 64 * 64) * 4)); OK(hipMalloc(&st.erow, ((size_t)nsel + 64 * 64) * 2)); OK(hipMalloc(&st.boff, ((size_t)nblk + 2) * 4));
  OK(hipMemset(st.ecol, 0, ((size_t)nsel + 64 * 64) * 4)); OK(hipMemset(st.erow, 0, ((size_t)nsel + 64 * 64) * 2)); OK(hipMemset(st.boff, 0xff, ((size_t)nblk + 2) * 4));
  k_entries<<<G, 256>>>(k1s, nsel, rbits, cbits, rowmin, st.ecol, st.erow, st.boff);
  OK(hipDeviceSynchronize());
  {
    std::vector<uint32_t> h((size_t)nblk + 1);
    OK(hipMemcpy(h.data(), st.boff, (size_t)nblk * 4, hipMemcpyDeviceToHost));
    h[nblk] = (uint32_t)nsel;
    for (int b = nblk - 1; b >= 0; b--) if (h[b] == 0xffffffffu) h[b] = h[b + 1];
    OK(hipMemcpy(st.boff, h.data(), ((size_t)nblk + 1) * 4, hipMemcpyHostToDevice));
  }
  printf("blocks of %d rows: %d blocks, %.1f entries per block and slice\n", RB, nblk, (double)nsel / nblk / S);
  const int wpc = RB >= 2048 ? 1 : RB >= 1024 ? 2 : 4;  // workgroups of 1024 threads per CU that the LDS allows (<= 32 waves per CU)
  run_one<RB, 4, 0>("blocked stream", st, x, y, yref, nrows, reps, 1024, wpc > 2 ? 2 : wpc);
  run_one<RB, 2, 0>("blocked stream", st, x, y, yref, nrows, reps, 1024, wpc > 2 ? 2 : wpc);
  run_one<RB, 8, 0>("blocked stream", st, x, y, yref, nrows, reps, 1024, wpc > 2 ? 2 : wpc);
  if (wpc >= 2) run_one<RB, 4, 0>("blocked stream, one workgroup per CU", st, x, y, yref, nrows, reps, 1024, 1);
  run_one<RB, 4, 1>("  ... no gathers (the stream's floor)", st, x, y, yref, nrows, reps, 1024, wpc > 2 ? 2 : wpc);
  run_one<RB, 4, 2>("  ... gathers, no fold", st, x, y, yref, nrows, reps, 1024, wpc > 2 ? 2 : wpc);
  OK(hipFree(st.ecol)); OK(hipFree(st.erow)); OK(hipFree(st.boff));
}

// ---- the workgroup-stationary form, workgroups kept in step ----------------------------------------------------------------
// A workgroup owns kRBW rows per pass (their running values fill its LDS); the grid's workgroups take the blocks of a pass side by
// side and walk the slices TOGETHER: a workgroup that has finished slice s tells its XCD's counter so, and nobody starts slice
// s + 1 before every workgroup of the XCD has finished slice s + 1 - window -- the L2 then holds `window` + 1 slices of x, whatever the
// workgroups' pace.  The wait is bounded: a workgroup that never arrives costs locality, not progress.
#ifndef BB_WAVES
#define BB_WAVES 16  // waves per workgroup (-DBB_WAVES=8: two workgroups of 512 threads and 16384 rows per CU, out of phase with each other)
#endif
constexpr int kWV = BB_WAVES, kRBW = kWV * 2048, kRBWBits = kWV == 16 ? 15 : 14, kWGT = kWV * 64, kGrid = 256 * (16 / kWV);
__global__ void k_entries_wg(const unsigned long long* __restrict__ key, int64_t n, int cbits, int S, const uint32_t* __restrict__ rowmin,
                             uint32_t* __restrict__ ecol, uint16_t* __restrict__ erow, uint32_t* __restrict__ toff) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long k = key[i];
    const uint32_t c = (uint32_t)(k & ((1ull << cbits) - 1));
    const uint32_t local = (uint32_t)(k >> cbits) & (kRBW - 1);
    const uint32_t slice = (uint32_t)(k >> (cbits + kRBWBits)) & 511u;
    const uint32_t blk = (uint32_t)(k >> (cbits + kRBWBits + 9));
    const uint32_t r = (blk << kRBWBits) | local;
    const bool first = c == rowmin[r] && (i == 0 || key[i - 1] != k);
    ecol[i] = c;
    erow[i] = (uint16_t)(local | (first ? 0x8000u : 0u));
    if (i == 0 || (key[i - 1] >> (cbits + kRBWBits)) != (k >> (cbits + kRBWBits))) toff[(size_t)blk * S + slice] = (uint32_t)i;
  }
}
// where the 16 waves of a workgroup start inside a (block, slice) segment: equal shares, moved forward to the next row border
__global__ void k_wave_offsets(const uint32_t* __restrict__ toff, const uint16_t* __restrict__ erow, size_t nseg, uint32_t* __restrict__ woff) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= nseg * (kWV + 1)) return;
  const size_t seg = t / (kWV + 1);
  const int w = (int)(t % (kWV + 1));
  const uint32_t a = toff[seg], e = toff[seg + 1];
  uint32_t p = a + (uint32_t)(((unsigned long long)(e - a) * (unsigned)w) / (unsigned)kWV);
  if (w == kWV) p = e;
  while (p > a && p < e && (erow[p] & 0x7fffu) == (erow[p - 1] & 0x7fffu)) p++;
  woff[t] = p;
}

template <int U, int NT>
__global__ void __launch_bounds__(1024) k_blocked_wg(const uint32_t* __restrict__ ecol, const uint16_t* __restrict__ erow, const uint32_t* __restrict__ woff, int S, int nblk,
                                                     const float* __restrict__ x, float* __restrict__ y, int nrows, unsigned int* __restrict__ cnt, int window,
                                                     unsigned int epoch) {
  extern __shared__ float acc[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7;
  const unsigned int nwg = gridDim.x >> 3;
  const int npass = (nblk + gridDim.x - 1) / gridDim.x;
  const int nsteps = npass * S;
  unsigned int* mycnt = cnt + (size_t)xcd * nsteps;
  for (int pass = 0; pass < npass; pass++) {
    const int blk = pass * gridDim.x + blockIdx.x;
    const bool has = blk < nblk;
    uint32_t ws = 0, we = 0;
    if (has) { const uint32_t* wo = woff + ((size_t)blk * S) * (kWV + 1) + wave; ws = wo[0]; we = wo[1]; }
    for (int s = 0; s < S; s++) {
      uint32_t nws = 0, nwe = 0;
      if (has && s + 1 < S) { const uint32_t* wo = woff + ((size_t)blk * S + s + 1) * (kWV + 1) + wave; nws = wo[0]; nwe = wo[1]; }  // (the next slice's range: requested now)
      for (uint32_t p0 = ws; p0 < we; p0 += 64 * U) {
        uint32_t c[U], r[U];
        float m[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint32_t p = p0 + u * 64 + lane;
          if (NT) { c[u] = __builtin_nontemporal_load(ecol + p); r[u] = __builtin_nontemporal_load(erow + p); }
          else { c[u] = ecol[p]; r[u] = erow[p]; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) m[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint32_t p = p0 + u * 64 + lane;
          const bool valid = p < we;
          const uint32_t id = valid ? (r[u] & 0x7fffu) : (0x10000u + lane);
          const uint32_t prev = __shfl_up(id, 1);
          const bool head = valid && (lane == 0 || id != prev);
          const unsigned long long H = __ballot(head) | (__ballot(!valid));
          const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1));
          const int runlen = above ? (__builtin_ctzll(above) + 1) : (64 - lane);
          float v = 0.f;
          if (head) v = (r[u] & 0x8000u) ? m[u] : acc[id] + m[u];
          for (int k = 1; __ballot(head && k < runlen); k++) {
            const float t = __shfl(m[u], (lane + k) & 63);
            if (head && k < runlen) v += t;
          }
          if (head) acc[id] = v;
        }
      }
      __syncthreads();
      const int step = pass * S + s;
      if (window >= 0) {
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&mycnt[step], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int need = step + 1 - window;
        if (need >= 0) {
          const unsigned int target = nwg * (epoch + 1);
          for (int tries = 0; tries < 20000; tries++)
            if (__hip_atomic_load(&mycnt[need], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
        }
      }
      ws = nws; we = nwe;
    }
    if (has) {
      const int r0 = blk * kRBW;
      for (int i = threadIdx.x; i < kRBW; i += kWGT) if (r0 + i < nrows) y[r0 + i] = acc[i];
    }
    __syncthreads();
  }
}

// ... and without a workgroup barrier: wave w of the workgroup owns the block's rows [w * 2048, (w + 1) * 2048) in EVERY slice (its part of a
// segment is found by row, not by share), so no two waves ever touch the same running value; the entries of the wave's next batch -- in this
// slice or the next -- are requested before the current batch is folded; wave 0 reports the slice done, every wave paces itself.
__global__ void k_wave_offsets_by_row(const uint32_t* __restrict__ toff, const uint16_t* __restrict__ erow, size_t nseg, uint32_t* __restrict__ woff) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= nseg * (kWV + 1)) return;
  const size_t seg = t / (kWV + 1);
  const uint32_t w = (uint32_t)(t % (kWV + 1));
  const uint32_t a = toff[seg], e = toff[seg + 1];
  uint32_t lo = a, hi = e;  // first entry whose local row is >= w * 2048
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if ((uint32_t)(erow[mid] & 0x7fffu) >= w * 2048u) hi = mid; else lo = mid + 1; }
  woff[t] = lo;
}

template <int U, bool BARRIER>
__global__ void __launch_bounds__(1024) k_blocked_wg2(const uint32_t* __restrict__ ecol, const uint16_t* __restrict__ erow, const uint32_t* __restrict__ woff, int S, int nblk,
                                                      const float* __restrict__ x, float* __restrict__ y, int nrows, unsigned int* __restrict__ cnt, int window,
                                                      unsigned int epoch) {
  extern __shared__ float acc[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7;
  const unsigned int nwg = gridDim.x >> 3;
  const int npass = (nblk + gridDim.x - 1) / gridDim.x;
  const int nsteps = npass * S;
  unsigned int* mycnt = cnt + (size_t)xcd * nsteps;
  for (int pass = 0; pass < npass; pass++) {
    const int blk = pass * gridDim.x + blockIdx.x;
    const bool has = blk < nblk;
    uint32_t ws = 0, we = 0;
    if (has) { const uint32_t* wo = woff + ((size_t)blk * S) * (kWV + 1) + wave; ws = wo[0]; we = wo[1]; }
    uint32_t c[U], r[U];
    uint32_t pre = ws;
#pragma unroll
    for (int u = 0; u < U; u++) { c[u] = ecol[pre + u * 64 + lane]; r[u] = erow[pre + u * 64 + lane]; }
    for (int s = 0; s < S; s++) {
      uint32_t nws = 0, nwe = 0;
      if (has && s + 1 < S) { const uint32_t* wo = woff + ((size_t)blk * S + s + 1) * (kWV + 1) + wave; nws = wo[0]; nwe = wo[1]; }
      if (pre != ws && ws < we) {
        pre = ws;
#pragma unroll
        for (int u = 0; u < U; u++) { c[u] = ecol[pre + u * 64 + lane]; r[u] = erow[pre + u * 64 + lane]; }
      }
      for (uint32_t p0 = ws; p0 < we;) {
        float m[U];
#pragma unroll
        for (int u = 0; u < U; u++) m[u] = x[c[u]];
        const uint32_t np0 = p0 + 64 * U;
        const uint32_t nxt = np0 < we ? np0 : nws;
        uint32_t nc[U], nr[U];
#pragma unroll
        for (int u = 0; u < U; u++) { nc[u] = ecol[nxt + u * 64 + lane]; nr[u] = erow[nxt + u * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint32_t p = p0 + u * 64 + lane;
          const bool valid = p < we;
          const uint32_t id = valid ? (r[u] & 0x7fffu) : (0x10000u + lane);
          const uint32_t prev = __shfl_up(id, 1);
          const bool head = valid && (lane == 0 || id != prev);
          const unsigned long long H = __ballot(head) | (__ballot(!valid));
          const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1));
          const int runlen = above ? (__builtin_ctzll(above) + 1) : (64 - lane);
          float v = 0.f;
          if (head) v = (r[u] & 0x8000u) ? m[u] : acc[id] + m[u];
          for (int k = 1; __ballot(head && k < runlen); k++) {
            const float t = __shfl(m[u], (lane + k) & 63);
            if (head && k < runlen) v += t;
          }
          if (head) acc[id] = v;
        }
#pragma unroll
        for (int u = 0; u < U; u++) { c[u] = nc[u]; r[u] = nr[u]; }
        pre = nxt;
        p0 = np0;
      }
      const int step = pass * S + s;
      if (BARRIER) __syncthreads();
      if (window >= 0) {
        // (BARRIER: ONE lane of the workgroup reports and waits, with a pause between two looks -- 512 waves of an XCD polling one line
        // of the L2 keep the reports themselves from getting through -- and a second barrier hands the result to the other waves)
        const int need = step + 1 - window;
        const unsigned int target = nwg * (epoch + 1);
        if (BARRIER) {
          if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&mycnt[step], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (need >= 0)
              for (int tries = 0; tries < 20000; tries++) {
                if (__hip_atomic_load(&mycnt[need], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                __builtin_amdgcn_s_sleep(4);
              }
          }
          __syncthreads();
        } else {
          if (threadIdx.x == 0) __hip_atomic_fetch_add(&mycnt[step], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (need >= 0)
            for (int tries = 0; tries < 20000; tries++)
              if (__hip_atomic_load(&mycnt[need], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
        }
      }
      ws = nws; we = nwe;
    }
    if (has) {
      const int r0 = blk * kRBW + wave * 2048;
      for (int i = lane; i < 2048; i += 64) if (r0 + i < nrows) y[r0 + i] = acc[wave * 2048 + i];
    }
    if (BARRIER) __syncthreads();
  }
}

template <int U, bool BARRIER>
static void run_wg2(const char* what, const uint32_t* ecol, const uint16_t* erow, const uint32_t* woff, int S, int nblk, int64_t n, const float* x, float* y, const float* yref,
                    int nrows, int reps, unsigned int* cnt, size_t cnt_words, int window) {
  const size_t lds = (size_t)kRBW * 4;
  OK(hipFuncSetAttribute((const void*)k_blocked_wg2<U, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  OK(hipMemset(cnt, 0, cnt_words * 4));
  OK(hipMemset(y, 0, (size_t)nrows * 4));
  float best = 1e9f, sum = 0.f;
  for (int it = 0; it < reps + 1; it++) {
    OK(hipEventRecord(ev0, 0));
    hipLaunchKernelGGL((k_blocked_wg2<U, BARRIER>), dim3(kGrid), dim3(kWGT), lds, 0, ecol, erow, woff, S, nblk, x, y, nrows, cnt, window, (unsigned int)it);
    OK(hipEventRecord(ev1, 0));
    OK(hipEventSynchronize(ev1));
    float ms; OK(hipEventElapsedTime(&ms, ev0, ev1));
    if (it == 0) continue;
    best = ms < best ? ms : best; sum += ms;
  }
  OK(hipGetLastError());
  unsigned long long* d; OK(hipMalloc(&d, 8)); OK(hipMemset(d, 0, 8));
  k_diff<<<(nrows + 255) / 256, 256>>>(y, yref, nrows, d);
  unsigned long long h = 0; OK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); OK(hipFree(d));
  printf("%-44s U %d, window %2d: best %.3f ms, mean %.3f ms = %.2f ps per edge, %.1f G edges/s; %llu of %d rows differ from the serial fold\n", what, U, window, best, sum / reps,
         best * 1e9 / (double)n, (double)n / best / 1e6, h, nrows);
  fflush(stdout);
}

template <int U, int NT>
static void run_wg(const char* what, const uint32_t* ecol, const uint16_t* erow, const uint32_t* woff, int S, int nblk, int64_t n, const float* x, float* y, const float* yref,
                   int nrows, int reps, unsigned int* cnt, size_t cnt_words, int window) {
  const size_t lds = (size_t)kRBW * 4;
  OK(hipFuncSetAttribute((const void*)k_blocked_wg<U, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  OK(hipMemset(cnt, 0, cnt_words * 4));
  OK(hipMemset(y, 0, (size_t)nrows * 4));
  float best = 1e9f, sum = 0.f;
  for (int it = 0; it < reps + 1; it++) {
    OK(hipEventRecord(ev0, 0));
    hipLaunchKernelGGL((k_blocked_wg<U, NT>), dim3(kGrid), dim3(kWGT), lds, 0, ecol, erow, woff, S, nblk, x, y, nrows, cnt, window, (unsigned int)it);
    OK(hipEventRecord(ev1, 0));
    OK(hipEventSynchronize(ev1));
    float ms; OK(hipEventElapsedTime(&ms, ev0, ev1));
    if (it == 0) continue;
    best = ms < best ? ms : best; sum += ms;
  }
  OK(hipGetLastError());
  unsigned long long* d; OK(hipMalloc(&d, 8)); OK(hipMemset(d, 0, 8));
  k_diff<<<(nrows + 255) / 256, 256>>>(y, yref, nrows, d);
  unsigned long long h = 0; OK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); OK(hipFree(d));
  printf("%-44s U %d%s, window %2d: best %.3f ms, mean %.3f ms = %.2f ps per edge, %.1f G edges/s; %llu of %d rows differ from the serial fold\n", what, U, NT ? ", stream non-temporal" : "",
         window, best, sum / reps, best * 1e9 / (double)n, (double)n / best / 1e6, h, nrows);
  fflush(stdout);
}

static void build_and_run_wg(const int32_t* src, const int32_t* dst, int64_t ne, int scale, const int32_t* rank_of, int nrows, const int32_t* bound, int S, const uint32_t* rowmin,
                             const float* x, int reps, unsigned long long* k1, unsigned long long* k1s, unsigned long long* scratch, int64_t nsel, const float* yref, float* y) {
  const int cbits = scale, G = 4096;
  k_edge_keys<<<G, 256>>>(src, dst, ne, rank_of, bound, S, kRBWBits, cbits, k1, scratch);
  sort_keys(k1, k1s, (size_t)ne, 64);
  const int nblk = (nrows + kRBW - 1) / kRBW;
  const size_t nseg = (size_t)nblk * S;
  uint32_t *ecol, *toff, *woff; uint16_t* erow;
  OK(hipMalloc(&ecol, ((size_t)nsel + 64 * 64) * 4)); OK(hipMalloc(&erow, ((size_t)nsel + 64 * 64) * 2)); OK(hipMalloc(&toff, (nseg + 2) * 4)); OK(hipMalloc(&woff, (nseg + 1) * (kWV + 1) * 4));
  OK(hipMemset(ecol, 0, ((size_t)nsel + 64 * 64) * 4)); OK(hipMemset(erow, 0, ((size_t)nsel + 64 * 64) * 2)); OK(hipMemset(toff, 0xff, (nseg + 2) * 4));
  k_entries_wg<<<G, 256>>>(k1s, nsel, cbits, S, rowmin, ecol, erow, toff);
  OK(hipDeviceSynchronize());
  {
    std::vector<uint32_t> h(nseg + 1);
    OK(hipMemcpy(h.data(), toff, nseg * 4, hipMemcpyDeviceToHost));
    h[nseg] = (uint32_t)nsel;
    for (size_t b = nseg; b-- > 0;) if (h[b] == 0xffffffffu) h[b] = h[b + 1];
    OK(hipMemcpy(toff, h.data(), (nseg + 1) * 4, hipMemcpyHostToDevice));
  }
  k_wave_offsets<<<(unsigned)((nseg * (kWV + 1) + 255) / 256), 256>>>(toff, erow, nseg, woff);
  OK(hipDeviceSynchronize());
  const int npass = (nblk + kGrid - 1) / kGrid;
  const size_t cnt_words = (size_t)8 * npass * S + 64;
  unsigned int* cnt; OK(hipMalloc(&cnt, cnt_words * 4));
  printf("workgroup-stationary form: blocks of %d rows: %d blocks = %d passes of %d workgroups, %.0f entries per block and slice (%.1f us of L2-hit gathers per pass and slice chip-wide)\n", kRBW, nblk,
         npass, kGrid, (double)nsel / nblk / S, (double)nsel / npass / S / 200e3);
  run_wg<4, 0>("workgroups in step", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 1);
  run_wg<4, 0>("workgroups in step", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg<4, 0>("workgroups in step", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 4);
  run_wg<4, 1>("workgroups in step", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg<2, 0>("workgroups in step", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg<4, 0>("workgroups not kept in step", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, -1);
  run_wg2<4, true>("in step, next batch prefetched", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 1);
  run_wg2<4, true>("in step, next batch prefetched", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg2<4, true>("in step, next batch prefetched", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 3);
  run_wg2<2, true>("in step, next batch prefetched", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg2<8, true>("in step, next batch prefetched", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg2<1, true>("in step, next batch prefetched", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg2<3, true>("in step, next batch prefetched", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  if (getenv("BLOCKED_BENCH_ALL") == nullptr) { OK(hipFree(ecol)); OK(hipFree(erow)); OK(hipFree(toff)); OK(hipFree(woff)); OK(hipFree(cnt)); return; }
  k_wave_offsets_by_row<<<(unsigned)((nseg * (kWV + 1) + 255) / 256), 256>>>(toff, erow, nseg, woff);
  OK(hipDeviceSynchronize());
  run_wg2<4, false>("waves own rows, paced, no barrier", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 1);
  run_wg2<4, false>("waves own rows, paced, no barrier", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg2<4, false>("waves own rows, paced, no barrier", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 3);
  run_wg2<2, false>("waves own rows, paced, no barrier", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg2<8, false>("waves own rows, paced, no barrier", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, 2);
  run_wg2<4, false>("waves own rows, not paced", ecol, erow, woff, S, nblk, nsel, x, y, yref, nrows, reps, cnt, cnt_words, -1);
  OK(hipFree(ecol)); OK(hipFree(erow)); OK(hipFree(toff)); OK(hipFree(woff)); OK(hipFree(cnt));
}

int main(int argc, char** argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 26;
  const int kind = argc > 2 ? atoi(argv[2]) : 0;
  const int S = argc > 3 ? atoi(argv[3]) : 128;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  const int row_hi = argc > 5 ? atoi(argv[5]) : 64;
  const int form = argc > 6 ? atoi(argv[6]) : 3;  // 1 = wave-stationary blocks, 2 = workgroup-stationary blocks kept in step, 3 = both
  if (S < 1 || S > kMaxS) { printf("slices: 1..%d\n", kMaxS); return 1; }
  const int nv = 1 << scale;
  const int64_t ne = 16ll * nv;
  const int G = 4096;
  OK(hipEventCreate(&ev0)); OK(hipEventCreate(&ev1));
  int32_t *src, *dst;
  OK(hipMalloc(&src, ne * 4)); OK(hipMalloc(&dst, ne * 4));
  if (kind == 0) {
    if (gm_rmat_generate(scale, 1, 0, ne, src, dst, nullptr, 0, nullptr) != 0) { printf("gm_rmat_generate: %s\n", gm_last_error()); return 1; }
  } else {
    k_uniform_edges<<<G, 256>>>(nv, src, dst);
  }
  OK(hipDeviceSynchronize());
  uint32_t *deg, *flag, *excl;
  OK(hipMalloc(&deg, (size_t)nv * 4)); OK(hipMalloc(&flag, (size_t)nv * 4)); OK(hipMalloc(&excl, (size_t)nv * 4));
  OK(hipMemset(deg, 0, (size_t)nv * 4));
  k_deg_in<<<G, 256>>>(dst, ne, deg);
  k_flag<<<(nv + 255) / 256, 256>>>(deg, nv, (uint32_t)row_hi, flag);
  scan(flag, excl, (size_t)nv, false);
  int32_t* rank_of; OK(hipMalloc(&rank_of, (size_t)nv * 4));
  k_rank<<<(nv + 255) / 256, 256>>>(flag, excl, nv, rank_of);
  uint32_t lastf = 0, laste = 0;
  OK(hipMemcpy(&lastf, flag + (nv - 1), 4, hipMemcpyDeviceToHost)); OK(hipMemcpy(&laste, excl + (nv - 1), 4, hipMemcpyDeviceToHost));
  const int nrows = (int)(lastf + laste);
  OK(hipFree(flag)); OK(hipFree(excl));
  unsigned long long *w, *pre;
  uint32_t* rowmin;
  OK(hipMalloc(&w, (size_t)nv * 8)); OK(hipMalloc(&pre, (size_t)nv * 8)); OK(hipMalloc(&rowmin, ((size_t)nrows + 1) * 4));
  OK(hipMemset(w, 0, (size_t)nv * 8)); OK(hipMemset(rowmin, 0xff, ((size_t)nrows + 1) * 4));
  k_col_weight<<<G, 256>>>(src, dst, ne, rank_of, w, rowmin);
  scan(w, pre, (size_t)nv, true);
  unsigned long long nsel_u = 0;
  OK(hipMemcpy(&nsel_u, pre + (nv - 1), 8, hipMemcpyDeviceToHost));
  const int64_t nsel = (int64_t)nsel_u;
  int32_t* bound; OK(hipMalloc(&bound, (kMaxS + 2) * 4));
  k_bounds<<<(S + 256) / 256, 256>>>(pre, nv, S, bound);
  OK(hipDeviceSynchronize());
  OK(hipFree(w)); OK(hipFree(pre));
  printf("%s scale %d: %d rows of 1 .. %d edges, %lld edges (%.1f %% of all), %d column slices of equal gather weight (%.2f MiB of x each on average)\n",
         kind == 0 ? "RMAT" : "uniform", scale, nrows, row_hi, (long long)nsel, 100.0 * nsel / ne, S, (double)nv * 4 / S / 1048576.0);
  if (nsel >= (1ll << 32) - 64 * 64 * 2) { printf("too many entries for 32-bit positions\n"); return 1; }
  float *x, *y, *yref;
  OK(hipMalloc(&x, (size_t)nv * 4)); OK(hipMalloc(&y, ((size_t)nrows + 1) * 4)); OK(hipMalloc(&yref, ((size_t)nrows + 1) * 4));
  k_fill_x<<<(nv + 255) / 256, 256>>>(x, nv);
  unsigned long long *k1, *k1s, *k2, *k2s;
  OK(hipMalloc(&k1, ne * 8)); OK(hipMalloc(&k1s, ne * 8)); OK(hipMalloc(&k2, ne * 8)); OK(hipMalloc(&k2s, ne * 8));
  // the serial fold
  {
    k_edge_keys<<<G, 256>>>(src, dst, ne, rank_of, bound, S, 11, scale, k1, k2);
    sort_keys(k2, k2s, (size_t)ne, 64);
    uint32_t* rowptr; OK(hipMalloc(&rowptr, ((size_t)nrows + 1) * 4));
    k_heads2<<<G, 256>>>(k2s, nsel, scale, rowptr);
    const uint32_t n32 = (uint32_t)nsel;
    OK(hipMemcpy(rowptr + nrows, &n32, 4, hipMemcpyHostToDevice));
    k_serial<<<(nrows + 255) / 256, 256>>>(k2s, rowptr, nrows, scale, x, yref);
    OK(hipDeviceSynchronize());
    OK(hipFree(rowptr));
  }
  if (form & 2) build_and_run_wg(src, dst, ne, scale, rank_of, nrows, bound, S, rowmin, x, reps, k1, k1s, k2, nsel, yref, y);
  if (!(form & 1)) return 0;
  build_and_run<2048>(src, dst, ne, nv, scale, rank_of, nrows, bound, S, rowmin, x, reps, k1, k1s, k2, nsel, yref, y);
  build_and_run<1024>(src, dst, ne, nv, scale, rank_of, nrows, bound, S, rowmin, x, reps, k1, k1s, k2, nsel, yref, y);
  build_and_run<512>(src, dst, ne, nv, scale, rank_of, nrows, bound, S, rowmin, x, reps, k1, k1s, k2, nsel, yref, y);
  return 0;
}
