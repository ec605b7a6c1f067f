// gm_graph.hip -- device-side construction of the adjacency (CSR by destination
// and CSR by source), replacing the reference's host pipeline
//   Graph::ReadEdgelist -> SpMat ctor (edge shuffle) -> DCSCTile ctor
//   (partition, __gnu_parallel::sort, column compaction) -> Transpose
// (include/Graph.h:210-246, include/GMDP/matrices/DCSCTile.h:241-381,
//  include/GMDP/matrices/SpMat.h:422-443).
//
// What must be preserved from the reference is the ORDER in which a row's
// messages are reduced: ascending native column id, duplicates adjacent
// (DCSCTile.h:41-58 sort key).  A stable 64-bit radix sort on (row << 32 | col)
// gives exactly that; everything else (row partitions per OpenMP thread, column
// compaction) is CPU-cache machinery that has no analogue here.
#include <string.h>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "gm_internal.hpp"

namespace gm {

int g_short_row = GM_SHORT_ROW;   // tunable through gm_set_option (experiments); defaults are the documented ones
int g_giant_row = 0;  // 0 = choose per graph (see pick_giant_threshold)
int g_rank_cap = 0;   // experiment: > 0 ranks only vertices of total degree >= cap; the others keep native order behind them
int g_rank_by = 0;    // experiment: 0 rank vertices by total degree, 1 by out-degree, 2 by in-degree
int g_tile_min_row = GM_TILE_MIN_ROW;  // rows of more than this many edges are tiled
int g_tile_balance = 1;  // column tiles serve equally many gathers (1) or hold equally many vertices with edges (0)
int g_long_mid = 0;  // > 0: experiment -- wave rows of more than this many edges get a wave each (0 = GM_LONG_MID rule)
int g_sweep_slices = 1;  // 1: cut the device order into ntiles * k <= 64 slices and build the row-stationary sweep of the medium rows (gm_graph_sweep)
int g_sort_tile_lists = 1;  // column tiles: the wave-row lists by descending piece length (sort_rows_by_length)
int g_sweep_border_factor = 4;  // the medium / long border is lowered while it exceeds this many times a wave's share of a block per slice
// cost of a stream row in the waves' shares of a block, percent of a medium row's.  A stream row is gathered like a medium row AND its products are stored, and a
// slice ends at a workgroup barrier: with the rows counted alike (100) the waves that hold the stream groups -- the block's last -- were what every slice waited
// for.  RMAT-26, ms per iteration at 100 / 150 / 200 / 250 / 300 / 350 / 400 / 500 / 700: 3.555 / 3.412 / 3.312 / 3.299 / 3.277 / 3.318 / 3.335 / 3.425 / 3.662;
// RMAT-27 at 100 / 200 / 300 / 500: 7.63 / 7.38 / 7.34 / 7.87; RMAT-25 at 100 / 300 / 500: 1.89 / 1.76 / 1.89 (gm_set_option("sweep_stream_weight"))
int g_sweep_stream_weight = 300;
int g_sweep_stream = 1;  // the short rows ride the sweep as stream groups (gm_sweep_t.nstream; round 6, last session); 0: none built
int g_sweep_waves = 16;  // waves per workgroup the sweep's blocks are dealt over (16; 12: a 768-thread sweep that leaves room on every CU for the short rows' kernel beside it -- experiment of round 6)
int g_sweep_fold_share = 50;  // share (percent of an equal share) of a block's groups that the waves folding the long rows get
int g_sweep_long_row = 0;  // the sweep's medium / long border (edges per row): 0 = chosen per graph (build_sweep)
int g_sweep_acc_limit = GM_SWEEP_ACC_ROWS, g_sweep_long_limit = GM_SWEEP_LONG_SLOTS;  // rows per workgroup and launch of the sweep (tests force several launches with small values)
int g_own_wave_row = 4096;  // column tiles: rows of more than this many edges (whole graph) keep the wave / giant kernels in every tile (0: classes per tile piece)
int g_col_tiles = 0;  // default number of column tiles for graphs whose descriptor says 0 (0 = environment GRAPHMAT_COL_TILES, else none)

constexpr int kT = 256;
inline int grid_for(int64_t n) { return (int)((n + kT - 1) / kT); }

// edge ids outside [lo, hi] (the reference only asserts this in __DEBUG builds, edgelist.h; here an
// out-of-range id would index device arrays out of bounds, so it is checked before anything else)
__global__ void __launch_bounds__(kT)
k_validate_ids(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t nnz, int lo, int hi,
               unsigned long long* __restrict__ bad /* [0] count, [1] first offending edge + 1 */) {
  const int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  bool b = false;
  if (e < nnz) {
    const int s = src[e], d = dst[e];
    b = s < lo || s > hi || d < lo || d > hi;
  }
  const unsigned long long m = __ballot(b);
  if (m && (threadIdx.x & 63) == __ffsll((long long)m) - 1) {
    atomicAdd(&bad[0], (unsigned long long)__popcll(m));
    atomicMin(&bad[1], (unsigned long long)e + 1ull);
  }
}

// total degree (in + out) per native vertex, for the GM_LAYOUT_DEGREE ranking
__global__ void __launch_bounds__(kT)
k_degree(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t nnz, int nparts, int nv,
         int ids_are_native, uint32_t* __restrict__ deg, int rank_by) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= nnz) return;
  int s = src[e], d = dst[e];
  int sn = ids_are_native ? s : to_native0(s, nparts, nv);
  int dn = ids_are_native ? d : to_native0(d, nparts, nv);
  // total degree decides who is "live"; the ranking experiments weight one side
  atomicAdd(&deg[sn], rank_by == 1 ? 65537u : rank_by == 2 ? 1u : 1u);
  atomicAdd(&deg[dn], rank_by == 2 ? 65537u : 1u);
}

__global__ void __launch_bounds__(kT)
k_count_nonzero(const uint32_t* __restrict__ deg, int nv, unsigned long long* __restrict__ out) {
  int v = blockIdx.x * kT + threadIdx.x;
  unsigned long long m = __ballot(v < nv && deg[v] != 0u);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

// out[0] += the degrees of all vertices, out[1] += the degrees of the vertices above `thr` (how much of the graph sits on busy vertices)
__global__ void __launch_bounds__(kT)
k_degree_mass(const uint32_t* __restrict__ deg, int nv, uint32_t thr, unsigned long long* __restrict__ out) {
  unsigned long long all = 0, heavy = 0;
  for (int v = blockIdx.x * kT + threadIdx.x; v < nv; v += gridDim.x * kT) { const uint32_t d = deg[v]; all += d; if (d > thr) heavy += d; }
  for (int o = 32; o > 0; o >>= 1) { all += __shfl_down(all, o); heavy += __shfl_down(heavy, o); }
  if ((threadIdx.x & 63) == 0) { if (all) atomicAdd(&out[0], all); if (heavy) atomicAdd(&out[1], heavy); }
}

__global__ void __launch_bounds__(kT)
k_rank_keys(const uint32_t* __restrict__ deg, int nv, uint32_t* __restrict__ keys, int32_t* __restrict__ ids, uint32_t cap) {
  int v = blockIdx.x * kT + threadIdx.x;
  if (v >= nv) return;
  uint32_t d = deg[v];
  if (cap > 0 && d > 0 && d < cap) d = 1;
  keys[v] = 0xffffffffu - d;  // ascending sort => descending degree; stable => ties by native id
  ids[v] = v;
}

__global__ void __launch_bounds__(kT)
k_live_flags(const uint32_t* __restrict__ deg, int nv, uint32_t* __restrict__ live) {
  int v = blockIdx.x * kT + threadIdx.x;
  if (v < nv) live[v] = deg[v] ? 1u : 0u;
}
// how often every vertex is a column of the GM_DIR_OUT adjacency (= a source): the gathers its x entry will serve
__global__ void __launch_bounds__(kT)
k_col_weight(const int32_t* __restrict__ src, int64_t nnz, int nparts, int nv, int ids_are_native, uint32_t* __restrict__ w) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= nnz) return;
  const int s = src[e];
  atomicAdd(&w[ids_are_native ? s : to_native0(s, nparts, nv)], 1u);
}
// final weight of a vertex from its column count: count^p (p = pw100 / 100) plus c for every vertex with edges
__global__ void __launch_bounds__(kT)
k_tile_weight(const uint32_t* __restrict__ deg, const uint32_t* __restrict__ cnt, int nv, uint32_t c, int pw100, int by_vertices,
              unsigned long long* __restrict__ w) {
  int v = blockIdx.x * kT + threadIdx.x;
  if (v >= nv) return;
  unsigned long long x = 0;
  if (deg[v]) {
    if (by_vertices) x = 1;
    else if (pw100 == 100) x = (unsigned long long)cnt[v] + c;
    else x = (unsigned long long)(powf((float)cnt[v], (float)pw100 * 0.01f) + 0.5f) + c;
  }
  w[v] = x;
}
// tile of the k-th ranked vertex: tiles are contiguous NATIVE ranges that serve equally many gathers (weight_before =
// column occurrences of the vertices before it in native order, `total` of all); 255 = no edges
__global__ void __launch_bounds__(kT)
k_tile_of_ranked(const int32_t* __restrict__ order, int nv, const uint32_t* __restrict__ deg, const unsigned long long* __restrict__ weight_before,
                 unsigned long long total, int ntiles, uint8_t* __restrict__ tile) {
  int k = blockIdx.x * kT + threadIdx.x;
  if (k >= nv) return;
  const int v = order[k];
  unsigned long long t = total ? (unsigned long long)((double)weight_before[v] * (double)ntiles / (double)total) : 0ull;
  if (t >= (unsigned long long)ntiles) t = (unsigned long long)ntiles - 1ull;
  tile[k] = deg[v] ? (uint8_t)t : (uint8_t)255;
}
// tile of a device id: the last t with base[t] <= d
__device__ __forceinline__ int tile_of_dev(int d, const int32_t* __restrict__ base, int ntiles) {
  int t = 0;
  while (t + 1 < ntiles && base[t + 1] <= d) t++;
  return t;
}

// rank k -> device id (k % nshards) * S + k / nshards
__global__ void __launch_bounds__(kT)
k_deal(const int32_t* __restrict__ order, int nv, int nshards, int S, int32_t* __restrict__ dev_of_native,
       int32_t* __restrict__ native_of_dev) {
  int k = blockIdx.x * kT + threadIdx.x;
  if (k >= nv) return;
  int v = order[k];
  int d = (k % nshards) * S + k / nshards;
  dev_of_native[v] = d;
  native_of_dev[d] = v;
}

// Sharded graphs with slices (graphmat_hip.h: gm_sweep_t.nsub): position k of the list sorted by (slice, degree rank) -- the j-th
// busiest vertex of slice t -- goes to owner (j + t) % G at position sb[t] + j / G of the owner's range (the rotation by t keeps
// the slices' busiest vertices from all landing on owner 0); vertices without edges (key 255) follow the live part of every range.
struct SliceDeal { int32_t hb[GM_MAX_SLICES + 2]; int32_t sb[GM_MAX_SLICES + 2]; };
__global__ void __launch_bounds__(kT)
k_deal_sliced(const int32_t* __restrict__ order, const uint8_t* __restrict__ key_sorted, int nv, int G, int S, int TS, SliceDeal sd,
              int32_t* __restrict__ dev_of_native, int32_t* __restrict__ native_of_dev) {
  const int k = blockIdx.x * kT + threadIdx.x;
  if (k >= nv) return;
  const int v = order[k];
  const int t = key_sorted[k] == 255 ? TS : (int)key_sorted[k];
  const int j = k - sd.hb[t];
  const int owner = t == TS ? j % G : (j + t) % G;
  const int d = owner * S + sd.sb[t] + j / G;
  dev_of_native[v] = d;
  native_of_dev[d] = v;
}

// key = (local device row, or nrows when the row is not in this shard) << 32 | NATIVE col:
// sorting by it stores a row's edges in the reference's reduction order.
__global__ void __launch_bounds__(kT)
k_make_keys(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t nnz, int by_dst, int nparts,
            int nv, int row_lo, int row_hi, int ids_are_native, const int32_t* __restrict__ dev_of_native,
            uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, unsigned long long* __restrict__ kept) {
  // grid-stride, one atomic per workgroup at the end: with a workgroup (let alone a wave) per 256 edges the single
  // `kept` counter took 16.7 M same-address atomics at RMAT-26 -- 200 ms for a kernel that moves 20 GB
  unsigned int mine_count = 0;
  for (int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * kT) {
    int s = src[e], d = dst[e];
    int sn = ids_are_native ? s : to_native0(s, nparts, nv);
    int dn = ids_are_native ? d : to_native0(d, nparts, nv);
    int rn = by_dst ? dn : sn;
    int c = by_dst ? sn : dn;
    int r = dev_of_native ? dev_of_native[rn] : rn;
    const bool mine = (r >= row_lo && r < row_hi);
    uint32_t rf = mine ? (uint32_t)(r - row_lo) : (uint32_t)(row_hi - row_lo);
    keys[e] = ((uint64_t)rf << 32) | (uint32_t)c;
    idx[e] = (uint32_t)e;
    mine_count += mine ? 1u : 0u;
  }
  __shared__ unsigned int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) mine_count += __shfl_down(mine_count, off, 64);
  if ((threadIdx.x & 63) == 0 && mine_count) atomicAdd(&s_cnt, mine_count);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(kept, (unsigned long long)s_cnt);
}

__global__ void __launch_bounds__(kT)
k_unpack(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, int64_t n, const void* __restrict__ val,
         int val_bytes, const int32_t* __restrict__ dev_of_native, int32_t* __restrict__ colidx,
         void* __restrict__ vals) {
  int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (k >= n) return;
  int32_t cn = (int32_t)(uint32_t)keys[k];
  colidx[k] = dev_of_native ? dev_of_native[cn] : cn;
  if (vals) {
    uint32_t e = idx[k];
    if (val_bytes == 4) {
      ((uint32_t*)vals)[k] = ((const uint32_t*)val)[e];
    } else if (val_bytes == 8) {
      ((uint64_t*)vals)[k] = ((const uint64_t*)val)[e];
    } else {
      const unsigned char* s = (const unsigned char*)val + (size_t)e * val_bytes;
      unsigned char* d = (unsigned char*)vals + (size_t)k * val_bytes;
      for (int b = 0; b < val_bytes; b++) d[b] = s[b];
    }
  }
}

// rowptr[r] = first sorted position whose row field >= r  (r in [0, nrows])
__global__ void __launch_bounds__(kT)
k_rowptr(const uint64_t* __restrict__ keys, int64_t n, int nrows, int64_t* __restrict__ rowptr) {
  int r = blockIdx.x * kT + threadIdx.x;
  if (r > nrows) return;
  uint64_t target = (uint64_t)(uint32_t)r << 32;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  rowptr[r] = lo;
}

// log2 histogram of the row lengths (bin b counts rows with 2^b <= edges < 2^(b+1))
__global__ void __launch_bounds__(kT)
k_deg_hist(const int64_t* __restrict__ rowptr, int nrows, unsigned int* __restrict__ hist) {
  __shared__ unsigned int sh[64];
  if (threadIdx.x < 64) sh[threadIdx.x] = 0;
  __syncthreads();
  int r = blockIdx.x * kT + threadIdx.x;
  if (r < nrows) {
    long long d = rowptr[r + 1] - rowptr[r];
    if (d > 0) atomicAdd(&sh[63 - __clzll(d)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64 && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

// segment starts (runs of rows, see gm_csr_t) and row classes
__global__ void __launch_bounds__(kT)
// (own_wave != null, column tiles with row classes fixed per ROW: a flagged row is a one-wave-per-row or giant row in
// every tile, however short its piece, and no other row is -- gm_csr_t.rows_keep_stream)
k_row_flags(const int64_t* __restrict__ rowptr, int nrows, int short_row, int giant_row, const unsigned char* __restrict__ own_wave,
            unsigned char* __restrict__ start, unsigned char* __restrict__ ismid, unsigned char* __restrict__ isgiant) {
  int r = blockIdx.x * kT + threadIdx.x;
  if (r >= nrows) return;
  int64_t a = rowptr[r], b = rowptr[r + 1];
  const bool flagged = own_wave != nullptr && own_wave[r] && b > a;
  bool lng = (b - a) > short_row || flagged;
  bool st = (r == 0) || ((r & 255) == 0) || lng;
  if (!st) {
    int64_t pa = rowptr[r - 1];
    bool prev_long = (a - pa) > short_row || (own_wave != nullptr && own_wave[r - 1] && a > pa);
    st = prev_long || (a / GM_BLOCK_NNZ != pa / GM_BLOCK_NNZ);
  }
  const bool giant = (b - a) > giant_row && (own_wave == nullptr || flagged);
  start[r] = st ? 1 : 0;
  ismid[r] = (lng && !giant) ? 1 : 0;
  isgiant[r] = giant ? 1 : 0;
}

// edges per kernel class of the multiply: [0] row-blocks (rows of <= short_row edges), [1] 16-rows-per-wave rows,
// [2] one-wave-per-row rows (more than long_limit edges), [3] giant rows
__global__ void __launch_bounds__(kT)
k_class_edges(const int64_t* __restrict__ rowptr, int nrows, int short_row, int64_t long_limit, int giant_row,
              const unsigned char* __restrict__ own_wave, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long s_c[4];
  if (threadIdx.x < 4) s_c[threadIdx.x] = 0ull;
  __syncthreads();
  const int r = blockIdx.x * kT + threadIdx.x;
  if (r < nrows) {
    const int64_t len = rowptr[r + 1] - rowptr[r];
    if (len > 0) {
      int cls = len <= short_row ? 0 : len > giant_row ? 3 : len > long_limit ? 2 : 1;
      if (own_wave != nullptr) cls = own_wave[r] ? (len > giant_row ? 3 : 2) : (len <= short_row ? 0 : 1);
      atomicAdd(&s_c[cls], (unsigned long long)len);
    }
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_c[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_c[threadIdx.x]);
}

// rows of more than `limit` edges in the whole graph (column tiles with row classes fixed per row)
__global__ void __launch_bounds__(kT)
k_own_wave_rows(const int64_t* __restrict__ rowptr, int nrows, int64_t limit, unsigned char* __restrict__ flag) {
  int r = blockIdx.x * kT + threadIdx.x;
  if (r < nrows) flag[r] = (rowptr[r + 1] - rowptr[r] > limit) ? 1 : 0;
}

// presence bits of the non-empty rows (bit r&31 of word r>>5)
__global__ void __launch_bounds__(kT)
k_rowbits(const int64_t* __restrict__ rowptr, int nrows, uint32_t* __restrict__ bits) {
  int r = blockIdx.x * kT + threadIdx.x;
  bool ne = r < nrows && rowptr[r + 1] > rowptr[r];
  unsigned long long m = __ballot(ne);
  if ((threadIdx.x & 63) == 0 && r < nrows) {
    bits[r >> 5] = (uint32_t)m;
    if (r + 32 < nrows) bits[(r >> 5) + 1] = (uint32_t)(m >> 32);
  }
}
__global__ void __launch_bounds__(kT)
k_or_words(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ o, int n) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) o[i] = a[i] | b[i];
}

// one past the last entry of the wave-row list whose row has more than `limit` edges
__global__ void __launch_bounds__(kT)
k_last_long(const int32_t* __restrict__ mid_row, int nmid, const int64_t* __restrict__ rowptr, int64_t limit, int* __restrict__ out) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nmid) return;
  const int r = mid_row[i];
  if (rowptr[r + 1] - rowptr[r] > limit) atomicMax(out, i + 1);
}

__global__ void __launch_bounds__(kT)
k_mid_long_flags(const int32_t* __restrict__ mid_row, int nmid, const int64_t* __restrict__ rowptr, int64_t limit, int64_t max_len,
                 const unsigned char* __restrict__ own_wave, unsigned char* __restrict__ is_long, unsigned char* __restrict__ is_rest) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nmid) return;
  const int r = mid_row[i];
  const int64_t len = rowptr[r + 1] - rowptr[r];
  const bool keep = len <= max_len, lg = own_wave != nullptr ? own_wave[r] != 0 : len > limit;
  is_long[i] = (keep && lg) ? 1 : 0;
  is_rest[i] = (keep && !lg) ? 1 : 0;
}

__global__ void __launch_bounds__(kT)
k_giant_extent(const int32_t* __restrict__ giant_row, int ngiant, const int64_t* __restrict__ rowptr,
               int64_t* __restrict__ ext) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i >= ngiant) return;
  ext[2 * i] = rowptr[giant_row[i]];
  ext[2 * i + 1] = rowptr[giant_row[i] + 1];
}

// a segment is a row-block iff its first row is short and it holds at least one edge
__global__ void __launch_bounds__(kT)
k_seg_flags(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ seg_row, int nseg, int short_row,
            const unsigned char* __restrict__ own_wave, unsigned char* __restrict__ isblk) {
  int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nseg) return;
  int r0 = seg_row[i], r1 = seg_row[i + 1];
  int64_t n = rowptr[r1] - rowptr[r0];
  bool longrow = (r1 - r0 == 1) && (n > short_row || (own_wave != nullptr && own_wave[r0]));
  isblk[i] = (!longrow && n > 0) ? 1 : 0;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) {
    if (p) { (void)hipFree(p); p = nullptr; }
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { set_error("hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); p = nullptr; return GM_ERR_NOMEM; }
    return GM_OK;
  }
  void free() { if (p) { (void)hipFree(p); p = nullptr; } }
  template <class T> T* as() { return (T*)p; }
  void* release() { void* q = p; p = nullptr; return q; }
};

static int bits_for(uint32_t maxval) {
  int b = 0;
  while ((1ull << b) <= (uint64_t)maxval) b++;
  return b;
}

// out = [entries of mid_row whose row has more than long_limit edges][the others], both in list order,
// keeping only rows of at most max_len edges; *n_long / *n_total = sizes
static int partition_mid(const int32_t* mid_row, int nmid, const int64_t* rowptr, int64_t long_limit, int64_t max_len,
                         int32_t* out, int* n_long, unsigned int* n_total, hipStream_t s, const unsigned char* own_wave = nullptr) {
  DevBuf fl, fs, tmp, cnt;
  int rc;
  if ((rc = fl.alloc((size_t)nmid)) || (rc = fs.alloc((size_t)nmid)) || (rc = cnt.alloc(16))) return rc;
  hipLaunchKernelGGL(k_mid_long_flags, dim3(grid_for(nmid)), dim3(kT), 0, s, mid_row, nmid, rowptr, long_limit, max_len,
                     own_wave, fl.as<unsigned char>(), fs.as<unsigned char>());
  size_t tb = 0;
  GM_TRY_HIP(rocprim::select(nullptr, tb, mid_row, fl.as<unsigned char>(), out, cnt.as<unsigned int>(), (size_t)nmid, s));
  if ((rc = tmp.alloc(tb + 256))) return rc;
  GM_TRY_HIP(rocprim::select(tmp.p, tb, mid_row, fl.as<unsigned char>(), out, cnt.as<unsigned int>(), (size_t)nmid, s));
  unsigned int nl = 0, ns = 0;
  GM_TRY_HIP(hipMemcpyAsync(&nl, cnt.p, 4, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  GM_TRY_HIP(rocprim::select(tmp.p, tb, mid_row, fs.as<unsigned char>(), out + nl, cnt.as<unsigned int>() + 1, (size_t)nmid, s));
  GM_TRY_HIP(hipMemcpyAsync(&ns, cnt.as<unsigned int>() + 1, 4, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  *n_long = (int)nl;
  *n_total = nl + ns;
  return GM_OK;
}

__global__ void __launch_bounds__(kT)
k_row_lengths(const int32_t* __restrict__ list, int n, const int64_t* __restrict__ rowptr, uint32_t* __restrict__ len) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const int r = list[i];
  len[i] = (uint32_t)(rowptr[r + 1] - rowptr[r]);
}
// list[0, n) reordered by descending row length (stable: rows of equal length keep their order).  The persistent
// 16-rows-per-wave kernel deals consecutive 16-entry groups to its waves round robin and a group takes as many 64-edge
// steps as its LONGEST row: in row order a column tile's pieces of neighbouring rows differ widely (RMAT-26: 77 % of the
// 16 x 64 slots of a step hold an edge, and the waves of one launch finish between 190 and 330 us); sorted, the rows of a
// group are equally long and every wave gets the same number of steps.
static int sort_rows_by_length(int32_t* list, int n, const int64_t* rowptr, hipStream_t s) {
  if (n < 2) return GM_OK;
  DevBuf k_in, k_out, v_out, tmp;
  int rc;
  if ((rc = k_in.alloc((size_t)n * 4)) || (rc = k_out.alloc((size_t)n * 4)) || (rc = v_out.alloc((size_t)n * 4))) return rc;
  hipLaunchKernelGGL(k_row_lengths, dim3(grid_for(n)), dim3(kT), 0, s, (const int32_t*)list, n, rowptr, k_in.as<uint32_t>());
  size_t tb = 0;
  GM_TRY_HIP(rocprim::radix_sort_pairs_desc(nullptr, tb, k_in.as<uint32_t>(), k_out.as<uint32_t>(), (const int32_t*)list, v_out.as<int32_t>(), (size_t)n, 0, 32, s));
  if ((rc = tmp.alloc(tb + 256))) return rc;
  GM_TRY_HIP(rocprim::radix_sort_pairs_desc(tmp.p, tb, k_in.as<uint32_t>(), k_out.as<uint32_t>(), (const int32_t*)list, v_out.as<int32_t>(), (size_t)n, 0, 32, s));
  GM_TRY_HIP(hipMemcpyAsync(list, v_out.p, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  return GM_OK;
}

static int finish_csr(gm_graph* g, const uint64_t* keys_sorted, const uint32_t* idx_sorted, unsigned long long kept,
                      const void* d_val, hipStream_t s, CsrOwned* out, int tile_split = -1, const unsigned char* own_wave = nullptr);
static int build_tiles(gm_graph* g, const uint64_t* keys_sorted, const uint32_t* idx_sorted, unsigned long long kept,
                       const void* d_val, hipStream_t s, const CsrOwned* whole);
static int build_sweep(gm_graph* g, const CsrOwned* whole, hipStream_t s);

// Sorts the edges of one direction into the reference's reduction order and builds its CSR
// (and, for GM_DIR_OUT of a tiled graph, the per-tile CSRs from the same sorted keys).
static int build_direction(gm_graph* g, int by_dst, int64_t nnz, const int32_t* d_src, const int32_t* d_dst,
                           const void* d_val, hipStream_t s, CsrOwned* out) {
  const gm_graph_desc_t& D = g->desc;
  const int nrows = D.row_hi - D.row_lo;
  DevBuf keys_in, keys_out, idx_in, idx_out, kept_d, tmp;
  int rc;
  if ((rc = keys_in.alloc((size_t)nnz * 8))) return rc;
  if ((rc = keys_out.alloc((size_t)nnz * 8))) return rc;
  if ((rc = idx_in.alloc((size_t)nnz * 4))) return rc;
  if ((rc = idx_out.alloc((size_t)nnz * 4))) return rc;
  if ((rc = kept_d.alloc(8))) return rc;
  GM_TRY_HIP(hipMemsetAsync(kept_d.p, 0, 8, s));
  if (nnz > 0) {
    hipLaunchKernelGGL(k_make_keys, dim3(grid_for(nnz) < 16384 ? grid_for(nnz) : 16384), dim3(kT), 0, s, d_src, d_dst, nnz, by_dst, D.nparts,
                       D.nvertices, D.row_lo, D.row_hi, D.ids_are_native, (const int32_t*)g->dev_of_native,
                       keys_in.as<uint64_t>(), idx_in.as<uint32_t>(), kept_d.as<unsigned long long>());
    GM_TRY_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    const unsigned end_bit = 32 + (unsigned)bits_for((uint32_t)nrows);
    GM_TRY_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(),
                                         idx_in.as<uint32_t>(), idx_out.as<uint32_t>(), (size_t)nnz, 0u, end_bit, s));
    if ((rc = tmp.alloc(tmp_bytes))) return rc;
    GM_TRY_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(),
                                         idx_in.as<uint32_t>(), idx_out.as<uint32_t>(), (size_t)nnz, 0u, end_bit, s));
  }
  unsigned long long kept = 0;
  GM_TRY_HIP(hipMemcpyAsync(&kept, kept_d.p, 8, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  keys_in.free();
  idx_in.free();
  tmp.free();
  const bool tiled = by_dst && g->ntiles > 1;
  if ((rc = finish_csr(g, keys_out.as<uint64_t>(), idx_out.as<uint32_t>(), kept, d_val, s, out, tiled ? g_tile_min_row : -1))) return rc;
  if (tiled) rc = build_tiles(g, keys_out.as<uint64_t>(), idx_out.as<uint32_t>(), kept, d_val, s, out);
  else if (by_dst && D.nshards > 1 && g->nslices > 1 && g_sweep_slices != 0) {  // a shard's rows through the sweep (gm_sweep_t.nsub)
    keys_out.free(); idx_out.free();
    rc = build_sweep(g, out, s);
  }
  return rc;
}

// ---- distributed build (gm_graph_desc_t.edges_local): this rank holds a part of the edge list ----------
// key = (row inside its owner's slice) << 32 | NATIVE col, owner = the shard whose slice holds the row
__global__ void __launch_bounds__(kT)
k_owner_keys(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t nnz, int by_dst, int nparts, int nv,
             int S, int ids_are_native, const int32_t* __restrict__ dev_of_native, uint64_t* __restrict__ keys,
             uint16_t* __restrict__ owner, uint32_t* __restrict__ pos) {
  const int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= nnz) return;
  const int s = src[e], d = dst[e];
  const int sn = ids_are_native ? s : to_native0(s, nparts, nv);
  const int dn = ids_are_native ? d : to_native0(d, nparts, nv);
  const int r = dev_of_native[by_dst ? dn : sn];
  const int q = r / S;
  keys[e] = ((uint64_t)(uint32_t)(r - q * S) << 32) | (uint32_t)(by_dst ? sn : dn);
  owner[e] = (uint16_t)q;
  pos[e] = (uint32_t)e;
}
// bounds[q] = first position of the sorted owner list whose owner is >= q  (q in [0, n_owners])
__global__ void k_owner_bounds(const uint16_t* __restrict__ owner_sorted, int64_t n, int n_owners, int64_t* __restrict__ bounds) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > n_owners) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int)owner_sorted[mid] < q) lo = mid + 1; else hi = mid;
  }
  bounds[q] = lo;
}
// the edges in bucket order (owner, then input position): keys and values ready to travel
__global__ void __launch_bounds__(kT)
k_bucket_gather(const uint64_t* __restrict__ keys, const void* __restrict__ val, int val_bytes, const uint32_t* __restrict__ pos,
                int64_t n, uint64_t* __restrict__ keys_b, void* __restrict__ val_b) {
  const int64_t j = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (j >= n) return;
  const uint32_t p = pos[j];
  keys_b[j] = keys[p];
  if (val_b) {
    if (val_bytes == 4) ((uint32_t*)val_b)[j] = ((const uint32_t*)val)[p];
    else if (val_bytes == 8) ((uint64_t*)val_b)[j] = ((const uint64_t*)val)[p];
    else {
      const unsigned char* a = (const unsigned char*)val + (size_t)p * val_bytes;
      unsigned char* b = (unsigned char*)val_b + (size_t)j * val_bytes;
      for (int i = 0; i < val_bytes; i++) b[i] = a[i];
    }
  }
}
__global__ void __launch_bounds__(kT)
k_iota(uint32_t* __restrict__ a, int64_t n) {
  const int64_t j = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (j < n) a[j] = (uint32_t)j;
}

// One direction from this rank's part of the edges: bucket them by the shard that owns their row, hand every
// shard its bucket (one padded all-gather per owner; a rank keeps what is addressed to it, in rank order, so
// duplicates of an edge stay in "rank, then input position" order), sort what arrived, build the CSR.
static int build_direction_local(gm_graph* g, int by_dst, int64_t nnz, const int32_t* d_src, const int32_t* d_dst,
                                 const void* d_val, hipStream_t s, CsrOwned* out) {
  const gm_graph_desc_t& D = g->desc;
  // (whether values travel is a property of the graph, not of this rank's part: a rank without edges may pass no array)
  const int N = D.nshards, me = D.shard, S = D.row_hi - D.row_lo, vb = D.val_bytes > 0 ? D.val_bytes : 0;
  const int64_t missing_vals = (vb && nnz > 0 && !d_val) ? 1 : 0;
  int rc;
  DevBuf keys, owner_in, owner_out, pos_in, pos_out, tmp, bounds_d, send_k, send_v;
  std::vector<int64_t> bounds((size_t)N + 1, 0);
  if (nnz > 0) {
    if ((rc = keys.alloc((size_t)nnz * 8)) || (rc = owner_in.alloc((size_t)nnz * 2)) || (rc = owner_out.alloc((size_t)nnz * 2)) ||
        (rc = pos_in.alloc((size_t)nnz * 4)) || (rc = pos_out.alloc((size_t)nnz * 4)) || (rc = bounds_d.alloc((size_t)(N + 1) * 8)))
      return rc;
    hipLaunchKernelGGL(k_owner_keys, dim3(grid_for(nnz)), dim3(kT), 0, s, d_src, d_dst, nnz, by_dst, D.nparts, D.nvertices, S,
                       D.ids_are_native, (const int32_t*)g->dev_of_native, keys.as<uint64_t>(), owner_in.as<uint16_t>(),
                       pos_in.as<uint32_t>());
    size_t tb = 0;
    const unsigned ob = (unsigned)std::max(1, bits_for((uint32_t)(N - 1)));
    GM_TRY_HIP(rocprim::radix_sort_pairs(nullptr, tb, owner_in.as<uint16_t>(), owner_out.as<uint16_t>(), pos_in.as<uint32_t>(),
                                         pos_out.as<uint32_t>(), (size_t)nnz, 0u, ob, s));
    if ((rc = tmp.alloc(tb))) return rc;
    GM_TRY_HIP(rocprim::radix_sort_pairs(tmp.p, tb, owner_in.as<uint16_t>(), owner_out.as<uint16_t>(), pos_in.as<uint32_t>(),
                                         pos_out.as<uint32_t>(), (size_t)nnz, 0u, ob, s));
    hipLaunchKernelGGL(k_owner_bounds, dim3((N + 1 + 63) / 64), dim3(64), 0, s, (const uint16_t*)owner_out.as<uint16_t>(), nnz, N,
                       bounds_d.as<int64_t>());
    GM_TRY_HIP(hipMemcpyAsync(bounds.data(), bounds_d.p, (size_t)(N + 1) * 8, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
  }
  // cnt[p * N + q] = edges on rank p whose row shard q owns
  std::vector<int64_t> mine((size_t)N + 1), cnt((size_t)N * N), bytes((size_t)N);
  for (int q = 0; q < N; q++) mine[(size_t)q] = bounds[(size_t)q + 1] - bounds[(size_t)q];
  mine[(size_t)N] = missing_vals;
  {
    void* all = nullptr;
    if ((rc = gm_dist_allgatherv_host(mine.data(), (int64_t)(N + 1) * 8, &all, bytes.data())) != GM_OK) return rc;
    int64_t bad = 0;
    for (int p = 0; p < N; p++) {
      memcpy(&cnt[(size_t)p * N], (const int64_t*)all + (size_t)p * (N + 1), (size_t)N * 8);
      bad += ((const int64_t*)all)[(size_t)p * (N + 1) + N];
    }
    gm_host_free(all);
    if (bad) { set_error("gm_graph_create: val_bytes = %d but %lld rank(s) passed edges without values", vb, (long long)bad); return GM_ERR_INVALID; }
  }
  int64_t maxm = 0, kept = 0;
  for (int q = 0; q < N; q++)
    for (int p = 0; p < N; p++) maxm = std::max(maxm, cnt[(size_t)p * N + q]);
  for (int p = 0; p < N; p++) kept += cnt[(size_t)p * N + me];
  if (kept >= (1ll << 32)) { set_error("gm_graph_create: shard %d would hold more than 2^32-1 edges", me); return GM_ERR_UNSUPPORTED; }
  // send buffers in bucket order, padded so that a full-size block can be read from every bucket start
  if ((rc = send_k.alloc((size_t)(nnz + maxm) * 8)) || (vb && (rc = send_v.alloc((size_t)(nnz + maxm) * vb)))) return rc;
  if (nnz > 0)
    hipLaunchKernelGGL(k_bucket_gather, dim3(grid_for(nnz)), dim3(kT), 0, s, (const uint64_t*)keys.as<uint64_t>(), d_val, vb,
                       (const uint32_t*)pos_out.as<uint32_t>(), nnz, send_k.as<uint64_t>(), vb ? send_v.p : nullptr);
  GM_TRY_HIP(hipStreamSynchronize(s));
  keys.free(); owner_in.free(); owner_out.free(); pos_in.free(); pos_out.free(); tmp.free();
  DevBuf rk_in, rk_out, ri_in, ri_out, rvals, gk, gv;
  if ((rc = rk_in.alloc((size_t)kept * 8)) || (rc = gk.alloc((size_t)N * maxm * 8)) ||
      (vb && ((rc = rvals.alloc((size_t)kept * vb)) || (rc = gv.alloc((size_t)N * maxm * vb)))))
    return rc;
  for (int q = 0; q < N; q++) {
    int64_t m = 0;
    for (int p = 0; p < N; p++) m = std::max(m, cnt[(size_t)p * N + q]);
    if (m == 0) continue;
    if ((rc = dist_all_gather_bytes(send_k.as<uint64_t>() + bounds[(size_t)q], gk.p, (size_t)m * 8, s))) return rc;
    if (vb && (rc = dist_all_gather_bytes((const char*)send_v.p + (size_t)bounds[(size_t)q] * vb, gv.p, (size_t)m * vb, s))) return rc;
    if (q == me) {
      int64_t off = 0;
      for (int p = 0; p < N; p++) {
        const int64_t c = cnt[(size_t)p * N + me];
        if (c > 0) {
          GM_TRY_HIP(hipMemcpyAsync(rk_in.as<uint64_t>() + off, gk.as<uint64_t>() + (size_t)p * m, (size_t)c * 8, hipMemcpyDeviceToDevice, s));
          if (vb) GM_TRY_HIP(hipMemcpyAsync((char*)rvals.p + (size_t)off * vb, (const char*)gv.p + (size_t)p * m * vb, (size_t)c * vb, hipMemcpyDeviceToDevice, s));
        }
        off += c;
      }
    }
    GM_TRY_HIP(hipStreamSynchronize(s));  // the gather buffers are reused by the next owner
  }
  gk.free(); gv.free(); send_k.free(); send_v.free();
  if ((rc = rk_out.alloc((size_t)kept * 8)) || (rc = ri_in.alloc((size_t)kept * 4)) || (rc = ri_out.alloc((size_t)kept * 4))) return rc;
  if (kept > 0) {
    hipLaunchKernelGGL(k_iota, dim3(grid_for(kept)), dim3(kT), 0, s, ri_in.as<uint32_t>(), kept);
    size_t tb = 0;
    const unsigned end_bit = 32 + (unsigned)bits_for((uint32_t)S);
    GM_TRY_HIP(rocprim::radix_sort_pairs(nullptr, tb, rk_in.as<uint64_t>(), rk_out.as<uint64_t>(), ri_in.as<uint32_t>(),
                                         ri_out.as<uint32_t>(), (size_t)kept, 0u, end_bit, s));
    if ((rc = tmp.alloc(tb))) return rc;
    GM_TRY_HIP(rocprim::radix_sort_pairs(tmp.p, tb, rk_in.as<uint64_t>(), rk_out.as<uint64_t>(), ri_in.as<uint32_t>(),
                                         ri_out.as<uint32_t>(), (size_t)kept, 0u, end_bit, s));
  }
  GM_TRY_HIP(hipStreamSynchronize(s));
  rk_in.free(); ri_in.free(); tmp.free();
  if ((rc = finish_csr(g, rk_out.as<uint64_t>(), ri_out.as<uint32_t>(), (unsigned long long)kept, vb ? rvals.p : nullptr, s, out, -1))) return rc;
  if (by_dst && N > 1 && g->nslices > 1 && g_sweep_slices != 0) {
    rk_out.free(); ri_out.free(); rvals.free();
    rc = build_sweep(g, out, s);
  }
  return rc;
}

// CSR arrays and work decomposition from `kept` sorted keys (row << 32 | native col) and, for the
// edge values, the input position of every sorted edge.
// tile_split >= 0 (whole-graph CSR of a tiled graph): also list the wave rows of at most tile_split edges.
// own_wave != null (a column tile): row classes fixed per row, see k_row_flags.
static int finish_csr(gm_graph* g, const uint64_t* keys_sorted, const uint32_t* idx_sorted, unsigned long long kept,
                      const void* d_val, hipStream_t s, CsrOwned* out, int tile_split, const unsigned char* own_wave) {
  const int short_row = g_short_row;
  const gm_graph_desc_t& D = g->desc;
  const int nrows = D.row_hi - D.row_lo;
  int rc;
  DevBuf tmp;
  DevBuf rowptr, colidx, vals, cnt;
  if ((rc = rowptr.alloc((size_t)(nrows + 1) * 8))) return rc;
  if ((rc = colidx.alloc((size_t)kept * 4))) return rc;
  const bool keep_vals = D.val_bytes > 0 && d_val != nullptr;
  if (keep_vals && (rc = vals.alloc((size_t)kept * D.val_bytes))) return rc;
  if (kept > 0) {
    hipLaunchKernelGGL(k_unpack, dim3(grid_for((int64_t)kept)), dim3(kT), 0, s, keys_sorted,
                       idx_sorted, (int64_t)kept, d_val, D.val_bytes, (const int32_t*)g->dev_of_native,
                       colidx.as<int32_t>(), keep_vals ? vals.p : nullptr);
  }
  hipLaunchKernelGGL(k_rowptr, dim3(grid_for(nrows + 1)), dim3(kT), 0, s, keys_sorted, (int64_t)kept,
                     nrows, rowptr.as<int64_t>());
  GM_TRY_HIP(hipGetLastError());

  DevBuf rbits;
  if ((rc = rbits.alloc(((size_t)(nrows + 31) / 32 + 2) * 4))) return rc;
  GM_TRY_HIP(hipMemsetAsync(rbits.p, 0, ((size_t)(nrows + 31) / 32 + 2) * 4, s));
  if (nrows > 0)
    hipLaunchKernelGGL(k_rowbits, dim3(grid_for(nrows)), dim3(kT), 0, s, rowptr.as<int64_t>(), nrows, rbits.as<uint32_t>());

  // Giant-row threshold.  The per-row workgroups of k_spmv_giant are the right tool for the few
  // rows whose serial chain would otherwise dominate, and the wrong one for thousands of merely
  // long rows, so the threshold is the smallest power of two >= 4096 that leaves at most 1024
  // rows above it (GM_GIANT_ROW documents the typical outcome at RMAT-24..27).
  int giant_row = g_giant_row;
  if (giant_row <= 0) {
    DevBuf hist;
    if ((rc = hist.alloc(64 * 4))) return rc;
    GM_TRY_HIP(hipMemsetAsync(hist.p, 0, 64 * 4, s));
    if (nrows > 0)
      hipLaunchKernelGGL(k_deg_hist, dim3(grid_for(nrows)), dim3(kT), 0, s, rowptr.as<int64_t>(), nrows,
                         hist.as<unsigned int>());
    unsigned int h[64];
    GM_TRY_HIP(hipMemcpyAsync(h, hist.p, 64 * 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    int b = 40;
    unsigned long long above = 0;
    while (b > 12 && above + h[b - 1] <= 1024) { above += h[b - 1]; b--; }
    giant_row = (b >= 31) ? 0x7fffffff : (1 << b);  // rows of >= 2^b edges (~ > giant_row) are giant
  }

  // work decomposition: segments, row-blocks, wave rows, giant rows
  DevBuf f0, f1, f2, seg, blkl, mid, giant;
  if ((rc = f0.alloc((size_t)nrows + 1))) return rc;
  if ((rc = f1.alloc((size_t)nrows + 1))) return rc;
  if ((rc = f2.alloc((size_t)nrows + 1))) return rc;
  if ((rc = seg.alloc((size_t)(nrows + 2) * 4))) return rc;
  if ((rc = mid.alloc((size_t)(nrows + 1) * 4))) return rc;
  if ((rc = giant.alloc((size_t)(nrows + 1) * 4))) return rc;
  if ((rc = cnt.alloc(32))) return rc;
  unsigned int nseg = 0, nblk = 0, nmid = 0, ngiant = 0;
  if (nrows > 0) {
    hipLaunchKernelGGL(k_row_flags, dim3(grid_for(nrows)), dim3(kT), 0, s, rowptr.as<int64_t>(), nrows, short_row,
                       giant_row, own_wave, f0.as<unsigned char>(), f1.as<unsigned char>(), f2.as<unsigned char>());
    GM_TRY_HIP(hipGetLastError());
    rocprim::counting_iterator<int32_t> ids(0);
    size_t tb = 0;
    GM_TRY_HIP(rocprim::select(nullptr, tb, ids, f0.as<unsigned char>(), seg.as<int32_t>(), cnt.as<unsigned int>(),
                               (size_t)nrows, s));
    if ((rc = tmp.alloc(tb + 256))) return rc;
    GM_TRY_HIP(rocprim::select(tmp.p, tb, ids, f0.as<unsigned char>(), seg.as<int32_t>(), cnt.as<unsigned int>(),
                               (size_t)nrows, s));
    GM_TRY_HIP(rocprim::select(tmp.p, tb, ids, f1.as<unsigned char>(), mid.as<int32_t>(), cnt.as<unsigned int>() + 1,
                               (size_t)nrows, s));
    GM_TRY_HIP(rocprim::select(tmp.p, tb, ids, f2.as<unsigned char>(), giant.as<int32_t>(),
                               cnt.as<unsigned int>() + 2, (size_t)nrows, s));
    unsigned int h[3] = {0, 0, 0};
    GM_TRY_HIP(hipMemcpyAsync(h, cnt.p, 12, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    nseg = h[0];
    nmid = h[1];
    ngiant = h[2];
    int32_t last = nrows;
    GM_TRY_HIP(hipMemcpyAsync(seg.as<int32_t>() + nseg, &last, 4, hipMemcpyHostToDevice, s));
    // row-blocks = the segments that hold short rows with at least one edge
    if ((rc = blkl.alloc((size_t)(nseg + 1) * 4))) return rc;
    hipLaunchKernelGGL(k_seg_flags, dim3(grid_for(nseg)), dim3(kT), 0, s, rowptr.as<int64_t>(), seg.as<int32_t>(),
                       (int)nseg, short_row, own_wave, f0.as<unsigned char>());
    GM_TRY_HIP(rocprim::select(tmp.p, tb, ids, f0.as<unsigned char>(), blkl.as<int32_t>(), cnt.as<unsigned int>() + 3,
                               (size_t)nseg, s));
    GM_TRY_HIP(hipMemcpyAsync(h, cnt.as<unsigned int>() + 3, 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    nblk = h[0];
  } else {
    if ((rc = blkl.alloc(16))) return rc;
  }

  // wave rows: where the long ones (more than GM_LONG_MID edges) end in the list
  int nmid_long = 0, numid_long = 0;
  unsigned int numid = 0;
  DevBuf umid;
  if (nmid > 0) {
    GM_TRY_HIP(hipMemsetAsync(cnt.p, 0, 4, s));
    // (with a million wave rows there are plenty of 16-row groups to keep the chip busy, and rows of up to
    // 4096 edges can be grouped too: RMAT-26 8.02 -> 7.87 ms; on RMAT-22 that limit costs 17 %)
    const int64_t long_limit = g_long_mid > 0 ? (int64_t)g_long_mid : nmid >= (1u << 20) ? 4 * (int64_t)GM_LONG_MID : (int64_t)GM_LONG_MID;
    if (g->ntiles > 1) {
      // tiled device order = (tile, degree rank): the long rows are no prefix of the row-ordered list,
      // so the list is partitioned instead: [long rows][the rest], each part in row order
      DevBuf mid2;
      if ((rc = mid2.alloc((size_t)(nrows + 1) * 4))) return rc;
      unsigned int n_all = 0;
      if ((rc = partition_mid(mid.as<int32_t>(), (int)nmid, rowptr.as<int64_t>(), long_limit, (int64_t)1 << 62, mid2.as<int32_t>(), &nmid_long, &n_all, s, own_wave))) return rc;
      if (tile_split >= 0) {  // the wave rows that stay untiled, same layout
        if ((rc = umid.alloc((size_t)(nmid + 1) * 4))) return rc;
        if ((rc = partition_mid(mid.as<int32_t>(), (int)nmid, rowptr.as<int64_t>(), long_limit, (int64_t)tile_split, umid.as<int32_t>(), &numid_long, &numid, s))) return rc;
      }
      // a column tile's lists by descending piece length (gm_set_option("sort_tile_lists", 0): row order, for A/B runs)
      if (own_wave != nullptr && g_sort_tile_lists != 0) {
        if ((rc = sort_rows_by_length(mid2.as<int32_t>(), nmid_long, rowptr.as<int64_t>(), s))) return rc;
        if ((rc = sort_rows_by_length(mid2.as<int32_t>() + nmid_long, (int)n_all - nmid_long, rowptr.as<int64_t>(), s))) return rc;
      }
      void* old = mid.release();
      mid.p = mid2.release();
      (void)hipFree(old);
    } else {
      hipLaunchKernelGGL(k_last_long, dim3(grid_for(nmid)), dim3(kT), 0, s, mid.as<int32_t>(), (int)nmid, rowptr.as<int64_t>(),
                         long_limit, cnt.as<int>());
      GM_TRY_HIP(hipMemcpyAsync(&nmid_long, cnt.p, 4, hipMemcpyDeviceToHost, s));
      GM_TRY_HIP(hipStreamSynchronize(s));
    }
  }

  // edges per kernel class (reporting: the per-kernel edge rates of bench.py)
  unsigned long long h_class[4] = {0, 0, 0, 0};
  {
    const int64_t long_limit = g_long_mid > 0 ? (int64_t)g_long_mid : nmid >= (1u << 20) ? 4 * (int64_t)GM_LONG_MID : (int64_t)GM_LONG_MID;
    DevBuf cls;
    if ((rc = cls.alloc(32))) return rc;
    GM_TRY_HIP(hipMemsetAsync(cls.p, 0, 32, s));
    if (nrows > 0)
      hipLaunchKernelGGL(k_class_edges, dim3(grid_for(nrows)), dim3(kT), 0, s, rowptr.as<int64_t>(), nrows, short_row, long_limit, giant_row,
                         own_wave, cls.as<unsigned long long>());
    GM_TRY_HIP(hipMemcpyAsync(h_class, cls.p, 32, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
  }

  // pieces of the giant rows for the parallel products pass
  DevBuf gcr, gce, gto;
  std::vector<int32_t> h_gcr;
  std::vector<int64_t> h_gce, h_gto(1, 0);
  if (ngiant > 0) {
    DevBuf ext;
    if ((rc = ext.alloc((size_t)ngiant * 16))) return rc;
    hipLaunchKernelGGL(k_giant_extent, dim3(grid_for(ngiant)), dim3(kT), 0, s, giant.as<int32_t>(), (int)ngiant,
                       rowptr.as<int64_t>(), ext.as<int64_t>());
    std::vector<int64_t> h_ext((size_t)ngiant * 2);
    GM_TRY_HIP(hipMemcpyAsync(h_ext.data(), ext.p, (size_t)ngiant * 16, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    for (unsigned i = 0; i < ngiant; i++) {
      for (int64_t e = h_ext[2 * i]; e < h_ext[2 * i + 1]; e += GM_GIANT_CHUNK) {
        h_gcr.push_back((int32_t)i);
        h_gce.push_back(e);
      }
      h_gto.push_back(h_gto.back() + ((h_ext[2 * i + 1] - h_ext[2 * i]) + 63) / 64 * 64);
    }
  }
  if ((rc = gcr.alloc(h_gcr.size() * 4)) || (rc = gce.alloc(h_gce.size() * 8)) || (rc = gto.alloc(h_gto.size() * 8))) return rc;
  if (!h_gcr.empty()) {
    GM_TRY_HIP(hipMemcpyAsync(gcr.p, h_gcr.data(), h_gcr.size() * 4, hipMemcpyHostToDevice, s));
    GM_TRY_HIP(hipMemcpyAsync(gce.p, h_gce.data(), h_gce.size() * 8, hipMemcpyHostToDevice, s));
  }
  GM_TRY_HIP(hipMemcpyAsync(gto.p, h_gto.data(), h_gto.size() * 8, hipMemcpyHostToDevice, s));
  DevBuf gst;  // 32-byte records of the giant-row kernels, one per 512-product sub-piece = 8 per piece (kernels.hpp: gchunk_state; no hints yet)
  const size_t gst_bytes = h_gcr.size() * (GM_GIANT_CHUNK / 512) * 32 + 64;  // (32-byte records: a hint and the maps of three binades)
  if ((rc = gst.alloc(gst_bytes))) return rc;
  GM_TRY_HIP(hipMemsetAsync(gst.p, 0, gst_bytes, s));
  GM_TRY_HIP(hipStreamSynchronize(s));

  out->rowptr = (int64_t*)rowptr.release();
  out->colidx = (int32_t*)colidx.release();
  out->vals = keep_vals ? vals.release() : nullptr;
  out->rowbits = (uint32_t*)rbits.release();
  out->seg_row = (int32_t*)seg.release();
  out->blk_seg = (int32_t*)blkl.release();
  out->mid_row = (int32_t*)mid.release();
  out->giant_row = (int32_t*)giant.release();
  out->gchunk_row = (int32_t*)gcr.release();
  out->gchunk_edge = (int64_t*)gce.release();
  out->gterm_off = (int64_t*)gto.release();
  out->umid_row = numid > 0 ? (int32_t*)umid.release() : nullptr;
  out->gchunk_state = gst.release();
  out->present = true;
  gm_csr_t& v = out->view;
  v.nnz = (int64_t)kept;
  v.nrows = nrows;
  v.row_base = D.row_lo;
  v.ncols = D.ndevice;
  v.val_bytes = keep_vals ? D.val_bytes : 0;
  v.rowptr = out->rowptr;
  v.colidx = out->colidx;
  v.vals = out->vals;
  v.rowbits = out->rowbits;
  v.seg_row = out->seg_row;
  v.nseg = (int32_t)nseg;
  v.blk_seg = out->blk_seg;
  v.nblk = (int32_t)nblk;
  v.mid_row = out->mid_row;
  v.nmid = (int32_t)nmid;
  v.giant_row = out->giant_row;
  v.ngiant = (int32_t)ngiant;
  v.gchunk_row = out->gchunk_row;
  v.gchunk_edge = out->gchunk_edge;
  v.gterm_off = out->gterm_off;
  v.ngchunk = (int32_t)h_gcr.size();
  v.giant_edges = h_gto.back();
  v.short_row = short_row;
  v.nmid_long = nmid_long;
  v.umid_row = out->umid_row;
  v.numid = (int32_t)numid;
  v.numid_long = numid_long;
  v.tile_min_row = tile_split;
  v.gchunk_state = out->gchunk_state;
  v.edges_blk = (int64_t)h_class[0];
  v.edges_wave16 = (int64_t)h_class[1];
  v.edges_wave = (int64_t)h_class[2];
  v.rows_keep_stream = own_wave != nullptr ? 1 : 0;
  v.cold_from = 0;
  v.hot_base = 0;
  v.hot_len = D.ndevice;
  v.hot_slices = 1;
  v.hot_stride = 0;
  if (D.layout == GM_LAYOUT_DEGREE && D.nshards > 1) {  // rank k lives in slice k % G at position k / G
    v.hot_slices = D.nshards;
    v.hot_stride = D.ndevice / D.nshards;
    v.hot_len = v.hot_stride;
  }
  return GM_OK;
}

// ---- column tiles (graphmat_hip.h: gm_graph_tile) ------------------------------------------------
// tile of every sorted edge: the tile whose device-id range holds its column; 255 = not tiled (short row)
struct TileBases { int32_t b[GM_MAX_TILES + 1]; };
__global__ void __launch_bounds__(kT)
k_tile_keys(const uint64_t* __restrict__ keys, int64_t n, const int64_t* __restrict__ rowptr, int short_row,
            const int32_t* __restrict__ dev_of_native, TileBases bases, int ntiles, uint8_t* __restrict__ tkey, uint32_t* __restrict__ pos) {
  const int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (k >= n) return;
  const uint64_t key = keys[k];
  const int row = (int)(key >> 32);
  const int cn = (int)(uint32_t)key;
  const bool tiled = rowptr[row + 1] - rowptr[row] > short_row;
  tkey[k] = tiled ? (uint8_t)tile_of_dev(dev_of_native[cn], bases.b, ntiles) : (uint8_t)255;
  pos[k] = (uint32_t)k;
}
// first position of every tile in the tile-sorted key array (lower bounds of 0..ntiles)
__global__ void k_tile_bounds(const uint8_t* __restrict__ tkey, int64_t n, int ntiles, int64_t* __restrict__ out) {
  const int t = threadIdx.x;
  if (t > ntiles) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int)tkey[mid] < t) lo = mid + 1; else hi = mid;
  }
  out[t] = lo;
}
__global__ void __launch_bounds__(kT)
k_tile_gather(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ pos, int64_t n,
              uint64_t* __restrict__ keys_t, uint32_t* __restrict__ idx_t) {
  const int64_t j = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (j >= n) return;
  const uint32_t p = pos[j];
  keys_t[j] = keys[p];
  idx_t[j] = idx[p];
}

// ---- gm_graph_sweep: the rows of more than short_row edges that are not giant, laid out for kernels.hpp: k_spmv_sell ----------
// (graphmat_hip.h: gm_sweep_t; prototype and measurements: tools/sell_bench.hip, profiles/r05_sell_prototype*.txt)
__global__ void __launch_bounds__(kT)
k_sweep_flag(const int64_t* __restrict__ rowptr, int nrows, int short_row, unsigned char* __restrict__ flag) {
  const int r = blockIdx.x * kT + threadIdx.x;
  if (r < nrows) flag[r] = (rowptr[r + 1] - rowptr[r] > short_row) ? 1 : 0;
}
__global__ void __launch_bounds__(kT) k_sweep_unflag(const int32_t* __restrict__ list, int n, unsigned char* __restrict__ flag) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) flag[list[i]] = 0;
}
__global__ void __launch_bounds__(kT) k_sweep_iota(int32_t* __restrict__ a, int n) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ void __launch_bounds__(kT)
k_sweep_lens(const int32_t* __restrict__ rows, int n, const int64_t* __restrict__ rowptr, uint32_t* __restrict__ len) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) len[i] = (uint32_t)(rowptr[rows[i] + 1] - rowptr[rows[i]]);
}
__global__ void __launch_bounds__(kT) k_sweep_widen(const uint32_t* __restrict__ a, int n, unsigned long long* __restrict__ o) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) o[i] = a[i];
}
// lens are sorted descending: the number of entries > limit
__global__ void k_sweep_count_above(const uint32_t* __restrict__ len_desc, int n, uint32_t limit, unsigned int* __restrict__ out) {
  int lo = 0, hi = n;  // first index with len <= limit
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (len_desc[mid] > limit) lo = mid + 1; else hi = mid; }
  *out = (unsigned int)lo;
}
struct SweepSlices { int32_t b[GM_MAX_SLICES + 2]; int32_t nsub, stride, hot_words, pad_; };
// slice of device column c: the last s with b[s] <= (position of c inside its owner's range)
__device__ __forceinline__ int sweep_slice_of(const SweepSlices& sl, int nslices, int c) {
  if (sl.nsub > 1) c -= (c / sl.stride) * sl.stride;
  int lo = 0, hi = nslices;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sl.b[mid] <= c) lo = mid; else hi = mid; }
  return lo;
}
// the entry the sweep keeps for column c of slice `slice`: byte offset into the message vector, or (sharded form) GM_SWEEP_HOT | byte
// offset into the workgroup's LDS copy of the slice's busiest entries (graphmat_hip.h: gm_sweep_t.nsub)
__device__ __forceinline__ uint32_t sweep_entry_of(const SweepSlices& sl, int slice, int c) {
  if (sl.nsub <= 1) return (uint32_t)c << 2;
  const int q = c / sl.stride, j = c - q * sl.stride - sl.b[slice];
  const int slen = sl.b[slice + 1] - sl.b[slice], cap = sl.hot_words / sl.nsub;
  const int hq = slen < cap ? slen : cap;
  return j < hq ? (GM_SWEEP_HOT | ((uint32_t)(q * hq + j) << 2)) : ((uint32_t)c << 2);
}
// one wave per swept row (in length-rank order, the long rows first): its edges, keyed (set * 256 + workgroup, slice, slot), with
// the CSR position as the value; the row's edges keep their CSR order in the key stream (ranked offsets `off`)
__global__ void __launch_bounds__(kT)
k_sweep_keys(const int32_t* __restrict__ rows_ranked, int nswept, int nlong, int nsets, const unsigned long long* __restrict__ off,
             const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, SweepSlices sl, int nslices, unsigned long long* __restrict__ key,
             uint32_t* __restrict__ val, int32_t* __restrict__ row_of_slot, int32_t* __restrict__ lrow_of_slot) {
  const int r = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (r >= nswept) return;
  const int lane = threadIdx.x & 63;
  const int row = rows_ranked[r];
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  const bool is_long = r < nlong;
  const unsigned q = (unsigned)(is_long ? r : r - nlong);
  const unsigned wg = q % 256u, l = q / 256u;
  const unsigned set = l % (unsigned)nsets, slot = l / (unsigned)nsets;
  const unsigned long long vw = (unsigned long long)set * 256ull + wg;
  if (lane == 0) {
    if (is_long) lrow_of_slot[vw * (unsigned long long)GM_SWEEP_LONG_SLOTS + slot] = row;
    else row_of_slot[vw * (unsigned long long)GM_SWEEP_ACC_ROWS + slot] = row;
  }
  const unsigned long long o = off[r];
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    const int lo = sweep_slice_of(sl, nslices, colidx[e]);
    key[o + (unsigned long long)(e - e0)] = (vw << 23) | ((unsigned long long)lo << 16) | slot;
    val[o + (unsigned long long)(e - e0)] = (uint32_t)e;
  }
}
__global__ void __launch_bounds__(kT)
k_sweep_heads(const unsigned long long* __restrict__ key, int64_t n, uint32_t* __restrict__ head) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i < n) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}
// pieces of the medium rows in (set, workgroup, slice, slot) order; rowmin[(set * 256 + w) * acc + slot] = the row's first slice
__global__ void __launch_bounds__(kT)
k_sweep_pieces(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ head, const uint32_t* __restrict__ pidx_incl, int64_t n,
               uint32_t* __restrict__ piece_start, uint16_t* __restrict__ piece_slot, uint32_t* __restrict__ piece_blk, int32_t* __restrict__ blk_first,
               int nslices, int* __restrict__ rowmin) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= n || !head[i]) return;
  const uint32_t p = pidx_incl[i] - 1;
  const unsigned long long k = key[i];
  const unsigned slot = (unsigned)(k & 0xffffull), slice = (unsigned)((k >> 16) & 127ull);
  const unsigned long long vw = k >> 23;
  const uint32_t blk = (uint32_t)(vw * (unsigned long long)nslices + slice);
  piece_start[p] = (uint32_t)i;
  piece_slot[p] = (uint16_t)slot;
  piece_blk[p] = blk;
  if (i == 0 || (key[i - 1] >> 16) != (k >> 16)) blk_first[blk] = (int32_t)p;
  atomicMin(&rowmin[vw * (unsigned long long)GM_SWEEP_ACC_ROWS + slot], (int)slice);
}
// sort key of a piece inside its block: the longest first
__global__ void __launch_bounds__(kT)
k_sweep_piece_keys(const uint32_t* __restrict__ piece_start, const uint32_t* __restrict__ piece_blk, uint32_t np, unsigned long long* __restrict__ pkey,
                   uint32_t* __restrict__ pid) {
  const uint32_t p = blockIdx.x * kT + threadIdx.x;
  if (p >= np) return;
  const uint32_t len = piece_start[p + 1] - piece_start[p];
  pkey[p] = ((unsigned long long)piece_blk[p] << 16) | (unsigned long long)(0xffffu - (len > 0xffffu ? 0xffffu : len));
  pid[p] = p;
}
__global__ void __launch_bounds__(kT) k_sweep_block_groups(const int32_t* __restrict__ blk_first, int nblk, uint32_t* __restrict__ ng) {
  const int b = blockIdx.x * kT + threadIdx.x;
  if (b < nblk) ng[b] = (uint32_t)(blk_first[b + 1] - blk_first[b] + 63) / 64u;
}
// group g of block b = pieces q0 .. q0 + 63 of the block's sorted order; entries = 64 x its first (longest) piece
__global__ void __launch_bounds__(64)
k_sweep_group_sizes(const int32_t* __restrict__ blk_first, const uint32_t* __restrict__ grp_first, const uint32_t* __restrict__ sp,
                    const uint32_t* __restrict__ piece_start, unsigned long long* __restrict__ gsize, uint32_t* __restrict__ gq0, uint32_t* __restrict__ gblk,
                    const uint32_t* __restrict__ stream_first /* [nstream_blk + 1] first stream entry of the block, or null */, int nstream_blk) {
  const int b = blockIdx.x;
  const uint32_t g0 = grp_first[b], g1 = grp_first[b + 1];
  const uint32_t nm = (uint32_t)(blk_first[b + 1] - blk_first[b] + 63) / 64u;  // medium groups; behind them the block's stream groups
  for (uint32_t g = g0 + threadIdx.x; g < g1; g += 64) {
    if (g - g0 >= nm) {  // stream group j of the block: GM_STREAM_WIDTH rows of 64 of the block's short-row edges (the last one: what is left)
      const uint32_t j = g - g0 - nm;
      const uint32_t rows = (stream_first[b + 1] - stream_first[b] + 63u) / 64u;
      const uint32_t w = rows - j * GM_STREAM_WIDTH < GM_STREAM_WIDTH ? rows - j * GM_STREAM_WIDTH : GM_STREAM_WIDTH;
      gsize[g] = ((unsigned long long)w + 1ull) * 64ull;
      gq0[g] = j;
      gblk[g] = (uint32_t)b | 0x80000000u;
      continue;
    }
    const uint32_t q0 = (uint32_t)blk_first[b] + (g - g0) * 64u;
    const uint32_t p = sp[q0];
    gsize[g] = ((unsigned long long)(piece_start[p + 1] - piece_start[p]) + 1ull) * 64ull;  // (a meta row, then the longest piece's rows)
    gq0[g] = q0;
    gblk[g] = (uint32_t)b;
  }
}
__global__ void __launch_bounds__(kT) k_sweep_narrow(const unsigned long long* __restrict__ a, size_t n, uint32_t* __restrict__ o) {
  const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
  if (i < n) o[i] = (uint32_t)a[i];
}
// a group's entries: row 0 = the meta row (GM_SWEEP_PAD | width << 16 | first-piece flag << 15 | slot, slot 0x7fff: no piece),
// then transposed columns: scol[gbase + (1 + k) * 64 + lane] = byte offset of the k-th column of the lane's piece, or padding
__global__ void __launch_bounds__(kT)
k_sweep_fill(uint32_t ngroups, const uint32_t* __restrict__ gbase, const uint32_t* __restrict__ gq0, const uint32_t* __restrict__ gblk,
             const int32_t* __restrict__ blk_first, const uint32_t* __restrict__ sp, const uint32_t* __restrict__ piece_start,
             const uint16_t* __restrict__ piece_slot, const uint32_t* __restrict__ pos_sorted, const int32_t* __restrict__ colidx,
             const uint32_t* __restrict__ vals, SweepSlices sl, int nslices, const int* __restrict__ rowmin, uint32_t* __restrict__ scol,
             uint32_t* __restrict__ sval, uint32_t* __restrict__ src_pos) {
  for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    if (gblk[g] & 0x80000000u) continue;  // (a stream group: k_stream_fill)
    const uint32_t b = gblk[g], q0 = gq0[g], qe = (uint32_t)blk_first[b + 1];
    const uint32_t base = gbase[g], n = gbase[g + 1] - base;
    const uint32_t width = (n >> 6) - 1u;
    const uint32_t slice = b % (uint32_t)nslices, vw = b / (uint32_t)nslices;
    const uint32_t pad = sl.nsub > 1 ? (GM_SWEEP_PAD | GM_SWEEP_HOT) : (GM_SWEEP_PAD | ((uint32_t)sl.b[slice] << 2));
    const int lane = threadIdx.x & 63;
    const uint32_t q = q0 + lane;
    uint32_t ps = 0, len = 0;
    uint32_t meta = GM_SWEEP_PAD | (width << 16) | 0x7fffu;
    if (q < qe) {
      const uint32_t p = sp[q];
      ps = piece_start[p];
      len = piece_start[p + 1] - ps;
      const unsigned slot = piece_slot[p];
      const bool first = rowmin[(size_t)vw * GM_SWEEP_ACC_ROWS + slot] == (int)slice;
      meta = GM_SWEEP_PAD | (width << 16) | (first ? 0x8000u : 0u) | slot;
    }
    for (uint32_t j = threadIdx.x; j < n; j += kT) {
      uint32_t c = pad, v = 0u, sp_ = 0xffffffffu;
      if (j < 64) {
        c = meta;
      } else {
        const uint32_t k = (j >> 6) - 1u;
        if (k < len) {
          sp_ = pos_sorted[ps + k];
          c = sweep_entry_of(sl, (int)slice, colidx[sp_]);
          if (vals) v = vals[sp_];
        }
      }
      scol[base + j] = c;
      if (sval) sval[base + j] = v;
      if (src_pos) src_pos[base + j] = sp_;
    }
  }
}
// ---- the short rows as stream groups of the sweep (graphmat_hip.h: gm_sweep_t.nstream) ----
__global__ void __launch_bounds__(kT) k_stream_flag(const int64_t* __restrict__ rowptr, int nrows, int short_row, unsigned char* __restrict__ flag) {
  const int r = blockIdx.x * kT + threadIdx.x;
  if (r < nrows) { const int64_t l = rowptr[r + 1] - rowptr[r]; flag[r] = (l >= 1 && l <= short_row) ? 1 : 0; }
}
// one thread per short row (device order): its edges keyed (workgroup = bin % 256, slice, bin), value = the edge's index in the short rows'
// own CSR order (soff), spos[that index] = its position in the graph's CSR
__global__ void __launch_bounds__(kT)
k_stream_keys(const int32_t* __restrict__ srow, int nshort, const uint32_t* __restrict__ soff, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
              SweepSlices sl, int nslices, unsigned long long* __restrict__ key, uint32_t* __restrict__ val, uint32_t* __restrict__ spos) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= nshort) return;
  const int row = srow[i];
  const int64_t e0 = rowptr[row];
  const uint32_t o = soff[i], n = soff[i + 1] - o;
  const unsigned long long bin = o / (uint32_t)GM_STREAM_BIN, wg = bin & 255ull;
  for (uint32_t k = 0; k < n; k++) {
    const int lo = sweep_slice_of(sl, nslices, colidx[e0 + k]);
    key[o + k] = (wg << 31) | ((unsigned long long)lo << 24) | bin;
    val[o + k] = o + k;
    spos[o + k] = (uint32_t)(e0 + k);
  }
}
__device__ __forceinline__ uint32_t stream_lower_bound(const unsigned long long* __restrict__ key, uint32_t n, unsigned long long want) {
  uint32_t lo = 0, hi = n;  // first index with key >= want
  while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (key[mid] < want) lo = mid + 1; else hi = mid; }
  return lo;
}
// first[b] = first sorted entry of block b = workgroup * nslices + slice (first[nblk] = n); rows[b] = 64-entry rows the block's products take
__global__ void __launch_bounds__(kT)
k_stream_block_starts(const unsigned long long* __restrict__ key, uint32_t n, int nslices, int nblk, uint32_t* __restrict__ first) {
  const int b = blockIdx.x * kT + threadIdx.x;
  if (b > nblk) return;
  if (b == nblk) { first[b] = n; return; }
  const unsigned long long wg = (unsigned)b / (unsigned)nslices, sl = (unsigned)b % (unsigned)nslices;
  first[b] = stream_lower_bound(key, n, (wg << 31) | (sl << 24));
}
__global__ void __launch_bounds__(kT) k_stream_block_rows(const uint32_t* __restrict__ first, int nblk, uint32_t* __restrict__ rows, uint32_t* __restrict__ groups) {
  const int b = blockIdx.x * kT + threadIdx.x;
  if (b > nblk) return;
  const uint32_t r = b < nblk ? (first[b + 1] - first[b] + 63u) / 64u : 0u;
  rows[b] = r;
  groups[b] = (r + GM_STREAM_WIDTH - 1) / GM_STREAM_WIDTH;
}
__global__ void __launch_bounds__(kT) k_stream_add_groups(const uint32_t* __restrict__ groups, int nblk, uint32_t* __restrict__ ng) {
  const int b = blockIdx.x * kT + threadIdx.x;
  if (b < nblk) ng[b] += groups[b];
}
// schunk[(bin * nslices + slice) * 2] = where the bin's products of that slice start in the products stream, [.. + 1] = how many
__global__ void __launch_bounds__(kT)
k_stream_chunks(const unsigned long long* __restrict__ key, uint32_t n, int nslices, int nbins, const uint32_t* __restrict__ first, const uint32_t* __restrict__ row_base,
                uint32_t* __restrict__ schunk) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= (int64_t)nbins * nslices) return;
  const unsigned long long bin = (unsigned long long)(i / nslices), sl = (unsigned long long)(i % nslices), wg = bin & 255ull;
  const unsigned long long k0 = (wg << 31) | (sl << 24) | bin;
  const uint32_t lo = stream_lower_bound(key, n, k0), hi = stream_lower_bound(key, n, k0 + 1ull);
  const uint32_t b = (uint32_t)(wg * (unsigned long long)nslices + sl);
  schunk[i * 2] = row_base[b] * 64u + (lo - first[b]);
  schunk[i * 2 + 1] = hi - lo;
}
__global__ void __launch_bounds__(kT) k_stream_bin_rows(const uint32_t* __restrict__ soff, int nshort, int nbins, uint32_t* __restrict__ bin_row) {
  const int b = blockIdx.x * kT + threadIdx.x;
  if (b > nbins) return;
  if (b == nbins) { bin_row[b] = (uint32_t)nshort; return; }
  const uint32_t want = (uint32_t)b * (uint32_t)GM_STREAM_BIN;
  uint32_t lo = 0, hi = (uint32_t)nshort;  // first row with soff >= want
  while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (soff[mid] < want) lo = mid + 1; else hi = mid; }
  bin_row[b] = lo;
}
// the entries of the stream groups (k_sweep_fill skips them), and where each product belongs in its bin (sinv)
__global__ void __launch_bounds__(kT)
k_stream_fill(uint32_t ngroups, const uint32_t* __restrict__ gbase, const uint32_t* __restrict__ gq0, const uint32_t* __restrict__ gblk,
              const uint32_t* __restrict__ first, const uint32_t* __restrict__ row_base, const unsigned long long* __restrict__ key_sorted,
              const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ spos, const int32_t* __restrict__ colidx, const uint32_t* __restrict__ vals,
              SweepSlices sl, int nslices, uint32_t* __restrict__ scol, uint32_t* __restrict__ sval, uint32_t* __restrict__ src_pos, uint16_t* __restrict__ sinv) {
  for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    if (!(gblk[g] & 0x80000000u)) continue;
    const uint32_t b = gblk[g] & 0x7fffffffu, j = gq0[g];
    const uint32_t base = gbase[g], n = gbase[g + 1] - base, width = (n >> 6) - 1u;
    const uint32_t slice = b % (uint32_t)nslices;
    const uint32_t e_lo = first[b] + j * (GM_STREAM_WIDTH * 64u), e_end = first[b + 1];
    const uint32_t drow = row_base[b] + j * GM_STREAM_WIDTH;
    const uint32_t pad = sl.nsub > 1 ? (GM_SWEEP_PAD | GM_SWEEP_HOT) : (GM_SWEEP_PAD | ((uint32_t)sl.b[slice] << 2));
    for (uint32_t t = threadIdx.x; t < n; t += kT) {
      uint32_t c = pad, v = 0u, sp_ = 0xffffffffu;
      if (t < 64) {
        c = GM_SWEEP_PAD | 0x40000000u | (width << 16) | 0x7fffu | ((t < 32 && ((drow >> t) & 1u)) ? 0x8000u : 0u);
      } else {
        const uint32_t i = e_lo + (t - 64u);
        if (i < e_end) {
          const uint32_t k = idx_sorted[i];
          sp_ = spos[k];
          c = sweep_entry_of(sl, (int)slice, colidx[sp_]);
          if (vals) v = vals[sp_];
          const uint32_t bin = (uint32_t)(key_sorted[i] & 0xffffffull);
          sinv[(size_t)drow * 64 + (t - 64u)] = (uint16_t)(k - bin * (uint32_t)GM_STREAM_BIN);
        }
      }
      scol[base + t] = c;
      if (sval) sval[base + t] = v;
      if (src_pos) src_pos[base + t] = sp_;
    }
  }
}
// contiguous ranges of a block's groups for the 16 waves (wfirst: groups; wrow: 64-entry rows of scol), balanced by rows + 1 per group; the last waves (they fold the long
// rows of the block first: as many waves as the workgroup's long rows need lanes) get fold_share percent of an equal share
__global__ void __launch_bounds__(64)
k_sweep_wave_ranges(const uint32_t* __restrict__ grp_first, int nblk, const uint32_t* __restrict__ gbase, uint32_t* __restrict__ wfirst, uint32_t* __restrict__ wrow,
                    int fold_share, int fold_waves, int nwaves, const uint32_t* __restrict__ skip_groups /* per block: groups at its end to leave out, or null */,
                    int nskip_blk, const uint32_t* __restrict__ gblk = nullptr /* bit 31: a stream group */, int stream_weight = 100 /* percent of a medium row's cost */) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= nblk) return;
  constexpr int WS = 16;  // (entries per block in wfirst / wrow: WS + 1, whatever W)
  const int W = nwaves;
  const int FW = fold_waves;
  const uint32_t g0 = grp_first[b], g1 = grp_first[b + 1] - ((skip_groups != nullptr && b < nskip_blk) ? skip_groups[b] : 0u);
  // (a stream row is gathered like a medium row and its products are stored as well: its cost in medium rows = stream_weight percent)
  auto cost = [&](uint32_t g_) -> unsigned long long {
    const unsigned long long rows = (gbase[g_ + 1] - gbase[g_]) / 64u + 1u;
    return (gblk != nullptr && (gblk[g_] & 0x80000000u)) ? (rows * (unsigned)stream_weight + 99ull) / 100ull : rows;
  };
  unsigned long long total = 0;
  for (uint32_t g = g0; g < g1; g++) total += cost(g);
  const unsigned long long units = (unsigned long long)(W - FW) * 100ull + (unsigned long long)FW * (unsigned)fold_share;
  uint32_t g = g0;
  unsigned long long acc = 0, share = 0;
  for (int w = 0; w < W; w++) {
    wfirst[(size_t)b * (WS + 1) + w] = g;
    wrow[(size_t)b * (WS + 1) + w] = gbase[g] >> 6;
    share += w >= W - FW ? (unsigned)fold_share : 100u;
    const unsigned long long want = total * share / units;
    while (g < g1 && acc + (cost(g) + 1u) / 2u <= want) { acc += cost(g); g++; }
  }
  for (int w = W; w <= WS; w++) {  // (entry W ends the block; the entries behind it repeat the end)
    wfirst[(size_t)b * (WS + 1) + w] = g1;
    wrow[(size_t)b * (WS + 1) + w] = gbase[g1] >> 6;
  }
}
// long rows: entry e = ((set * 256 + w) * nslices + slice) * long_slots + j -> first position of that piece in the sorted keys
__global__ void __launch_bounds__(kT)
k_sweep_long_starts(const unsigned long long* __restrict__ key, int64_t n, int nslices, size_t nent, uint32_t* __restrict__ lps) {
  const size_t e = (size_t)blockIdx.x * kT + threadIdx.x;
  if (e > nent) return;
  if (e == nent) { lps[e] = (uint32_t)n; return; }
  const size_t b = e / GM_SWEEP_LONG_SLOTS;
  const unsigned long long vw = b / (size_t)nslices, slice = b % (size_t)nslices, j = e % GM_SWEEP_LONG_SLOTS;
  const unsigned long long want = (vw << 23) | (slice << 16) | j;
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (key[mid] >= want) hi = mid; else lo = mid + 1; }
  lps[e] = (uint32_t)lo;
}
__global__ void __launch_bounds__(kT)
k_sweep_long_fill(const unsigned long long* __restrict__ key_sorted, const uint32_t* __restrict__ pos_sorted, int64_t n, const int32_t* __restrict__ colidx,
                  const uint32_t* __restrict__ vals, SweepSlices sl, uint32_t* __restrict__ lcol, uint32_t* __restrict__ lval) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = pos_sorted[i];
  lcol[i] = sweep_entry_of(sl, (int)((key_sorted[i] >> 16) & 127ull), colidx[p]);
  if (lval) lval[i] = vals[p];
}
__global__ void __launch_bounds__(kT)
k_sweep_long_max(const uint32_t* __restrict__ lps, size_t nblk, unsigned int* __restrict__ out) {
  const size_t b = (size_t)blockIdx.x * kT + threadIdx.x;
  if (b < nblk) atomicMax(out, lps[(b + 1) * GM_SWEEP_LONG_SLOTS] - lps[b * GM_SWEEP_LONG_SLOTS]);
}
// The giant rows' edges for the sweep (gm_sweep_t.gcol / gdst): one wave per giant row writes, per edge, key = slice of its
// column, value = CSR position, and the edge's place in the products stream (gterm_off + position in the row); the stable
// sort by slice keeps (row, column) order inside a slice.
__global__ void __launch_bounds__(kT)
k_sweep_giant_keys(const int32_t* __restrict__ giant_row, int ngiant, const int64_t* __restrict__ gterm_off, const int64_t* __restrict__ rowptr,
                   const int32_t* __restrict__ colidx, SweepSlices sl, int nslices, uint8_t* __restrict__ key, uint32_t* __restrict__ pos,
                   uint32_t* __restrict__ dst) {
  const int gi = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (gi >= ngiant) return;
  const int lane = threadIdx.x & 63;
  const int row = giant_row[gi];
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1], o = gterm_off[gi];
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    const int lo = sweep_slice_of(sl, nslices, colidx[e]);
    key[o + (e - e0)] = (uint8_t)lo;
    pos[o + (e - e0)] = (uint32_t)e;
    dst[o + (e - e0)] = (uint32_t)(o + (e - e0));
  }
}
// (gterm_off is padded per row to multiples of 64: slots between rows carry key 255 and sort to the end)
__global__ void __launch_bounds__(kT)
k_sweep_giant_fill(const uint8_t* __restrict__ key_sorted, const uint32_t* __restrict__ pos_sorted, int64_t n, const int32_t* __restrict__ colidx,
                   const uint32_t* __restrict__ vals, SweepSlices sl, uint32_t* __restrict__ gcol, uint32_t* __restrict__ gval) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = pos_sorted[i];
  gcol[i] = sweep_entry_of(sl, (int)key_sorted[i], colidx[p]);
  if (gval) gval[i] = vals[p];
}
__global__ void k_sweep_giant_bounds(const uint8_t* __restrict__ key_sorted, int64_t n, int nslices, uint32_t* __restrict__ out) {
  const int t = threadIdx.x;
  if (t > nslices) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int)key_sorted[mid] < t) lo = mid + 1; else hi = mid; }
  out[t] = (uint32_t)lo;
}
// edge values rewritten in the CSR: the sweep's copies follow (gm_graph_sync_tile_vals)
__global__ void __launch_bounds__(kT)
k_sweep_sync_vals(const uint32_t* __restrict__ src_pos, size_t n, const uint32_t* __restrict__ vals, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
  if (i < n && src_pos[i] != 0xffffffffu) out[i] = vals[src_pos[i]];
}
static void free_sweep(gm_graph* g) {
  gm_sweep_t& S = g->sweep;
  const void* owned[] = {S.gcol, S.gval, S.gdst, S.gslice, S.gsrc_pos, S.scol, S.sval, S.gbase, S.wrow, S.wfirst, S.row_of_slot, S.lcol, S.lval, S.lps, S.lrow_of_slot, S.src_pos, S.lsrc_pos,
                         S.srow, S.soff, S.sbin_row, S.schunk, S.sinv, S.wrow_stream};
  for (const void* q : owned)
    if (q) (void)hipFree((void*)q);
  if (g->d_slice_base) (void)hipFree(g->d_slice_base);
  memset(&S, 0, sizeof(S));
  g->d_slice_base = nullptr;
}
template <class K, class V>
static int sweep_sort_pairs(K* kin, K* kout, V* vin, V* vout, size_t n, int bits, hipStream_t s) {
  DevBuf tmp;
  size_t tb = 0;
  int rc;
  GM_TRY_HIP(rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, 0, bits, s));
  if ((rc = tmp.alloc(tb + 256))) return rc;
  GM_TRY_HIP(rocprim::radix_sort_pairs(tmp.p, tb, kin, kout, vin, vout, n, 0, bits, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  return GM_OK;
}
template <class X>
static int sweep_excl_scan(X* in, X* out, size_t n, hipStream_t s) {
  DevBuf tmp;
  size_t tb = 0;
  int rc;
  GM_TRY_HIP(rocprim::exclusive_scan(nullptr, tb, in, out, X(0), n, rocprim::plus<X>(), s));
  if ((rc = tmp.alloc(tb + 256))) return rc;
  GM_TRY_HIP(rocprim::exclusive_scan(tmp.p, tb, in, out, X(0), n, rocprim::plus<X>(), s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  return GM_OK;
}
// LDS words of the long rows' stage (the rest of the pool holds a slice's hot entries): the largest block in one round where that
// leaves most of the pool to the hot set
static int sweep_stage_words(int nrows_long, int max_long_block) {
  if (nrows_long <= 0) return 64;
  int stage = (max_long_block + 63) / 64 * 64;
  if (stage > GM_SWEEP_MAX_STAGE) stage = GM_SWEEP_MAX_STAGE;
  if (stage < 1024) stage = 1024;
  return stage;
}
static int build_sweep(gm_graph* g, const CsrOwned* whole, hipStream_t s) {
  memset(&g->sweep, 0, sizeof(g->sweep));
  const int TS = g->nslices;
  const int nrows = g->desc.row_hi - g->desc.row_lo;
  if (TS < 2 || TS > GM_MAX_SLICES) return GM_OK;
  {  // the slices of the device order are reported even when no row ends up in the sweep (nrows = 0)
    DevBuf sb0;
    int rc0;
    if ((rc0 = sb0.alloc((size_t)(GM_MAX_SLICES + 2) * 4))) return rc0;
    GM_TRY_HIP(hipMemcpyAsync(sb0.p, g->slice_base, (size_t)(TS + 1) * 4, hipMemcpyHostToDevice, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    g->d_slice_base = (int32_t*)sb0.release();
    g->sweep.nslices = TS;
    g->sweep.slice_base = g->d_slice_base;
  }
  const int vb = whole->view.val_bytes;
  const int nsub = g->desc.nshards > 1 ? g->desc.nshards : 1;  // (sharded: slice_base holds positions inside an owner's range)
  if (nrows <= 0 || whole->view.nnz >= ((int64_t)1 << 32) || g->desc.ndevice >= (nsub > 1 ? (1 << 28) : (1 << 29)) || (whole->vals != nullptr && vb != 4)) return GM_OK;
  const int64_t* rowptr = (const int64_t*)whole->rowptr;
  const int32_t* colidx = (const int32_t*)whole->colidx;
  const uint32_t* vals = (const uint32_t*)whole->vals;
  int rc;
  DevBuf flag, iota, rows, cnt, len_in, len_out, ranked, tmp;
  if ((rc = flag.alloc((size_t)nrows)) || (rc = iota.alloc((size_t)nrows * 4)) || (rc = rows.alloc((size_t)nrows * 4)) || (rc = cnt.alloc(16))) return rc;
  hipLaunchKernelGGL(k_sweep_flag, dim3(grid_for(nrows)), dim3(kT), 0, s, rowptr, nrows, whole->view.short_row, flag.as<unsigned char>());
  // (giant rows keep their own passes -- k_giant_terms + the exact replay: a lane folding a 13 000-edge piece per slice would hold
  // its workgroup's slice barrier for the whole multiply)
  if (whole->view.ngiant > 0)
    hipLaunchKernelGGL(k_sweep_unflag, dim3(grid_for(whole->view.ngiant)), dim3(kT), 0, s, (const int32_t*)whole->giant_row, whole->view.ngiant, flag.as<unsigned char>());
  hipLaunchKernelGGL(k_sweep_iota, dim3(grid_for(nrows)), dim3(kT), 0, s, iota.as<int32_t>(), nrows);
  size_t tb = 0;
  GM_TRY_HIP(rocprim::select(nullptr, tb, iota.as<int32_t>(), flag.as<unsigned char>(), rows.as<int32_t>(), cnt.as<unsigned int>(), (size_t)nrows, s));
  if ((rc = tmp.alloc(tb + 256))) return rc;
  GM_TRY_HIP(rocprim::select(tmp.p, tb, iota.as<int32_t>(), flag.as<unsigned char>(), rows.as<int32_t>(), cnt.as<unsigned int>(), (size_t)nrows, s));
  unsigned int nswept = 0;
  GM_TRY_HIP(hipMemcpyAsync(&nswept, cnt.p, 4, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  iota.free(); flag.free();
  if (nswept == 0) return GM_OK;
  // rank by length (descending; stable: ties by row id), the long rows come first
  if ((rc = len_in.alloc((size_t)nswept * 4)) || (rc = len_out.alloc((size_t)nswept * 4)) || (rc = ranked.alloc((size_t)nswept * 4))) return rc;
  hipLaunchKernelGGL(k_sweep_lens, dim3(grid_for((int)nswept)), dim3(kT), 0, s, (const int32_t*)rows.as<int32_t>(), (int)nswept, rowptr, len_in.as<uint32_t>());
  tb = 0;
  GM_TRY_HIP(rocprim::radix_sort_pairs_desc(nullptr, tb, len_in.as<uint32_t>(), len_out.as<uint32_t>(), rows.as<int32_t>(), ranked.as<int32_t>(), (size_t)nswept, 0, 32, s));
  if ((rc = tmp.alloc(tb + 256))) return rc;
  GM_TRY_HIP(rocprim::radix_sort_pairs_desc(tmp.p, tb, len_in.as<uint32_t>(), len_out.as<uint32_t>(), rows.as<int32_t>(), ranked.as<int32_t>(), (size_t)nswept, 0, 32, s));
  rows.free(); len_in.free();
  // The border between medium rows (64-wide groups) and long rows (staged through LDS, one lane folds a piece).  A piece is folded
  // by ONE lane, row of entries after row: the longest medium piece of a slice is a chain that one wave has to walk while its
  // workgroup's other waves wait at the slice barrier, so it should not exceed a wave's fair share of the block -- (medium
  // entries per workgroup / 64) / 16 waves per slice, against ~L / nslices for a row of L edges: L <= medium edges / 262144 (phase
  // clocks of the kernel, tools/sweep_lib_bench.hip: with the border at 4096 the slowest wave of an RMAT-26 workgroup spends
  // 2.0 ms in its groups against 1.5 ms on average, at RMAT-24 1.2 against 0.33).  Candidates: powers of two from 256 to 4096;
  // the long rows must fit their slots (long_slots per workgroup and launch) in as many launches as the medium rows need.
  uint32_t long_limit = 4096;
  if (g_sweep_long_row > 0) {
    long_limit = (uint32_t)std::min(g_sweep_long_row, 8191);  // (13 bits for a group's width)
  } else {
    std::vector<uint32_t> hl(nswept);
    GM_TRY_HIP(hipMemcpyAsync(hl.data(), len_out.p, (size_t)nswept * 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    // hl is descending: for a border L, the long rows are the prefix with len > L
    auto rows_above = [&](uint32_t L) { size_t lo = 0, hi = nswept; while (lo < hi) { const size_t mid = (lo + hi) / 2; if (hl[mid] > L) lo = mid + 1; else hi = mid; } return lo; };
    std::vector<unsigned long long> suffix((size_t)nswept + 1, 0ull);  // edges of the rows from position i on
    for (size_t i = nswept; i-- > 0;) suffix[i] = suffix[i + 1] + hl[i];
    // launches a border needs: the medium rows must fit the accumulator slots, the long rows theirs
    auto sets_for = [&](uint32_t L) {
      const size_t nl = rows_above(L), nm = (size_t)nswept - nl;
      const size_t a = ((nm + 255) / 256 + (size_t)g_sweep_acc_limit - 1) / (size_t)g_sweep_acc_limit, b = ((nl + 255) / 256 + (size_t)g_sweep_long_limit - 1) / (size_t)g_sweep_long_limit;
      return std::max<size_t>(1, std::max(a, b));
    };
    size_t fewest = ~(size_t)0;
    for (uint32_t L = 4096; L >= 256; L >>= 1) fewest = std::min(fewest, sets_for(L));
    long_limit = 0;
    uint32_t smallest_ok = 4096;
    for (uint32_t L = 4096; L >= 256; L >>= 1) {
      if (sets_for(L) != fewest) continue;
      smallest_ok = L;
      const unsigned long long emed = suffix[rows_above(L)];
      if ((unsigned long long)L * 262144ull <= emed * (unsigned long long)g_sweep_border_factor) { long_limit = L; break; }
    }
    if (long_limit == 0) long_limit = smallest_ok;
  }
  hipLaunchKernelGGL(k_sweep_count_above, dim3(1), dim3(1), 0, s, (const uint32_t*)len_out.as<uint32_t>(), (int)nswept, long_limit, cnt.as<unsigned int>());
  unsigned int nlong = 0;
  GM_TRY_HIP(hipMemcpyAsync(&nlong, cnt.p, 4, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  const unsigned int nmed = nswept - nlong;
  const int per_wg_med = (int)((nmed + 255u) / 256u), per_wg_long = (int)((nlong + 255u) / 256u);
  int nsets = std::max(1, std::max((per_wg_med + g_sweep_acc_limit - 1) / g_sweep_acc_limit, (per_wg_long + g_sweep_long_limit - 1) / g_sweep_long_limit));
  if (nsets > 16) return GM_OK;  // (12 key bits for set * 256 + workgroup)
  // edge offsets of the ranked rows
  DevBuf l64, off;
  if ((rc = l64.alloc(((size_t)nswept + 1) * 8)) || (rc = off.alloc(((size_t)nswept + 1) * 8))) return rc;
  GM_TRY_HIP(hipMemsetAsync(l64.p, 0, ((size_t)nswept + 1) * 8, s));
  hipLaunchKernelGGL(k_sweep_widen, dim3(grid_for((int)nswept)), dim3(kT), 0, s, (const uint32_t*)len_out.as<uint32_t>(), (int)nswept, l64.as<unsigned long long>());
  if ((rc = sweep_excl_scan(l64.as<unsigned long long>(), off.as<unsigned long long>(), (size_t)nswept + 1, s))) return rc;
  unsigned long long tot_edges = 0, long_edges = 0;
  GM_TRY_HIP(hipMemcpyAsync(&tot_edges, off.as<unsigned long long>() + nswept, 8, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipMemcpyAsync(&long_edges, off.as<unsigned long long>() + nlong, 8, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  l64.free(); len_out.free();
  const int64_t nedges_all = (int64_t)tot_edges, nedges_long = (int64_t)long_edges, nedges = nedges_all - nedges_long;
  if (nedges_all <= 0 || nedges_all >= ((int64_t)1 << 32)) return GM_OK;
  const size_t nvw = (size_t)nsets * 256;
  const size_t nblk = nvw * (size_t)TS;
  DevBuf k_in, k_out, v_in, v_out, rslot, lrslot;
  if ((rc = k_in.alloc((size_t)nedges_all * 8)) || (rc = k_out.alloc((size_t)nedges_all * 8)) || (rc = v_in.alloc((size_t)nedges_all * 4)) ||
      (rc = v_out.alloc((size_t)nedges_all * 4)) || (rc = rslot.alloc(nvw * GM_SWEEP_ACC_ROWS * 4)) || (rc = lrslot.alloc(nvw * GM_SWEEP_LONG_SLOTS * 4)))
    return rc;
  SweepSlices sl;
  memset(&sl, 0, sizeof(sl));
  for (int t = 0; t <= TS; t++) sl.b[t] = g->slice_base[t];
  sl.nsub = nsub;
  sl.stride = nsub > 1 ? nrows : 0;
  sl.hot_words = GM_SWEEP_POOL - 64;  // (settled below, once the long rows' largest block -- the stage it needs -- is known)
  GM_TRY_HIP(hipMemsetAsync(rslot.p, 0xff, nvw * GM_SWEEP_ACC_ROWS * 4, s));
  GM_TRY_HIP(hipMemsetAsync(lrslot.p, 0xff, nvw * GM_SWEEP_LONG_SLOTS * 4, s));
  hipLaunchKernelGGL(k_sweep_keys, dim3((nswept + (kT / 64) - 1) / (kT / 64)), dim3(kT), 0, s, (const int32_t*)ranked.as<int32_t>(), (int)nswept, (int)nlong, nsets,
                     (const unsigned long long*)off.as<unsigned long long>(), rowptr, colidx, sl, TS, k_in.as<unsigned long long>(), v_in.as<uint32_t>(),
                     rslot.as<int32_t>(), lrslot.as<int32_t>());
  GM_TRY_HIP(hipGetLastError());
  off.free(); ranked.free();
  // stable sorts: inside (set, workgroup, slice, slot) the edges keep their ascending native column order
  const int key_bits = 23 + bits_for((uint32_t)nvw);
  if (nedges_long > 0 && (rc = sweep_sort_pairs(k_in.as<unsigned long long>(), k_out.as<unsigned long long>(), v_in.as<uint32_t>(), v_out.as<uint32_t>(), (size_t)nedges_long, key_bits, s))) return rc;
  if (nedges > 0 && (rc = sweep_sort_pairs(k_in.as<unsigned long long>() + nedges_long, k_out.as<unsigned long long>() + nedges_long, v_in.as<uint32_t>() + nedges_long,
                                            v_out.as<uint32_t>() + nedges_long, (size_t)nedges, key_bits, s)))
    return rc;
  k_in.free(); v_in.free();
  gm_sweep_t& S = g->sweep;
  const bool keep_pos = vals != nullptr;
  // ---- the long rows
  DevBuf lps, lcol, lval, lpos;
  unsigned int max_block = 0;
  {
    const size_t nent = nblk * GM_SWEEP_LONG_SLOTS;
    if ((rc = lps.alloc((nent + 1) * 4)) || (rc = lcol.alloc(((size_t)nedges_long + 64) * 4))) return rc;
    if (vals && (rc = lval.alloc(((size_t)nedges_long + 64) * 4))) return rc;
    hipLaunchKernelGGL(k_sweep_long_starts, dim3(grid_for((int64_t)nent + 1)), dim3(kT), 0, s, (const unsigned long long*)k_out.as<unsigned long long>(), nedges_long, TS, nent,
                       lps.as<uint32_t>());
    GM_TRY_HIP(hipMemsetAsync(cnt.p, 0, 4, s));
    hipLaunchKernelGGL(k_sweep_long_max, dim3(grid_for((int64_t)nblk)), dim3(kT), 0, s, (const uint32_t*)lps.as<uint32_t>(), nblk, cnt.as<unsigned int>());
    GM_TRY_HIP(hipMemcpyAsync(&max_block, cnt.p, 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    // the LDS pool is shared by a slice's hot entries and the long rows' stage (engine.hpp: multiply_out_swept chooses the same stage)
    sl.hot_words = GM_SWEEP_POOL - sweep_stage_words((int)nlong, (int)max_block);
    if (nedges_long > 0)
      hipLaunchKernelGGL(k_sweep_long_fill, dim3(grid_for(nedges_long)), dim3(kT), 0, s, (const unsigned long long*)k_out.as<unsigned long long>(), (const uint32_t*)v_out.as<uint32_t>(),
                         nedges_long, colidx, vals, sl, lcol.as<uint32_t>(), vals ? lval.as<uint32_t>() : (uint32_t*)nullptr);
    if (keep_pos && nedges_long > 0) {
      if ((rc = lpos.alloc((size_t)nedges_long * 4))) return rc;
      GM_TRY_HIP(hipMemcpyAsync(lpos.p, v_out.p, (size_t)nedges_long * 4, hipMemcpyDeviceToDevice, s));
    }
    GM_TRY_HIP(hipStreamSynchronize(s));
  }
  // ---- the medium rows: pieces, groups of 64 pieces of similar length, transposed entries
  DevBuf scol, sval, spos, gbase, wrow, wfirst;
  DevBuf st_srow, st_soff, st_keys, st_idx, st_spos, st_first, st_rowbase, st_binrow, st_chunk, st_inv, st_sgroups, st_wrow, st_wfirst;  // the short rows' stream groups (below)
  uint32_t st_nshort = 0, st_edges = 0, st_rows_total = 0;
  int st_nbins = 0;
  bool st_built = false;
  uint32_t ngroups = 0;
  unsigned long long nentries = 0;
  if (nedges > 0) {
    const unsigned long long* mkey = k_out.as<unsigned long long>() + nedges_long;
    const uint32_t* mpos = v_out.as<uint32_t>() + nedges_long;
    DevBuf head, pidx;
    if ((rc = head.alloc((size_t)nedges * 4)) || (rc = pidx.alloc((size_t)nedges * 4))) return rc;
    hipLaunchKernelGGL(k_sweep_heads, dim3(grid_for(nedges)), dim3(kT), 0, s, mkey, nedges, head.as<uint32_t>());
    tb = 0;
    GM_TRY_HIP(rocprim::inclusive_scan(nullptr, tb, head.as<uint32_t>(), pidx.as<uint32_t>(), (size_t)nedges, rocprim::plus<uint32_t>(), s));
    if ((rc = tmp.alloc(tb + 256))) return rc;
    GM_TRY_HIP(rocprim::inclusive_scan(tmp.p, tb, head.as<uint32_t>(), pidx.as<uint32_t>(), (size_t)nedges, rocprim::plus<uint32_t>(), s));
    uint32_t npieces = 0;
    GM_TRY_HIP(hipMemcpyAsync(&npieces, pidx.as<uint32_t>() + (nedges - 1), 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    DevBuf pstart, pslot16, pblk, bfirst, rowmin;
    if ((rc = pstart.alloc(((size_t)npieces + 1) * 4)) || (rc = pslot16.alloc(((size_t)npieces + 1) * 2)) || (rc = pblk.alloc(((size_t)npieces + 1) * 4)) ||
        (rc = bfirst.alloc((nblk + 1) * 4)) || (rc = rowmin.alloc(nvw * GM_SWEEP_ACC_ROWS * 4)))
      return rc;
    GM_TRY_HIP(hipMemsetAsync(bfirst.p, 0xff, (nblk + 1) * 4, s));
    GM_TRY_HIP(hipMemsetAsync(rowmin.p, 0x7f, nvw * GM_SWEEP_ACC_ROWS * 4, s));
    hipLaunchKernelGGL(k_sweep_pieces, dim3(grid_for(nedges)), dim3(kT), 0, s, mkey, (const uint32_t*)head.as<uint32_t>(), (const uint32_t*)pidx.as<uint32_t>(), nedges,
                       pstart.as<uint32_t>(), pslot16.as<uint16_t>(), pblk.as<uint32_t>(), bfirst.as<int32_t>(), TS, rowmin.as<int>());
    const uint32_t ne32 = (uint32_t)nedges;
    GM_TRY_HIP(hipMemcpyAsync(pstart.as<uint32_t>() + npieces, &ne32, 4, hipMemcpyHostToDevice, s));
    {
      std::vector<int32_t> h(nblk + 1);
      GM_TRY_HIP(hipMemcpyAsync(h.data(), bfirst.p, (nblk + 1) * 4, hipMemcpyDeviceToHost, s));
      GM_TRY_HIP(hipStreamSynchronize(s));
      h[nblk] = (int32_t)npieces;
      for (int64_t b = (int64_t)nblk - 1; b >= 0; b--) if (h[b] < 0) h[b] = h[b + 1];
      GM_TRY_HIP(hipMemcpyAsync(bfirst.p, h.data(), (nblk + 1) * 4, hipMemcpyHostToDevice, s));
      GM_TRY_HIP(hipStreamSynchronize(s));
    }
    head.free(); pidx.free();
    // the pieces of a block, longest first (stable: equal lengths keep their slot order)
    DevBuf pk_in, pk_out, pid_in, sp;
    if ((rc = pk_in.alloc((size_t)npieces * 8)) || (rc = pk_out.alloc((size_t)npieces * 8)) || (rc = pid_in.alloc((size_t)npieces * 4)) || (rc = sp.alloc((size_t)npieces * 4))) return rc;
    hipLaunchKernelGGL(k_sweep_piece_keys, dim3(grid_for((int64_t)npieces)), dim3(kT), 0, s, (const uint32_t*)pstart.as<uint32_t>(), (const uint32_t*)pblk.as<uint32_t>(), npieces,
                       pk_in.as<unsigned long long>(), pid_in.as<uint32_t>());
    if ((rc = sweep_sort_pairs(pk_in.as<unsigned long long>(), pk_out.as<unsigned long long>(), pid_in.as<uint32_t>(), sp.as<uint32_t>(), (size_t)npieces,
                               16 + bits_for((uint32_t)nblk), s)))
      return rc;
    pk_in.free(); pk_out.free(); pid_in.free(); pblk.free();
    DevBuf ng, grp_first;
    if ((rc = ng.alloc((nblk + 1) * 4)) || (rc = grp_first.alloc((nblk + 1) * 4))) return rc;
    GM_TRY_HIP(hipMemsetAsync(ng.p, 0, (nblk + 1) * 4, s));
    hipLaunchKernelGGL(k_sweep_block_groups, dim3(grid_for((int64_t)nblk)), dim3(kT), 0, s, (const int32_t*)bfirst.as<int32_t>(), (int)nblk, ng.as<uint32_t>());
    // ---- the short rows (1 .. short_row edges) as STREAM groups behind the medium groups of the first launch's blocks (gm_sweep_t.nstream):
    // listed in device order, cut into bins of ~GM_STREAM_BIN edges, bin b -> workgroup b % 256; their edges sorted (workgroup, slice, bin, row,
    // column) -- stable, so a row's edges keep their ascending native column order inside every (bin, slice) chunk
    if (g_sweep_stream != 0) {  // (shards too: the entries carry the build-time hot-set decision like every other entry of a sharded structure)
      const int nsblk = 256 * TS;
      DevBuf sflag, siota, scnt, slen, kin, vin, srows_;
      DevBuf& sgroups = st_sgroups;
      if ((rc = sflag.alloc((size_t)nrows)) || (rc = siota.alloc((size_t)nrows * 4)) || (rc = st_srow.alloc((size_t)nrows * 4 + 4)) || (rc = scnt.alloc(16))) return rc;
      hipLaunchKernelGGL(k_stream_flag, dim3(grid_for(nrows)), dim3(kT), 0, s, rowptr, nrows, whole->view.short_row, sflag.as<unsigned char>());
      hipLaunchKernelGGL(k_sweep_iota, dim3(grid_for(nrows)), dim3(kT), 0, s, siota.as<int32_t>(), nrows);
      tb = 0;
      GM_TRY_HIP(rocprim::select(nullptr, tb, siota.as<int32_t>(), sflag.as<unsigned char>(), st_srow.as<int32_t>(), scnt.as<unsigned int>(), (size_t)nrows, s));
      if ((rc = tmp.alloc(tb + 256))) return rc;
      GM_TRY_HIP(rocprim::select(tmp.p, tb, siota.as<int32_t>(), sflag.as<unsigned char>(), st_srow.as<int32_t>(), scnt.as<unsigned int>(), (size_t)nrows, s));
      GM_TRY_HIP(hipMemcpyAsync(&st_nshort, scnt.p, 4, hipMemcpyDeviceToHost, s));
      GM_TRY_HIP(hipStreamSynchronize(s));
      sflag.free(); siota.free();
      if (st_nshort > 0) {
        if ((rc = slen.alloc(((size_t)st_nshort + 1) * 4)) || (rc = st_soff.alloc(((size_t)st_nshort + 1) * 4))) return rc;
        GM_TRY_HIP(hipMemsetAsync(slen.p, 0, ((size_t)st_nshort + 1) * 4, s));
        hipLaunchKernelGGL(k_sweep_lens, dim3(grid_for((int)st_nshort)), dim3(kT), 0, s, (const int32_t*)st_srow.as<int32_t>(), (int)st_nshort, rowptr, slen.as<uint32_t>());
        if ((rc = sweep_excl_scan(slen.as<uint32_t>(), st_soff.as<uint32_t>(), (size_t)st_nshort + 1, s))) return rc;
        GM_TRY_HIP(hipMemcpyAsync(&st_edges, st_soff.as<uint32_t>() + st_nshort, 4, hipMemcpyDeviceToHost, s));
        GM_TRY_HIP(hipStreamSynchronize(s));
        slen.free();
      }
      st_nbins = (int)(((unsigned long long)st_edges + GM_STREAM_BIN - 1) / GM_STREAM_BIN);
      if (st_edges > 0 && st_nbins < (1 << 24)) {
        if ((rc = kin.alloc((size_t)st_edges * 8)) || (rc = st_keys.alloc((size_t)st_edges * 8)) || (rc = vin.alloc((size_t)st_edges * 4)) ||
            (rc = st_idx.alloc((size_t)st_edges * 4)) || (rc = st_spos.alloc((size_t)st_edges * 4)))
          return rc;
        hipLaunchKernelGGL(k_stream_keys, dim3(grid_for((int)st_nshort)), dim3(kT), 0, s, (const int32_t*)st_srow.as<int32_t>(), (int)st_nshort, (const uint32_t*)st_soff.as<uint32_t>(),
                           rowptr, colidx, sl, TS, kin.as<unsigned long long>(), vin.as<uint32_t>(), st_spos.as<uint32_t>());
        GM_TRY_HIP(hipGetLastError());
        if ((rc = sweep_sort_pairs(kin.as<unsigned long long>(), st_keys.as<unsigned long long>(), vin.as<uint32_t>(), st_idx.as<uint32_t>(), (size_t)st_edges, 39, s))) return rc;
        kin.free(); vin.free();
        if ((rc = st_first.alloc(((size_t)nsblk + 2) * 4)) || (rc = srows_.alloc(((size_t)nsblk + 2) * 4)) || (rc = sgroups.alloc(((size_t)nsblk + 2) * 4)) ||
            (rc = st_rowbase.alloc(((size_t)nsblk + 2) * 4)) || (rc = st_binrow.alloc(((size_t)st_nbins + 2) * 4)) || (rc = st_chunk.alloc((size_t)st_nbins * TS * 8 + 64)))
          return rc;
        hipLaunchKernelGGL(k_stream_block_starts, dim3(grid_for(nsblk + 1)), dim3(kT), 0, s, (const unsigned long long*)st_keys.as<unsigned long long>(), st_edges, TS, nsblk,
                           st_first.as<uint32_t>());
        hipLaunchKernelGGL(k_stream_block_rows, dim3(grid_for(nsblk + 1)), dim3(kT), 0, s, (const uint32_t*)st_first.as<uint32_t>(), nsblk, srows_.as<uint32_t>(), sgroups.as<uint32_t>());
        if ((rc = sweep_excl_scan(srows_.as<uint32_t>(), st_rowbase.as<uint32_t>(), (size_t)nsblk + 1, s))) return rc;
        GM_TRY_HIP(hipMemcpyAsync(&st_rows_total, st_rowbase.as<uint32_t>() + nsblk, 4, hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(k_stream_add_groups, dim3(grid_for(nsblk)), dim3(kT), 0, s, (const uint32_t*)sgroups.as<uint32_t>(), nsblk, ng.as<uint32_t>());
        hipLaunchKernelGGL(k_stream_bin_rows, dim3(grid_for(st_nbins + 1)), dim3(kT), 0, s, (const uint32_t*)st_soff.as<uint32_t>(), (int)st_nshort, st_nbins, st_binrow.as<uint32_t>());
        hipLaunchKernelGGL(k_stream_chunks, dim3(grid_for((int64_t)st_nbins * TS)), dim3(kT), 0, s, (const unsigned long long*)st_keys.as<unsigned long long>(), st_edges, TS, st_nbins,
                           (const uint32_t*)st_first.as<uint32_t>(), (const uint32_t*)st_rowbase.as<uint32_t>(), st_chunk.as<uint32_t>());
        GM_TRY_HIP(hipGetLastError());
        GM_TRY_HIP(hipStreamSynchronize(s));
        if ((rc = st_inv.alloc((size_t)st_rows_total * 64 * 2 + 64))) return rc;
        GM_TRY_HIP(hipMemsetAsync(st_inv.p, 0xff, (size_t)st_rows_total * 64 * 2 + 64, s));
        st_built = true;
      }
    }
    if ((rc = sweep_excl_scan(ng.as<uint32_t>(), grp_first.as<uint32_t>(), nblk + 1, s))) return rc;
    GM_TRY_HIP(hipMemcpyAsync(&ngroups, grp_first.as<uint32_t>() + nblk, 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    DevBuf gsize, gb64, gq0, gblk;
    if ((rc = gsize.alloc(((size_t)ngroups + 1) * 8)) || (rc = gb64.alloc(((size_t)ngroups + 1) * 8)) || (rc = gq0.alloc(((size_t)ngroups + 1) * 4)) ||
        (rc = gblk.alloc(((size_t)ngroups + 1) * 4)) || (rc = gbase.alloc(((size_t)ngroups + 1) * 4)))
      return rc;
    GM_TRY_HIP(hipMemsetAsync(gsize.p, 0, ((size_t)ngroups + 1) * 8, s));
    hipLaunchKernelGGL(k_sweep_group_sizes, dim3((unsigned)nblk), dim3(64), 0, s, (const int32_t*)bfirst.as<int32_t>(), (const uint32_t*)grp_first.as<uint32_t>(),
                       (const uint32_t*)sp.as<uint32_t>(), (const uint32_t*)pstart.as<uint32_t>(), gsize.as<unsigned long long>(), gq0.as<uint32_t>(), gblk.as<uint32_t>(),
                       st_built ? (const uint32_t*)st_first.as<uint32_t>() : (const uint32_t*)nullptr, 256 * TS);
    if ((rc = sweep_excl_scan(gsize.as<unsigned long long>(), gb64.as<unsigned long long>(), (size_t)ngroups + 1, s))) return rc;
    GM_TRY_HIP(hipMemcpyAsync(&nentries, gb64.as<unsigned long long>() + ngroups, 8, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    if (nentries >= (1ull << 32)) return GM_OK;  // (32-bit entry positions; the graph keeps its tile passes)
    hipLaunchKernelGGL(k_sweep_narrow, dim3(grid_for((int64_t)ngroups + 1)), dim3(kT), 0, s, (const unsigned long long*)gb64.as<unsigned long long>(), (size_t)ngroups + 1,
                       gbase.as<uint32_t>());
    gsize.free(); gb64.free();
    if ((rc = scol.alloc(((size_t)nentries + 64 * 64) * 4)) || (rc = wrow.alloc(nblk * 17 * 4)) || (rc = wfirst.alloc(nblk * 17 * 4))) return rc;
    if (vals && ((rc = sval.alloc(((size_t)nentries + 64 * 64) * 4)) || (rc = spos.alloc(((size_t)nentries + 64) * 4)))) return rc;
    hipLaunchKernelGGL(k_sweep_fill, dim3(65536), dim3(kT), 0, s, ngroups, (const uint32_t*)gbase.as<uint32_t>(), (const uint32_t*)gq0.as<uint32_t>(),
                       (const uint32_t*)gblk.as<uint32_t>(), (const int32_t*)bfirst.as<int32_t>(), (const uint32_t*)sp.as<uint32_t>(), (const uint32_t*)pstart.as<uint32_t>(),
                       (const uint16_t*)pslot16.as<uint16_t>(), mpos, colidx, vals, sl, TS, (const int*)rowmin.as<int>(), scol.as<uint32_t>(),
                       vals ? sval.as<uint32_t>() : (uint32_t*)nullptr, vals ? spos.as<uint32_t>() : (uint32_t*)nullptr);
    if (st_built)
      hipLaunchKernelGGL(k_stream_fill, dim3(65536), dim3(kT), 0, s, ngroups, (const uint32_t*)gbase.as<uint32_t>(), (const uint32_t*)gq0.as<uint32_t>(),
                         (const uint32_t*)gblk.as<uint32_t>(), (const uint32_t*)st_first.as<uint32_t>(), (const uint32_t*)st_rowbase.as<uint32_t>(),
                         (const unsigned long long*)st_keys.as<unsigned long long>(), (const uint32_t*)st_idx.as<uint32_t>(), (const uint32_t*)st_spos.as<uint32_t>(), colidx, vals, sl, TS,
                         scol.as<uint32_t>(), vals ? sval.as<uint32_t>() : (uint32_t*)nullptr, vals ? spos.as<uint32_t>() : (uint32_t*)nullptr, st_inv.as<uint16_t>());
    hipLaunchKernelGGL(k_sweep_wave_ranges, dim3((unsigned)((nblk + 63) / 64)), dim3(64), 0, s, (const uint32_t*)grp_first.as<uint32_t>(), (int)nblk,
                       (const uint32_t*)gbase.as<uint32_t>(), wfirst.as<uint32_t>(), wrow.as<uint32_t>(), nlong > 0 ? g_sweep_fold_share : 100,
                       (int)std::min(8, (((per_wg_long + nsets - 1) / nsets) + 63) / 64), (g_sweep_waves == 12 || g_sweep_waves == 8) ? g_sweep_waves : 16,
                       st_built ? (const uint32_t*)st_sgroups.as<uint32_t>() : (const uint32_t*)nullptr, 256 * TS);
    if (st_built) {  // the ranges WITH the stream groups (gm_sweep_t.wrow_stream): the kernel form that stores their products walks these
      if ((rc = st_wrow.alloc(nblk * 17 * 4)) || (rc = st_wfirst.alloc(nblk * 17 * 4))) return rc;
      hipLaunchKernelGGL(k_sweep_wave_ranges, dim3((unsigned)((nblk + 63) / 64)), dim3(64), 0, s, (const uint32_t*)grp_first.as<uint32_t>(), (int)nblk,
                         (const uint32_t*)gbase.as<uint32_t>(), st_wfirst.as<uint32_t>(), st_wrow.as<uint32_t>(), nlong > 0 ? g_sweep_fold_share : 100,
                         (int)std::min(8, (((per_wg_long + nsets - 1) / nsets) + 63) / 64), 16, (const uint32_t*)nullptr, 0, (const uint32_t*)gblk.as<uint32_t>(), g_sweep_stream_weight);
    }
    GM_TRY_HIP(hipGetLastError());
    GM_TRY_HIP(hipStreamSynchronize(s));
  } else {
    // no medium row: empty blocks for every wave
    // (the kernel requests a wave's first batch of entries -- and their values -- before it knows that the stream is empty: both arrays exist)
    if ((rc = gbase.alloc(64)) || (rc = wrow.alloc(nblk * 17 * 4)) || (rc = wfirst.alloc(nblk * 17 * 4)) || (rc = scol.alloc(64 * 64 * 4))) return rc;
    GM_TRY_HIP(hipMemsetAsync(scol.p, 0xff, 64 * 64 * 4, s));
    if (vals) {
      if ((rc = sval.alloc(64 * 64 * 4))) return rc;
      GM_TRY_HIP(hipMemsetAsync(sval.p, 0, 64 * 64 * 4, s));
    }
    GM_TRY_HIP(hipMemsetAsync(gbase.p, 0, 64, s));
    GM_TRY_HIP(hipMemsetAsync(wfirst.p, 0, nblk * 17 * 4, s));
    GM_TRY_HIP(hipMemsetAsync(wrow.p, 0, nblk * 17 * 4, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
  }
  // ---- the giant rows' edges by slice: the sweep gathers their messages with its hot sets and writes the products where the
  // giant rows' fold passes expect them (gterm_off), so those passes need not gather at all
  DevBuf gcol, gval, gdst, gslice, gpos;
  int64_t ngiant_edges = 0;
  if (whole->view.ngiant > 0 && whole->view.giant_edges > 0 && whole->view.giant_edges < ((int64_t)1 << 32)) {
    const int64_t ge = whole->view.giant_edges;  // (slots, rows padded to multiples of 64)
    DevBuf gk_in, gk_out, gp_in, gp_out, gd_in, gd_out;
    if ((rc = gk_in.alloc((size_t)ge)) || (rc = gk_out.alloc((size_t)ge)) || (rc = gp_in.alloc((size_t)ge * 4)) || (rc = gp_out.alloc((size_t)ge * 4)) ||
        (rc = gd_in.alloc((size_t)ge * 4)) || (rc = gd_out.alloc((size_t)ge * 4)) || (rc = gslice.alloc((GM_MAX_SLICES + 2) * 4)))
      return rc;
    GM_TRY_HIP(hipMemsetAsync(gk_in.p, 0xff, (size_t)ge, s));
    GM_TRY_HIP(hipMemsetAsync(gp_in.p, 0, (size_t)ge * 4, s));
    GM_TRY_HIP(hipMemsetAsync(gd_in.p, 0, (size_t)ge * 4, s));
    hipLaunchKernelGGL(k_sweep_giant_keys, dim3((whole->view.ngiant + (kT / 64) - 1) / (kT / 64)), dim3(kT), 0, s, (const int32_t*)whole->giant_row, whole->view.ngiant,
                       (const int64_t*)whole->gterm_off, rowptr, colidx, sl, TS, gk_in.as<uint8_t>(), gp_in.as<uint32_t>(), gd_in.as<uint32_t>());
    // two stable sorts by the same keys: positions and destinations travel together
    if ((rc = sweep_sort_pairs(gk_in.as<uint8_t>(), gk_out.as<uint8_t>(), gp_in.as<uint32_t>(), gp_out.as<uint32_t>(), (size_t)ge, 8, s))) return rc;
    if ((rc = sweep_sort_pairs(gk_in.as<uint8_t>(), gk_out.as<uint8_t>(), gd_in.as<uint32_t>(), gd_out.as<uint32_t>(), (size_t)ge, 8, s))) return rc;
    hipLaunchKernelGGL(k_sweep_giant_bounds, dim3(1), dim3(256), 0, s, (const uint8_t*)gk_out.as<uint8_t>(), ge, TS, gslice.as<uint32_t>());
    uint32_t nreal = 0;
    GM_TRY_HIP(hipMemcpyAsync(&nreal, gslice.as<uint32_t>() + TS, 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    ngiant_edges = (int64_t)nreal;
    if ((rc = gcol.alloc(((size_t)nreal + 64) * 4))) return rc;
    if (vals && (rc = gval.alloc(((size_t)nreal + 64) * 4))) return rc;
    if (nreal > 0)
      hipLaunchKernelGGL(k_sweep_giant_fill, dim3(grid_for((int64_t)nreal)), dim3(kT), 0, s, (const uint8_t*)gk_out.as<uint8_t>(), (const uint32_t*)gp_out.as<uint32_t>(), (int64_t)nreal,
                         colidx, vals, sl, gcol.as<uint32_t>(), vals ? gval.as<uint32_t>() : (uint32_t*)nullptr);
    GM_TRY_HIP(hipGetLastError());
    GM_TRY_HIP(hipStreamSynchronize(s));
    gdst.p = gd_out.release();
    if (keep_pos) gpos.p = gp_out.release();
  }
  S.ngiant_edges = ngiant_edges;
  S.gcol = (const uint32_t*)gcol.release(); S.gval = (const uint32_t*)gval.release(); S.gdst = (const uint32_t*)gdst.release();
  S.gslice = (const uint32_t*)gslice.release(); S.gsrc_pos = (const uint32_t*)gpos.release();
  S.nrows = (int32_t)nswept; S.nrows_long = (int32_t)nlong; S.nsets = nsets; S.nslices = TS;
  S.acc_rows = GM_SWEEP_ACC_ROWS; S.long_slots = GM_SWEEP_LONG_SLOTS; S.max_long_block = (int32_t)max_block; S.val_bytes = vals ? 4 : 0;
  S.short_row = whole->view.short_row; S.long_row = (int32_t)long_limit;
  S.nedges = nedges; S.nedges_long = nedges_long; S.nentries = (int64_t)nentries; S.ngroups = (int64_t)ngroups;
  S.scol = (const uint32_t*)scol.release(); S.sval = (const uint32_t*)sval.release(); S.gbase = (const uint32_t*)gbase.release();
  S.wrow = (const uint32_t*)wrow.release(); S.wfirst = (const uint32_t*)wfirst.release(); S.row_of_slot = (const int32_t*)rslot.release();
  S.lcol = (const uint32_t*)lcol.release(); S.lval = (const uint32_t*)lval.release(); S.lps = (const uint32_t*)lps.release();
  S.lrow_of_slot = (const int32_t*)lrslot.release(); S.slice_base = g->d_slice_base;
  S.src_pos = (const uint32_t*)spos.release(); S.lsrc_pos = (const uint32_t*)lpos.release();
  S.nsub = nsub; S.stride = sl.stride; S.hot_words = sl.hot_words;
  S.waves = (g_sweep_waves == 12 || g_sweep_waves == 8) ? g_sweep_waves : 16;
  if (st_built) {
    S.nstream = (int64_t)st_edges; S.nstream_slots = (int64_t)st_rows_total * 64; S.nshort_rows = (int32_t)st_nshort; S.nbins = st_nbins;
    S.bin_cap = GM_STREAM_BIN; S.stream_width = GM_STREAM_WIDTH;
    S.srow = (const int32_t*)st_srow.release(); S.soff = (const uint32_t*)st_soff.release(); S.sbin_row = (const uint32_t*)st_binrow.release();
    S.schunk = (const uint32_t*)st_chunk.release(); S.sinv = (const uint16_t*)st_inv.release(); S.wrow_stream = (const uint32_t*)st_wrow.release();
  }
  return GM_OK;
}

// ---- the short rows of a graph without skew as a column-blocked stream (graphmat_hip.h: gm_blocked_t; kernels.hpp: k_spmv_blocked) ----
int g_blocked_rows = 0;  // 0: automatic, 1: whenever the graph has slices, -1: never (gm_set_option("blocked_rows"))
__global__ void __launch_bounds__(kT) k_blocked_flag(const int64_t* __restrict__ rowptr, int nrows, int short_row, unsigned char* __restrict__ flag) {
  const int r = blockIdx.x * kT + threadIdx.x;
  if (r < nrows) { const int64_t l = rowptr[r + 1] - rowptr[r]; flag[r] = (l >= 1 && l <= short_row) ? 1 : 0; }
}
// The short rows are dealt over the blocks in runs of 64 consecutive rows (run j -> block j % nblk): the device order is degree-ranked inside
// a slice, so blocks of CONSECUTIVE rows would hold up to twice as many entries at the top of a slice as at its bottom -- and workgroups
// that walk the slices in step wait for the heaviest block of the pass (measured on the uniform 2^26 graph: 14.3 ms against 11 for the
// prototype's equal blocks).  A run's results are 256 contiguous bytes of y.
__global__ void __launch_bounds__(kT) k_blocked_deal(const int32_t* __restrict__ rows, int ns, int nblk, int32_t* __restrict__ dealt) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= ns) return;
  const int j = i >> 6;
  dealt[(size_t)(j % nblk) * GM_BLOCKED_ROWS + (size_t)(j / nblk) * 64 + (i & 63)] = rows[i];
}
// (slots: the blocks' row slots, -1 = no row)
__global__ void __launch_bounds__(kT) k_blocked_lens(const int32_t* __restrict__ slots, int n, const int64_t* __restrict__ rowptr, unsigned long long* __restrict__ len) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i < n) len[i] = slots[i] >= 0 ? (unsigned long long)(rowptr[slots[i] + 1] - rowptr[slots[i]]) : 0ull;
}
// key = block | slice (7 bits) | row slot inside the block (15 bits); value = the edge's position in the CSR
__global__ void __launch_bounds__(kT)
k_blocked_keys(const int32_t* __restrict__ slots, int n, const unsigned long long* __restrict__ off, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
               SweepSlices sl, int TS, unsigned long long* __restrict__ key, uint32_t* __restrict__ pos) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const int r = slots[i];
  if (r < 0) return;
  const int64_t a = rowptr[r], e = rowptr[r + 1];
  const unsigned long long hi = ((unsigned long long)(i / GM_BLOCKED_ROWS) << 22), lo = (unsigned long long)(i % GM_BLOCKED_ROWS);
  unsigned long long o = off[i];
  for (int64_t k = a; k < e; k++, o++) {
    const int c = colidx[k];
    int s0 = 0, s1 = TS;  // slice of column c: sl.b[s] <= c < sl.b[s + 1]
    while (s1 - s0 > 1) { const int mid = (s0 + s1) / 2; if (sl.b[mid] <= c) s0 = mid; else s1 = mid; }
    key[o] = hi | ((unsigned long long)s0 << 15) | lo;
    pos[o] = (uint32_t)k;
  }
}
__global__ void __launch_bounds__(kT)
k_blocked_fill(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ pos, int64_t n, const int32_t* __restrict__ slots, const int64_t* __restrict__ rowptr,
               const int32_t* __restrict__ colidx, const uint32_t* __restrict__ vals, int TS, uint32_t* __restrict__ ecol, uint16_t* __restrict__ erow,
               uint32_t* __restrict__ eval, uint32_t* __restrict__ toff) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = key[i];
  const uint32_t local = (uint32_t)(k & 0x7fffu), slice = (uint32_t)(k >> 15) & 127u;
  const size_t blk = (size_t)(k >> 22);
  const int r = slots[blk * GM_BLOCKED_ROWS + local];
  const uint32_t p = pos[i];
  ecol[i] = (uint32_t)colidx[p];
  if (eval) eval[i] = vals[p];
  erow[i] = (uint16_t)(local | ((int64_t)p == rowptr[r] ? 0x8000u : 0u));
  if (i == 0 || (key[i - 1] >> 15) != (k >> 15)) toff[blk * (size_t)TS + slice] = (uint32_t)i;
}
// where the 16 waves of a workgroup start inside a (block, slice) segment: equal shares, moved forward to the next row border
__global__ void __launch_bounds__(kT) k_blocked_wave_offsets(const uint32_t* __restrict__ toff, const uint16_t* __restrict__ erow, size_t nseg, uint32_t* __restrict__ woff) {
  const size_t t = (size_t)blockIdx.x * kT + threadIdx.x;
  if (t >= nseg * 17) return;
  const size_t seg = t / 17;
  const int w = (int)(t % 17);
  const uint32_t a = toff[seg], e = toff[seg + 1];
  uint32_t p = w == 16 ? e : a + (uint32_t)(((unsigned long long)(e - a) * (unsigned)w) / 16u);
  while (p > a && p < e && (erow[p] & 0x7fffu) == (erow[p - 1] & 0x7fffu)) p++;
  woff[t] = p;
}
static void free_blocked(gm_graph* g) {
  gm_blocked_t& B = g->blocked;
  const void* owned[] = {B.ecol, B.erow, B.woff, B.row_of, B.step_count, B.eval, B.epos};
  for (const void* q : owned)
    if (q) (void)hipFree((void*)q);
  memset(&B, 0, sizeof(B));
}
static int build_blocked(gm_graph* g, const CsrOwned* whole, hipStream_t s) {
  memset(&g->blocked, 0, sizeof(g->blocked));
  const int TS = g->nslices;
  const int nrows = g->desc.row_hi - g->desc.row_lo;
  if (g_blocked_rows < 0 || TS < 2 || TS > GM_MAX_SLICES || nrows <= 0 || (whole->vals != nullptr && whole->view.val_bytes != 4) || whole->view.nnz <= 0 || whole->view.nnz >= ((int64_t)1 << 32) - 65536 ||
      g->desc.nshards > 1 || whole->view.short_row <= 0 || whole->view.short_row > 16384)
    return GM_OK;
  if (g_blocked_rows == 0 && (double)g->nlive * 4.0 < 48.0 * 1048576.0) return GM_OK;
  const int64_t* rowptr = (const int64_t*)whole->rowptr;
  const int32_t* colidx = (const int32_t*)whole->colidx;
  const uint32_t* vals = (const uint32_t*)whole->vals;
  int rc;
  DevBuf flag, iota, rows, cnt, tmp, l64, off, dealt;
  if ((rc = flag.alloc((size_t)nrows)) || (rc = iota.alloc((size_t)nrows * 4)) || (rc = rows.alloc((size_t)nrows * 4)) || (rc = cnt.alloc(16))) return rc;
  hipLaunchKernelGGL(k_blocked_flag, dim3(grid_for(nrows)), dim3(kT), 0, s, rowptr, nrows, whole->view.short_row, flag.as<unsigned char>());
  hipLaunchKernelGGL(k_sweep_iota, dim3(grid_for(nrows)), dim3(kT), 0, s, iota.as<int32_t>(), nrows);
  size_t tb = 0;
  GM_TRY_HIP(rocprim::select(nullptr, tb, iota.as<int32_t>(), flag.as<unsigned char>(), rows.as<int32_t>(), cnt.as<unsigned int>(), (size_t)nrows, s));
  if ((rc = tmp.alloc(tb + 256))) return rc;
  GM_TRY_HIP(rocprim::select(tmp.p, tb, iota.as<int32_t>(), flag.as<unsigned char>(), rows.as<int32_t>(), cnt.as<unsigned int>(), (size_t)nrows, s));
  unsigned int ns = 0;
  GM_TRY_HIP(hipMemcpyAsync(&ns, cnt.p, 4, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  iota.free(); flag.free();
  if (ns == 0) return GM_OK;
  const int nblk = (int)(((size_t)ns + GM_BLOCKED_ROWS - 1) / GM_BLOCKED_ROWS);
  if (nblk >= (1 << 16)) return GM_OK;
  const int nslots = nblk * GM_BLOCKED_ROWS;
  if ((rc = dealt.alloc((size_t)nslots * 4))) return rc;
  GM_TRY_HIP(hipMemsetAsync(dealt.p, 0xff, (size_t)nslots * 4, s));
  hipLaunchKernelGGL(k_blocked_deal, dim3(grid_for((int)ns)), dim3(kT), 0, s, (const int32_t*)rows.as<int32_t>(), (int)ns, nblk, dealt.as<int32_t>());
  GM_TRY_HIP(hipStreamSynchronize(s));
  rows.free();
  if ((rc = l64.alloc(((size_t)nslots + 1) * 8)) || (rc = off.alloc(((size_t)nslots + 1) * 8))) return rc;
  GM_TRY_HIP(hipMemsetAsync(l64.p, 0, ((size_t)nslots + 1) * 8, s));
  hipLaunchKernelGGL(k_blocked_lens, dim3(grid_for(nslots)), dim3(kT), 0, s, (const int32_t*)dealt.as<int32_t>(), nslots, rowptr, l64.as<unsigned long long>());
  if ((rc = sweep_excl_scan(l64.as<unsigned long long>(), off.as<unsigned long long>(), (size_t)nslots + 1, s))) return rc;
  unsigned long long tot = 0;
  GM_TRY_HIP(hipMemcpyAsync(&tot, off.as<unsigned long long>() + nslots, 8, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  l64.free();
  const int64_t nent = (int64_t)tot;
  // automatic: a graph without skew -- (nearly) every edge in a short row; with hot vertices the row-blocks and the sweep are faster
  // (profiles/r05_short_rows_blocked_stream_prototype.md: RMAT-26's short rows 2.8-3.2 ms this way against 1.27-1.30)
  if (nent <= 0 || (g_blocked_rows == 0 && (double)nent < 0.9 * (double)whole->view.nnz)) return GM_OK;
  const size_t nseg = (size_t)nblk * TS;
  DevBuf k_in, k_out, v_in, v_out;
  if ((rc = k_in.alloc((size_t)nent * 8)) || (rc = k_out.alloc((size_t)nent * 8)) || (rc = v_in.alloc((size_t)nent * 4)) || (rc = v_out.alloc((size_t)nent * 4))) return rc;
  SweepSlices sl;
  memset(&sl, 0, sizeof(sl));
  for (int t = 0; t <= TS; t++) sl.b[t] = g->slice_base[t];
  hipLaunchKernelGGL(k_blocked_keys, dim3(grid_for(nslots)), dim3(kT), 0, s, (const int32_t*)dealt.as<int32_t>(), nslots, (const unsigned long long*)off.as<unsigned long long>(), rowptr,
                     colidx, sl, TS, k_in.as<unsigned long long>(), v_in.as<uint32_t>());
  GM_TRY_HIP(hipGetLastError());
  GM_TRY_HIP(hipStreamSynchronize(s));
  off.free();
  // stable: inside a (block, slice, row) the edges keep their CSR order = ascending native column
  if ((rc = sweep_sort_pairs(k_in.as<unsigned long long>(), k_out.as<unsigned long long>(), v_in.as<uint32_t>(), v_out.as<uint32_t>(), (size_t)nent, 22 + bits_for((uint32_t)nblk), s))) return rc;
  k_in.free(); v_in.free();
  DevBuf ecol, erow, toff, woff, steps, eval;
  if (vals && (rc = eval.alloc(((size_t)nent + 64 * 64) * 4))) return rc;
  if (vals) GM_TRY_HIP(hipMemsetAsync(eval.p, 0, ((size_t)nent + 64 * 64) * 4, s));
  const int npass = (nblk + 255) / 256;
  const int nsteps = npass * TS;
  if ((rc = ecol.alloc(((size_t)nent + 64 * 64) * 4)) || (rc = erow.alloc(((size_t)nent + 64 * 64) * 2)) || (rc = toff.alloc((nseg + 2) * 4)) || (rc = woff.alloc((nseg + 1) * 17 * 4)) ||
      (rc = steps.alloc((size_t)8 * nsteps * 4 + 256)))
    return rc;
  GM_TRY_HIP(hipMemsetAsync(ecol.p, 0, ((size_t)nent + 64 * 64) * 4, s));
  GM_TRY_HIP(hipMemsetAsync(erow.p, 0, ((size_t)nent + 64 * 64) * 2, s));
  GM_TRY_HIP(hipMemsetAsync(toff.p, 0xff, (nseg + 2) * 4, s));
  GM_TRY_HIP(hipMemsetAsync(steps.p, 0, (size_t)8 * nsteps * 4 + 256, s));
  hipLaunchKernelGGL(k_blocked_fill, dim3(grid_for(nent)), dim3(kT), 0, s, (const unsigned long long*)k_out.as<unsigned long long>(), (const uint32_t*)v_out.as<uint32_t>(), nent,
                     (const int32_t*)dealt.as<int32_t>(), rowptr, colidx, vals, TS, ecol.as<uint32_t>(), erow.as<uint16_t>(), vals ? eval.as<uint32_t>() : (uint32_t*)nullptr,
                     toff.as<uint32_t>());
  GM_TRY_HIP(hipGetLastError());
  {
    std::vector<uint32_t> h(nseg + 1);
    GM_TRY_HIP(hipMemcpyAsync(h.data(), toff.p, nseg * 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    h[nseg] = (uint32_t)nent;
    for (size_t b = nseg; b-- > 0;) if (h[b] == 0xffffffffu) h[b] = h[b + 1];
    GM_TRY_HIP(hipMemcpyAsync(toff.p, h.data(), (nseg + 1) * 4, hipMemcpyHostToDevice, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
  }
  k_out.free();
  if (!vals) v_out.free();  // (with edge values: the entries' CSR positions stay, for gm_graph_sync_tile_vals)
  hipLaunchKernelGGL(k_blocked_wave_offsets, dim3((unsigned)((nseg * 17 + kT - 1) / kT)), dim3(kT), 0, s, (const uint32_t*)toff.as<uint32_t>(), (const uint16_t*)erow.as<uint16_t>(), nseg,
                     woff.as<uint32_t>());
  GM_TRY_HIP(hipGetLastError());
  GM_TRY_HIP(hipStreamSynchronize(s));
  gm_blocked_t& B = g->blocked;
  B.nrows = (int32_t)ns; B.nblocks = nblk; B.nslices = TS; B.short_row = whole->view.short_row; B.nsteps = nsteps; B.nentries = nent;
  B.ecol = (const uint32_t*)ecol.release(); B.erow = (const uint16_t*)erow.release(); B.woff = (const uint32_t*)woff.release();
  B.row_of = (const int32_t*)dealt.release(); B.step_count = (uint32_t*)steps.release();
  B.val_bytes = vals ? 4 : 0; B.eval = (const uint32_t*)eval.release(); B.epos = vals ? (const uint32_t*)v_out.release() : nullptr;
  return GM_OK;
}

static int build_tiles(gm_graph* g, const uint64_t* keys_sorted, const uint32_t* idx_sorted, unsigned long long kept,
                       const void* d_val, hipStream_t s, const CsrOwned* whole) {
  const int T = g->ntiles;
  const int nrows = g->desc.row_hi - g->desc.row_lo;
  const int nw = (nrows + 31) / 32 + 2;
  int rc;
  g->out_tiles = new CsrOwned[T];
  g->out_tile_prev = new uint32_t*[T];
  for (int t = 0; t < T; t++) g->out_tile_prev[t] = nullptr;
  DevBuf tk_in, tk_out, pos_in, pos_out, tmp, bounds;
  if ((rc = tk_in.alloc((size_t)kept)) || (rc = tk_out.alloc((size_t)kept)) || (rc = pos_in.alloc((size_t)kept * 4)) ||
      (rc = pos_out.alloc((size_t)kept * 4)) || (rc = bounds.alloc((size_t)(GM_MAX_TILES + 2) * 8)))
    return rc;
  std::vector<int64_t> h_bounds((size_t)T + 1, 0);
  TileBases bases;
  memcpy(bases.b, g->tile_base, sizeof(bases.b));
  if (kept > 0) {
    hipLaunchKernelGGL(k_tile_keys, dim3(grid_for((int64_t)kept)), dim3(kT), 0, s, keys_sorted, (int64_t)kept,
                       (const int64_t*)whole->rowptr, whole->view.tile_min_row, (const int32_t*)g->dev_of_native, bases, T,
                       tk_in.as<uint8_t>(), pos_in.as<uint32_t>());
    GM_TRY_HIP(hipGetLastError());
    size_t tb = 0;  // stable: inside a tile the edges keep their (row, native col) order
    GM_TRY_HIP(rocprim::radix_sort_pairs(nullptr, tb, tk_in.as<uint8_t>(), tk_out.as<uint8_t>(), pos_in.as<uint32_t>(),
                                         pos_out.as<uint32_t>(), (size_t)kept, 0u, 8u, s));
    if ((rc = tmp.alloc(tb))) return rc;
    GM_TRY_HIP(rocprim::radix_sort_pairs(tmp.p, tb, tk_in.as<uint8_t>(), tk_out.as<uint8_t>(), pos_in.as<uint32_t>(),
                                         pos_out.as<uint32_t>(), (size_t)kept, 0u, 8u, s));
    hipLaunchKernelGGL(k_tile_bounds, dim3(1), dim3(128), 0, s, (const uint8_t*)tk_out.p, (int64_t)kept, T, bounds.as<int64_t>());
    GM_TRY_HIP(hipMemcpyAsync(h_bounds.data(), bounds.p, (size_t)(T + 1) * 8, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
  }
  tk_in.free();
  pos_in.free();
  tmp.free();
  int64_t largest = 0;
  for (int t = 0; t < T; t++) largest = std::max(largest, h_bounds[t + 1] - h_bounds[t]);
  DevBuf keys_t, idx_t;
  if ((rc = keys_t.alloc((size_t)largest * 8)) || (rc = idx_t.alloc((size_t)largest * 4))) return rc;
  // Row classes fixed per row (g_own_wave_row > 0): a row of more than that many edges in the WHOLE graph is worked on
  // by the one-wave-per-row / giant kernels in every tile, all other rows by the row-block / 16-row kernels in every
  // tile -- the two groups then never touch the same y entry, and the engine can run them on two streams through all
  // tiles of an iteration without joining them after every tile.
  DevBuf own;
  const unsigned char* own_wave = nullptr;
  int own_limit = g_own_wave_row;
  if (own_limit == 0) { const char* e = getenv("GRAPHMAT_OWN_WAVE_ROW"); if (e) own_limit = atoi(e); }
  if (own_limit > 0 && nrows > 0) {
    if ((rc = own.alloc((size_t)nrows))) return rc;
    hipLaunchKernelGGL(k_own_wave_rows, dim3(grid_for(nrows)), dim3(kT), 0, s, (const int64_t*)whole->rowptr, nrows, (int64_t)own_limit,
                       own.as<unsigned char>());
    own_wave = own.as<unsigned char>();
  }
  uint32_t* prev = nullptr;
  GM_TRY_HIP(hipMalloc((void**)&prev, (size_t)nw * 4));
  GM_TRY_HIP(hipMemsetAsync(prev, 0, (size_t)nw * 4, s));
  g->out_tile_prev[0] = prev;
  for (int t = 0; t < T; t++) {
    const int64_t n = h_bounds[t + 1] - h_bounds[t];
    if (n > 0)
      hipLaunchKernelGGL(k_tile_gather, dim3(grid_for(n)), dim3(kT), 0, s, keys_sorted, idx_sorted,
                         (const uint32_t*)pos_out.as<uint32_t>() + h_bounds[t], n, keys_t.as<uint64_t>(), idx_t.as<uint32_t>());
    if ((rc = finish_csr(g, keys_t.as<uint64_t>(), idx_t.as<uint32_t>(), (unsigned long long)n, d_val, s, &g->out_tiles[t], -1, own_wave))) return rc;
    gm_csr_t& v = g->out_tiles[t].view;
    v.hot_base = g->tile_base[t];
    v.hot_len = g->tile_base[t + 1] - g->tile_base[t];
    if (t + 1 < T) {
      uint32_t* nxt = nullptr;
      GM_TRY_HIP(hipMalloc((void**)&nxt, (size_t)nw * 4));
      hipLaunchKernelGGL(k_or_words, dim3(grid_for(nw)), dim3(kT), 0, s, (const uint32_t*)g->out_tile_prev[t],
                         (const uint32_t*)g->out_tiles[t].rowbits, nxt, nw);
      g->out_tile_prev[t + 1] = nxt;
    }
  }
  GM_TRY_HIP(hipStreamSynchronize(s));
  if (g_sweep_slices != 0 && (rc = build_sweep(g, whole, s))) return rc;
  if (g_sweep_slices != 0 && (rc = build_blocked(g, whole, s))) return rc;
  return GM_OK;
}

// GM_LAYOUT_DEGREE: rank vertices by total degree, deal the ranks round-robin over the shards
static int build_degree_layout(gm_graph* g, int64_t nnz, const int32_t* d_src, const int32_t* d_dst, hipStream_t s, bool keeps_values) {
  gm_graph_desc_t& D = g->desc;
  const int nv = D.nvertices, G = D.nshards;
  int S = (nv + G - 1) / G;
  if (G > 1) S = (S + 63) / 64 * 64;
  int vd = (G > 1) ? G * S : nv;  // (sharded graphs with slices: S and vd are settled once the slices are known, below)
  DevBuf deg, keys_in, keys_out, ids_in, order, tmp, don, nod;
  int rc;
  if ((rc = deg.alloc((size_t)nv * 4)) || (rc = keys_in.alloc((size_t)nv * 4)) || (rc = keys_out.alloc((size_t)nv * 4)) ||
      (rc = ids_in.alloc((size_t)nv * 4)) || (rc = order.alloc((size_t)nv * 4)) || (rc = don.alloc((size_t)nv * 4)))
    return rc;
  GM_TRY_HIP(hipMemsetAsync(deg.p, 0, (size_t)nv * 4, s));
  if (nnz > 0)
    hipLaunchKernelGGL(k_degree, dim3(grid_for(nnz)), dim3(kT), 0, s, d_src, d_dst, nnz, D.nparts, nv, D.ids_are_native,
                       deg.as<uint32_t>(), g_rank_by);
  if (D.edges_local && (rc = dist_all_reduce_sum_u32(deg.as<uint32_t>(), (size_t)nv, s))) return rc;  // counts of all parts
  hipLaunchKernelGGL(k_rank_keys, dim3(grid_for(nv)), dim3(kT), 0, s, deg.as<uint32_t>(), nv, keys_in.as<uint32_t>(),
                     ids_in.as<int32_t>(), (uint32_t)(g_rank_by == 0 ? g_rank_cap : 0));
  size_t tb = 0;
  GM_TRY_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in.as<uint32_t>(), keys_out.as<uint32_t>(), ids_in.as<int32_t>(),
                                       order.as<int32_t>(), (size_t)nv, 0u, 32u, s));
  if ((rc = tmp.alloc(tb))) return rc;
  GM_TRY_HIP(rocprim::radix_sort_pairs(tmp.p, tb, keys_in.as<uint32_t>(), keys_out.as<uint32_t>(), ids_in.as<int32_t>(),
                                       order.as<int32_t>(), (size_t)nv, 0u, 32u, s));
  // vertices with at least one edge come first in every slice: the rest is never gathered
  DevBuf nzd;
  if ((rc = nzd.alloc(8))) return rc;
  GM_TRY_HIP(hipMemsetAsync(nzd.p, 0, 8, s));
  hipLaunchKernelGGL(k_count_nonzero, dim3(grid_for(nv)), dim3(kT), 0, s, deg.as<uint32_t>(), nv,
                     nzd.as<unsigned long long>());
  unsigned long long nz = 0;
  GM_TRY_HIP(hipMemcpyAsync(&nz, nzd.p, 8, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  g->nlive = (int32_t)nz;
  // column tiles: contiguous NATIVE ranges; the device order becomes (tile, degree rank inside the tile),
  // vertices without edges last.  A stable sort of the ranked list by tile does it.
  int T = D.col_tiles;
  const double mib_live = (double)nz * 4.0 / 1048576.0;
  if (T == 0) T = g_col_tiles;
  if (T == 0) { const char* e = getenv("GRAPHMAT_COL_TILES"); if (e) T = atoi(e); }
  if (T == 0) {
    // automatic: tiles pay once the live part of a 4-byte message vector outgrows what the caches hold.  Measured,
    // PageRank on RMAT with tiles that serve equally many gathers and row classes fixed per row (g_own_wave_row; ms per
    // iteration by tile count): RMAT-24 (34 MiB live) 1 / 3 / 4: 1.67 / 1.85 / 1.84; RMAT-25 (65 MiB) 4 / 5 / 6 / 7: 3.41 /
    // 3.38 / 3.39 / 3.46; RMAT-26 (125 MiB) 5 / 6 / 7 / 8 / 9 / 10 / 12: 6.31 / 6.18 / 6.24 / 6.09-6.13 / 6.14 / 6.09 / 6.21;
    // RMAT-27 (239 MiB) 10 / 12 / 14 / 16: 13.24 / 13.24 / 12.99 / 13.03; RMAT-25 with the persistent kernels 4 / 5 / 6 / 7 / 8:
    // 3.18 / 3.10 / 3.15 / 3.31 / 3.31 -- about one tile per 17 MiB on top of one, from 60 MiB on
    // (with the auxiliary stream joined after every tile the optimum was fewer, larger tiles: 4 / 6 / 10)
    const double mib = (double)nz * 4.0 / 1048576.0;
    T = mib >= 60.0 ? (int)(1.0 + mib / 17.0 + 0.5) : 1;
    // With the row-stationary sweep (g_sweep_slices != 0; the closing session of round 4, profiles/r04_tile_counts_with_sweep.md) the
    // tiles only cut the one-wave-per-row and giant rows -- the medium rows walk ~64 slices whatever the tile count -- and every tile
    // is a pass of two latency-bound kernel chains: FEWER tiles win, and tiling pays from smaller graphs on because it brings the
    // sweep with it.  ms per iteration by tile count: RMAT-23 (18 MiB) 1 / 2 / 3: 0.820 / 0.842 / 0.857; RMAT-24 (34 MiB) 1 / 2 / 3:
    // 1.66 / 1.33 / 1.45; RMAT-25 (65 MiB) 2 / 3 / 4 / 5 / 6: 2.61 / 2.47 / 2.54 / 2.56 / 2.60; RMAT-26 (125 MiB) 2 .. 9: 5.26 / 4.84 / 4.92 /
    // 4.99 / 5.00 / 5.09 / 5.10 / 5.12 (seeds 2 and 3: 3 tiles best as well, 4.83 / 4.81); RMAT-27 (239 MiB) 3 / 4 / 5 / 6 / 8 / 10 / 12 / 15:
    // 11.84 / 11.20 / 11.27 / 11.15 / 11.14 / 11.30 / 11.38 / 11.49 -- two tiles from 25 MiB, three from 50 MiB, one per 40 MiB from 180 MiB on
    // (An adjacency that keeps its edge values is not swept -- the sweep's copy of the medium rows holds column ids only -- and all
    // its rows above 64 edges go through the tiles as before: the old rule stays for it, from 100 MiB on.  The reference's unchanged
    // PageRank.cpp -- int edge values, ordered fold -- by tile count: RMAT-24 1 / 2 / 3 / 5: 149 / 161 / 162 / 165 ms; RMAT-25 1 / 2 / 3 / 5:
    // 228 / 308 / 283 / 281 ms; RMAT-26 1 / 3 / 8: 641 / 582 / 506 ms.)
    // (round 5: the sweep -- kernels.hpp: k_spmv_sell -- also takes the rows of more than own_wave_row edges and adjacencies that keep
    // 4-byte edge values; the tiles are only walked by programs the sweep does not cover, and get the same count)
    // (round 6: with the giant rows' passes down to a third -- they were what the small graphs' iterations waited for -- the sweep pays from ~7 MiB on:
    // RMAT-22 (9 MiB) 0.445 -> 0.384 ms, RMAT-23 (18 MiB) 0.813 -> 0.589 ms with 2 x 8 slices; RMAT-21 (5 MiB) 0.295 -> 0.321, RMAT-20 0.207 -> 0.307:
    // profiles/r06_small_scales_sweep.txt.  The [slice][degree rank] order the sweep needs spreads the hub columns over the slices, which the
    // gather kernels of the programs that do NOT take the sweep pay for: unchanged SSSP.cpp RMAT-22 6.4 -> 7.8 ms, RMAT-23 11.4 -> 13.4 ms,
    // BFS.cpp +3..5 %, against PageRank.cpp 24.3 -> 24.8 ms (RMAT-22) and 52.3 -> 44.4 ms (RMAT-23) -- so the rule starts at 12 MiB, where PageRank
    // gains 28 %, not at 7.  Shards keep the 25 MiB rule: nothing smaller was measured on them.)
    const double sweep_from = G > 1 ? 25.0 : 12.0;
    if (g_sweep_slices != 0 && (!keeps_values || D.val_bytes == 4)) T = mib < sweep_from ? 1 : mib < 50.0 ? 2 : mib < 180.0 ? 3 : (int)(mib / 40.0 + 0.5);
    else if (keeps_values && mib < 100.0) T = 1;
  }
  // Sharded graphs (round 6): no column tiles, but the same SLICES -- every owner's range of the device order becomes [slice][degree rank
  // inside it], so that a slice of the message vector is the same sub-range of all G owners' ranges and the row-stationary sweep can walk
  // a shard's rows (gm_sweep_t.nsub).  Whenever a single-shard graph of this size would be tiled (or the caller asks for tiles).
  const bool sweepable_ = !keeps_values || D.val_bytes == 4;
  const bool shard_slices = G > 1 && T > 1 && nz >= 2 && g_sweep_slices != 0 && sweepable_ && nnz >= 0 && nv < (1 << 28);
  if (T < 1 || G > 1 || nz < 2) T = 1;
  if (T > GM_MAX_TILES) T = GM_MAX_TILES;
  DevBuf slice_keys;  // sharded graphs with slices: the slice of every position of the sorted list (255 = no edges)
  SliceDeal sd;
  memset(&sd, 0, sizeof(sd));
  int shard_TS = 0;
  if (T > 1 || shard_slices) {
    // Tiles are contiguous NATIVE ranges cut so that every tile serves about the same number of gathers (a vertex weighs
    // as often as it is a column of the GM_DIR_OUT adjacency; on RMAT the busy tiles then hold fewer vertices, i.e. a
    // smaller slice of x where most gathers go).  gm_set_option("tile_balance", 0) cuts them into equally many vertices
    // with edges instead: RMAT-26 best 4 such tiles 7.07 ms against 6.74 ms with 6 gather-balanced ones.
    DevBuf cnt, w, wpre, tk_in, tk_out, order2, bnd;
    if ((rc = cnt.alloc((size_t)nv * 4)) || (rc = w.alloc((size_t)(nv + 1) * 8)) || (rc = wpre.alloc((size_t)(nv + 1) * 8)) ||
        (rc = tk_in.alloc((size_t)nv)) || (rc = tk_out.alloc((size_t)nv)) || (rc = order2.alloc((size_t)nv * 4)) ||
        (rc = bnd.alloc((size_t)(GM_MAX_SLICES + 2) * 8)))
      return rc;
    GM_TRY_HIP(hipMemsetAsync(cnt.p, 0, (size_t)nv * 4, s));
    GM_TRY_HIP(hipMemsetAsync(w.p, 0, (size_t)(nv + 1) * 8, s));
    // g_tile_balance: 0 = vertices; 1 = gathers; 2..65536 = gathers + (value - 1) per vertex; 100000 + X = gathers^(X/100)  (experiments)
    const int by_vertices = (g_tile_balance == 0 || nnz == 0) ? 1 : 0;
    const int pw100 = g_tile_balance >= 100000 ? g_tile_balance - 100000 : 100;
    const uint32_t addc = (g_tile_balance >= 2 && g_tile_balance <= 65536) ? (uint32_t)(g_tile_balance - 1) : 0u;
    if (!by_vertices && nnz > 0)
      hipLaunchKernelGGL(k_col_weight, dim3(grid_for(nnz)), dim3(kT), 0, s, d_src, nnz, D.nparts, nv, D.ids_are_native, cnt.as<uint32_t>());
    if (!by_vertices && D.edges_local && (rc = dist_all_reduce_sum_u32(cnt.as<uint32_t>(), (size_t)nv, s))) return rc;  // (every rank must cut the same slices)
    hipLaunchKernelGGL(k_tile_weight, dim3(grid_for(nv)), dim3(kT), 0, s, deg.as<uint32_t>(), cnt.as<uint32_t>(), nv, addc, pw100, by_vertices,
                       w.as<unsigned long long>());
    size_t sb = 0;
    GM_TRY_HIP(rocprim::exclusive_scan(nullptr, sb, w.as<unsigned long long>(), wpre.as<unsigned long long>(), 0ull, (size_t)nv + 1,
                                       rocprim::plus<unsigned long long>(), s));
    if ((rc = tmp.alloc(sb))) return rc;
    GM_TRY_HIP(rocprim::exclusive_scan(tmp.p, sb, w.as<unsigned long long>(), wpre.as<unsigned long long>(), 0ull, (size_t)nv + 1,
                                       rocprim::plus<unsigned long long>(), s));
    unsigned long long total = 0;
    GM_TRY_HIP(hipMemcpyAsync(&total, wpre.as<unsigned long long>() + nv, 8, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    // (gm_graph_sweep: the order is cut k times finer than the tiles, a tile = k consecutive slices; about 1.3 MiB of live
    // messages per slice -- measured with the sweep's prototype, tools/sell_bench.hip: RMAT-26 (125 MiB) 64 / 80 / 96 / 128 slices
    // 2.17 / 2.09 / 2.05 / 2.15 ms for the swept rows, RMAT-25 (65 MiB) 32 / 64: 1.02 / 0.94 ms, RMAT-24 (34 MiB) 16 / 32 / 64: 0.79 / 0.80 / 0.92)
    // (a graph WITHOUT skew -- next to no edge on a vertex of more than 2 x short_row edges: the shape of the reference's test/generator.h --
    // is multiplied by the column-blocked stream of its short rows, k_spmv_blocked, whose synchronised steps want fewer, larger slices:
    // uniform 2^26, 32 / 48 / 56 / 64 / 72 / 96 / 128 slices: 12.9 / 10.5 / 9.7 / 9.0 / 10.0 / 10.1 / 12.0 ms; 2^25, 32 / 99: 4.24 / 4.71 -- about 4 MiB each)
    bool no_skew = false;
    if (g_sweep_slices == 1 && g_blocked_rows >= 0 && !keeps_values && nnz > 0 && G == 1) {
      DevBuf mass;
      if ((rc = mass.alloc(16))) return rc;
      GM_TRY_HIP(hipMemsetAsync(mass.p, 0, 16, s));
      hipLaunchKernelGGL(k_degree_mass, dim3(1024), dim3(kT), 0, s, deg.as<uint32_t>(), nv, (uint32_t)(2 * g_short_row), mass.as<unsigned long long>());
      unsigned long long hm[2] = {0, 0};
      GM_TRY_HIP(hipMemcpyAsync(hm, mass.p, 16, hipMemcpyDeviceToHost, s));
      GM_TRY_HIP(hipStreamSynchronize(s));
      no_skew = hm[0] > 0 && hm[1] * 10ull <= hm[0];
    }
    int want_slices = g_sweep_slices >= 8 ? g_sweep_slices : (int)(mib_live / (no_skew ? 4.0 : 1.3) + 0.5);
    // (a shard of G holds 1 / G of the edges but walks the WHOLE message vector: what a slice costs a workgroup whatever it holds --
    // two barriers, the hot-set load, the staging round -- weighs G times more against the gathers it makes cheaper, so a shard wants
    // fewer, larger slices.  Shard 0 of RMAT-26, compute only, ms per iteration by slice count -- of 8: 16 / 20 / 24 / 28 / 32 / 48 / 64 / 96:
    // 1.15 / 1.05 / 1.07 / 1.12 / 1.14 / 1.33 / 1.44 / 1.81; of 4: 16 / 24 / 32 / 96: 1.78 / 1.58 / 1.60 / 2.07; of 2: 24 / 32 / 48 / 96: 2.65 /
    // 2.23 / 2.18 / 2.37 -- about G^-0.75 of the single shard's count; profiles/r06_shard_slice_counts.txt)
    if (shard_slices && g_sweep_slices < 8) want_slices = (int)((double)want_slices / pow((double)G, 0.75) + 0.5);
    want_slices = std::max(16, std::min(GM_MAX_SLICES, want_slices));
    // (an adjacency the sweep cannot take -- edge values that are not 4 bytes wide -- is only ever walked tile by tile: its order stays
    // degree-ranked inside a whole TILE, so that the tile kernels' LDS hot sets hold the tile's busiest vertices, not one slice's)
    const bool sweepable = !keeps_values || D.val_bytes == 4;
    const int sub = (g_sweep_slices != 0 && sweepable) ? std::max(1, std::min(GM_MAX_SLICES / T, (want_slices + T / 2) / T)) : 1;
    const int TS = T * sub;
    hipLaunchKernelGGL(k_tile_of_ranked, dim3(grid_for(nv)), dim3(kT), 0, s, order.as<int32_t>(), nv, deg.as<uint32_t>(),
                       wpre.as<unsigned long long>(), total, TS, tk_in.as<uint8_t>());
    sb = 0;
    GM_TRY_HIP(rocprim::radix_sort_pairs(nullptr, sb, tk_in.as<uint8_t>(), tk_out.as<uint8_t>(), order.as<int32_t>(),
                                         order2.as<int32_t>(), (size_t)nv, 0u, 8u, s));
    if ((rc = tmp.alloc(sb))) return rc;
    GM_TRY_HIP(rocprim::radix_sort_pairs(tmp.p, sb, tk_in.as<uint8_t>(), tk_out.as<uint8_t>(), order.as<int32_t>(),
                                         order2.as<int32_t>(), (size_t)nv, 0u, 8u, s));
    GM_TRY_HIP(hipMemcpyAsync(order.p, order2.p, (size_t)nv * 4, hipMemcpyDeviceToDevice, s));
    // where every tile starts in the new order (vertices without edges carry key 255 and sort to the end)
    hipLaunchKernelGGL(k_tile_bounds, dim3(1), dim3(256), 0, s, (const uint8_t*)tk_out.p, (int64_t)nv, TS, bnd.as<int64_t>());
    int64_t hb[GM_MAX_SLICES + 2];
    GM_TRY_HIP(hipMemcpyAsync(hb, bnd.p, (size_t)(TS + 1) * 8, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    if (!shard_slices) {
      for (int t = 0; t <= T; t++) g->tile_base[t] = (int32_t)hb[t * sub];
      g->nslices = sub > 1 ? TS : 0;
      for (int t = 0; t <= TS; t++) g->slice_base[t] = (int32_t)hb[t];
    } else {
      // positions inside an owner's range: slice t takes ceil(n_t / G) of them in every owner
      shard_TS = TS;
      long long pos = 0;
      for (int t = 0; t < TS; t++) { sd.hb[t] = (int32_t)hb[t]; sd.sb[t] = (int32_t)pos; pos += (hb[t + 1] - hb[t] + G - 1) / G; }
      sd.hb[TS] = (int32_t)hb[TS];  // (= vertices with edges; the rest of the list has none)
      sd.sb[TS] = (int32_t)pos;
      const long long ndead = (long long)nv - hb[TS];
      const long long need = pos + (ndead + G - 1) / G;
      S = (int)((need + 63) / 64 * 64);
      vd = G * S;
      g->tile_base[0] = 0; g->tile_base[1] = (int32_t)pos;
      g->nslices = TS;
      for (int t = 0; t <= TS; t++) g->slice_base[t] = sd.sb[t];
      slice_keys.p = tk_out.release();
    }
  }
  g->ntiles = T;
  D.col_tiles = T;
  if ((rc = nod.alloc((size_t)vd * 4))) return rc;
  GM_TRY_HIP(hipMemsetAsync(nod.p, 0xff, (size_t)vd * 4, s));  // -1 = unused slot
  if (shard_TS > 0)
    hipLaunchKernelGGL(k_deal_sliced, dim3(grid_for(nv)), dim3(kT), 0, s, order.as<int32_t>(), (const uint8_t*)slice_keys.as<uint8_t>(), nv, G, S, shard_TS, sd,
                       don.as<int32_t>(), nod.as<int32_t>());
  else
    hipLaunchKernelGGL(k_deal, dim3(grid_for(nv)), dim3(kT), 0, s, order.as<int32_t>(), nv, G, S, don.as<int32_t>(),
                       nod.as<int32_t>());
  GM_TRY_HIP(hipGetLastError());
  GM_TRY_HIP(hipStreamSynchronize(s));
  {
    long long per = shard_TS > 0 ? (long long)sd.sb[shard_TS] : ((long long)nz + G - 1) / G;
    per = (per + 63) / 64 * 64;
    const long long slice = (G > 1) ? S : nv;
    D.xchg_rows = (int32_t)(per < slice ? per : slice);
    if (D.xchg_rows < 64 && slice >= 64) D.xchg_rows = 64;
  }
  g->dev_of_native = (int32_t*)don.release();
  g->native_of_dev = (int32_t*)nod.release();
  D.ndevice = vd;
  D.row_lo = (G > 1) ? D.shard * S : 0;
  D.row_hi = (G > 1) ? D.row_lo + S : nv;
  return GM_OK;
}

// does any endpoint (0-based native ids) of these edges have a device id >= limit?
__global__ void __launch_bounds__(kT)
k_any_dev_beyond(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t nnz, const int32_t* __restrict__ dev_of_native,
                 int limit, unsigned int* __restrict__ flag) {
  const int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (k >= nnz) return;
  if (dev_of_native[src[k]] >= limit || dev_of_native[dst[k]] >= limit) *flag = 1u;
}

// edges of a CSR direction back as native (src, dst) pairs, in CSR order
__global__ void __launch_bounds__(kT)
k_csr_to_coo(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, int nrows, int row_base,
             const int32_t* __restrict__ native_of_dev, int rows_are_dst, int32_t* __restrict__ src,
             int32_t* __restrict__ dst) {
  const int row = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int lane = threadIdx.x & 63;
  const int rn = native_of_dev ? native_of_dev[row_base + row] : row_base + row;
  for (int64_t e = rowptr[row] + lane; e < rowptr[row + 1]; e += 64) {
    int c = colidx[e];
    int cn = native_of_dev ? native_of_dev[c] : c;
    src[e] = rows_are_dst ? cn : rn;
    dst[e] = rows_are_dst ? rn : cn;
  }
}

// tile copies of the edge values from the whole-graph CSR: a row's edges there are the concatenation of its pieces in
// tile order, so `done[r]` (edges of row r already handed to earlier tiles) locates a piece.  One wave per row.
__global__ void __launch_bounds__(kT)
k_tile_vals(const int64_t* __restrict__ rowptr_whole, const unsigned char* __restrict__ vals_whole, const int64_t* __restrict__ rowptr_tile,
            unsigned char* __restrict__ vals_tile, int nrows, int val_bytes, int64_t* __restrict__ done) {
  const int row = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int64_t a = rowptr_tile[row], len = rowptr_tile[row + 1] - a;
  if (len == 0) return;
  const int64_t from = rowptr_whole[row] + done[row];
  const int64_t bytes = len * val_bytes;
  const unsigned char* src = vals_whole + from * val_bytes;
  unsigned char* dst = vals_tile + a * val_bytes;
  for (int64_t b = threadIdx.x & 63; b < bytes; b += 64) dst[b] = src[b];
  if ((threadIdx.x & 63) == 0) done[row] += len;
}

static void free_csr(CsrOwned* c) {
  if (c->rowptr) (void)hipFree(c->rowptr);
  if (c->colidx) (void)hipFree(c->colidx);
  if (c->vals) (void)hipFree(c->vals);
  if (c->rowbits) (void)hipFree(c->rowbits);
  if (c->seg_row) (void)hipFree(c->seg_row);
  if (c->blk_seg) (void)hipFree(c->blk_seg);
  if (c->mid_row) (void)hipFree(c->mid_row);
  if (c->giant_row) (void)hipFree(c->giant_row);
  if (c->gchunk_row) (void)hipFree(c->gchunk_row);
  if (c->gchunk_edge) (void)hipFree(c->gchunk_edge);
  if (c->gterm_off) (void)hipFree(c->gterm_off);
  if (c->umid_row) (void)hipFree(c->umid_row);
  if (c->gchunk_state) (void)hipFree(c->gchunk_state);
  *c = CsrOwned();
}

static void free_tiles(gm_graph* g) {
  free_sweep(g);
  free_blocked(g);
  if (g->out_tiles) {
    for (int t = 0; t < g->ntiles; t++) free_csr(&g->out_tiles[t]);
    delete[] g->out_tiles;
    g->out_tiles = nullptr;
  }
  if (g->out_tile_prev) {
    for (int t = 0; t < g->ntiles; t++)
      if (g->out_tile_prev[t]) (void)hipFree(g->out_tile_prev[t]);
    delete[] g->out_tile_prev;
    g->out_tile_prev = nullptr;
  }
}

}  // namespace gm

extern "C" {

int gm_graph_tiles(const gm_graph_t* g, int direction, int* ntiles) {
  if (!g || !ntiles) { gm::set_error("gm_graph_tiles: null argument"); return GM_ERR_INVALID; }
  *ntiles = (direction == GM_DIR_OUT && g->out_tiles && g->ntiles > 1) ? g->ntiles : 1;
  return GM_OK;
}

int gm_graph_sweep(const gm_graph_t* g, gm_sweep_t* out) {
  if (!g || !out) { gm::set_error("gm_graph_sweep: null argument"); return GM_ERR_INVALID; }
  *out = g->sweep;
  return GM_OK;
}

int gm_graph_blocked(const gm_graph_t* g, gm_blocked_t* out) {
  if (!g || !out) { gm::set_error("gm_graph_blocked: null argument"); return GM_ERR_INVALID; }
  *out = g->blocked;
  return GM_OK;
}
int gm_graph_tile(const gm_graph_t* g, int direction, int tile, gm_csr_t* out, const uint32_t** d_prev_bits) {
  if (!g || !out) { gm::set_error("gm_graph_tile: null argument"); return GM_ERR_INVALID; }
  if (direction != GM_DIR_OUT || !g->out_tiles || tile < 0 || tile >= g->ntiles) {
    gm::set_error("gm_graph_tile: no tile %d in direction %d", tile, direction);
    return GM_ERR_INVALID;
  }
  *out = g->out_tiles[tile].view;
  if (d_prev_bits) *d_prev_bits = g->out_tile_prev[tile];
  return GM_OK;
}

// Collective builds (edges_local): a failure on one rank must fail every rank, or the others wait for ever in the next
// collective.  Returns GM_OK only when every rank arrived with rc == GM_OK (one all-reduce of a failure count).
static int agree_on_status(int rc, hipStream_t s) {
  gm::DevBuf w;
  if (w.alloc(64) != GM_OK) return rc != GM_OK ? rc : GM_ERR_NOMEM;  // (cannot even ask: report locally)
  const uint32_t mine = rc != GM_OK ? 1u : 0u;
  uint32_t total = 0;
  if (hipMemcpyAsync(w.p, &mine, 4, hipMemcpyHostToDevice, s) != hipSuccess) return rc != GM_OK ? rc : GM_ERR_HIP;
  int rc2 = gm::dist_all_reduce_sum_u32(w.as<uint32_t>(), 1, s);
  if (rc2 != GM_OK) return rc != GM_OK ? rc : rc2;
  if (hipMemcpyAsync(&total, w.p, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return rc != GM_OK ? rc : GM_ERR_HIP;
  if (rc != GM_OK) return rc;
  if (total != 0) { gm::set_error("gm_graph_create: the distributed build failed on %u other rank(s)", total); return GM_ERR_INVALID; }
  return GM_OK;
}

int gm_graph_create(gm_graph_t** gout, const gm_graph_desc_t* desc, int64_t nnz, const int32_t* src,
                    const int32_t* dst, const void* val, gm_stream_t stream) {
  if (!gout || !desc) { gm::set_error("gm_graph_create: null argument"); return GM_ERR_INVALID; }
  *gout = nullptr;
  if (desc->nvertices <= 0 || nnz < 0 || desc->nparts <= 0 || desc->row_lo < 0 || desc->row_hi > desc->nvertices ||
      desc->row_lo > desc->row_hi || (desc->directions & (GM_DIR_OUT | GM_DIR_IN)) == 0 || desc->val_bytes < 0 ||
      (nnz > 0 && (!src || !dst))) {
    gm::set_error("gm_graph_create: invalid descriptor (nv=%d nnz=%lld nparts=%d rows=[%d,%d) dirs=%d)", desc->nvertices,
                  (long long)nnz, desc->nparts, desc->row_lo, desc->row_hi, desc->directions);
    return GM_ERR_INVALID;
  }
  if (desc->layout == GM_LAYOUT_NATIVE &&
      ((desc->row_lo & 63) != 0 || ((desc->row_hi & 63) != 0 && desc->row_hi != desc->nvertices))) {
    gm::set_error("gm_graph_create: shard boundaries must be multiples of 64 (got [%d,%d))", desc->row_lo, desc->row_hi);
    return GM_ERR_INVALID;
  }
  if (desc->layout != GM_LAYOUT_NATIVE && desc->layout != GM_LAYOUT_DEGREE) {
    gm::set_error("gm_graph_create: unknown layout %d", desc->layout);
    return GM_ERR_INVALID;
  }
  if (desc->layout == GM_LAYOUT_DEGREE && (desc->nshards < 1 || desc->shard < 0 || desc->shard >= desc->nshards)) {
    gm::set_error("gm_graph_create: invalid shard %d of %d", desc->shard, desc->nshards);
    return GM_ERR_INVALID;
  }
  if (nnz >= (1ll << 32)) { gm::set_error("gm_graph_create: more than 2^32-1 edges per call is unsupported"); return GM_ERR_UNSUPPORTED; }
  if (desc->edges_local) {
    int wr = 0, wn = 0;
    if (desc->layout != GM_LAYOUT_DEGREE) { gm::set_error("gm_graph_create: edges_local needs GM_LAYOUT_DEGREE"); return GM_ERR_INVALID; }
    if (desc->nshards > 65535) { gm::set_error("gm_graph_create: edges_local supports at most 65535 shards"); return GM_ERR_UNSUPPORTED; }
    if (!gm::dist_world(&wr, &wn) || wn != desc->nshards || wr != desc->shard) {
      gm::set_error("gm_graph_create: edges_local is a collective over the gm_dist communicator (have rank %d of %d, graph is shard %d of %d)",
                    wr, wn, desc->shard, desc->nshards);
      return GM_ERR_INVALID;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  gm_graph* g = new gm_graph();
  memset((void*)g, 0, sizeof(*g));
  g->desc = *desc;
  g->out = gm::CsrOwned();
  g->in = gm::CsrOwned();

  gm::DevBuf usrc, udst, uval;
  const int32_t* d_src = src;
  const int32_t* d_dst = dst;
  const void* d_val = val;
  int rc = GM_OK;
  if (!desc->ids_on_device && nnz > 0) {
    if ((rc = usrc.alloc((size_t)nnz * 4)) || (rc = udst.alloc((size_t)nnz * 4))) { if (desc->edges_local) (void)agree_on_status(rc, s); delete g; return rc; }
    hipError_t e1 = hipMemcpyAsync(usrc.p, src, (size_t)nnz * 4, hipMemcpyHostToDevice, s);
    hipError_t e2 = hipMemcpyAsync(udst.p, dst, (size_t)nnz * 4, hipMemcpyHostToDevice, s);
    if (e1 != hipSuccess || e2 != hipSuccess) { gm::set_error("edge upload failed"); if (desc->edges_local) (void)agree_on_status(GM_ERR_HIP, s); delete g; return GM_ERR_HIP; }
    d_src = usrc.as<int32_t>();
    d_dst = udst.as<int32_t>();
    if (val && desc->val_bytes > 0) {
      if ((rc = uval.alloc((size_t)nnz * desc->val_bytes))) { if (desc->edges_local) (void)agree_on_status(rc, s); delete g; return rc; }
      if (hipMemcpyAsync(uval.p, val, (size_t)nnz * desc->val_bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
        gm::set_error("edge value upload failed"); if (desc->edges_local) (void)agree_on_status(GM_ERR_HIP, s); delete g; return GM_ERR_HIP;
      }
      d_val = uval.p;
    }
  }
  if (nnz > 0) {  // every id must name a vertex: 1..nvertices (0..nvertices-1 when ids_are_native)
    gm::DevBuf badc;
    if ((rc = badc.alloc(16))) { if (desc->edges_local) (void)agree_on_status(rc, s); delete g; return rc; }
    const unsigned long long init[2] = {0ull, ~0ull};
    const int lo = desc->ids_are_native ? 0 : 1, hi = desc->ids_are_native ? desc->nvertices - 1 : desc->nvertices;
    unsigned long long bad[2] = {0ull, 0ull};
    if (hipMemcpyAsync(badc.p, init, 16, hipMemcpyHostToDevice, s) != hipSuccess) { gm::set_error("edge id check: upload failed"); if (desc->edges_local) (void)agree_on_status(GM_ERR_HIP, s); delete g; return GM_ERR_HIP; }
    hipLaunchKernelGGL(gm::k_validate_ids, dim3(gm::grid_for(nnz)), dim3(gm::kT), 0, s, d_src, d_dst, nnz, lo, hi, badc.as<unsigned long long>());
    if (hipMemcpyAsync(bad, badc.p, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      gm::set_error("edge id check failed: %s", hipGetErrorString(hipGetLastError()));
      if (desc->edges_local) (void)agree_on_status(GM_ERR_HIP, s);
      delete g;
      return GM_ERR_HIP;
    }
    if (bad[0] != 0) {
      gm::set_error("gm_graph_create: %llu edge(s) name a vertex outside [%d, %d] (first: edge %llu); ids are %s", bad[0], lo, hi,
                    bad[1] - 1ull, desc->ids_are_native ? "0-based native ids" : "1-based vertex ids as in the .mtx file");
      rc = GM_ERR_INVALID;
    }
  }
  if (desc->edges_local) rc = agree_on_status(rc, s);  // (bad ids / failed uploads on ANY rank stop all of them here)
  if (rc != GM_OK) { delete g; return rc; }
  g->desc.ndevice = desc->nvertices;
  g->desc.xchg_rows = desc->row_hi - desc->row_lo;
  g->ntiles = 1;
  if (desc->layout == GM_LAYOUT_DEGREE) rc = gm::build_degree_layout(g, nnz, d_src, d_dst, s, desc->val_bytes > 0 && d_val != nullptr);
  else g->desc.col_tiles = 1;
  if (rc != GM_OK) { gm_graph_destroy(g); return rc; }
  if (desc->edges_local) {
    if (desc->directions & GM_DIR_OUT) rc = gm::build_direction_local(g, 1, nnz, d_src, d_dst, d_val, s, &g->out);
    if (rc == GM_OK && (desc->directions & GM_DIR_IN)) rc = gm::build_direction_local(g, 0, nnz, d_src, d_dst, d_val, s, &g->in);
  } else {
    if (desc->directions & GM_DIR_OUT) rc = gm::build_direction(g, 1, nnz, d_src, d_dst, d_val, s, &g->out);
    if (rc == GM_OK && (desc->directions & GM_DIR_IN)) rc = gm::build_direction(g, 0, nnz, d_src, d_dst, d_val, s, &g->in);
  }
  if (rc != GM_OK) { gm_graph_destroy(g); return rc; }
  if (g->out.present && g->in.present) {
    const int nw = (g->desc.row_hi - g->desc.row_lo + 31) / 32 + 2;
    if (hipMalloc((void**)&g->rowbits_all, (size_t)nw * 4) != hipSuccess) { gm::set_error("rowbits alloc failed"); gm_graph_destroy(g); return GM_ERR_NOMEM; }
    hipLaunchKernelGGL(gm::k_or_words, dim3(gm::grid_for(nw)), dim3(gm::kT), 0, s, (const uint32_t*)g->out.rowbits,
                       (const uint32_t*)g->in.rowbits, g->rowbits_all, nw);
  }
  if (hipStreamSynchronize(s) != hipSuccess) { gm::set_error("graph build: stream sync failed"); gm_graph_destroy(g); return GM_ERR_HIP; }
  void* unused[4];
  if ((rc = gm_graph_run_resources(g, &unused[0], &unused[1], &unused[2], &unused[3])) != GM_OK) { gm_graph_destroy(g); return rc; }
  void* flag = nullptr;
  if (gm_graph_workspace(g, 0, 4096, &flag) == GM_OK) gm::warm_program_kernels(flag);
  *gout = g;
  return GM_OK;
}

int gm_graph_run_resources(gm_graph_t* g, void** aux_stream, void** fork_event, void** join_event, void** pinned) {
  if (!g || !aux_stream || !fork_event || !join_event || !pinned) { gm::set_error("gm_graph_run_resources: null argument"); return GM_ERR_INVALID; }
  if (!g->aux_stream) {
    GM_TRY_HIP(hipStreamCreateWithFlags(&g->aux_stream, hipStreamNonBlocking));
    GM_TRY_HIP(hipEventCreateWithFlags(&g->aux_fork, hipEventDisableTiming));
    GM_TRY_HIP(hipEventCreateWithFlags(&g->aux_join, hipEventDisableTiming));
    GM_TRY_HIP(hipHostMalloc(&g->pinned_flag, 4096, hipHostMallocDefault));
    memset(g->pinned_flag, 0, 4096);
    // first device-to-pinned-host copy of a process sets up the copy path (milliseconds): do it now
    void* d = nullptr;
    GM_TRY_HIP(hipMalloc(&d, 64));
    GM_TRY_HIP(hipMemsetAsync(d, 0, 64, 0));
    GM_TRY_HIP(hipMemcpyAsync(g->pinned_flag, d, 64, hipMemcpyDeviceToHost, 0));
    GM_TRY_HIP(hipStreamSynchronize(0));
    (void)hipFree(d);
  }
  *aux_stream = (void*)g->aux_stream;
  *fork_event = (void*)g->aux_fork;
  *join_event = (void*)g->aux_join;
  *pinned = g->pinned_flag;
  return GM_OK;
}

int gm_graph_destroy(gm_graph_t* g) {
  if (!g) return GM_OK;
  gm::free_csr(&g->out);
  gm::free_csr(&g->in);
  if (g->dev_of_native) (void)hipFree(g->dev_of_native);
  if (g->native_of_dev) (void)hipFree(g->native_of_dev);
  if (g->rowbits_all) (void)hipFree(g->rowbits_all);
  gm::free_tiles(g);
  gm::free_native_exchange(g);
  for (int i = 0; i < GM_WS_SLOTS; i++)
    if (g->ws[i] && !g->ws_external[i]) (void)hipFree(g->ws[i]);
  if (g->aux_stream) {
    (void)hipStreamSynchronize(g->aux_stream);
    (void)hipStreamDestroy(g->aux_stream);
    (void)hipEventDestroy(g->aux_fork);
    (void)hipEventDestroy(g->aux_join);
    (void)hipHostFree(g->pinned_flag);
  }
  delete g;
  return GM_OK;
}

int gm_graph_desc(const gm_graph_t* g, gm_graph_desc_t* out) {
  if (!g || !out) { gm::set_error("gm_graph_desc: null argument"); return GM_ERR_INVALID; }
  *out = g->desc;
  return GM_OK;
}

int gm_graph_csr(const gm_graph_t* g, int direction, gm_csr_t* out) {
  if (!g || !out) { gm::set_error("gm_graph_csr: null argument"); return GM_ERR_INVALID; }
  const gm::CsrOwned* c = direction == GM_DIR_OUT ? &g->out : direction == GM_DIR_IN ? &g->in : nullptr;
  if (!c || !c->present) { gm::set_error("gm_graph_csr: direction %d not built", direction); return GM_ERR_INVALID; }
  *out = c->view;
  return GM_OK;
}

int gm_graph_relayout_like(gm_graph_t* g, const gm_graph_t* like, gm_stream_t stream) {
  if (!g || !like) { gm::set_error("gm_graph_relayout_like: null graph"); return GM_ERR_INVALID; }
  if (g == like) return GM_OK;
  const gm_graph_desc_t &a = g->desc, &b = like->desc;
  // Sharded graphs (one process per shard, both graphs cut the same way): a COLLECTIVE over the gm_dist communicator.
  // In the other graph's order a row may belong to another shard, so every rank hands the edges it holds back to the
  // distributed build (build_direction_local: the owner of each edge's row under the adopted order receives it).
  const bool sharded = a.nshards > 1 || b.nshards > 1;
  if (sharded) {
    int wr = 0, wn = 0;
    if (a.nvertices != b.nvertices || a.nparts != b.nparts || a.nshards != b.nshards || a.shard != b.shard || a.ndevice != b.ndevice ||
        a.row_lo != b.row_lo || a.row_hi != b.row_hi || a.layout != GM_LAYOUT_DEGREE || b.layout != GM_LAYOUT_DEGREE ||
        !gm::dist_world(&wr, &wn) || wn != a.nshards || wr != a.shard) {
      gm::set_error("gm_graph_relayout_like: sharded graphs must be the same shard of the same cut, with the gm_dist communicator up");
      return GM_ERR_INVALID;
    }
  } else if (a.nvertices != b.nvertices || a.nparts != b.nparts || a.row_lo != 0 || a.row_hi != a.ndevice || b.row_lo != 0 ||
             b.row_hi != b.ndevice || a.ndevice != b.ndevice) {
    gm::set_error("gm_graph_relayout_like: graphs must have the same vertices and the same cut");
    return GM_ERR_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  gm::CsrOwned* from = g->in.present ? &g->in : &g->out;
  const int64_t nnz = from->view.nnz;
  gm::DevBuf src, dst;
  int rc;
  if ((rc = src.alloc((size_t)nnz * 4)) || (rc = dst.alloc((size_t)nnz * 4))) return rc;
  if (from->view.nrows > 0)
    hipLaunchKernelGGL(gm::k_csr_to_coo, dim3((from->view.nrows + 3) / 4), dim3(gm::kT), 0, s, from->view.rowptr,
                       from->view.colidx, from->view.nrows, from->view.row_base, (const int32_t*)g->native_of_dev,
                       from == &g->out ? 1 : 0, src.as<int32_t>(), dst.as<int32_t>());
  GM_TRY_HIP(hipGetLastError());
  // the edge values travel in the same (CSR) order; keep them alive while rebuilding
  void* vals = from->vals;
  from->vals = nullptr;
  const int val_bytes = from->view.val_bytes;
  gm::CsrOwned old_out = g->out, old_in = g->in;
  g->out = gm::CsrOwned();
  g->in = gm::CsrOwned();
  // adopt the other graph's device order
  if (g->dev_of_native) (void)hipFree(g->dev_of_native);
  if (g->native_of_dev) (void)hipFree(g->native_of_dev);
  g->dev_of_native = nullptr;
  g->native_of_dev = nullptr;
  if (like->dev_of_native) {
    GM_TRY_HIP(hipMalloc((void**)&g->dev_of_native, (size_t)b.nvertices * 4));
    GM_TRY_HIP(hipMalloc((void**)&g->native_of_dev, (size_t)b.ndevice * 4));
    GM_TRY_HIP(hipMemcpyAsync(g->dev_of_native, like->dev_of_native, (size_t)b.nvertices * 4, hipMemcpyDeviceToDevice, s));
    GM_TRY_HIP(hipMemcpyAsync(g->native_of_dev, like->native_of_dev, (size_t)b.ndevice * 4, hipMemcpyDeviceToDevice, s));
  }
  g->desc.layout = b.layout;
  // ... and its column tiles (the tile of a column is a function of its device id) -- but only if every vertex that has
  // an edge in g lies inside like's tiles: a vertex without edges in `like` sits behind the last tile in like's order
  // (device id >= tile_base[T]) whatever its native id, so with such columns the tiles would no longer be contiguous
  // native ranges -- the tile-after-tile fold would leave the ascending-native-column order and the tile copies of the
  // edge values would not line up with the whole CSR.  Such a graph simply keeps the untiled multiply.
  gm::free_tiles(g);
  int like_tiles = like->ntiles > 1 ? like->ntiles : 1;
  if (sharded) like_tiles = 1;
  if (like_tiles > 1 && nnz > 0) {
    gm::DevBuf flag;
    if ((rc = flag.alloc(4))) return rc;
    GM_TRY_HIP(hipMemsetAsync(flag.p, 0, 4, s));
    hipLaunchKernelGGL(gm::k_any_dev_beyond, dim3(gm::grid_for(nnz)), dim3(gm::kT), 0, s, src.as<int32_t>(), dst.as<int32_t>(), nnz,
                       (const int32_t*)like->dev_of_native, like->tile_base[like_tiles], flag.as<unsigned int>());
    unsigned int outside = 0;
    GM_TRY_HIP(hipMemcpyAsync(&outside, flag.p, 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    if (outside) like_tiles = 1;
  }
  g->ntiles = like_tiles;
  memcpy(g->tile_base, like->tile_base, sizeof(g->tile_base));
  // (... and the finer cut of the adopted order: the slices are native ranges of LIKE's order -- this graph's own cuts, made for the order it
  // had, are not: a row folded slice by slice through them would leave the ascending-native-column order, and the column-blocked stream,
  // whose "first edge of the row" is the CSR's first, would assign it in the middle of a row)
  g->nslices = like_tiles > 1 ? like->nslices : 0;
  memcpy(g->slice_base, like->slice_base, sizeof(g->slice_base));
  g->nlive = like->nlive;
  g->desc.col_tiles = g->ntiles;
  // the other graph's order ranks ITS edges: this graph's vertices with edges may sit anywhere in it
  g->desc.xchg_rows = b.row_hi - b.row_lo;
  const int saved_native = g->desc.ids_are_native, saved_vb = g->desc.val_bytes;
  g->desc.ids_are_native = 1;
  g->desc.val_bytes = val_bytes;
  rc = GM_OK;
  if (sharded) {  // this rank's edges (the rows it owned under the old order) go to their new owners
    if (g->desc.directions & GM_DIR_OUT) rc = gm::build_direction_local(g, 1, nnz, src.as<int32_t>(), dst.as<int32_t>(), vals, s, &g->out);
    if (rc == GM_OK && (g->desc.directions & GM_DIR_IN)) rc = gm::build_direction_local(g, 0, nnz, src.as<int32_t>(), dst.as<int32_t>(), vals, s, &g->in);
  } else {
    if (g->desc.directions & GM_DIR_OUT) rc = gm::build_direction(g, 1, nnz, src.as<int32_t>(), dst.as<int32_t>(), vals, s, &g->out);
    if (rc == GM_OK && (g->desc.directions & GM_DIR_IN)) rc = gm::build_direction(g, 0, nnz, src.as<int32_t>(), dst.as<int32_t>(), vals, s, &g->in);
  }
  g->desc.ids_are_native = saved_native;
  g->desc.val_bytes = saved_vb;
  if (vals) (void)hipFree(vals);
  gm::free_csr(&old_out);
  gm::free_csr(&old_in);
  memset(g->split_memo, 0, sizeof(g->split_memo));  // answers about the old adjacency
  memset(g->note_set, 0, sizeof(g->note_set));
  if (rc != GM_OK) return rc;
  if (g->rowbits_all && g->out.present && g->in.present) {
    const int nw = (g->desc.row_hi - g->desc.row_lo + 31) / 32 + 2;
    hipLaunchKernelGGL(gm::k_or_words, dim3(gm::grid_for(nw)), dim3(gm::kT), 0, s, (const uint32_t*)g->out.rowbits,
                       (const uint32_t*)g->in.rowbits, g->rowbits_all, nw);
  }
  GM_TRY_HIP(hipStreamSynchronize(s));
  return GM_OK;
}

int gm_graph_rowbits_all(const gm_graph_t* g, const uint32_t** d_bits) {
  if (!g || !d_bits) { gm::set_error("gm_graph_rowbits_all: null argument"); return GM_ERR_INVALID; }
  if (!g->rowbits_all) { gm::set_error("gm_graph_rowbits_all: graph was not built with both directions"); return GM_ERR_INVALID; }
  *d_bits = g->rowbits_all;
  return GM_OK;
}

int gm_graph_maps(const gm_graph_t* g, const int32_t** d_dev_of_native, const int32_t** d_native_of_dev) {
  if (!g) { gm::set_error("gm_graph_maps: null graph"); return GM_ERR_INVALID; }
  if (d_dev_of_native) *d_dev_of_native = g->dev_of_native;
  if (d_native_of_dev) *d_native_of_dev = g->native_of_dev;
  return GM_OK;
}

int gm_graph_maps_to_host(const gm_graph_t* g, int32_t* h_dev_of_native, int32_t* h_native_of_dev) {
  if (!g) { gm::set_error("gm_graph_maps_to_host: null graph"); return GM_ERR_INVALID; }
  const int nv = g->desc.nvertices, vd = g->desc.ndevice;
  if (h_dev_of_native) {
    if (g->dev_of_native) GM_TRY_HIP(hipMemcpy(h_dev_of_native, g->dev_of_native, (size_t)nv * 4, hipMemcpyDeviceToHost));
    else for (int i = 0; i < nv; i++) h_dev_of_native[i] = i;
  }
  if (h_native_of_dev) {
    if (g->native_of_dev) GM_TRY_HIP(hipMemcpy(h_native_of_dev, g->native_of_dev, (size_t)vd * 4, hipMemcpyDeviceToHost));
    else for (int i = 0; i < vd; i++) h_native_of_dev[i] = i;
  }
  return GM_OK;
}

int gm_graph_csr_to_host(const gm_graph_t* g, int direction, int64_t* h_rowptr, int32_t* h_colidx, void* h_vals) {
  gm_csr_t v;
  int rc = gm_graph_csr(g, direction, &v);
  if (rc) return rc;
  if (h_rowptr) GM_TRY_HIP(hipMemcpy(h_rowptr, v.rowptr, (size_t)(v.nrows + 1) * 8, hipMemcpyDeviceToHost));
  if (h_colidx && v.nnz) GM_TRY_HIP(hipMemcpy(h_colidx, v.colidx, (size_t)v.nnz * 4, hipMemcpyDeviceToHost));
  if (h_vals && v.vals && v.nnz) GM_TRY_HIP(hipMemcpy(h_vals, v.vals, (size_t)v.nnz * v.val_bytes, hipMemcpyDeviceToHost));
  return GM_OK;
}

int gm_graph_set_vals(gm_graph_t* g, int direction, const void* h_vals) {
  if (!g || !h_vals) { gm::set_error("gm_graph_set_vals: null argument"); return GM_ERR_INVALID; }
  gm::CsrOwned* c = direction == GM_DIR_OUT ? &g->out : direction == GM_DIR_IN ? &g->in : nullptr;
  if (!c || !c->present || !c->vals) { gm::set_error("gm_graph_set_vals: direction %d has no edge values", direction); return GM_ERR_INVALID; }
  if (c->view.nnz) GM_TRY_HIP(hipMemcpy(c->vals, h_vals, (size_t)c->view.nnz * c->view.val_bytes, hipMemcpyHostToDevice));
  return direction == GM_DIR_OUT ? gm_graph_sync_tile_vals(g, nullptr) : GM_OK;
}

int gm_graph_sync_tile_vals(gm_graph_t* g, gm_stream_t stream) {
  if (!g) { gm::set_error("gm_graph_sync_tile_vals: null graph"); return GM_ERR_INVALID; }
  if (!g->out.present || !g->out.vals) return GM_OK;
  hipStream_t s = (hipStream_t)stream;
  {  // the sweep's copies of the values (gm_sweep_t.sval / lval), through the CSR positions its entries came from
    const gm_sweep_t& S = g->sweep;
    if (S.nrows > 0 && S.val_bytes == 4 && S.sval && S.src_pos && S.nentries > 0)
      hipLaunchKernelGGL(gm::k_sweep_sync_vals, dim3(gm::grid_for(S.nentries)), dim3(gm::kT), 0, s, S.src_pos, (size_t)S.nentries, (const uint32_t*)g->out.vals,
                         const_cast<uint32_t*>(S.sval));
    if (S.nrows > 0 && S.val_bytes == 4 && S.gval && S.gsrc_pos && S.ngiant_edges > 0)
      hipLaunchKernelGGL(gm::k_sweep_sync_vals, dim3(gm::grid_for(S.ngiant_edges)), dim3(gm::kT), 0, s, S.gsrc_pos, (size_t)S.ngiant_edges, (const uint32_t*)g->out.vals,
                         const_cast<uint32_t*>(S.gval));
    if (S.nrows > 0 && S.val_bytes == 4 && S.lval && S.lsrc_pos && S.nedges_long > 0)
      hipLaunchKernelGGL(gm::k_sweep_sync_vals, dim3(gm::grid_for(S.nedges_long)), dim3(gm::kT), 0, s, S.lsrc_pos, (size_t)S.nedges_long, (const uint32_t*)g->out.vals,
                         const_cast<uint32_t*>(S.lval));
  }
  {  // ... and the column-blocked stream's (gm_blocked_t.eval)
    const gm_blocked_t& B = g->blocked;
    if (B.nrows > 0 && B.val_bytes == 4 && B.eval && B.epos && B.nentries > 0)
      hipLaunchKernelGGL(gm::k_sweep_sync_vals, dim3(gm::grid_for(B.nentries)), dim3(gm::kT), 0, s, B.epos, (size_t)B.nentries, (const uint32_t*)g->out.vals,
                         const_cast<uint32_t*>(B.eval));
  }
  if (g->ntiles <= 1 || !g->out_tiles) { GM_TRY_HIP(hipGetLastError()); GM_TRY_HIP(hipStreamSynchronize(s)); return GM_OK; }
  const int nrows = g->out.view.nrows, vb = g->out.view.val_bytes;
  gm::DevBuf done;
  int rc;
  if ((rc = done.alloc((size_t)(nrows + 1) * 8))) return rc;
  GM_TRY_HIP(hipMemsetAsync(done.p, 0, (size_t)(nrows + 1) * 8, s));
  // (rows of at most tile_min_row edges have no tile pieces; the tiled rows' pieces cover them completely)
  for (int t = 0; t < g->ntiles; t++) {
    gm::CsrOwned& T = g->out_tiles[t];
    if (!T.vals || T.view.nnz == 0) continue;
    hipLaunchKernelGGL(gm::k_tile_vals, dim3((nrows + gm::kT / 64 - 1) / (gm::kT / 64)), dim3(gm::kT), 0, s, (const int64_t*)g->out.rowptr,
                       (const unsigned char*)g->out.vals, (const int64_t*)T.rowptr, (unsigned char*)T.vals, nrows, vb, done.as<int64_t>());
  }
  GM_TRY_HIP(hipGetLastError());
  GM_TRY_HIP(hipStreamSynchronize(s));
  return GM_OK;
}

int gm_graph_workspace(gm_graph_t* g, int slot, size_t bytes, void** d_ptr) {
  if (!g || slot < 0 || slot >= GM_WS_SLOTS || !d_ptr) { gm::set_error("gm_graph_workspace: invalid argument"); return GM_ERR_INVALID; }
  if (g->ws_bytes[slot] < bytes) {
    if (g->ws_external[slot]) {
      gm::set_error("adopted workspace slot %d is too small (%zu < %zu bytes)", slot, g->ws_bytes[slot], bytes);
      return GM_ERR_INVALID;
    }
    if (g->ws[slot]) (void)hipFree(g->ws[slot]);
    g->ws[slot] = nullptr;
    g->ws_bytes[slot] = 0;
    hipError_t e = hipMalloc(&g->ws[slot], bytes);
    if (e != hipSuccess) { gm::set_error("workspace hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return GM_ERR_NOMEM; }
    g->ws_bytes[slot] = bytes;
  }
  *d_ptr = g->ws[slot];
  return GM_OK;
}

int gm_graph_note_set(gm_graph_t* g, int slot, int64_t value) {
  if (!g || slot < 0 || slot >= GM_NOTE_SLOTS) { gm::set_error("gm_graph_note_set: invalid argument"); return GM_ERR_INVALID; }
  g->note_val[slot] = value;
  g->note_set[slot] = 1;
  return GM_OK;
}
int gm_graph_note_get(const gm_graph_t* g, int slot, int64_t* value) {
  if (!g || slot < 0 || slot >= GM_NOTE_SLOTS || !value || !g->note_set[slot]) return GM_ERR_INVALID;
  *value = g->note_val[slot];
  return GM_OK;
}

int gm_graph_workspace_info(const gm_graph_t* g, int slot, void** d_ptr, size_t* bytes, int* external) {
  if (!g || slot < 0 || slot >= GM_WS_SLOTS || !d_ptr || !bytes || !external) { gm::set_error("gm_graph_workspace_info: invalid argument"); return GM_ERR_INVALID; }
  *d_ptr = g->ws[slot];
  *bytes = g->ws_bytes[slot];
  *external = g->ws_external[slot];
  return GM_OK;
}

int gm_graph_split(const gm_graph_t* g, int direction, int head_permille, int32_t* row_split, int32_t* blk_split,
                   int32_t* mid_split) {
  if (!g || !row_split || !blk_split || !mid_split || head_permille < 1 || head_permille > 999) { gm::set_error("gm_graph_split: invalid argument"); return GM_ERR_INVALID; }
  const gm::CsrOwned* c = direction == GM_DIR_OUT ? &g->out : direction == GM_DIR_IN ? &g->in : nullptr;
  if (!c || !c->present) { gm::set_error("gm_graph_split: direction %d not built", direction); return GM_ERR_INVALID; }
  const gm_csr_t& A = c->view;
  gm_graph::SplitMemo* memo = const_cast<gm_graph::SplitMemo*>(g->split_memo[direction == GM_DIR_OUT ? 0 : 1]);
  for (int k = 0; k < 2; k++)
    if (memo[k].valid && memo[k].permille == head_permille && memo[k].asked == *row_split) {
      *row_split = memo[k].rs; *blk_split = memo[k].bs; *mid_split = memo[k].ms;
      return GM_OK;
    }
  const int32_t asked = *row_split;
  auto rd64 = [&](const int64_t* p, int64_t i, int64_t* out) { return hipMemcpy(out, p + i, 8, hipMemcpyDeviceToHost); };
  auto rd32 = [&](const int32_t* p, int64_t i, int32_t* out) { return hipMemcpy(out, p + i, 4, hipMemcpyDeviceToHost); };
  const int64_t target = (A.nnz * head_permille + 999) / 1000;
  // smallest multiple of 64 (capped at nrows) with rowptr[row] >= target
  int64_t lo = 0, hi = ((int64_t)A.nrows + 63) / 64;
  while (lo < hi) {
    const int64_t mid = (lo + hi) / 2;
    const int64_t r = mid * 64 < A.nrows ? mid * 64 : A.nrows;
    int64_t v = 0;
    GM_TRY_HIP(rd64(A.rowptr, r, &v));
    if (v >= target) hi = mid; else lo = mid + 1;
  }
  int64_t rs = lo * 64 < A.nrows ? lo * 64 : A.nrows;
  if (*row_split > 0) rs = (int64_t)(*row_split) < A.nrows ? (int64_t)(*row_split) / 64 * 64 : A.nrows;  // caller's choice
  // first row-block whose rows end beyond rs
  int64_t bl = 0, bh = A.nblk;
  while (bl < bh) {
    const int64_t mid = (bl + bh) / 2;
    int32_t sg = 0, r1 = 0;
    GM_TRY_HIP(rd32(A.blk_seg, mid, &sg));
    GM_TRY_HIP(rd32(A.seg_row, (int64_t)sg + 1, &r1));
    if ((int64_t)r1 > rs) bh = mid; else bl = mid + 1;
  }
  int64_t ml = 0, mh = A.nmid;
  while (ml < mh) {
    const int64_t mid = (ml + mh) / 2;
    int32_t r = 0;
    GM_TRY_HIP(rd32(A.mid_row, mid, &r));
    if ((int64_t)r >= rs) mh = mid; else ml = mid + 1;
  }
  if (A.ngiant > 0) {
    int32_t last = 0;
    GM_TRY_HIP(rd32(A.giant_row, (int64_t)A.ngiant - 1, &last));
    if ((int64_t)last >= rs) { gm::set_error("gm_graph_split: a giant row lies in the tail"); return GM_ERR_INVALID; }
  }
  *row_split = (int32_t)rs;
  *blk_split = (int32_t)bl;
  *mid_split = (int32_t)ml;
  gm_graph::SplitMemo& m = memo[asked == 0 ? 0 : 1];
  m.valid = 1; m.permille = head_permille; m.asked = asked; m.rs = (int32_t)rs; m.bs = (int32_t)bl; m.ms = (int32_t)ml;
  return GM_OK;
}

int gm_graph_adopt_workspace(gm_graph_t* g, int slot, void* d_ptr, size_t bytes) {
  if (!g || slot < 0 || slot >= GM_WS_SLOTS) { gm::set_error("gm_graph_adopt_workspace: invalid argument"); return GM_ERR_INVALID; }
  if (g->ws[slot] && !g->ws_external[slot]) (void)hipFree(g->ws[slot]);
  g->ws[slot] = d_ptr;
  g->ws_bytes[slot] = d_ptr ? bytes : 0;
  g->ws_external[slot] = d_ptr ? 1 : 0;
  return GM_OK;
}

int gm_graph_set_exchange(gm_graph_t* g, gm_exchange_fn fn, void* ctx) {
  if (!g) { gm::set_error("gm_graph_set_exchange: null graph"); return GM_ERR_INVALID; }
  g->xfn = fn;
  g->xctx = ctx;
  g->xcaps = 0;
  return GM_OK;
}
int gm_graph_set_exchange_caps(gm_graph_t* g, int caps) {
  if (!g) { gm::set_error("gm_graph_set_exchange_caps: null graph"); return GM_ERR_INVALID; }
  g->xcaps = caps;
  return GM_OK;
}
int gm_graph_exchange_caps(const gm_graph_t* g) { return (g && g->xfn) ? g->xcaps : 0; }
int gm_graph_has_exchange(const gm_graph_t* g) { return (g && g->xfn) ? 1 : 0; }
int gm_graph_exchange(gm_graph_t* g, int kind, void* d_ptr, int64_t elt_bytes, uint32_t* d_bits, int* h_flag) {
  if (!g || !g->xfn) return GM_OK;
  return g->xfn(g->xctx, kind, d_ptr, elt_bytes, d_bits, h_flag);
}
int gm_graph_enable_timing(gm_graph_t* g, int on) {
  if (!g) return GM_ERR_INVALID;
  g->timing = on ? 1 : 0;
  return GM_OK;
}
int gm_graph_timing_enabled(const gm_graph_t* g) { return g ? g->timing : 0; }
int gm_graph_record_stats(gm_graph_t* g, const gm_run_stats_t* st) {
  if (!g || !st) return GM_ERR_INVALID;
  g->stats = *st;
  return GM_OK;
}
int gm_graph_last_stats(const gm_graph_t* g, gm_run_stats_t* out) {
  if (!g || !out) return GM_ERR_INVALID;
  *out = g->stats;
  return GM_OK;
}

}  // extern "C"
