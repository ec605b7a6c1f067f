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
#include <vector>

#include "gm_internal.hpp"

namespace gm {

constexpr int kT = 256;
inline int grid_for(int64_t n) { return (int)((n + kT - 1) / kT); }

// key = (local row + dropped-flag) << 32 | col.  Edges whose row is outside the shard
// get row field = nrows (sorts last).
__global__ void __launch_bounds__(kT)
k_make_keys(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t nnz, int by_dst, int nparts,
            int nv, int row_lo, int row_hi, int ids_are_native, uint64_t* __restrict__ keys,
            uint32_t* __restrict__ idx, unsigned long long* __restrict__ kept) {
  int64_t e = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (e >= nnz) return;
  int s = src[e], d = dst[e];
  int sn = ids_are_native ? s : to_native0(s, nparts, nv);
  int dn = ids_are_native ? d : to_native0(d, nparts, nv);
  int r = by_dst ? dn : sn;
  int c = by_dst ? sn : dn;
  uint32_t rf;
  if (r >= row_lo && r < row_hi) {
    rf = (uint32_t)(r - row_lo);
    atomicAdd(kept, 1ull);
  } else {
    rf = (uint32_t)(row_hi - row_lo);
  }
  keys[e] = ((uint64_t)rf << 32) | (uint32_t)c;
  idx[e] = (uint32_t)e;
}

__global__ void __launch_bounds__(kT)
k_unpack(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, int64_t n, const void* __restrict__ val,
         int val_bytes, int32_t* __restrict__ colidx, void* __restrict__ vals) {
  int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (k >= n) return;
  colidx[k] = (int32_t)(uint32_t)keys[k];
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

// row-block starts and long-row flags (see include/graphmat/kernels.hpp for the scheme)
__global__ void __launch_bounds__(kT)
k_block_flags(const int64_t* __restrict__ rowptr, int nrows, unsigned char* __restrict__ start,
              unsigned char* __restrict__ islong) {
  int r = blockIdx.x * kT + threadIdx.x;
  if (r >= nrows) return;
  int64_t a = rowptr[r], b = rowptr[r + 1];
  bool lng = (b - a) > GM_LONG_ROW;
  bool st = (r == 0) || ((r & 255) == 0) || lng;
  if (!st) {
    int64_t pa = rowptr[r - 1];
    bool prev_long = (a - pa) > GM_LONG_ROW;
    st = prev_long || (a / GM_BLOCK_NNZ != pa / GM_BLOCK_NNZ);
  }
  start[r] = st ? 1 : 0;
  islong[r] = lng ? 1 : 0;
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
  template <class T> T* as() { return (T*)p; }
  void* release() { void* q = p; p = nullptr; return q; }
};

static int bits_for(uint32_t maxval) {
  int b = 0;
  while ((1ull << b) <= (uint64_t)maxval) b++;
  return b;
}

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
    hipLaunchKernelGGL(k_make_keys, dim3(grid_for(nnz)), dim3(kT), 0, s, d_src, d_dst, nnz, by_dst, D.nparts,
                       D.nvertices, D.row_lo, D.row_hi, D.ids_are_native, keys_in.as<uint64_t>(),
                       idx_in.as<uint32_t>(), kept_d.as<unsigned long long>());
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
  keys_in.alloc(0);
  idx_in.alloc(0);
  tmp.alloc(0);

  DevBuf rowptr, colidx, vals, start, islong, blk, lng, cnt;
  if ((rc = rowptr.alloc((size_t)(nrows + 1) * 8))) return rc;
  if ((rc = colidx.alloc((size_t)kept * 4))) return rc;
  const bool keep_vals = D.val_bytes > 0 && d_val != nullptr;
  if (keep_vals && (rc = vals.alloc((size_t)kept * D.val_bytes))) return rc;
  if (kept > 0) {
    hipLaunchKernelGGL(k_unpack, dim3(grid_for((int64_t)kept)), dim3(kT), 0, s, keys_out.as<uint64_t>(),
                       idx_out.as<uint32_t>(), (int64_t)kept, d_val, D.val_bytes, colidx.as<int32_t>(),
                       keep_vals ? vals.p : nullptr);
  }
  hipLaunchKernelGGL(k_rowptr, dim3(grid_for(nrows + 1)), dim3(kT), 0, s, keys_out.as<uint64_t>(), (int64_t)kept,
                     nrows, rowptr.as<int64_t>());
  GM_TRY_HIP(hipGetLastError());

  // row-blocks and long rows
  if ((rc = start.alloc((size_t)nrows + 1))) return rc;
  if ((rc = islong.alloc((size_t)nrows + 1))) return rc;
  if ((rc = blk.alloc((size_t)(nrows + 2) * 4))) return rc;
  if ((rc = lng.alloc((size_t)(nrows + 1) * 4))) return rc;
  if ((rc = cnt.alloc(16))) return rc;
  unsigned int nblk = 0, nlong = 0;
  if (nrows > 0) {
    hipLaunchKernelGGL(k_block_flags, dim3(grid_for(nrows)), dim3(kT), 0, s, rowptr.as<int64_t>(), nrows,
                       start.as<unsigned char>(), islong.as<unsigned char>());
    GM_TRY_HIP(hipGetLastError());
    rocprim::counting_iterator<int32_t> rows(0);
    size_t tb = 0, tb2 = 0;
    GM_TRY_HIP(rocprim::select(nullptr, tb, rows, start.as<unsigned char>(), blk.as<int32_t>(), cnt.as<unsigned int>(),
                               (size_t)nrows, s));
    GM_TRY_HIP(rocprim::select(nullptr, tb2, rows, islong.as<unsigned char>(), lng.as<int32_t>(),
                               cnt.as<unsigned int>() + 1, (size_t)nrows, s));
    if ((rc = tmp.alloc(std::max(tb, tb2)))) return rc;
    GM_TRY_HIP(rocprim::select(tmp.p, tb, rows, start.as<unsigned char>(), blk.as<int32_t>(), cnt.as<unsigned int>(),
                               (size_t)nrows, s));
    GM_TRY_HIP(rocprim::select(tmp.p, tb2, rows, islong.as<unsigned char>(), lng.as<int32_t>(),
                               cnt.as<unsigned int>() + 1, (size_t)nrows, s));
    unsigned int h[2] = {0, 0};
    GM_TRY_HIP(hipMemcpyAsync(h, cnt.p, 8, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    nblk = h[0];
    nlong = h[1];
    int32_t last = nrows;
    GM_TRY_HIP(hipMemcpyAsync(blk.as<int32_t>() + nblk, &last, 4, hipMemcpyHostToDevice, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
  }

  out->rowptr = (int64_t*)rowptr.release();
  out->colidx = (int32_t*)colidx.release();
  out->vals = keep_vals ? vals.release() : nullptr;
  out->blk_row = (int32_t*)blk.release();
  out->long_row = (int32_t*)lng.release();
  out->present = true;
  gm_csr_t& v = out->view;
  v.nnz = (int64_t)kept;
  v.nrows = nrows;
  v.row_base = D.row_lo;
  v.ncols = D.nvertices;
  v.val_bytes = keep_vals ? D.val_bytes : 0;
  v.rowptr = out->rowptr;
  v.colidx = out->colidx;
  v.vals = out->vals;
  v.blk_row = out->blk_row;
  v.nblk = (int32_t)nblk;
  v.long_row = out->long_row;
  v.nlong = (int32_t)nlong;
  return GM_OK;
}

static void free_csr(CsrOwned* c) {
  if (c->rowptr) (void)hipFree(c->rowptr);
  if (c->colidx) (void)hipFree(c->colidx);
  if (c->vals) (void)hipFree(c->vals);
  if (c->blk_row) (void)hipFree(c->blk_row);
  if (c->long_row) (void)hipFree(c->long_row);
  *c = CsrOwned();
}

}  // namespace gm

extern "C" {

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
  if ((desc->row_lo & 63) != 0 || ((desc->row_hi & 63) != 0 && desc->row_hi != desc->nvertices)) {
    gm::set_error("gm_graph_create: shard boundaries must be multiples of 64 (got [%d,%d))", desc->row_lo, desc->row_hi);
    return GM_ERR_INVALID;
  }
  if (nnz >= (1ll << 32)) { gm::set_error("gm_graph_create: more than 2^32-1 edges per call is unsupported"); return GM_ERR_UNSUPPORTED; }
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
    if ((rc = usrc.alloc((size_t)nnz * 4)) || (rc = udst.alloc((size_t)nnz * 4))) { delete g; return rc; }
    hipError_t e1 = hipMemcpyAsync(usrc.p, src, (size_t)nnz * 4, hipMemcpyHostToDevice, s);
    hipError_t e2 = hipMemcpyAsync(udst.p, dst, (size_t)nnz * 4, hipMemcpyHostToDevice, s);
    if (e1 != hipSuccess || e2 != hipSuccess) { gm::set_error("edge upload failed"); delete g; return GM_ERR_HIP; }
    d_src = usrc.as<int32_t>();
    d_dst = udst.as<int32_t>();
    if (val && desc->val_bytes > 0) {
      if ((rc = uval.alloc((size_t)nnz * desc->val_bytes))) { delete g; return rc; }
      if (hipMemcpyAsync(uval.p, val, (size_t)nnz * desc->val_bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
        gm::set_error("edge value upload failed"); delete g; return GM_ERR_HIP;
      }
      d_val = uval.p;
    }
  }
  if (desc->directions & GM_DIR_OUT) rc = gm::build_direction(g, 1, nnz, d_src, d_dst, d_val, s, &g->out);
  if (rc == GM_OK && (desc->directions & GM_DIR_IN)) rc = gm::build_direction(g, 0, nnz, d_src, d_dst, d_val, s, &g->in);
  if (rc != GM_OK) { gm_graph_destroy(g); return rc; }
  if (hipStreamSynchronize(s) != hipSuccess) { gm::set_error("graph build: stream sync failed"); gm_graph_destroy(g); return GM_ERR_HIP; }
  *gout = g;
  return GM_OK;
}

int gm_graph_destroy(gm_graph_t* g) {
  if (!g) return GM_OK;
  gm::free_csr(&g->out);
  gm::free_csr(&g->in);
  for (int i = 0; i < GM_WS_SLOTS; i++)
    if (g->ws[i]) (void)hipFree(g->ws[i]);
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
  return GM_OK;
}

int gm_graph_workspace(gm_graph_t* g, int slot, size_t bytes, void** d_ptr) {
  if (!g || slot < 0 || slot >= GM_WS_SLOTS || !d_ptr) { gm::set_error("gm_graph_workspace: invalid argument"); return GM_ERR_INVALID; }
  if (g->ws_bytes[slot] < bytes) {
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

int gm_graph_set_exchange(gm_graph_t* g, gm_exchange_fn fn, void* ctx) {
  if (!g) { gm::set_error("gm_graph_set_exchange: null graph"); return GM_ERR_INVALID; }
  g->xfn = fn;
  g->xctx = ctx;
  return GM_OK;
}
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
