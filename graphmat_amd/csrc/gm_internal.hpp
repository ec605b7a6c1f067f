// gm_internal.hpp -- shared by the translation units of libgraphmat_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "graphmat_hip.h"

namespace gm {

void set_error(const char* fmt, ...);

#define GM_TRY_HIP(expr)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      gm::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GM_ERR_HIP;                                                                  \
    }                                                                                     \
  } while (0)

// 0-based native id of 1-based vertex id; include/Graph.h:111-130 of the reference
__host__ __device__ inline int to_native0(int vertex1, int nparts, int len) {
  int v = vertex1 - 1;
  int height = len / nparts;
  int vmax = height * nparts;
  if (v >= vmax) return v;
  return (v / nparts) + (v % nparts) * height;
}
// 1-based vertex id of 0-based native id; include/Graph.h:132-150
__host__ __device__ inline int to_vertex1(int native0, int nparts, int len) {
  int v = native0;
  int height = len / nparts;
  int vmax = height * nparts;
  if (v >= vmax) return v + 1;
  return (v / height) + (v % height) * nparts + 1;
}

struct CsrOwned {
  gm_csr_t view;
  int64_t* rowptr = nullptr;
  int32_t* colidx = nullptr;
  void* vals = nullptr;
  uint32_t* rowbits = nullptr;
  int32_t* seg_row = nullptr;
  int32_t* blk_seg = nullptr;
  int32_t* mid_row = nullptr;
  int32_t* giant_row = nullptr;
  int32_t* gchunk_row = nullptr;
  int64_t* gchunk_edge = nullptr;
  int64_t* gterm_off = nullptr;
  int32_t* umid_row = nullptr;
  void* gchunk_state = nullptr;
  bool present = false;
};

}  // namespace gm

namespace gm { void warm_program_kernels(void* d_scratch256); }

struct gm_graph {
  gm_graph_desc_t desc;
  gm::CsrOwned out, in;
  int32_t* dev_of_native;  // nullptr = identity
  int32_t* native_of_dev;
  uint32_t* rowbits_all;
  void* ws[GM_WS_SLOTS];
  size_t ws_bytes[GM_WS_SLOTS];
  int ws_external[GM_WS_SLOTS];
  gm_exchange_fn xfn;
  void* xctx;
  int timing;
  gm_run_stats_t stats;
  // per-graph run resources (gm_graph_run_resources): created with the graph so that no
  // stream / pinned-memory creation lands inside a timed run_graph_program call
  hipStream_t aux_stream;
  hipEvent_t aux_fork, aux_join;
  void* pinned_flag;
  // small per-graph memo for the header layer (gm_graph_note_*): e.g. the row split the shards agreed on
  int64_t note_val[GM_NOTE_SLOTS];
  int note_set[GM_NOTE_SLOTS];
  // last answers of gm_graph_split per direction (the search costs ~70 small device reads)
  struct SplitMemo { int valid, permille; int32_t asked, rs, bs, ms; } split_memo[2][2];
  // column tiles of the GM_DIR_OUT adjacency (gm_graph_tile); ntiles <= 1: none
  int ntiles;
  int32_t tile_base[GM_MAX_TILES + 1];  // device ids [tile_base[t], tile_base[t+1]) belong to tile t (tile_base[ntiles] = nlive)
  int32_t nlive;                // vertices with at least one edge (they come first in the device order)
  gm::CsrOwned* out_tiles;      // [ntiles]
  uint32_t** out_tile_prev;     // [ntiles] presence bits of the rows with an edge in an earlier tile
  // finer cut of the device order (gm_graph_sweep): nslices = ntiles * k slices, slice_base[nslices] = nlive; 0: none
  int nslices;
  int32_t slice_base[GM_MAX_SLICES + 2];
  gm_sweep_t sweep;             // device arrays owned by the graph (nrows = 0: not built)
  int32_t* d_slice_base;
  gm_blocked_t blocked;         // the short rows of a graph without skew as a column-blocked stream (nrows = 0: not built)
  // native RCCL exchange (gm_dist.hip): state behind xfn/xctx when gm_graph_use_rccl installed it
  int xcaps;                    // GM_XCAP_* of the installed exchange
  void* native_xchg;
  hipStream_t run_stream;       // the stream of the run in progress (gm_graph_set_run_stream): collectives are enqueued on it
  // engine options this graph overrides (gm_graph_set_option); bit i of opt_set = field i of gm_engine_options_t
  gm_engine_options_t opt;
  uint32_t opt_set;
};
namespace gm {
void free_native_exchange(gm_graph* g);
// collectives over the gm_dist communicator for a distributed graph build (gm_dist.hip)
int dist_world(int* rank, int* nranks);  // 1 when a communicator exists
int dist_all_reduce_sum_u32(uint32_t* d, size_t n, hipStream_t s);
int dist_all_gather_bytes(const void* d_send, void* d_recv, size_t bytes, hipStream_t s);
// one step of a ring pipeline: send `send_bytes` to rank+1 and receive `recv_bytes` from rank-1 (either may be 0: the
// ends of the pipeline, or a rank idle in this step).  Every rank of the communicator calls it for every step.
int dist_ring_step(const void* d_send, size_t send_bytes, void* d_recv, size_t recv_bytes, hipStream_t s);
int dist_broadcast(void* d_buf, size_t bytes, int root, hipStream_t s);
}
