// gm_dist.hip -- the message exchange of row-sharded graphs, natively on RCCL.
//
// Replaces the reference's multi-rank SpMSpV transport (include/GMDP/multinode/spmspv.h:62-116: x
// segments broadcast along tile columns with MPI point-to-point, then the y row-reduce of :141-203) and
// its convergence Allreduce (include/GraphMatRuntime.h:226).  With 1-D row sharding every shard owns
// whole rows, so the only data-path collective is making x global between send and multiply: an
// all-gather of the shards' slices (values + presence words), plus a 1-int MIN all-reduce for the
// convergence flag.  Everything here is enqueued on HIP streams -- no Python, no host round trip per
// iteration (the flag all-reduce aside, whose result the host loop needs anyway).
//
// RCCL is bound when gm_dist_init is called (dlopen of librccl.so.1), so single-GPU processes never map
// the 570 MB library.
#include <dlfcn.h>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <vector>

#include "gm_internal.hpp"

namespace gm {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static ncclComm_t g_comm = nullptr;
static int g_rank = 0, g_nranks = 1;

static int bind_rccl() {
  if (g_rccl.handle) return GM_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { set_error("gm_dist: cannot load librccl.so.1: %s", dlerror()); return GM_ERR_UNSUPPORTED; }
#define GM_BIND(field, sym)                                                                  \
  *(void**)(&g_rccl.field) = dlsym(h, sym);                                                  \
  if (!g_rccl.field) { set_error("gm_dist: librccl has no symbol %s", sym); dlclose(h); return GM_ERR_UNSUPPORTED; }
  GM_BIND(GetUniqueId, "ncclGetUniqueId")
  GM_BIND(CommInitRank, "ncclCommInitRank")
  GM_BIND(CommDestroy, "ncclCommDestroy")
  GM_BIND(AllGather, "ncclAllGather")
  GM_BIND(AllReduce, "ncclAllReduce")
  GM_BIND(GetErrorString, "ncclGetErrorString")
#undef GM_BIND
  g_rccl.handle = h;
  return GM_OK;
}

#define GM_TRY_NCCL(expr)                                                                         \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess) {                                                                      \
      gm::set_error("%s: %s (%s:%d)", #expr, gm::g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
      return GM_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

// Per-graph state of the native exchange.
struct RcclExchange {
  gm_graph* g = nullptr;
  hipStream_t xs = nullptr;        // side stream of the overlapped (GM_XCHG_PART) transfers
  hipEvent_t ready = nullptr;      // "the part's messages have been written" (recorded on the run stream)
  hipEvent_t arrived = nullptr;    // "all parts issued so far have arrived" (recorded on the side stream)
  void* stage = nullptr;           // compact staging of the live prefixes (plain exchange)
  size_t stage_bytes = 0;
  void* stage_bits = nullptr;
  size_t stage_bits_bytes = 0;
  void* part_stage = nullptr;      // staging of the parts (overlapped exchange)
  size_t part_stage_bytes = 0;
  int* d_flag = nullptr;           // convergence flag on the device
  struct Part { void* buf; int64_t first, count, elt; };
  std::vector<Part> pending;
  long long calls = 0, parts = 0, sparse_gathers = 0;
  unsigned long long bytes_sent = 0;  // bytes this rank contributed to all-gathers
};

static int grow(void** p, size_t* have, size_t need) {
  if (*have >= need) return GM_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  hipError_t e = hipMalloc(p, need);
  if (e != hipSuccess) { set_error("gm_dist: hipMalloc(%zu): %s", need, hipGetErrorString(e)); return GM_ERR_NOMEM; }
  *have = need;
  return GM_OK;
}

// slices are equal (GM_LAYOUT_DEGREE): shard r owns device rows [r*S, (r+1)*S); only the first L rows of a
// slice are ever read by other shards (gm_graph_desc_t.xchg_rows)
static int exchange_messages(RcclExchange* X, void* d_x, int64_t elt, uint32_t* d_bits) {
  gm_graph* g = X->g;
  const int n = g_nranks, r = g_rank;
  const int64_t S = g->desc.row_hi - g->desc.row_lo;
  int64_t L = g->desc.xchg_rows;
  if (L <= 0 || L > S || (L & 63)) L = S;
  hipStream_t s = g->run_stream;
  char* x = (char*)d_x;
  if (n == 1 && (S & 63) != 0) return GM_OK;  // a lone shard whose size is no multiple of 64: nothing to move
  if (L == S && (S & 31) == 0) {
    // whole slices: one in-place all-gather per array (send buffer = this rank's slot of the receive buffer)
    GM_TRY_NCCL(g_rccl.AllGather(x + (size_t)r * S * elt, x, (size_t)(S * elt), ncclChar, g_comm, s));
    if (d_bits) GM_TRY_NCCL(g_rccl.AllGather(d_bits + (size_t)r * (S / 32), d_bits, (size_t)(S / 32), ncclUint32, g_comm, s));
  } else {
    // only the live prefix travels: gather into a compact staging buffer, then put every shard's prefix at
    // the front of its slice (2-D copy: n rows of L*elt bytes, source pitch L*elt, destination pitch S*elt)
    int rc;
    if ((rc = grow(&X->stage, &X->stage_bytes, (size_t)n * L * elt))) return rc;
    GM_TRY_NCCL(g_rccl.AllGather(x + (size_t)r * S * elt, X->stage, (size_t)(L * elt), ncclChar, g_comm, s));
    GM_TRY_HIP(hipMemcpy2DAsync(x, (size_t)(S * elt), X->stage, (size_t)(L * elt), (size_t)(L * elt), (size_t)n,
                                hipMemcpyDeviceToDevice, s));
    if (d_bits) {
      const int64_t W = S / 32, WL = L / 32;  // S and L are multiples of 64
      if ((rc = grow(&X->stage_bits, &X->stage_bits_bytes, (size_t)n * WL * 4))) return rc;
      GM_TRY_NCCL(g_rccl.AllGather(d_bits + (size_t)r * W, X->stage_bits, (size_t)WL, ncclUint32, g_comm, s));
      GM_TRY_HIP(hipMemcpy2DAsync(d_bits, (size_t)W * 4, X->stage_bits, (size_t)WL * 4, (size_t)WL * 4, (size_t)n,
                                  hipMemcpyDeviceToDevice, s));
    }
  }
  X->bytes_sent += (unsigned long long)(L * elt) + (d_bits ? (unsigned long long)(L / 8) : 0ull);
  return GM_OK;
}

// every shard's rows [first, first+count) of its slice of `buf` start travelling on the side stream
static int start_part(RcclExchange* X, void* buf, int64_t first, int64_t count, int64_t elt) {
  gm_graph* g = X->g;
  const int n = g_nranks, r = g_rank;
  const int64_t S = g->desc.row_hi - g->desc.row_lo;
  int64_t L = g->desc.xchg_rows;
  if (L <= 0 || L > S || (L & 63)) L = S;
  if (n == 1 && (S & 63) != 0) return GM_OK;
  if (first < 0 || count <= 0 || first + count > L) { set_error("gm_dist: part [%lld,+%lld) outside the live rows %lld", (long long)first, (long long)count, (long long)L); return GM_ERR_INVALID; }
  int rc;
  if (X->pending.empty()) {  // (never grown while transfers into it are in flight)
    if ((rc = grow(&X->part_stage, &X->part_stage_bytes, (size_t)n * L * elt))) return rc;
  } else if (X->part_stage_bytes < (size_t)n * L * elt) {
    set_error("gm_dist: part staging buffer too small for a second element size");
    return GM_ERR_INVALID;
  }
  // each part owns its own region of the staging buffer: n blocks of count*elt bytes
  char* st = (char*)X->part_stage + (size_t)n * first * elt;
  GM_TRY_HIP(hipEventRecord(X->ready, g->run_stream));
  GM_TRY_HIP(hipStreamWaitEvent(X->xs, X->ready, 0));
  GM_TRY_NCCL(g_rccl.AllGather((char*)buf + (size_t)(r * S + first) * elt, st, (size_t)(count * elt), ncclChar, g_comm, X->xs));
  X->pending.push_back({buf, first, count, elt});
  X->parts++;
  X->bytes_sent += (unsigned long long)(count * elt);
  return GM_OK;
}

// work enqueued on the run stream afterwards sees all parts in place
static int wait_parts(RcclExchange* X) {
  if (X->pending.empty()) return GM_OK;
  gm_graph* g = X->g;
  const int n = g_nranks;
  const int64_t S = g->desc.row_hi - g->desc.row_lo;
  hipStream_t s = g->run_stream;
  GM_TRY_HIP(hipEventRecord(X->arrived, X->xs));
  GM_TRY_HIP(hipStreamWaitEvent(s, X->arrived, 0));
  for (const RcclExchange::Part& p : X->pending) {
    const char* st = (const char*)X->part_stage + (size_t)n * p.first * p.elt;
    GM_TRY_HIP(hipMemcpy2DAsync((char*)p.buf + (size_t)p.first * p.elt, (size_t)(S * p.elt), st, (size_t)(p.count * p.elt),
                                (size_t)(p.count * p.elt), (size_t)n, hipMemcpyDeviceToDevice, s));
  }
  X->pending.clear();
  // the side stream must not overwrite the staging buffer before these copies have read it
  GM_TRY_HIP(hipEventRecord(X->ready, s));
  GM_TRY_HIP(hipStreamWaitEvent(X->xs, X->ready, 0));
  return GM_OK;
}

static int all_reduce_min(RcclExchange* X, int* h_flag) {
  hipStream_t s = X->g->run_stream;
  GM_TRY_HIP(hipMemcpyAsync(X->d_flag, h_flag, sizeof(int), hipMemcpyHostToDevice, s));
  GM_TRY_NCCL(g_rccl.AllReduce(X->d_flag, X->d_flag, 1, ncclInt32, ncclMin, g_comm, s));
  GM_TRY_HIP(hipMemcpyAsync(h_flag, X->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  return GM_OK;
}

// convergence flag (AND) + size of the local active set (max, sum) in one all-gather of two ints per shard
static int exchange_state(RcclExchange* X, int* h_flag) {
  hipStream_t s = X->g->run_stream;
  const int n = g_nranks, r = g_rank;
  int rc;
  if ((rc = grow(&X->stage_bits, &X->stage_bits_bytes, (size_t)n * 8 + 64))) return rc;
  int* d = (int*)X->stage_bits;
  GM_TRY_HIP(hipMemcpyAsync(d + 2 * r, h_flag, 8, hipMemcpyHostToDevice, s));
  GM_TRY_NCCL(g_rccl.AllGather(d + 2 * r, d, 2, ncclInt32, g_comm, s));
  std::vector<int> all((size_t)2 * n);
  GM_TRY_HIP(hipMemcpyAsync(all.data(), d, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  GM_TRY_HIP(hipStreamSynchronize(s));
  int conv = 1, mx = 0;
  long long sum = 0;
  for (int i = 0; i < n; i++) {
    conv = conv && all[2 * i] != 0;
    mx = all[2 * i + 1] > mx ? all[2 * i + 1] : mx;
    sum += all[2 * i + 1];
  }
  h_flag[0] = conv ? 1 : 0;
  h_flag[1] = mx;
  h_flag[2] = sum > 0x7fffffffll ? 0x7fffffff : (int)sum;
  return GM_OK;
}

// in-place all-gather of equal blocks of `cap` entries
static int gather_blocks(RcclExchange* X, void* buf, int64_t elt, int cap) {
  if (cap <= 0 || elt <= 0) { set_error("gm_dist: empty gather"); return GM_ERR_INVALID; }
  const size_t block = (size_t)cap * (size_t)elt;
  GM_TRY_NCCL(g_rccl.AllGather((char*)buf + (size_t)g_rank * block, buf, block, ncclChar, g_comm, X->g->run_stream));
  X->bytes_sent += (unsigned long long)block;
  X->sparse_gathers++;
  return GM_OK;
}

// the gm_exchange_fn the library installs on itself (graphmat_hip.h: GM_XCHG_*)
static int native_exchange(void* ctx, int kind, void* d_ptr, int64_t elt_bytes, uint32_t* d_bits, int* h_flag) {
  RcclExchange* X = (RcclExchange*)ctx;
  X->calls++;
  int rc = GM_ERR_INVALID;
  if (kind == GM_XCHG_MESSAGES) rc = exchange_messages(X, d_ptr, elt_bytes, d_bits);
  else if (kind == GM_XCHG_PART) rc = h_flag ? start_part(X, d_ptr, h_flag[0], h_flag[1], elt_bytes) : GM_ERR_INVALID;
  else if (kind == GM_XCHG_WAIT) rc = wait_parts(X);
  else if (kind == GM_XCHG_CONVERGED) rc = h_flag ? all_reduce_min(X, h_flag) : GM_ERR_INVALID;
  else if (kind == GM_XCHG_STATE) rc = h_flag ? exchange_state(X, h_flag) : GM_ERR_INVALID;
  else if (kind == GM_XCHG_GATHER) rc = h_flag ? gather_blocks(X, d_ptr, elt_bytes, h_flag[0]) : GM_ERR_INVALID;
  else set_error("gm_dist: unknown exchange kind %d", kind);
  if (rc != GM_OK) fprintf(stderr, "GraphMat(HIP): RCCL exchange failed: %s\n", gm_last_error());
  return rc;
}

void free_native_exchange(gm_graph* g) {
  RcclExchange* X = (RcclExchange*)g->native_xchg;
  if (!X) return;
  if (X->xs) { (void)hipStreamSynchronize(X->xs); (void)hipStreamDestroy(X->xs); }
  if (X->ready) (void)hipEventDestroy(X->ready);
  if (X->arrived) (void)hipEventDestroy(X->arrived);
  if (X->stage) (void)hipFree(X->stage);
  if (X->stage_bits) (void)hipFree(X->stage_bits);
  if (X->part_stage) (void)hipFree(X->part_stage);
  if (X->d_flag) (void)hipFree(X->d_flag);
  delete X;
  g->native_xchg = nullptr;
}

}  // namespace gm

extern "C" {

int gm_dist_unique_id(void* out, size_t bytes) {
  if (!out || bytes < sizeof(ncclUniqueId)) { gm::set_error("gm_dist_unique_id: need a buffer of %zu bytes", sizeof(ncclUniqueId)); return GM_ERR_INVALID; }
  int rc;
  if ((rc = gm::bind_rccl())) return rc;
  ncclUniqueId id;
  GM_TRY_NCCL(gm::g_rccl.GetUniqueId(&id));
  memset(out, 0, bytes);
  memcpy(out, &id, sizeof(id));
  return GM_OK;
}

int gm_dist_init(int rank, int nranks, const void* unique_id, size_t bytes) {
  if (nranks < 1 || rank < 0 || rank >= nranks || !unique_id || bytes < sizeof(ncclUniqueId)) { gm::set_error("gm_dist_init: invalid argument (rank %d of %d)", rank, nranks); return GM_ERR_INVALID; }
  if (gm::g_comm) { gm::set_error("gm_dist_init: already initialised (rank %d of %d)", gm::g_rank, gm::g_nranks); return GM_ERR_INVALID; }
  int rc;
  if ((rc = gm::bind_rccl())) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  GM_TRY_NCCL(gm::g_rccl.CommInitRank(&gm::g_comm, nranks, id, rank));
  gm::g_rank = rank;
  gm::g_nranks = nranks;
  return GM_OK;
}

int gm_dist_finalize(void) {
  if (gm::g_comm) {
    (void)hipDeviceSynchronize();
    (void)gm::g_rccl.CommDestroy(gm::g_comm);
    gm::g_comm = nullptr;
  }
  gm::g_rank = 0;
  gm::g_nranks = 1;
  return GM_OK;
}

int gm_dist_info(int* rank, int* nranks) {
  if (rank) *rank = gm::g_rank;
  if (nranks) *nranks = gm::g_comm ? gm::g_nranks : 0;
  return GM_OK;
}

int gm_graph_use_rccl(gm_graph_t* g) {
  if (!g) { gm::set_error("gm_graph_use_rccl: null graph"); return GM_ERR_INVALID; }
  if (!gm::g_comm) { gm::set_error("gm_graph_use_rccl: call gm_dist_init first"); return GM_ERR_INVALID; }
  const gm_graph_desc_t& d = g->desc;
  const int64_t S = d.row_hi - d.row_lo;
  if (d.layout != GM_LAYOUT_DEGREE || d.nshards != gm::g_nranks || d.shard != gm::g_rank || d.row_lo != (int64_t)d.shard * S ||
      (int64_t)d.ndevice != S * d.nshards || (d.nshards > 1 && (S & 63) != 0)) {
    gm::set_error("gm_graph_use_rccl: the graph must be shard %d of %d of a GM_LAYOUT_DEGREE graph (it is shard %d of %d, rows [%d,%d))",
                  gm::g_rank, gm::g_nranks, d.shard, d.nshards, d.row_lo, d.row_hi);
    return GM_ERR_INVALID;
  }
  gm::free_native_exchange(g);
  gm::RcclExchange* X = new gm::RcclExchange();
  X->g = g;
  g->native_xchg = X;
  GM_TRY_HIP(hipStreamCreateWithFlags(&X->xs, hipStreamNonBlocking));
  GM_TRY_HIP(hipEventCreateWithFlags(&X->ready, hipEventDisableTiming));
  GM_TRY_HIP(hipEventCreateWithFlags(&X->arrived, hipEventDisableTiming));
  GM_TRY_HIP(hipMalloc((void**)&X->d_flag, 64));
  g->xfn = gm::native_exchange;
  g->xctx = X;
  g->xcaps = GM_XCAP_SPARSE;
  return GM_OK;
}

int gm_graph_exchange_is_native(const gm_graph_t* g) { return (g && g->native_xchg && g->xfn == gm::native_exchange) ? 1 : 0; }

int gm_graph_exchange_counters(const gm_graph_t* g, int64_t out[4]) {
  if (!g || !out) return GM_ERR_INVALID;
  const gm::RcclExchange* X = (const gm::RcclExchange*)g->native_xchg;
  out[0] = X ? X->calls : 0;
  out[1] = X ? X->parts : 0;
  out[2] = X ? (int64_t)X->bytes_sent : 0;
  out[3] = X ? X->sparse_gathers : 0;
  return GM_OK;
}

int gm_graph_set_run_stream(gm_graph_t* g, gm_stream_t stream) {
  if (!g) return GM_ERR_INVALID;
  g->run_stream = (hipStream_t)stream;
  return GM_OK;
}

}  // extern "C"
