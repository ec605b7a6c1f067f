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
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <string>

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
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
};
static RcclApi g_rccl;
static ncclComm_t g_comm = nullptr;
static int g_rank = 0, g_nranks = 1;

// librccl is bound with dlopen when gm_dist_init is first called.  GRAPHMAT_RCCL_LIBRARY names another library with
// the same entry points (the test suite's host shared-memory stand-in, tests/support/, so that several ranks can
// run on a 1-GPU box; nothing of it is compiled into this library).
static int bind_rccl() {
  if (g_rccl.handle) return GM_OK;
  const char* name = getenv("GRAPHMAT_RCCL_LIBRARY");
  void* h = nullptr;
  if (name && *name) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_error("gm_dist: cannot load GRAPHMAT_RCCL_LIBRARY=%s: %s", name, dlerror()); return GM_ERR_UNSUPPORTED; }
  } else {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { set_error("gm_dist: cannot load librccl.so.1: %s", dlerror()); return GM_ERR_UNSUPPORTED; }
  }
#define GM_BIND(field, sym)                                                                  \
  *(void**)(&g_rccl.field) = dlsym(h, sym);                                                  \
  if (!g_rccl.field) { set_error("gm_dist: librccl has no symbol %s", sym); dlclose(h); return GM_ERR_UNSUPPORTED; }
  GM_BIND(GetUniqueId, "ncclGetUniqueId")
  GM_BIND(CommInitRank, "ncclCommInitRank")
  GM_BIND(CommDestroy, "ncclCommDestroy")
  GM_BIND(AllGather, "ncclAllGather")
  GM_BIND(AllReduce, "ncclAllReduce")
  GM_BIND(GetErrorString, "ncclGetErrorString")
  GM_BIND(Send, "ncclSend")
  GM_BIND(Recv, "ncclRecv")
  GM_BIND(GroupStart, "ncclGroupStart")
  GM_BIND(GroupEnd, "ncclGroupEnd")
  GM_BIND(Broadcast, "ncclBroadcast")
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

// ---- collectives of a distributed graph build (gm_graph.hip, gm_graph_desc_t.edges_local) ----------------
int dist_world(int* rank, int* nranks) {
  if (rank) *rank = g_comm ? g_rank : 0;
  if (nranks) *nranks = g_comm ? g_nranks : 0;
  return g_comm ? 1 : 0;
}
int dist_all_reduce_sum_u32(uint32_t* d, size_t n, hipStream_t s) {
  if (!g_comm) { set_error("distributed build: call gm_dist_init first"); return GM_ERR_INVALID; }
  if (n == 0 || g_nranks == 1) return GM_OK;
  GM_TRY_NCCL(g_rccl.AllReduce(d, d, n, ncclUint32, ncclSum, g_comm, s));
  return GM_OK;
}
int dist_ring_step(const void* d_send, size_t send_bytes, void* d_recv, size_t recv_bytes, hipStream_t s) {
  if (!g_comm) { set_error("gm_dist: call gm_dist_init first"); return GM_ERR_INVALID; }
  if (g_nranks == 1) return GM_OK;
  GM_TRY_NCCL(g_rccl.GroupStart());
  if (send_bytes > 0 && g_rank + 1 < g_nranks) GM_TRY_NCCL(g_rccl.Send(d_send, send_bytes, ncclChar, g_rank + 1, g_comm, s));
  if (recv_bytes > 0 && g_rank > 0) GM_TRY_NCCL(g_rccl.Recv(d_recv, recv_bytes, ncclChar, g_rank - 1, g_comm, s));
  GM_TRY_NCCL(g_rccl.GroupEnd());
  return GM_OK;
}
int dist_broadcast(void* d_buf, size_t bytes, int root, hipStream_t s) {
  if (!g_comm) { set_error("gm_dist: call gm_dist_init first"); return GM_ERR_INVALID; }
  if (g_nranks == 1 || bytes == 0) return GM_OK;
  GM_TRY_NCCL(g_rccl.Broadcast(d_buf, d_buf, bytes, ncclChar, root, g_comm, s));
  return GM_OK;
}
// recv = the ranks' `bytes` bytes at d_send, in rank order (separate buffers)
int dist_all_gather_bytes(const void* d_send, void* d_recv, size_t bytes, hipStream_t s) {
  if (!g_comm) { set_error("distributed build: call gm_dist_init first"); return GM_ERR_INVALID; }
  if (bytes == 0) return GM_OK;
  if (g_nranks == 1) { GM_TRY_HIP(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, s)); return GM_OK; }
  GM_TRY_NCCL(g_rccl.AllGather(d_send, d_recv, bytes, ncclChar, g_comm, s));
  return GM_OK;
}

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
#ifdef GM_TEST_HOOKS
  // Fault injection of the TEST-HOOKS build only (graphmat_amd/build.py: build_hooks -> build/hooks/libgraphmat_hip.so; the product
  // library has no such switch): GRAPHMAT_DEBUG_DROP_WAIT=1 leaves this dependency out -- the run stream then copies the parts
  // out of the staging buffer without waiting for the side stream's all-gathers, the bug a blocking transport can never show
  // and tests/test_gpu_multi.py::test_missing_stream_wait_is_caught proves the stream-ordered test transport does
  static const bool drop_wait = getenv("GRAPHMAT_DEBUG_DROP_WAIT") != nullptr && getenv("GRAPHMAT_DEBUG_DROP_WAIT")[0] == '1';
  if (!drop_wait) GM_TRY_HIP(hipStreamWaitEvent(s, X->arrived, 0));
#else
  GM_TRY_HIP(hipStreamWaitEvent(s, X->arrived, 0));
#endif
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

// rank / size / local device from the launcher's environment, the unique id through a rendezvous file.
//
// A process joins a communicator only on an UNAMBIGUOUS multi-rank launch: GRAPHMAT_NRANKS (+ GRAPHMAT_RANK), or a
// launcher's own pair of variables both present (torchrun RANK + WORLD_SIZE, Open MPI OMPI_COMM_WORLD_*, PMI_*, and
// for Slurm the per-STEP variables SLURM_STEP_NUM_TASKS + SLURM_PROCID that only `srun` sets -- SLURM_NTASKS describes
// the allocation, so a single process started inside an sbatch script is one rank, not eight).  A lone WORLD_SIZE in
// a shell is ignored.  GRAPHMAT_NRANKS=1 forces a single-process run whatever else is set.
int gm_dist_init_from_env(int* rank_out, int* nranks_out) {
  auto env = [](const char* n) -> const char* { const char* v = getenv(n); return (v && *v) ? v : nullptr; };
  struct Family { const char *size, *rank, *local; };
  static const Family fam[] = {{"GRAPHMAT_NRANKS", "GRAPHMAT_RANK", "GRAPHMAT_LOCAL_RANK"},
                               {"WORLD_SIZE", "RANK", "LOCAL_RANK"},
                               {"OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_LOCAL_RANK"},
                               {"PMI_SIZE", "PMI_RANK", "MPI_LOCALRANKID"},
                               {"SLURM_STEP_NUM_TASKS", "SLURM_PROCID", "SLURM_LOCALID"}};
  int nranks = 1, rank = 0, local = -1;
  for (const Family& f : fam) {
    const char *sz = env(f.size), *rk = env(f.rank);
    if (!sz) continue;
    if (f.size == fam[0].size && atoi(sz) <= 1) break;  // GRAPHMAT_NRANKS=1: explicit single process
    if (!rk) continue;                                   // a size without its rank is not a launch
    nranks = atoi(sz);
    rank = atoi(rk);
    if (const char* l = env(f.local)) local = atoi(l);
    else if (const char* l2 = env("GRAPHMAT_LOCAL_RANK")) local = atoi(l2);
    break;
  }
  if (rank_out) *rank_out = nranks > 1 ? rank : 0;
  if (nranks_out) *nranks_out = nranks > 1 ? nranks : 1;
  if (nranks <= 1) return GM_OK;  // single process: nothing to set up
  if (gm::g_comm) return GM_OK;
  if (rank < 0 || rank >= nranks) { gm::set_error("gm_dist_init_from_env: rank %d of %d", rank, nranks); return GM_ERR_INVALID; }
  int ndev = 0;
  GM_TRY_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1) { gm::set_error("gm_dist_init_from_env: no GPU"); return GM_ERR_HIP; }
  GM_TRY_HIP(hipSetDevice((local >= 0 ? local : rank) % ndev));
  // Rendezvous file: GRAPHMAT_RENDEZVOUS (must be on a file system every rank sees; the default below is node-local,
  // which is this library's scope: the GPUs of one node), else /tmp/graphmat_rdv_<uid>_<job>, where <job> names THIS
  // launch as well as the launcher lets us: Slurm job + step, the PMIx / Open MPI job id, torchrun's run id + port, else
  // the parent pid (ranks of one mpirun / torchrun agent share their parent).  Rank 0 removes whatever an earlier,
  // crashed launch may have left under that name BEFORE publishing, creates the file exclusively (no symlink is
  // followed: O_EXCL | O_NOFOLLOW, mode 0600) and writes a header with its start time; the other ranks accept only a
  // file that is complete, carries the magic, and was written no earlier than 120 s before they started -- a stale file
  // of an older launch is ignored and waited out.  A wrong id can still only lead to ncclCommInitRank not completing,
  // and that is bounded too (GRAPHMAT_INIT_TIMEOUT seconds, default 180): an error instead of a hang.
  std::string path;
  if (const char* e = env("GRAPHMAT_RENDEZVOUS")) path = e;
  else {
    std::string job;
    if (env("SLURM_JOB_ID")) job = std::string("slurm") + env("SLURM_JOB_ID") + "_" + (env("SLURM_STEP_ID") ? env("SLURM_STEP_ID") : "0");
    else if (env("PMIX_NAMESPACE")) job = std::string("pmix") + env("PMIX_NAMESPACE");
    else if (env("OMPI_MCA_ess_base_jobid")) job = std::string("ompi") + env("OMPI_MCA_ess_base_jobid");
    else if (env("MASTER_PORT")) job = std::string("port") + env("MASTER_PORT") + "_" + (env("TORCHELASTIC_RUN_ID") ? env("TORCHELASTIC_RUN_ID") : "none");
    else job = std::string("ppid") + std::to_string((long)getppid());
    for (char& c : job) if (!((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_' || c == '-')) c = '_';
    path = std::string("/tmp/graphmat_rdv_") + std::to_string((long)getuid()) + "_" + job;
  }
  struct Header { char magic[8]; int64_t written_at; int32_t nranks; int32_t pad; };
  const time_t started = time(nullptr);
  char id[GM_DIST_ID_BYTES];
  if (rank == 0) {
    int rc = gm_dist_unique_id(id, sizeof(id));
    if (rc) return rc;
    (void)unlink(path.c_str());  // an earlier launch's leftover
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    Header h;
    memcpy(h.magic, "GMRDV01", 8);
    h.written_at = (int64_t)started;
    h.nranks = nranks;
    h.pad = 0;
    bool ok = fd >= 0 && write(fd, &h, sizeof(h)) == (ssize_t)sizeof(h) && write(fd, id, sizeof(id)) == (ssize_t)sizeof(id);
    if (fd >= 0) ok = (close(fd) == 0) && ok;
    if (!ok) { (void)unlink(tmp.c_str()); gm::set_error("gm_dist_init_from_env: cannot write %s", tmp.c_str()); return GM_ERR_IO; }
    if (rename(tmp.c_str(), path.c_str()) != 0) { (void)unlink(tmp.c_str()); gm::set_error("gm_dist_init_from_env: cannot publish %s", path.c_str()); return GM_ERR_IO; }
  } else {
    // A file written within 2 s of this rank's own start is this launch's for sure.  An OLDER one (but inside the 120 s
    // window) may be rank 0 having started early -- or the leftover of a launch that crashed a moment ago under the same
    // key, which this launch's rank 0 is about to unlink and replace: such a candidate is only accepted after the file
    // has stayed the same (inode and time stamp) for 5 more seconds; a replacement that shows up meanwhile is judged afresh.
    bool got = false;
    bool have_cand = false;
    int64_t cand_written = 0, cand_since_ms = 0;
    unsigned long long cand_ino = 0;
    char cand_id[GM_DIST_ID_BYTES];
    for (int64_t waited_ms = 0; waited_ms < 65000 && !got; waited_ms++) {  // up to ~65 s
      const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW);
      if (fd >= 0) {
        Header h;
        struct stat sb;
        char buf[GM_DIST_ID_BYTES];
        const bool valid = fstat(fd, &sb) == 0 && read(fd, &h, sizeof(h)) == (ssize_t)sizeof(h) && memcmp(h.magic, "GMRDV01", 8) == 0 && h.nranks == nranks &&
                           h.written_at >= (int64_t)started - 120 && read(fd, buf, sizeof(buf)) == (ssize_t)sizeof(buf);
        (void)close(fd);
        if (valid) {
          if (h.written_at >= (int64_t)started - 2) {
            memcpy(id, buf, sizeof(id));
            got = true;
          } else if (!have_cand || cand_ino != (unsigned long long)sb.st_ino || cand_written != h.written_at) {
            have_cand = true;
            cand_ino = (unsigned long long)sb.st_ino;
            cand_written = h.written_at;
            cand_since_ms = waited_ms;
            memcpy(cand_id, buf, sizeof(cand_id));
          }
        }
      }
      if (!got && have_cand && waited_ms - cand_since_ms >= 5000) {
        memcpy(id, cand_id, sizeof(id));
        got = true;
      }
      if (!got) usleep(1000);
    }
    if (!got) { gm::set_error("gm_dist_init_from_env: rank 0 never published a fresh %s (GRAPHMAT_RENDEZVOUS names the file)", path.c_str()); return GM_ERR_IO; }
  }
  // ncclCommInitRank blocks until every rank has arrived: bound it
  int timeout_s = 180;
  if (const char* t = env("GRAPHMAT_INIT_TIMEOUT")) timeout_s = atoi(t) > 0 ? atoi(t) : timeout_s;
  struct InitJob { std::atomic<int> done{0}; int rc = GM_OK; int rank, nranks, dev; char id[GM_DIST_ID_BYTES]; std::string err; };
  InitJob* job = new InitJob();  // (leaked on a timeout: the worker may still be inside RCCL)
  job->rank = rank;
  job->nranks = nranks;
  GM_TRY_HIP(hipGetDevice(&job->dev));
  memcpy(job->id, id, sizeof(id));
  pthread_t th;
  auto worker = [](void* p) -> void* {
    InitJob* j = (InitJob*)p;
    (void)hipSetDevice(j->dev);
    j->rc = gm_dist_init(j->rank, j->nranks, j->id, sizeof(j->id));
    if (j->rc != GM_OK) j->err = gm_last_error();
    j->done.store(1, std::memory_order_release);
    return nullptr;
  };
  if (pthread_create(&th, nullptr, worker, job) != 0) { delete job; gm::set_error("gm_dist_init_from_env: cannot start a thread"); return GM_ERR_INVALID; }
  for (int64_t waited_ms = 0; !job->done.load(std::memory_order_acquire); waited_ms += 5) {
    if (waited_ms > (int64_t)timeout_s * 1000) {
      pthread_detach(th);
      gm::set_error("gm_dist_init_from_env: rank %d of %d: the communicator was not set up within %d s -- are all %d ranks running, and is %s "
                    "this launch's rendezvous file?  (GRAPHMAT_NRANKS=1 runs a single process; GRAPHMAT_INIT_TIMEOUT changes the limit)",
                    rank, nranks, timeout_s, nranks, path.c_str());
      return GM_ERR_UNSUPPORTED;
    }
    usleep(5000);
  }
  pthread_join(th, nullptr);
  int rc = job->rc;
  if (rc != GM_OK) gm::set_error("%s", job->err.c_str());
  delete job;
  if (rc) return rc;
  rc = gm_dist_barrier();  // everybody has read the file
  if (rank == 0) (void)unlink(path.c_str());
  return rc;
}

int gm_dist_barrier(void) {
  if (!gm::g_comm || gm::g_nranks <= 1) return GM_OK;
  static int* d = nullptr;
  if (!d) { GM_TRY_HIP(hipMalloc((void**)&d, 64)); GM_TRY_HIP(hipMemset(d, 0, 64)); }
  GM_TRY_NCCL(gm::g_rccl.AllReduce(d, d, 1, ncclInt32, ncclMax, gm::g_comm, (hipStream_t)0));
  GM_TRY_HIP(hipStreamSynchronize((hipStream_t)0));
  return GM_OK;
}

// all-gather of host buffers of different sizes: *all = malloc'ed concatenation in rank order (gm_host_free),
// counts[r] = bytes of rank r.  (Edge lists read per rank, per-rank partial results of a map-reduce.)
int gm_dist_allgatherv_host(const void* mine, int64_t my_bytes, void** all, int64_t* counts) {
  if (!all || !counts || my_bytes < 0 || (my_bytes && !mine)) { gm::set_error("gm_dist_allgatherv_host: invalid argument"); return GM_ERR_INVALID; }
  const int n = gm::g_comm ? gm::g_nranks : 1, r = gm::g_comm ? gm::g_rank : 0;
  if (n == 1) {
    counts[0] = my_bytes;
    *all = malloc((size_t)(my_bytes > 0 ? my_bytes : 1));
    if (my_bytes) memcpy(*all, mine, (size_t)my_bytes);
    return GM_OK;
  }
  // sizes first (as two int32 halves per rank), then the payload padded to the largest size
  int* d_sz = nullptr;
  GM_TRY_HIP(hipMalloc((void**)&d_sz, (size_t)n * 8));
  const int half[2] = {(int)(my_bytes & 0x7fffffff), (int)(my_bytes >> 31)};
  GM_TRY_HIP(hipMemcpy(d_sz + 2 * r, half, 8, hipMemcpyHostToDevice));
  GM_TRY_NCCL(gm::g_rccl.AllGather(d_sz + 2 * r, d_sz, 2, ncclInt32, gm::g_comm, (hipStream_t)0));
  std::vector<int> h((size_t)2 * n);
  GM_TRY_HIP(hipMemcpy(h.data(), d_sz, (size_t)n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d_sz);
  int64_t mx = 0, total = 0;
  for (int i = 0; i < n; i++) { counts[i] = (int64_t)h[2 * i] | ((int64_t)h[2 * i + 1] << 31); mx = counts[i] > mx ? counts[i] : mx; total += counts[i]; }
  const size_t pad = ((size_t)mx + 255) / 256 * 256;
  char* out = (char*)malloc((size_t)(total > 0 ? total : 1));
  if (!out) { gm::set_error("gm_dist_allgatherv_host: out of host memory"); return GM_ERR_NOMEM; }
  if (pad > 0) {
    char* d = nullptr;
    GM_TRY_HIP(hipMalloc((void**)&d, pad * (size_t)n));
    if (my_bytes) GM_TRY_HIP(hipMemcpy(d + (size_t)r * pad, mine, (size_t)my_bytes, hipMemcpyHostToDevice));
    GM_TRY_NCCL(gm::g_rccl.AllGather(d + (size_t)r * pad, d, pad, ncclChar, gm::g_comm, (hipStream_t)0));
    GM_TRY_HIP(hipStreamSynchronize((hipStream_t)0));
    size_t off = 0;
    for (int i = 0; i < n; i++) {
      if (counts[i]) GM_TRY_HIP(hipMemcpy(out + off, d + (size_t)i * pad, (size_t)counts[i], hipMemcpyDeviceToHost));
      off += (size_t)counts[i];
    }
    (void)hipFree(d);
  }
  *all = out;
  return GM_OK;
}

int gm_graph_set_run_stream(gm_graph_t* g, gm_stream_t stream) {
  if (!g) return GM_ERR_INVALID;
  g->run_stream = (hipStream_t)stream;
  return GM_OK;
}

}  // extern "C"
