// shm_transport.hip -- TEST INFRASTRUCTURE, not part of the product.
//
// RCCL wants one rank per GPU, so the library's native multi-rank exchange (graphmat_amd/csrc/gm_dist.hip) cannot be
// tried with it on a 1-GPU box.  This shared object exports the RCCL entry points gm_dist binds (same names, same
// signatures) but moves the bytes through a POSIX shared-memory segment: device -> host -> barrier -> device,
// blocking.  Slow, and loaded only when a test sets GRAPHMAT_RCCL_LIBRARY to this file -- the role gloo plays for the
// torch.distributed callback path.  Limits (fine for tests): all ranks of the communicator must call every collective
// AND every ncclGroupEnd together (the barrier is global); a point-to-point payload must fit a rank's mailbox
// (GRAPHMAT_SHM_MB / nranks).
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {

struct Header {
  std::atomic<int> count;
  std::atomic<int> generation;
  char pad[56];
};
struct Mail {  // one pending point-to-point message in a rank's mailbox
  int64_t bytes;
  int32_t dest, pad;
};
struct State {
  Header* hdr = nullptr;
  char* data = nullptr;
  size_t data_bytes = 0;
  int rank = 0, nranks = 1;
  std::string name;
};
State g;
struct P2P { bool send; void* ptr; size_t bytes; int peer; hipStream_t s; };
std::vector<P2P> g_group;
int g_group_depth = 0;

void barrier() {
  Header* h = g.hdr;
  const int gen = h->generation.load(std::memory_order_acquire);
  if (h->count.fetch_add(1, std::memory_order_acq_rel) == g.nranks - 1) {
    h->count.store(0, std::memory_order_relaxed);
    h->generation.store(gen + 1, std::memory_order_release);
  } else {
    while (h->generation.load(std::memory_order_acquire) == gen) usleep(20);
  }
}
size_t type_bytes(ncclDataType_t t) { return (t == ncclChar || t == ncclUint8) ? 1 : (t == ncclInt64 || t == ncclUint64 || t == ncclFloat64) ? 8 : 4; }

ncclResult_t all_reduce_sum32(const void* send, void* recv, size_t count, hipStream_t s) {
  const int n = g.nranks, r = g.rank;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  const size_t chunk = ((g.data_bytes / (size_t)(n + 1)) & ~(size_t)63) / 4;  // words per rank slot (+1 result slot)
  uint32_t* slots = (uint32_t*)g.data;
  for (size_t off = 0; off < count; off += chunk) {
    const size_t len = count - off < chunk ? count - off : chunk;
    if (hipMemcpy(slots + (size_t)r * chunk, (const uint32_t*)send + off, len * 4, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier();
    if (r == 0) {
      uint32_t* out = slots + (size_t)n * chunk;
      for (size_t i = 0; i < len; i++) {
        uint32_t a = slots[i];
        for (int q = 1; q < n; q++) a += slots[(size_t)q * chunk + i];
        out[i] = a;
      }
    }
    barrier();
    if (hipMemcpy((uint32_t*)recv + off, slots + (size_t)n * chunk, len * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier();
  }
  return ncclSuccess;
}

// every rank calls this together: deliver the queued sends, complete the queued receives
ncclResult_t run_group() {
  const int n = g.nranks, r = g.rank;
  const size_t box = (g.data_bytes / (size_t)n) & ~(size_t)63;
  for (const P2P& op : g_group)
    if (hipStreamSynchronize(op.s) != hipSuccess) return ncclUnhandledCudaError;
  char* mine = g.data + (size_t)r * box;
  size_t off = 0;
  int nsend = 0;
  for (const P2P& op : g_group) {
    if (!op.send) continue;
    if (off + sizeof(Mail) + op.bytes + sizeof(Mail) > box) { fprintf(stderr, "shm_transport: point-to-point payload exceeds the mailbox (raise GRAPHMAT_SHM_MB)\n"); return ncclInvalidArgument; }
    Mail m = {(int64_t)op.bytes, op.peer, 0};
    memcpy(mine + off, &m, sizeof(m));
    if (op.bytes && hipMemcpy(mine + off + sizeof(m), op.ptr, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    off += sizeof(m) + ((op.bytes + 63) & ~(size_t)63);
    nsend++;
  }
  Mail end = {-1, -1, 0};
  memcpy(mine + off, &end, sizeof(end));
  barrier();
  for (const P2P& op : g_group) {
    if (op.send) continue;
    // the first not-yet-consumed message from op.peer addressed to this rank (messages between a pair arrive in order)
    char* box_p = g.data + (size_t)op.peer * box;
    size_t o = 0;
    bool found = false;
    for (;;) {
      Mail m;
      memcpy(&m, box_p + o, sizeof(m));
      if (m.bytes < 0) break;
      if (m.dest == r && m.pad == 0) {
        if ((size_t)m.bytes != op.bytes) { fprintf(stderr, "shm_transport: receive of %zu bytes meets a send of %lld\n", op.bytes, (long long)m.bytes); return ncclInvalidArgument; }
        if (op.bytes && hipMemcpy(op.ptr, box_p + o + sizeof(m), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        m.pad = 1;  // consumed (only this rank touches messages addressed to it)
        memcpy(box_p + o, &m, sizeof(m));
        found = true;
        break;
      }
      o += sizeof(m) + (((size_t)m.bytes + 63) & ~(size_t)63);
    }
    if (!found) { fprintf(stderr, "shm_transport: rank %d: no message from rank %d in this group\n", r, op.peer); return ncclInvalidArgument; }
  }
  barrier();
  g_group.clear();
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/graphmat_shm_%d_%ld", (int)getpid(), (long)time(nullptr));
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  const char* mb = getenv("GRAPHMAT_SHM_MB");
  const size_t bytes = sizeof(Header) + (size_t)(mb ? atoi(mb) : 64) * 1024 * 1024;
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) return ncclSystemError;
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  g.hdr = (Header*)p;  // a fresh segment is zero-filled: counters start at 0
  g.data = (char*)p + sizeof(Header);
  g.data_bytes = bytes - sizeof(Header);
  g.rank = rank;
  g.nranks = nranks;
  g.name = id.internal;
  *comm = (ncclComm_t)&g;
  barrier();
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t) {
  if (g.hdr) {
    barrier();
    munmap((void*)g.hdr, g.data_bytes + sizeof(Header));
    if (g.rank == 0) shm_unlink(g.name.c_str());
    g.hdr = nullptr;
  }
  return ncclSuccess;
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t, hipStream_t s) {
  const size_t bytes = count * type_bytes(t);
  const int n = g.nranks, r = g.rank;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  const size_t chunk = (g.data_bytes / (size_t)n) & ~(size_t)63;
  for (size_t off = 0; off < bytes; off += chunk) {
    const size_t len = bytes - off < chunk ? bytes - off : chunk;
    if (hipMemcpy(g.data + (size_t)r * chunk, (const char*)send + off, len, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier();
    for (int q = 0; q < n; q++)
      if (hipMemcpy((char*)recv + (size_t)q * bytes + off, g.data + (size_t)q * chunk, len, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier();
  }
  return ncclSuccess;
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t, hipStream_t s) {
  if ((t == ncclUint32 || t == ncclInt32) && op == ncclSum && (count != 1 || t == ncclUint32)) return all_reduce_sum32(send, recv, count, s);
  if (t != ncclInt32 || count != 1 || (op != ncclMin && op != ncclMax && op != ncclSum)) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  int v = 0;
  if (hipMemcpy(&v, send, 4, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  ((int*)g.data)[g.rank] = v;
  barrier();
  int acc = ((int*)g.data)[0];
  for (int q = 1; q < g.nranks; q++) {
    const int o = ((int*)g.data)[q];
    acc = op == ncclMin ? (o < acc ? o : acc) : op == ncclMax ? (o > acc ? o : acc) : acc + o;
  }
  barrier();
  if (hipMemcpy(recv, &acc, 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t, hipStream_t s) {
  const size_t bytes = count * type_bytes(t);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  const size_t chunk = g.data_bytes & ~(size_t)63;
  for (size_t off = 0; off < bytes; off += chunk) {
    const size_t len = bytes - off < chunk ? bytes - off : chunk;
    if (g.rank == root && hipMemcpy(g.data, (const char*)send + off, len, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier();
    if ((g.rank != root || recv != send) && hipMemcpy((char*)recv + off, g.data, len, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier();
  }
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (--g_group_depth > 0) return ncclSuccess;
  g_group_depth = 0;
  return run_group();
}
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t t, int peer, ncclComm_t, hipStream_t s) {
  g_group.push_back({true, const_cast<void*>(send), count * type_bytes(t), peer, s});
  return g_group_depth > 0 ? ncclSuccess : run_group();
}
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t t, int peer, ncclComm_t, hipStream_t s) {
  g_group.push_back({false, recv, count * type_bytes(t), peer, s});
  return g_group_depth > 0 ? ncclSuccess : run_group();
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "shared-memory test transport error"; }

}  // extern "C"
