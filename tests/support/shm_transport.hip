// shm_transport.hip -- TEST INFRASTRUCTURE, not part of the product.
//
// RCCL wants one rank per GPU, so the library's native multi-rank exchange (graphmat_amd/csrc/gm_dist.hip) cannot be
// tried with it on a 1-GPU box.  This shared object exports the RCCL entry points gm_dist binds (same names, same
// signatures) and moves the bytes through a POSIX shared-memory segment that every rank maps and registers with HIP.
//
// STREAM-ORDERED, like the real thing: a collective only ENQUEUES work on the stream it is handed and returns --
// copies into / out of the segment (hipMemcpyAsync on that stream), reductions (small kernels reading the segment),
// and the rendezvous between the ranks, which is a one-thread kernel on that stream that adds itself to a counter in the
// segment (system-scope atomic) and spins until every rank has arrived.  Nothing here calls hipStreamSynchronize,
// hipDeviceSynchronize or a blocking hipMemcpy after the communicator is set up, so a kernel of the caller that reads a
// receive buffer without having waited for the collective's stream really does read it too early -- which is what
// tests/test_gpu_multi.py::test_missing_stream_wait_is_caught relies on.  As RCCL does, the collectives of one
// communicator are executed in the order they were issued, whatever streams they were issued on (each op first waits for
// an event recorded behind the previous one).
//
// All ranks share one GPU; kernels of different processes run side by side, so a spinning rendezvous kernel does not
// keep the other ranks' kernels from reaching theirs.  A rendezvous gives up after GRAPHMAT_SHM_TIMEOUT_S seconds
// (default 20) and raises the segment's error flag -- every later call then fails -- instead of hanging the GPU.
// Limits (fine for tests): all ranks must issue the same sequence of collectives and ncclGroupEnd calls; a
// point-to-point payload must fit a rank's mailbox (GRAPHMAT_SHM_MB / nranks).
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
  std::atomic<int> count;            // host-side barrier of communicator set-up / tear-down
  std::atomic<int> generation;
  char pad0[56];
  unsigned long long arrivals;       // device-side rendezvous: total arrivals so far (monotone)
  char pad1[56];
  int error;                         // raised by a rendezvous that timed out
  char pad2[60];
};
struct Mail {  // one pending point-to-point message in a rank's mailbox
  int64_t bytes;
  int32_t dest, consumed;
};
struct State {
  Header* hdr = nullptr;       // host mapping
  char* data = nullptr;
  Header* d_hdr = nullptr;     // the same bytes as the GPU sees them
  char* d_data = nullptr;
  size_t data_bytes = 0;
  int rank = 0, nranks = 1;
  unsigned long long rendezvous = 0;  // rendezvous issued so far (the same number on every rank)
  unsigned long long timeout_ticks = 0;
  hipEvent_t last_op = nullptr;        // recorded behind the most recently issued collective
  bool have_last = false;
  hipStream_t last_stream = nullptr;   // ... and the stream it was issued on
  std::string name;
};
State g;
struct P2P { bool send; void* ptr; size_t bytes; int peer; hipStream_t s; };
std::vector<P2P> g_group;
int g_group_depth = 0;

void host_barrier() {
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

// ---- device side ---------------------------------------------------------------------------------------------
// one thread: arrive, then wait until `target` arrivals have been counted
__global__ void k_rendezvous(Header* h, unsigned long long target, unsigned long long timeout_ticks) {
  __hip_atomic_fetch_add(&h->arrivals, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(&h->arrivals, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > timeout_ticks) {
      __hip_atomic_store(&h->error, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    if (__hip_atomic_load(&h->error, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) return;
  }
}
// out[i] = op over the ranks of slots[q * stride + i]   (32-bit integers; op 0 sum (wrapping), 1 min, 2 max)
__global__ void k_reduce_slots(const uint32_t* slots, size_t stride, int nranks, size_t n, uint32_t* out, int op, int is_signed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t a = slots[i];
    for (int q = 1; q < nranks; q++) {
      const uint32_t b = slots[(size_t)q * stride + i];
      if (op == 0) a += b;
      else if (is_signed) a = (uint32_t)(op == 1 ? ((int32_t)b < (int32_t)a ? (int32_t)b : (int32_t)a) : ((int32_t)b > (int32_t)a ? (int32_t)b : (int32_t)a));
      else a = op == 1 ? (b < a ? b : a) : (b > a ? b : a);
    }
    out[i] = a;
  }
}
__global__ void k_write_mail(Mail* m, int64_t bytes, int dest) {
  m->bytes = bytes;
  m->dest = dest;
  m->consumed = 0;
}
// one workgroup: find the first message in `box` addressed to `me` that has not been consumed, copy its payload
__global__ void __launch_bounds__(1024) k_take_mail(char* box, int me, char* dst, size_t bytes, Header* h) {
  __shared__ long long s_off;
  if (threadIdx.x == 0) {
    long long found = -1;
    size_t o = 0;
    for (;;) {
      Mail* m = (Mail*)(box + o);
      if (m->bytes < 0) break;
      if (m->dest == me && m->consumed == 0) {
        if ((size_t)m->bytes == bytes) { found = (long long)o; m->consumed = 1; }
        break;
      }
      o += sizeof(Mail) + (((size_t)m->bytes + 63) & ~(size_t)63);
    }
    if (found < 0) __hip_atomic_store(&h->error, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    s_off = found;
  }
  __syncthreads();
  if (s_off < 0) return;
  const char* src = box + s_off + sizeof(Mail);
  if ((bytes & 3) == 0 && (((uintptr_t)dst) & 3) == 0) {
    for (size_t i = threadIdx.x; i < bytes / 4; i += 1024) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
  } else {
    for (size_t i = threadIdx.x; i < bytes; i += 1024) dst[i] = src[i];
  }
}

// ---- host side: everything below only enqueues -------------------------------------------------------------------
#define TRY(e) do { if ((e) != hipSuccess) return ncclUnhandledCudaError; } while (0)

bool failed() { return g.hdr == nullptr || g.hdr->error != 0; }

// the collectives of a communicator run in issue order, on whatever streams they were issued
ncclResult_t begin_op(hipStream_t s) {
  if (failed()) return ncclSystemError;
  if (g.have_last) TRY(hipStreamWaitEvent(s, g.last_op, 0));
  return ncclSuccess;
}
ncclResult_t end_op(hipStream_t s) {
  TRY(hipEventRecord(g.last_op, s));
  g.have_last = true;
  g.last_stream = s;
  return ncclSuccess;
}
ncclResult_t rendezvous(hipStream_t s) {
  g.rendezvous++;
  hipLaunchKernelGGL(k_rendezvous, dim3(1), dim3(1), 0, s, g.d_hdr, g.rendezvous * (unsigned long long)g.nranks, g.timeout_ticks);
  TRY(hipGetLastError());
  return ncclSuccess;
}
#define TRYN(e) do { ncclResult_t r_ = (e); if (r_ != ncclSuccess) return r_; } while (0)

ncclResult_t all_reduce_32(const void* send, void* recv, size_t count, int op, int is_signed, hipStream_t s) {
  const int n = g.nranks, r = g.rank;
  const size_t chunk = ((g.data_bytes / (size_t)n) & ~(size_t)63) / 4;  // words per rank slot
  TRYN(begin_op(s));
  for (size_t off = 0; off < count; off += chunk) {
    const size_t len = count - off < chunk ? count - off : chunk;
    TRY(hipMemcpyAsync(g.d_data + (size_t)r * chunk * 4, (const uint32_t*)send + off, len * 4, hipMemcpyDefault, s));
    TRYN(rendezvous(s));
    const int grid = (int)((len + 255) / 256 < 1024 ? (len + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_reduce_slots, dim3(grid), dim3(256), 0, s, (const uint32_t*)g.d_data, chunk, n, len, (uint32_t*)recv + off, op, is_signed);
    TRYN(rendezvous(s));  // nobody refills its slot before everybody has read all of them
  }
  return end_op(s);
}

// every rank calls this together: deliver the queued sends, complete the queued receives
ncclResult_t run_group() {
  const int n = g.nranks, r = g.rank;
  const size_t box = (g.data_bytes / (size_t)n) & ~(size_t)63;
  // (a rank with nothing to send or receive in this group -- the ends of a ring pipeline -- still takes part in the
  // group's two rendezvous: the others count on its arrivals.  It has no stream of its own to say: the last one used.)
  hipStream_t s = g_group.empty() ? g.last_stream : g_group[0].s;
  for (const P2P& op : g_group)
    if (op.s != s) { fprintf(stderr, "shm_transport: the operations of one group must share a stream\n"); g_group.clear(); return ncclInvalidArgument; }
  TRYN(begin_op(s));
  char* mine = g.d_data + (size_t)r * box;
  size_t off = 0;
  for (const P2P& op : g_group) {
    if (!op.send) continue;
    if (off + sizeof(Mail) + op.bytes + sizeof(Mail) > box) { fprintf(stderr, "shm_transport: point-to-point payload exceeds the mailbox (raise GRAPHMAT_SHM_MB)\n"); g_group.clear(); return ncclInvalidArgument; }
    hipLaunchKernelGGL(k_write_mail, dim3(1), dim3(1), 0, s, (Mail*)(mine + off), (int64_t)op.bytes, op.peer);
    if (op.bytes) TRY(hipMemcpyAsync(mine + off + sizeof(Mail), op.ptr, op.bytes, hipMemcpyDefault, s));
    off += sizeof(Mail) + ((op.bytes + 63) & ~(size_t)63);
  }
  hipLaunchKernelGGL(k_write_mail, dim3(1), dim3(1), 0, s, (Mail*)(mine + off), (int64_t)-1, -1);
  TRYN(rendezvous(s));
  for (const P2P& op : g_group) {
    if (op.send) continue;
    // the first not-yet-consumed message from op.peer addressed to this rank (messages between a pair arrive in order)
    hipLaunchKernelGGL(k_take_mail, dim3(1), dim3(1024), 0, s, g.d_data + (size_t)op.peer * box, r, (char*)op.ptr, op.bytes, g.d_hdr);
  }
  TRYN(rendezvous(s));
  g_group.clear();
  return end_op(s);
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
  // pinned and mapped: the GPU reads and writes the segment directly, coherently with the other processes
  void* dp = nullptr;
  if (hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess || hipHostGetDevicePointer(&dp, p, 0) != hipSuccess) {
    fprintf(stderr, "shm_transport: hipHostRegister of the shared segment failed: %s\n", hipGetErrorString(hipGetLastError()));
    munmap(p, bytes);
    return ncclUnhandledCudaError;
  }
  g.hdr = (Header*)p;  // a fresh segment is zero-filled: counters start at 0
  g.data = (char*)p + sizeof(Header);
  g.d_hdr = (Header*)dp;
  g.d_data = (char*)dp + sizeof(Header);
  g.data_bytes = bytes - sizeof(Header);
  g.rank = rank;
  g.nranks = nranks;
  g.name = id.internal;
  g.rendezvous = 0;
  const char* to = getenv("GRAPHMAT_SHM_TIMEOUT_S");
  g.timeout_ticks = (unsigned long long)(to && atoi(to) > 0 ? atoi(to) : 20) * 100000000ull;  // wall_clock64 ticks at 100 MHz
  if (hipEventCreateWithFlags(&g.last_op, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  g.have_last = false;
  *comm = (ncclComm_t)&g;
  host_barrier();
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t) {
  if (g.hdr) {
    (void)hipDeviceSynchronize();  // (tear-down: everything this rank enqueued has run)
    host_barrier();
    if (g.last_op) (void)hipEventDestroy(g.last_op);
    g.last_op = nullptr;
    (void)hipHostUnregister((void*)g.hdr);
    munmap((void*)g.hdr, g.data_bytes + sizeof(Header));
    if (g.rank == 0) shm_unlink(g.name.c_str());
    g.hdr = nullptr;
  }
  return ncclSuccess;
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t, hipStream_t s) {
  const size_t bytes = count * type_bytes(t);
  const int n = g.nranks, r = g.rank;
  const size_t chunk = (g.data_bytes / (size_t)n) & ~(size_t)63;
  TRYN(begin_op(s));
  for (size_t off = 0; off < bytes; off += chunk) {
    const size_t len = bytes - off < chunk ? bytes - off : chunk;
    TRY(hipMemcpyAsync(g.d_data + (size_t)r * chunk, (const char*)send + off, len, hipMemcpyDefault, s));
    TRYN(rendezvous(s));
    for (int q = 0; q < n; q++)
      TRY(hipMemcpyAsync((char*)recv + (size_t)q * bytes + off, g.d_data + (size_t)q * chunk, len, hipMemcpyDefault, s));
    TRYN(rendezvous(s));
  }
  return end_op(s);
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t, hipStream_t s) {
  if ((t != ncclUint32 && t != ncclInt32) || (op != ncclMin && op != ncclMax && op != ncclSum) || count == 0) return ncclInvalidArgument;
  return all_reduce_32(send, recv, count, op == ncclSum ? 0 : op == ncclMin ? 1 : 2, t == ncclInt32 ? 1 : 0, s);
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t, hipStream_t s) {
  const size_t bytes = count * type_bytes(t);
  const size_t chunk = g.data_bytes & ~(size_t)63;
  TRYN(begin_op(s));
  for (size_t off = 0; off < bytes; off += chunk) {
    const size_t len = bytes - off < chunk ? bytes - off : chunk;
    if (g.rank == root) TRY(hipMemcpyAsync(g.d_data, (const char*)send + off, len, hipMemcpyDefault, s));
    TRYN(rendezvous(s));
    if (g.rank != root || recv != send) TRY(hipMemcpyAsync((char*)recv + off, g.d_data, len, hipMemcpyDefault, s));
    TRYN(rendezvous(s));
  }
  return end_op(s);
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
const char* ncclGetErrorString(ncclResult_t r) {
  if (r == ncclSuccess) return "ok";
  if (g.hdr && g.hdr->error == 1) return "shared-memory test transport: a rendezvous timed out (a rank did not issue the same collectives?)";
  if (g.hdr && g.hdr->error == 2) return "shared-memory test transport: a receive found no matching message in its group";
  return "shared-memory test transport error";
}

}  // extern "C"
