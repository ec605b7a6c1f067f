// gm_core.hip -- error state, id permutation, .mtx reader, RMAT generator, reductions.
#include <stdlib.h>
#include <string.h>

#include "gm_internal.hpp"

namespace gm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int kT = 256;
inline int grid_for(int64_t n) { return (int)((n + kT - 1) / kT); }

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// thresholds floor(p*2^32) for a=.57, a+b=.76, a+b+c=.95 (graphmat_amd/generators.py)
constexpr uint32_t kTA = 2448131358u, kTAB = 3264175144u, kTABC = 4080218931u;

__global__ void __launch_bounds__(kT)
k_rmat(int scale, uint64_t key, int64_t first, int64_t count, int32_t* __restrict__ src, int32_t* __restrict__ dst,
       int32_t* __restrict__ val, int weights_mode) {
  int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= count) return;
  uint64_t e = (uint64_t)(first + i);
  uint32_t s = 0, d = 0;
  uint64_t h = 0;
  for (int lvl = 0; lvl < scale; lvl++) {
    uint32_t r;
    if ((lvl & 1) == 0) {
      h = splitmix64(key + e * 32ull + (uint64_t)(lvl >> 1));
      r = (uint32_t)h;
    } else {
      r = (uint32_t)(h >> 32);
    }
    uint32_t sb = r >= kTAB ? 1u : 0u;
    uint32_t db = ((r >= kTA && r < kTAB) || r >= kTABC) ? 1u : 0u;
    int sh = scale - 1 - lvl;
    s |= sb << sh;
    d |= db << sh;
  }
  src[i] = (int32_t)(s + 1);
  dst[i] = (int32_t)(d + 1);
  if (val) val[i] = weights_mode ? (int32_t)(1 + (splitmix64(key ^ e) % 127ull)) : 1;
}

// ---- reductions: fixed-shape two-level tree (deterministic for a given n) ----------------
template <class T, class Acc, class Map>
__global__ void __launch_bounds__(kT)
k_reduce_partial(const T* __restrict__ x, int64_t n, int64_t stride, Acc* __restrict__ partial, Map map) {
  __shared__ Acc sm[kT];
  Acc a = Acc(0);
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) a += map(x[i * stride]);
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int s = kT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

template <class T, class Acc, class Map>
int reduce_to_host(const T* d_x, int64_t n, int64_t stride, Acc* h_out, Map map, hipStream_t s) {
  const int nb = 1024;
  Acc* d_part = nullptr;
  GM_TRY_HIP(hipMalloc((void**)&d_part, nb * sizeof(Acc)));
  hipLaunchKernelGGL((k_reduce_partial<T, Acc, Map>), dim3(nb), dim3(kT), 0, s, d_x, n, stride, d_part, map);
  Acc h[nb];
  hipError_t e = hipMemcpyAsync(h, d_part, nb * sizeof(Acc), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d_part);
  if (e != hipSuccess) { set_error("reduce: %s", hipGetErrorString(e)); return GM_ERR_HIP; }
  Acc t = Acc(0);
  for (int i = 0; i < nb; i++) t += h[i];
  *h_out = t;
  return GM_OK;
}

struct MapF64 { __device__ double operator()(double v) const { return v; } };
struct MapF32 { __device__ double operator()(float v) const { return (double)v; } };
struct MapLess { uint32_t bound; __device__ long long operator()(uint32_t v) const { return v < bound ? 1ll : 0ll; } };
struct MapPop { __device__ long long operator()(uint32_t v) const { return (long long)__popc(v); } };

}  // namespace gm

extern "C" {

const char* gm_last_error(void) { return gm::g_err; }
int gm_version(void) { return 100; }

int gm_device_count(int* count) {
  if (!count) { gm::set_error("gm_device_count: null argument"); return GM_ERR_INVALID; }
  GM_TRY_HIP(hipGetDeviceCount(count));
  return GM_OK;
}
int gm_set_device(int device) {
  GM_TRY_HIP(hipSetDevice(device));
  return GM_OK;
}

int gm_vertex_to_native(int vertex, int nparts, int len) { return gm::to_native0(vertex, nparts, len) + 1; }
int gm_native_to_vertex(int native, int nparts, int len) { return gm::to_vertex1(native - 1, nparts, len); }

int gm_mtx_read(const char* path, int val_bytes, int* nv, int64_t* nnz, int32_t** h_src, int32_t** h_dst,
                void** h_val) {
  if (!path || !nv || !nnz || !h_src || !h_dst || val_bytes < 0) { gm::set_error("gm_mtx_read: invalid argument"); return GM_ERR_INVALID; }
  FILE* fp = fopen(path, "rb");
  if (!fp) { gm::set_error("Could not open file: %s", path); return GM_ERR_IO; }
  int32_t hdr[3];
  if (fread(hdr, 4, 3, fp) != 3 || hdr[0] <= 0 || hdr[1] <= 0 || hdr[2] < 0) {
    fclose(fp);
    gm::set_error("%s: bad binary .mtx header", path);
    return GM_ERR_IO;
  }
  const int64_t n = hdr[2];
  const size_t rec = 8 + (size_t)val_bytes;
  unsigned char* raw = (unsigned char*)malloc((size_t)n * rec + 1);
  int32_t* s = (int32_t*)malloc((size_t)n * 4 + 4);
  int32_t* d = (int32_t*)malloc((size_t)n * 4 + 4);
  void* v = (h_val && val_bytes) ? malloc((size_t)n * val_bytes + 4) : nullptr;
  if (!raw || !s || !d || (h_val && val_bytes && !v)) {
    fclose(fp); free(raw); free(s); free(d); free(v);
    gm::set_error("gm_mtx_read: out of host memory");
    return GM_ERR_NOMEM;
  }
  size_t got = fread(raw, rec, (size_t)n, fp);  // header count governs; trailing records ignored
  fclose(fp);
  if ((int64_t)got != n) {
    free(raw); free(s); free(d); free(v);
    gm::set_error("%s: header says %lld edges, file holds %zu", path, (long long)n, got);
    return GM_ERR_IO;
  }
  for (int64_t i = 0; i < n; i++) {
    memcpy(&s[i], raw + i * rec, 4);
    memcpy(&d[i], raw + i * rec + 4, 4);
    if (v) memcpy((unsigned char*)v + i * val_bytes, raw + i * rec + 8, val_bytes);
  }
  free(raw);
  *nv = hdr[0] > hdr[1] ? hdr[0] : hdr[1];  // Graph::ReadMTX squares the matrix (Graph.h:253-257)
  *nnz = n;
  *h_src = s;
  *h_dst = d;
  if (h_val) *h_val = v;
  return GM_OK;
}
void gm_host_free(void* p) { free(p); }

int gm_rmat_generate(int scale, uint64_t seed, int64_t first_edge, int64_t count, int32_t* d_src, int32_t* d_dst,
                     int32_t* d_val, int weights_mode, gm_stream_t stream) {
  if (scale < 1 || scale > 30 || count < 0 || first_edge < 0 || !d_src || !d_dst) {
    gm::set_error("gm_rmat_generate: invalid argument (scale=%d count=%lld)", scale, (long long)count);
    return GM_ERR_INVALID;
  }
  if (count == 0) return GM_OK;
  hipLaunchKernelGGL(gm::k_rmat, dim3(gm::grid_for(count)), dim3(gm::kT), 0, (hipStream_t)stream, scale,
                     gm::splitmix64(seed), first_edge, count, d_src, d_dst, d_val, weights_mode);
  GM_TRY_HIP(hipGetLastError());
  return GM_OK;
}

int gm_reduce_sum_f64(const double* d_x, int64_t n, int64_t stride, double* h_out, gm_stream_t stream) {
  if (!d_x || !h_out || n < 0) { gm::set_error("gm_reduce_sum_f64: invalid argument"); return GM_ERR_INVALID; }
  return gm::reduce_to_host<double, double>(d_x, n, stride, h_out, gm::MapF64(), (hipStream_t)stream);
}
int gm_reduce_sum_f32(const float* d_x, int64_t n, int64_t stride, double* h_out, gm_stream_t stream) {
  if (!d_x || !h_out || n < 0) { gm::set_error("gm_reduce_sum_f32: invalid argument"); return GM_ERR_INVALID; }
  return gm::reduce_to_host<float, double>(d_x, n, stride, h_out, gm::MapF32(), (hipStream_t)stream);
}
int gm_count_less_u32(const uint32_t* d_x, int64_t n, int64_t stride, uint32_t bound, int64_t* h_out,
                      gm_stream_t stream) {
  if (!d_x || !h_out || n < 0) { gm::set_error("gm_count_less_u32: invalid argument"); return GM_ERR_INVALID; }
  long long t = 0;
  gm::MapLess m{bound};
  int rc = gm::reduce_to_host<uint32_t, long long>(d_x, n, stride, &t, m, (hipStream_t)stream);
  *h_out = t;
  return rc;
}
int gm_popcount_bits(const uint32_t* d_bits, int64_t nbits, int64_t* h_out, gm_stream_t stream) {
  if (!d_bits || !h_out || nbits < 0) { gm::set_error("gm_popcount_bits: invalid argument"); return GM_ERR_INVALID; }
  long long t = 0;
  // callers keep bits past nbits clear (Graph::setAll sets exactly n bits)
  int rc = gm::reduce_to_host<uint32_t, long long>(d_bits, (nbits + 31) / 32, 1, &t, gm::MapPop(), (hipStream_t)stream);
  *h_out = t;
  return rc;
}

}  // extern "C"
