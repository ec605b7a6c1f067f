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
int gm_version(void) { return 107; }  // 100 + the round whose ABI this is ("ABI changes by round")

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

namespace {
int val_kind_bytes(int kind) {
  if (kind == GM_VAL_I32 || kind == GM_VAL_U32 || kind == GM_VAL_F32) return 4;
  if (kind == GM_VAL_F64) return 8;
  if (kind > 0x100 && kind <= 0x100 + 4096) return kind - 0x100;
  return -1;
}
void store_one(void* dst, int kind) {  // the value an unweighted edge gets: (T)1; opaque kinds: zero bytes (the C++ layer constructs T(1) over them)
  if (kind > 0x100) { memset(dst, 0, (size_t)(kind - 0x100)); return; }
  if (kind == GM_VAL_I32) { int32_t v = 1; memcpy(dst, &v, 4); }
  else if (kind == GM_VAL_U32) { uint32_t v = 1; memcpy(dst, &v, 4); }
  else if (kind == GM_VAL_F32) { float v = 1.f; memcpy(dst, &v, 4); }
  else if (kind == GM_VAL_F64) { double v = 1.0; memcpy(dst, &v, 8); }
}
struct Growing {  // three parallel malloc'ed arrays
  int32_t *s = nullptr, *d = nullptr;
  unsigned char* v = nullptr;
  int64_t n = 0, cap = 0;
  int vb;
  explicit Growing(int vb_) : vb(vb_) {}
  bool reserve(int64_t want) {
    if (want <= cap) return true;
    int64_t nc = cap ? cap * 2 : 1024;
    if (nc < want) nc = want;
    int32_t* ns = (int32_t*)realloc(s, (size_t)nc * 4 + 4);
    if (ns) s = ns;
    int32_t* nd = (int32_t*)realloc(d, (size_t)nc * 4 + 4);
    if (nd) d = nd;
    unsigned char* nv = (unsigned char*)realloc(v, (size_t)nc * vb + 8);
    if (nv) v = nv;
    if (!ns || !nd || !nv) return false;
    cap = nc;
    return true;
  }
  void release() { free(s); free(d); free(v); s = d = nullptr; v = nullptr; }
};
}  // namespace

int gm_edgelist_read(const char* path, int binary, int header, int weights, int val_kind, int* m, int* n, int64_t* nnz,
                     int32_t** h_src, int32_t** h_dst, void** h_val) {
  const int vb = val_kind_bytes(val_kind);
  if (!path || !m || !n || !nnz || !h_src || !h_dst || !h_val || vb <= 0) { gm::set_error("gm_edgelist_read: invalid argument"); return GM_ERR_INVALID; }
  if (!binary && weights && val_kind > 0x100) { gm::set_error("gm_edgelist_read: opaque values need a binary file"); return GM_ERR_INVALID; }
  FILE* fp = fopen(path, binary ? "rb" : "r");
  if (!fp) { gm::set_error("Could not open file: %s", path); return GM_ERR_IO; }
  fseek(fp, 0, SEEK_END);
  const long fsize = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  char* buf = (char*)malloc((size_t)(fsize > 0 ? fsize : 0) + 2);
  if (!buf) { fclose(fp); gm::set_error("gm_edgelist_read: out of host memory"); return GM_ERR_NOMEM; }
  const size_t len = fsize > 0 ? fread(buf, 1, (size_t)fsize, fp) : 0;
  fclose(fp);
  buf[len] = 0;
  Growing g(vb);
  int hm = 0, hn = 0;
  int64_t hcount = -1;
  int maxs = 0, maxd = 0;
  int rc = GM_OK;
  if (binary) {
    size_t pos = 0;
    if (header) {
      int32_t h[3];
      if (len < 12) { free(buf); gm::set_error("%s: bad binary header", path); return GM_ERR_IO; }
      memcpy(h, buf, 12);
      hm = h[0]; hn = h[1]; hcount = (uint32_t)h[2];
      pos = 12;
    }
    const size_t rec = 8 + (weights ? (size_t)vb : 0);
    int64_t avail = (int64_t)((len - pos) / rec);
    if (hcount >= 0 && avail < hcount) { free(buf); gm::set_error("%s: header says %lld edges, file holds %lld", path, (long long)hcount, (long long)avail); return GM_ERR_IO; }
    const int64_t cnt = hcount >= 0 ? hcount : avail;
    if (!g.reserve(cnt > 0 ? cnt : 1)) rc = GM_ERR_NOMEM;
    for (int64_t i = 0; rc == GM_OK && i < cnt; i++) {
      const char* r = buf + pos + (size_t)i * rec;
      memcpy(&g.s[i], r, 4);
      memcpy(&g.d[i], r + 4, 4);
      if (weights) memcpy(g.v + (size_t)i * vb, r + 8, vb);
      else store_one(g.v + (size_t)i * vb, val_kind);
      if (g.s[i] > maxs) maxs = g.s[i];
      if (g.d[i] > maxd) maxd = g.d[i];
    }
    g.n = cnt;
  } else {
    char* p = buf;
    auto next_long = [&](long long* out) {
      char* e = nullptr;
      long long v = strtoll(p, &e, 10);
      if (e == p) return false;
      p = e;
      *out = v;
      return true;
    };
    if (header) {
      long long a, b, c;
      if (!next_long(&a) || !next_long(&b) || !next_long(&c)) { free(buf); gm::set_error("%s: bad text header", path); return GM_ERR_IO; }
      hm = (int)a; hn = (int)b; hcount = c;
    }
    while (hcount < 0 || g.n < hcount) {
      long long a, b;
      if (!next_long(&a) || !next_long(&b)) break;
      if (!g.reserve(g.n + 1)) { rc = GM_ERR_NOMEM; break; }
      unsigned char* vd = g.v + (size_t)g.n * vb;
      if (weights) {
        char* e = nullptr;
        if (val_kind == GM_VAL_I32) { int32_t v = (int32_t)strtol(p, &e, 10); memcpy(vd, &v, 4); }
        else if (val_kind == GM_VAL_U32) { uint32_t v = (uint32_t)strtoul(p, &e, 10); memcpy(vd, &v, 4); }
        else if (val_kind == GM_VAL_F32) { float v = strtof(p, &e); memcpy(vd, &v, 4); }
        else { double v = strtod(p, &e); memcpy(vd, &v, 8); }
        if (e == p) break;  // a line without its value ends the list, like a failed fscanf
        p = e;
      } else {
        store_one(vd, val_kind);
      }
      g.s[g.n] = (int32_t)a;
      g.d[g.n] = (int32_t)b;
      if (a > maxs) maxs = (int)a;
      if (b > maxd) maxd = (int)b;
      g.n++;
    }
    if (rc == GM_OK && hcount >= 0 && g.n < hcount) {
      gm::set_error("%s: header says %lld edges, file holds %lld", path, (long long)hcount, (long long)g.n);
      rc = GM_ERR_IO;
    }
    if (rc == GM_OK && !g.reserve(1)) rc = GM_ERR_NOMEM;
  }
  free(buf);
  if (rc != GM_OK) {
    if (rc == GM_ERR_NOMEM) gm::set_error("gm_edgelist_read: out of host memory");
    g.release();
    return rc;
  }
  *m = header ? hm : maxs;
  *n = header ? hn : maxd;
  *nnz = g.n;
  *h_src = g.s;
  *h_dst = g.d;
  *h_val = g.v;
  return GM_OK;
}

int gm_edgelist_write(const char* path, int binary, int header, int weights, int val_kind, int m, int n, int64_t nnz,
                      const int32_t* h_src, const int32_t* h_dst, const void* h_val) {
  const int vb = val_kind_bytes(val_kind);
  if (!path || nnz < 0 || (nnz && (!h_src || !h_dst)) || vb <= 0 || (weights && nnz && !h_val)) { gm::set_error("gm_edgelist_write: invalid argument"); return GM_ERR_INVALID; }
  if (!binary && weights && val_kind > 0x100) { gm::set_error("gm_edgelist_write: opaque values need a binary file"); return GM_ERR_INVALID; }
  FILE* fp = fopen(path, binary ? "wb" : "w");
  if (!fp) { gm::set_error("Could not open file for writing: %s", path); return GM_ERR_IO; }
  bool ok = true;
  const unsigned char* vals = (const unsigned char*)h_val;
  if (binary) {
    if (header) {
      int32_t h[3] = {m, n, (int32_t)nnz};
      ok = fwrite(h, 4, 3, fp) == 3;
    }
    const size_t rec = 8 + (weights ? (size_t)vb : 0);
    const int64_t chunk = 1 << 16;
    unsigned char* out = (unsigned char*)malloc((size_t)chunk * rec);
    if (!out) { fclose(fp); gm::set_error("gm_edgelist_write: out of host memory"); return GM_ERR_NOMEM; }
    for (int64_t base = 0; ok && base < nnz; base += chunk) {
      const int64_t cnt = nnz - base < chunk ? nnz - base : chunk;
      for (int64_t i = 0; i < cnt; i++) {
        unsigned char* r = out + (size_t)i * rec;
        memcpy(r, &h_src[base + i], 4);
        memcpy(r + 4, &h_dst[base + i], 4);
        if (weights) memcpy(r + 8, vals + (size_t)(base + i) * vb, vb);
      }
      ok = fwrite(out, rec, (size_t)cnt, fp) == (size_t)cnt;
    }
    free(out);
  } else {
    if (header) ok = fprintf(fp, "%d %d %u\n", m, n, (unsigned)nnz) > 0;
    for (int64_t i = 0; ok && i < nnz; i++) {
      int w;
      if (!weights) w = fprintf(fp, "%d %d\n", h_src[i], h_dst[i]);
      else if (val_kind == GM_VAL_I32) { int32_t v; memcpy(&v, vals + i * 4, 4); w = fprintf(fp, "%d %d %d\n", h_src[i], h_dst[i], v); }
      else if (val_kind == GM_VAL_U32) { uint32_t v; memcpy(&v, vals + i * 4, 4); w = fprintf(fp, "%d %d %u\n", h_src[i], h_dst[i], v); }
      else if (val_kind == GM_VAL_F32) { float v; memcpy(&v, vals + i * 4, 4); w = fprintf(fp, "%d %d %.8f\n", h_src[i], h_dst[i], v); }
      else { double v; memcpy(&v, vals + i * 8, 8); w = fprintf(fp, "%d %d %.15lf\n", h_src[i], h_dst[i], v); }
      ok = w > 0;
    }
  }
  if (fclose(fp) != 0) ok = false;
  if (!ok) { gm::set_error("%s: write failed", path); return GM_ERR_IO; }
  return GM_OK;
}

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
