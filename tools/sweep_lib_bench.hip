// tools/sweep_lib_bench.hip -- the library's sweep kernel (kernels.hpp: k_spmv_sell) timed ALONE on the structure the library
// builds (gm_graph_sweep), with measurement forms of the same source (no gathers / every gather from LDS / no long rows) to see
// what bounds it.  The plain form is checked against a serial fold of the CSR rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Igraphmat_amd/csrc -DGRAPHMAT_NO_MPI tools/sweep_lib_bench.hip -Lgraphmat_amd -lgraphmat_hip -o build/sweep_lib_bench
//   LD_LIBRARY_PATH=graphmat_amd build/sweep_lib_bench [scale 26] [reps 5] [key=value library options ...]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <hip/hip_runtime.h>
#define GM_SELL_PHASE_TIMES 1
#include "GraphMatRuntime.h"
#include <rocprim/rocprim.hpp>

#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); exit(1); } } while (0)
#define GOK(e) do { int r_ = (e); if (r_ != 0) { printf("%s:%d %s: %s\n", __FILE__, __LINE__, #e, gm_last_error()); exit(1); } } while (0)
#define HD __host__ __device__

struct Vp { float a; int b; HD Vp() : a(0), b(0) {} };
struct SumP : GraphMat::GraphProgram<float, float, Vp, int> {
  SumP() { this->activity = GraphMat::ALL_VERTICES; this->process_message_requires_vertexprop = false; }
  HD void reduce_function(float& a, const float& b) const { a += b; }
  HD void process_message(const float& m, const int, const Vp&, float& res) const { res = m; }
  HD bool send_message(const Vp& v, float& m) const { m = v.a; return true; }
  HD void apply(const float& y, Vp& v) { v.a = y; }
};

__global__ void k_keep_rows(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int64_t ne, int k, int32_t* __restrict__ s2, int32_t* __restrict__ d2,
                            unsigned long long* __restrict__ cnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
    if ((((uint32_t)dst[i] * 2654435761u) >> 12) % (uint32_t)k == 0u) {  // (a hash: the low bits of an RMAT id are not independent of its degree)
      const unsigned long long p = atomicAdd(cnt, 1ull);
      s2[p] = src[i];
      d2[p] = dst[i];
    }
  }
}
__global__ void k_fill_x(float* __restrict__ x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = (float)(h >> 8) * (1.0f / 16777216.0f) + 1e-3f; }
}
// serial fold of the swept rows (row_of_slot lists) straight from the CSR
__global__ void k_ref(const int32_t* __restrict__ rows, int n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ x, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = rows[i];
  if (r < 0) return;
  const int64_t e0 = rowptr[r], e1 = rowptr[r + 1];
  float acc = x[col[e0]];
  for (int64_t k = e0 + 1; k < e1; k++) acc += x[col[k]];
  y[r] = acc;
}

// ---- feasibility of running the short rows NEXT TO the sweep (round 5): a sliced-ELLPACK form of the rows of 1..64 edges that
// needs no LDS and few registers (lane = row, whole rows, groups of 64 rows of nearly equal length), built here from the library's CSR
__global__ void k_short_lens(const int64_t* __restrict__ rowptr, int nrows, int short_row, uint32_t* __restrict__ key, int32_t* __restrict__ id) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int64_t l = rowptr[r + 1] - rowptr[r];
  key[r] = (l >= 1 && l <= short_row) ? (uint32_t)(short_row - l) : 0xffffffffu;  // ascending key = descending length; others last
  id[r] = r;
}
__global__ void k_short_gsize(const int32_t* __restrict__ rows, uint32_t ngroups, const int64_t* __restrict__ rowptr, uint32_t* __restrict__ gsize) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  const int r = rows[(size_t)g * 64];
  gsize[g] = (uint32_t)(rowptr[r + 1] - rowptr[r]) * 64u;
}
__global__ void k_short_fill(const int32_t* __restrict__ rows, int nshort, uint32_t ngroups, const uint32_t* __restrict__ gbase, const int64_t* __restrict__ rowptr,
                             const int32_t* __restrict__ col, uint32_t* __restrict__ scol) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= ngroups) return;
  const size_t q = (size_t)g * 64 + lane;
  int64_t e0 = 0, e1 = 0;
  if (q < (size_t)nshort) { const int r = rows[q]; e0 = rowptr[r]; e1 = rowptr[r + 1]; }
  const uint32_t b = gbase[g], w = (gbase[g + 1] - b) >> 6;
  for (uint32_t k = 0; k < w; k++) scol[(size_t)b + (size_t)k * 64 + lane] = (int64_t)k < e1 - e0 ? ((uint32_t)col[e0 + k] << 2) : 0x80000000u;
}
template <int UBS>
__global__ void __launch_bounds__(256)
k_sell_short(const uint32_t* __restrict__ scol, const uint32_t* __restrict__ gbase, uint32_t ngroups, const int32_t* __restrict__ rows, int nshort,
             const float* __restrict__ x, float* __restrict__ y) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= ngroups) return;
  const uint32_t b = gbase[g], w = (gbase[g + 1] - b) >> 6;
  const char* __restrict__ xb = (const char*)x;
  float acc = 0.f;
  bool has = false;
  for (uint32_t k = 0; k < w; k += UBS) {
    uint32_t c[UBS];
#pragma unroll
    for (int j = 0; j < UBS; j++) c[j] = __builtin_nontemporal_load(&scol[(size_t)b + (size_t)(k + j < w ? k + j : w - 1) * 64 + lane]);
    float m[UBS];
#pragma unroll
    for (int j = 0; j < UBS; j++) m[j] = *(const float*)(xb + (c[j] & 0x7fffffffu));
#pragma unroll
    for (int j = 0; j < UBS; j++)
      if (k + j < w && (int32_t)c[j] >= 0) { acc = has ? acc + m[j] : m[j]; has = true; }
  }
  const size_t q = (size_t)g * 64 + lane;
  if (q < (size_t)nshort && has) y[rows[q]] = acc;
}
__global__ void k_ref_rows(const int32_t* __restrict__ rows, int n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ x, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = rows[i];
  const int64_t e0 = rowptr[r], e1 = rowptr[r + 1];
  float acc = x[col[e0]];
  for (int64_t k = e0 + 1; k < e1; k++) acc += x[col[k]];
  y[r] = acc;
}

int main(int argc, char** argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 26;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  int rows_of = 1;  // rows_of=K: only the in-edges of every K-th vertex are kept -- the rows a shard of K would own, against the WHOLE message vector
  for (int a = 3; a < argc; a++) {
    char* eq = strchr(argv[a], '=');
    if (!eq) continue;
    *eq = 0;
    if (!strcmp(argv[a], "rows_of")) { rows_of = atoi(eq + 1); continue; }
    if (gm_set_option(argv[a], atoi(eq + 1)) != 0) { printf("option %s: %s\n", argv[a], gm_last_error()); return 1; }
  }
  const int nv = 1 << scale;
  int64_t ne = 16ll * nv;
  int32_t *src, *dst;
  OK(hipMalloc(&src, ne * 4)); OK(hipMalloc(&dst, ne * 4));
  GOK(gm_rmat_generate(scale, 1, 0, ne, src, dst, nullptr, 0, nullptr));
  OK(hipDeviceSynchronize());
  if (rows_of > 1) {
    int32_t *s2, *d2; unsigned long long* cnt;
    OK(hipMalloc(&s2, ne * 4)); OK(hipMalloc(&d2, ne * 4)); OK(hipMalloc(&cnt, 8)); OK(hipMemset(cnt, 0, 8));
    k_keep_rows<<<65536, 256>>>(src, dst, ne, rows_of, s2, d2, cnt);
    unsigned long long kept = 0;
    OK(hipMemcpy(&kept, cnt, 8, hipMemcpyDeviceToHost));
    OK(hipFree(src)); OK(hipFree(dst)); OK(hipFree(cnt));
    src = s2; dst = d2;
    printf("rows_of=%d: %llu of %lld edges kept (the in-edges of one vertex in %d, by a hash of its id)\n", rows_of, kept, (long long)ne, rows_of);
    ne = (int64_t)kept;
  }
  gm_graph_desc_t d;
  memset(&d, 0, sizeof(d));
  d.nvertices = nv; d.nparts = 16; d.row_lo = 0; d.row_hi = nv; d.directions = GM_DIR_OUT | GM_DIR_IN; d.val_bytes = 0; d.ids_on_device = 1;
  d.layout = GM_LAYOUT_DEGREE; d.nshards = 1;
  gm_graph_t* g = nullptr;
  GOK(gm_graph_create(&g, &d, ne, src, dst, nullptr, nullptr));
  OK(hipFree(src)); OK(hipFree(dst));
  gm_graph_desc_t gd; GOK(gm_graph_desc(g, &gd));
  gm_csr_t A; GOK(gm_graph_csr(g, GM_DIR_OUT, &A));
  gm_sweep_t S; GOK(gm_graph_sweep(g, &S));
  printf("RMAT-%d: %d tiles, %d slices, sweep: %d rows (%d long) in %d set(s), %lld + %lld edges, %lld entries (+%.1f %%), %lld groups (%.2f rows each + meta), largest long block %d\n",
         scale, gd.col_tiles, S.nslices, S.nrows, S.nrows_long, S.nsets, (long long)S.nedges, (long long)S.nedges_long, (long long)S.nentries,
         S.nedges ? 100.0 * ((double)S.nentries / S.nedges - 1.0) : 0.0, (long long)S.ngroups, S.ngroups ? (double)S.nentries / 64 / S.ngroups - 1.0 : 0.0, S.max_long_block); printf("medium / long border: %d edges\n", S.long_row);
  if (S.nrows <= 0) { printf("no sweep structure\n"); return 0; }
  float *x, *y, *yref;
  OK(hipMalloc(&x, (size_t)gd.ndevice * 4)); OK(hipMalloc(&y, (size_t)gd.ndevice * 4)); OK(hipMalloc(&yref, (size_t)gd.ndevice * 4));
  k_fill_x<<<(gd.ndevice + 255) / 256, 256>>>(x, gd.ndevice);
  OK(hipMemset(y, 0, (size_t)gd.ndevice * 4)); OK(hipMemset(yref, 0, (size_t)gd.ndevice * 4));
  SumP prog;
  GraphMat::dev::ProgArg<SumP> pa = GraphMat::dev::make_prog_arg(&prog);
  int stage = 64;
  if (S.nrows_long > 0) { stage = (S.max_long_block + 63) / 64 * 64; if (stage > GM_SWEEP_MAX_STAGE) stage = GM_SWEEP_MAX_STAGE; if (stage < 1024) stage = 1024; }
  float* gterms = nullptr;  // (set below for the form that also gathers for the giant rows)
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  const int64_t nedges = S.nedges + S.nedges_long;
  auto time_it = [&](auto launch, const char* name) {
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < reps + 1; r++) {
      OK(hipEventRecord(e0));
      launch();
      OK(hipEventRecord(e1));
      OK(hipEventSynchronize(e1));
      float ms; OK(hipEventElapsedTime(&ms, e0, e1));
      if (r) { best = ms < best ? ms : best; sum += ms; }
    }
    OK(hipGetLastError());
    printf("%-58s best %.3f ms, mean %.3f ms  = %.2f ps per edge, %.1f G edges/s\n", name, best, sum / reps, best * 1e9 / nedges, nedges / best * 1e-6);
    fflush(stdout);
  };
#define LAUNCH(ABL, STG) LAUNCHP(ABL, STG, 7, 2)
#define LAUNCHU(ABL, STG, UBAT) LAUNCHP(ABL, STG, UBAT, 1)
#define LAUNCHP(ABL, STG, UBAT, PIPE) for (int set = 0; set < S.nsets; set++) hipLaunchKernelGGL((GraphMat::dev::k_spmv_sell<SumP, float, float, Vp, int, false, ABL, UBAT, PIPE>), dim3(256), dim3(1024), 0, 0, pa, set, STG, \
      S.nslices, S.nrows_long, S.slice_base, S.scol, (const uint32_t*)nullptr, S.wrow, S.row_of_slot, S.lcol, (const uint32_t*)nullptr, S.lps, S.lrow_of_slot, \
      S.gcol, (const uint32_t*)nullptr, S.gdst, S.gslice, gterms, (const float*)x, y)
  time_it([&]() { LAUNCH(0, stage); }, "k_spmv_sell (the library's form)");
  {
    const size_t nm = (size_t)S.nsets * 256 * S.acc_rows, nl = (size_t)S.nsets * 256 * S.long_slots;
    k_ref<<<(unsigned)((nm + 255) / 256), 256>>>(S.row_of_slot, (int)nm, A.rowptr, A.colidx, x, yref);
    k_ref<<<(unsigned)((nl + 255) / 256), 256>>>(S.lrow_of_slot, (int)nl, A.rowptr, A.colidx, x, yref);
    OK(hipDeviceSynchronize());
    std::vector<float> a(gd.ndevice), b(gd.ndevice);
    OK(hipMemcpy(a.data(), y, (size_t)gd.ndevice * 4, hipMemcpyDeviceToHost));
    OK(hipMemcpy(b.data(), yref, (size_t)gd.ndevice * 4, hipMemcpyDeviceToHost));
    int64_t bad = 0, set_ = 0;
    for (int i = 0; i < gd.ndevice; i++) { bad += memcmp(&a[i], &b[i], 4) != 0; set_ += b[i] != 0.f; }
    printf("   against the serial fold of the CSR rows: %lld of %lld rows differ (bit compare)\n", (long long)bad, (long long)set_);
  }
  auto check = [&](const char* what) {
    OK(hipDeviceSynchronize());
    std::vector<float> a(gd.ndevice), b(gd.ndevice);
    OK(hipMemcpy(a.data(), y, (size_t)gd.ndevice * 4, hipMemcpyDeviceToHost));
    OK(hipMemcpy(b.data(), yref, (size_t)gd.ndevice * 4, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (int i = 0; i < gd.ndevice; i++) bad += memcmp(&a[i], &b[i], 4) != 0;
    printf("   %s: %lld rows differ from the serial fold\n", what, (long long)bad);
  };
  OK(hipMemset(y, 0, (size_t)gd.ndevice * 4));
  time_it([&]() { LAUNCHP(0, stage, 8, 2); }, "  ... two batches deep, batches of 8 rows");
  check("two batches deep, 8 rows");
  time_it([&]() { LAUNCHP(0, stage, 4, 2); }, "  ... two batches deep, batches of 4 rows");
  time_it([&]() { LAUNCHP(0, stage, 5, 2); }, "  ... two batches deep, batches of 5 rows");
  time_it([&]() { LAUNCHP(0, stage, 6, 2); }, "  ... two batches deep, batches of 6 rows");
  time_it([&]() { LAUNCHP(0, stage, 8, 1); }, "  ... one batch deep (round 5's first form), batches of 8 rows");
  time_it([&]() { LAUNCHP(0, stage, 12, 1); }, "  ... one batch deep, batches of 12 rows");
  if (S.ngiant_edges > 0) {
    OK(hipMalloc(&gterms, (size_t)A.giant_edges * 4 + 256));
    time_it([&]() { LAUNCH(0, stage); }, "  ... gathering for the giant rows as well");
    OK(hipFree(gterms));
    gterms = nullptr;
  }
  time_it([&]() { LAUNCH(0, GM_SWEEP_MAX_STAGE); }, "  ... with the largest stage (smallest hot set)");
  time_it([&]() { LAUNCH(4, 64); }, "  ... without the long rows' phase, largest hot set");
  time_it([&]() { LAUNCH(4, stage); }, "  ... without the long rows' phase, same hot set");
  time_it([&]() { LAUNCH(2, stage); }, "  ... every gather served from LDS");
  time_it([&]() { LAUNCH(1, stage); }, "  ... no gathers at all");
  time_it([&]() { LAUNCH(5, stage); }, "  ... no gathers, no long rows");
  {  // ---- the short rows next to the sweep
    const int nr = A.nrows;
    uint32_t *key, *key2; int32_t *id, *rows;
    OK(hipMalloc(&key, (size_t)nr * 4)); OK(hipMalloc(&key2, (size_t)nr * 4)); OK(hipMalloc(&id, (size_t)nr * 4)); OK(hipMalloc(&rows, (size_t)nr * 4));
    k_short_lens<<<(nr + 255) / 256, 256>>>(A.rowptr, nr, A.short_row, key, id);
    {
      size_t tb = 0;
      OK(rocprim::radix_sort_pairs(nullptr, tb, key, key2, id, rows, (size_t)nr, 0, 32, (hipStream_t)0));
      void* tmp; OK(hipMalloc(&tmp, tb + 256));
      OK(rocprim::radix_sort_pairs(tmp, tb, key, key2, id, rows, (size_t)nr, 0, 32, (hipStream_t)0));
      OK(hipDeviceSynchronize()); OK(hipFree(tmp));
    }
    std::vector<uint32_t> hk(nr);
    OK(hipMemcpy(hk.data(), key2, (size_t)nr * 4, hipMemcpyDeviceToHost));
    int nshort = 0;
    while (nshort < nr && hk[nshort] != 0xffffffffu) nshort++;
    const uint32_t ngroups = (uint32_t)((nshort + 63) / 64);
    uint32_t *gsize, *gbase;
    OK(hipMalloc(&gsize, ((size_t)ngroups + 1) * 4)); OK(hipMalloc(&gbase, ((size_t)ngroups + 1) * 4));
    OK(hipMemset(gsize, 0, ((size_t)ngroups + 1) * 4));
    k_short_gsize<<<(ngroups + 255) / 256, 256>>>(rows, ngroups, A.rowptr, gsize);
    {
      size_t tb = 0;
      OK(rocprim::exclusive_scan(nullptr, tb, gsize, gbase, 0u, (size_t)ngroups + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
      void* tmp; OK(hipMalloc(&tmp, tb + 256));
      OK(rocprim::exclusive_scan(tmp, tb, gsize, gbase, 0u, (size_t)ngroups + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
      OK(hipDeviceSynchronize()); OK(hipFree(tmp));
    }
    uint32_t total = 0; OK(hipMemcpy(&total, gbase + ngroups, 4, hipMemcpyDeviceToHost));
    uint32_t* scs; OK(hipMalloc(&scs, ((size_t)total + 64) * 4));
    k_short_fill<<<(ngroups + 3) / 4, 256>>>(rows, nshort, ngroups, gbase, A.rowptr, A.colidx, scs);
    k_ref_rows<<<(nshort + 255) / 256, 256>>>(rows, nshort, A.rowptr, A.colidx, x, yref);
    OK(hipDeviceSynchronize());
    printf("short rows 1..%d: %d rows, %u entries in %u groups\n", A.short_row, nshort, total, ngroups);
    const int64_t keep_nedges = nedges;
    hipStream_t s1, s2; OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ea, eb, ec; OK(hipEventCreate(&ea)); OK(hipEventCreate(&eb)); OK(hipEventCreate(&ec));
    auto wall = [&](auto both, const char* name) {
      float best = 1e9f;
      for (int r = 0; r < reps + 1; r++) {
        OK(hipDeviceSynchronize());
        OK(hipEventRecord(ea, s1));
        OK(hipStreamWaitEvent(s2, ea, 0));
        both();
        OK(hipEventRecord(eb, s2));
        OK(hipStreamWaitEvent(s1, eb, 0));
        OK(hipEventRecord(ec, s1));
        OK(hipEventSynchronize(ec));
        float ms; OK(hipEventElapsedTime(&ms, ea, ec));
        if (r) best = ms < best ? ms : best;
      }
      OK(hipGetLastError());
      printf("%-70s best %.3f ms\n", name, best);
      fflush(stdout);
    };
#define SWEEP_ON(ABL, STREAM) for (int set = 0; set < S.nsets; set++) hipLaunchKernelGGL((GraphMat::dev::k_spmv_sell<SumP, float, float, Vp, int, false, ABL, 7, 2>), dim3(256), dim3(1024), 0, STREAM, pa, set, stage, \
      S.nslices, S.nrows_long, S.slice_base, S.scol, (const uint32_t*)nullptr, S.wrow, S.row_of_slot, S.lcol, (const uint32_t*)nullptr, S.lps, S.lrow_of_slot, \
      S.gcol, (const uint32_t*)nullptr, S.gdst, S.gslice, (float*)nullptr, (const float*)x, y)
    wall([&]() { SWEEP_ON(0, s1); }, "sweep alone (121 VGPRs)");
    wall([&]() { SWEEP_ON(4, s1); }, "sweep without its long rows alone (fewer VGPRs)");
    wall([&]() { k_sell_short<2><<<(ngroups + 3) / 4, 256, 0, s2>>>(scs, gbase, ngroups, rows, nshort, x, y); }, "short rows (sliced-ELLPACK form, batch 2) alone");
    wall([&]() { k_sell_short<4><<<(ngroups + 3) / 4, 256, 0, s2>>>(scs, gbase, ngroups, rows, nshort, x, y); }, "short rows (batch 4) alone");
    wall([&]() { SWEEP_ON(0, s1); k_sell_short<2><<<(ngroups + 3) / 4, 256, 0, s2>>>(scs, gbase, ngroups, rows, nshort, x, y); }, "sweep + short rows on two streams");
    wall([&]() { k_sell_short<2><<<(ngroups + 3) / 4, 256, 0, s2>>>(scs, gbase, ngroups, rows, nshort, x, y); SWEEP_ON(0, s1); }, "short rows + sweep on two streams (short launched first)");
    wall([&]() { SWEEP_ON(4, s1); k_sell_short<2><<<(ngroups + 3) / 4, 256, 0, s2>>>(scs, gbase, ngroups, rows, nshort, x, y); }, "sweep without long rows + short rows on two streams");
    wall([&]() { SWEEP_ON(4, s1); k_sell_short<4><<<(ngroups + 3) / 4, 256, 0, s2>>>(scs, gbase, ngroups, rows, nshort, x, y); }, "sweep without long rows + short rows (batch 4) on two streams");
    // the library's own short-row kernel (k_spmv_rowblock: 17 KB of LDS per workgroup) next to a sweep that leaves it LDS and registers
    {
      gm_csr_t As = A;
      As.nmid = 0; As.nmid_long = 0; As.ngiant = 0; As.ngchunk = 0;
      Vp* novp = nullptr;
#define SWEEP_SMALL(STREAM) for (int set = 0; set < S.nsets; set++) hipLaunchKernelGGL((GraphMat::dev::k_spmv_sell<SumP, float, float, Vp, int, false, 4, 7, 2, GM_SWEEP_POOL - 9216>), dim3(256), dim3(1024), 0, STREAM, pa, set, stage, \
      S.nslices, S.nrows_long, S.slice_base, S.scol, (const uint32_t*)nullptr, S.wrow, S.row_of_slot, S.lcol, (const uint32_t*)nullptr, S.lps, S.lrow_of_slot, \
      S.gcol, (const uint32_t*)nullptr, S.gdst, S.gslice, (float*)nullptr, (const float*)x, y)
#define ROWBLOCK(STREAM) hipLaunchKernelGGL((GraphMat::dev::k_spmv_rowblock<SumP, float, float, Vp, int, false, true, GraphMat::REDUCE_F32_ADD>), dim3(As.nblk), dim3(GraphMat::dev::kBlock), 0, STREAM, pa, As, \
      (const float*)x, (const uint32_t*)nullptr, (const Vp*)novp, y, (uint32_t*)nullptr, (int)GraphMat::dev::ACC_STATIC_BITS GM_DBG_ARG(0), (const uint32_t*)nullptr)
      wall([&]() { ROWBLOCK(s2); }, "k_spmv_rowblock alone");
      wall([&]() { SWEEP_SMALL(s1); }, "sweep without long rows, pool cut by 9216 words (36 KB of LDS free), alone");
      wall([&]() { SWEEP_SMALL(s1); ROWBLOCK(s2); }, "that sweep + k_spmv_rowblock on two streams");
      wall([&]() { ROWBLOCK(s2); SWEEP_SMALL(s1); }, "k_spmv_rowblock + that sweep on two streams (row-blocks launched first)");
    }
    {
      OK(hipDeviceSynchronize());
      std::vector<float> a(gd.ndevice), b(gd.ndevice);
      std::vector<int32_t> hr(nshort);
      OK(hipMemcpy(a.data(), y, (size_t)gd.ndevice * 4, hipMemcpyDeviceToHost));
      OK(hipMemcpy(b.data(), yref, (size_t)gd.ndevice * 4, hipMemcpyDeviceToHost));
      OK(hipMemcpy(hr.data(), rows, (size_t)nshort * 4, hipMemcpyDeviceToHost));
      int64_t bad = 0;
      for (int i = 0; i < nshort; i++) bad += memcmp(&a[hr[i]], &b[hr[i]], 4) != 0;
      printf("   short rows against the serial fold: %lld of %d differ\n", (long long)bad, nshort);
    }
    (void)keep_nedges;
  }
  {  // where a wave's time goes (100 MHz ticks summed over the slices of ONE launch)
    static unsigned long long h[4096][8];
    memset(h, 0, sizeof(h));
    OK(hipMemcpyToSymbol(HIP_SYMBOL(GraphMat::dev::g_sell_phase_ticks), h, sizeof(h)));
    LAUNCH(8, stage);
    OK(hipDeviceSynchronize());
    OK(hipMemcpyFromSymbol(h, HIP_SYMBOL(GraphMat::dev::g_sell_phase_ticks), sizeof(h)));
    const char* names[6] = {"wait at the slice's end", "hot set (load + barrier)", "long rows: staging", "long rows: fold", "groups", "prefetch issue"};
    for (int cls = 0; cls < 2; cls++) {
      double sum[6] = {0}, mx[6] = {0};
      int n = 0;
      for (int w = 0; w < 4096; w++) {
        const bool folder = (w % 16) >= 16 - (((S.nrows_long + 255) / 256 + S.nsets - 1) / S.nsets + 63) / 64;
        if (folder != (cls == 1)) continue;
        n++;
        for (int k = 0; k < 6; k++) { sum[k] += (double)h[w][k]; if ((double)h[w][k] > mx[k]) mx[k] = (double)h[w][k]; }
      }
      printf("%s waves: per-wave mean us (max) over one launch:", cls ? "folding (last of 16)" : "other");
      double tot = 0;
      for (int k = 0; k < 6; k++) { printf("  %s %.0f (%.0f)", names[k], sum[k] / n * 0.01, mx[k] * 0.01); tot += sum[k] / n * 0.01; }
      printf("  | total %.0f us\n", tot);
    }
  }
  gm_graph_destroy(g);
  return 0;
}
