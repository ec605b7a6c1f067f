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

int main(int argc, char** argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 26;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  for (int a = 3; a < argc; a++) {
    char* eq = strchr(argv[a], '=');
    if (!eq) continue;
    *eq = 0;
    if (gm_set_option(argv[a], atoi(eq + 1)) != 0) { printf("option %s: %s\n", argv[a], gm_last_error()); return 1; }
  }
  const int nv = 1 << scale;
  const int64_t ne = 16ll * nv;
  int32_t *src, *dst;
  OK(hipMalloc(&src, ne * 4)); OK(hipMalloc(&dst, ne * 4));
  GOK(gm_rmat_generate(scale, 1, 0, ne, src, dst, nullptr, 0, nullptr));
  OK(hipDeviceSynchronize());
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
