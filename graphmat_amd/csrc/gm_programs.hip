// gm_programs.hip -- the fixed-menu vertex programs behind the C-ABI.
//
// Each program is an ordinary GraphMat::GraphProgram subclass (this project's own
// restatement of the programs in the reference's src/PageRank.cpp, src/BFS.cpp,
// src/SSSP.cpp, src/SGD.cpp) run by the same engine (include/graphmat/engine.hpp)
// that user programs compiled against include/GraphMatRuntime.h use.  The only
// difference to a user program is that the methods are annotated __host__
// __device__ here, so this file builds without --hipstdpar.
// Built with -ffp-contract=off: floating-point expressions are evaluated as
// written (no fused multiply-add), like the oracle.
#include <limits.h>
#include <math.h>

#include <string>

#include "GraphMatRuntime.h"
#include "gm_internal.hpp"

#define HD __host__ __device__

namespace gm {

extern int g_short_row, g_giant_row, g_rank_by, g_rank_cap, g_col_tiles, g_tile_min_row, g_long_mid, g_tile_balance, g_own_wave_row, g_sort_tile_lists, g_sweep_slices, g_sweep_acc_limit, g_sweep_long_limit, g_sweep_long_row, g_sweep_fold_share, g_sweep_border_factor, g_blocked_rows, g_sweep_waves, g_sweep_stream, g_sweep_stream_weight;
static int g_force_ordered = 0;

// ---------------- PageRank (reference: src/PageRank.cpp:34-112) ----------------------------
struct PRv : gm_pr_t {
  HD PRv() { pagerank = 0.3; degree = 0; }
  // tolerance test of the reference's PR::operator!= (:44-46)
  HD int operator!=(const PRv& p) { return ((double)fabsf(p.pagerank - pagerank) > 1e-5); }
};

struct DegreeP : GraphMat::GraphProgram<int, int, PRv, int> {
  DegreeP() {
    this->order = GraphMat::IN_EDGES;
    this->process_message_requires_vertexprop = false;
  }
  HD bool send_message(const PRv&, int& message) const { message = 1; return true; }
  HD void process_message(const int& message, const int, const PRv&, int& result) const { result = message; }
  HD void reduce_function(int& a, const int& b) const { a += b; }
  HD void apply(const int& message_out, PRv& vertexprop) { vertexprop.degree = message_out; }
};

template <int ORDERED>
struct PageRankP : GraphMat::GraphProgram<float, float, PRv, int> {
  float alpha;
  explicit PageRankP(float a) : alpha(a) {
    this->activity = GraphMat::ALL_VERTICES;
    this->process_message_requires_vertexprop = false;
  }
  HD void reduce_function(float& a, const float& b) const { a += b; }
  HD void process_message(const float& message, const int, const PRv&, float& res) const { res = message; }
  HD bool send_message(const PRv& vertexprop, float& message) const {
    if (vertexprop.degree == 0) message = 0.0f;
    else message = vertexprop.pagerank / (float)vertexprop.degree;
    return true;
  }
  // double arithmetic on float operands, narrowed on store (:108-110)
  HD void apply(const float& message_out, PRv& vertexprop) { vertexprop.pagerank = alpha + (1.0 - alpha) * message_out; }
};

// ---------------- BFS (reference: src/BFS.cpp:36-99) -----------------------------------------
struct BFSv : gm_bfs_t {
  HD BFSv() { depth = UINT_MAX; pad_ = 0; parent = (uint64_t)-1; id = (uint64_t)-1; }
  HD bool operator!=(const BFSv& p) { return this->depth != p.depth; }
};
struct BfsP : GraphMat::GraphProgram<unsigned long long, unsigned long long, BFSv, int> {
  unsigned int current_depth;
  BfsP() : current_depth(1) {
    this->order = GraphMat::OUT_EDGES;
    this->process_message_requires_vertexprop = false;
  }
  HD void reduce_function(unsigned long long& a, const unsigned long long& b) const { a = b; }
  HD void process_message(const unsigned long long& m, const int, const BFSv&, unsigned long long& res) const { res = m; }
  HD bool send_message(const BFSv& vp, unsigned long long& m) const { m = vp.id; return vp.depth == current_depth - 1; }
  HD void apply(const unsigned long long& y, BFSv& vp) {
    if (vp.depth == UINT_MAX) { vp.depth = current_depth; vp.parent = y; }
  }
  void do_every_iteration(int) { current_depth++; }
};

// ---------------- SSSP (reference: src/SSSP.cpp:36-90) -----------------------------------------
struct SSSPv {
  unsigned int distance;
  HD SSSPv() : distance(UINT_MAX) {}
  HD bool operator!=(const SSSPv& p) { return this->distance != p.distance; }
};
struct SsspP : GraphMat::GraphProgram<unsigned int, unsigned int, SSSPv, int> {
  SsspP() {
    this->order = GraphMat::OUT_EDGES;
    this->process_message_requires_vertexprop = false;
  }
  HD void reduce_function(unsigned int& a, const unsigned int& b) const { a = (a <= b) ? a : b; }
  HD void process_message(const unsigned int& m, const int e, const SSSPv&, unsigned int& res) const { res = m + e; }
  HD bool send_message(const SSSPv& vp, unsigned int& m) const { m = vp.distance; return true; }
  HD void apply(const unsigned int& y, SSSPv& vp) { vp.distance = vp.distance < y ? vp.distance : y; }
};

// ---------------- SGD / RMSE (reference: src/SGD.cpp:36-156) --------------------------------------
template <class R, int K>
struct Latent {
  R lv[K];
  R sqerr;
  HD Latent() {}
  HD bool operator!=(const Latent& p) {
    bool result = false;
    for (int i = 0; i < K; i++)
      if (fabs(p.lv[i] - lv[i]) > 1e-7) result = true;
    return result;
  }
};
template <class R, int K>
struct SgdP : GraphMat::GraphProgram<Latent<R, K>, Latent<R, K>, Latent<R, K>, int> {
  typedef Latent<R, K> L;
  R lambda, step;
  SgdP(R l, R s) : lambda(l), step(s) {
    this->order = GraphMat::ALL_EDGES;
    this->activity = GraphMat::ALL_VERTICES;
  }
  HD void reduce_function(L& v, const L& w) const { for (int i = 0; i < K; i++) v.lv[i] += w.lv[i]; }
  HD void process_message(const L& m, const int e, const L& vp, L& res) const {
    R estimate = 0;
    for (int i = 0; i < K; i++) estimate += m.lv[i] * vp.lv[i];
    R error = e - estimate;
    for (int i = 0; i < K; i++) res.lv[i] = m.lv[i] * error;
    res.sqerr = 0;
  }
  HD bool send_message(const L& vp, L& m) const { m = vp; return true; }
  HD void apply(const L& y, L& vp) { for (int i = 0; i < K; i++) vp.lv[i] += step * (-lambda * vp.lv[i] + y.lv[i]); }
};
template <class R, int K>
struct RmseP : GraphMat::GraphProgram<Latent<R, K>, R, Latent<R, K>, int> {
  typedef Latent<R, K> L;
  RmseP() { this->order = GraphMat::IN_EDGES; }
  HD void reduce_function(R& v, const R& w) const { v += w; }
  HD void process_message(const L& m, const int e, const L& vp, R& res) const {
    R est = 0;
    for (int i = 0; i < K; i++) est += m.lv[i] * vp.lv[i];
    R error = e - est;
    res = error * error;
  }
  HD bool send_message(const L& vp, L& m) const { m = vp; return true; }
  HD void apply(const R& y, L& vp) { vp.sqerr = y; }
};

}  // namespace gm

// what the runtime may assume about each program's reduce_function
namespace GraphMat {
template <> struct program_traits<gm::DegreeP> { static constexpr reduce_kind reduce = REDUCE_COMMUTATIVE; };
template <> struct program_traits<gm::PageRankP<0> > { static constexpr reduce_kind reduce = REDUCE_F32_ADD; };
template <> struct program_traits<gm::PageRankP<1> > { static constexpr reduce_kind reduce = REDUCE_ORDERED; };
template <> struct program_traits<gm::BfsP> { static constexpr reduce_kind reduce = REDUCE_LAST; };
template <> struct program_traits<gm::SsspP> { static constexpr reduce_kind reduce = REDUCE_COMMUTATIVE; };
// BFS apply() ignores messages once a vertex has a depth: such rows skip the multiply
template <> struct program_row_filter<gm::BfsP> {
  static constexpr bool enabled = true;
  HD static bool wants(const gm::BfsP&, const gm::BFSv& v) { return v.depth == UINT_MAX; }
};
}  // namespace GraphMat

namespace gm {

// ---------------- dedicated SGD / RMSE kernels for wide fp32 latent vectors -----------------------
// (BASELINE config 5: K = 128).  Same arithmetic, in the same order, as SgdP / RmseP above (and
// as the reference's src/SGD.cpp:93-115,135-152): per edge a sequential K-term dot product, the
// scaled message, and a sequential accumulation over a row's edges -- so results are identical
// to the generic kernels'; only the mapping to the machine differs:
//   * x (the message vector = a copy of the latent vectors) is laid out with a row stride of K
//     floats (512 B for K = 128), without the sqerr field, so a row is one aligned burst;
//   * one wave per row, 16 edges at a time, their x rows fetched once as coalesced bursts into an
//     LDS tile (next tile in flight meanwhile): phase A gives every lane one edge and walks its x
//     row sequentially against the row's own vector (the ordered dot product); phase B gives
//     every lane K/64 components and walks the edges in order, accumulating error-scaled messages.
// HBM-bound: 4K flop against one 4K-byte row per edge; MFMA has nothing to chew on (each dot is a
// 1 x K by K x 1 product with no operand shared between edges), see DESIGN.md.
constexpr int kSgdBlock = 256;

template <int K>
__global__ void __launch_bounds__(kSgdBlock)
k_sgd_send(const float* __restrict__ vp, float* __restrict__ x, int n) {  // x[v][0..K) = vp[v].lv (x: this shard's slice)
  const int64_t i = (int64_t)blockIdx.x * kSgdBlock + threadIdx.x;
  if (i >= (int64_t)n * K) return;
  const int v = (int)(i / K), c = (int)(i % K);
  x[i] = vp[(int64_t)v * (K + 1) + c];
}

// x holds every shard's slice (ndevice rows); ours was just written, the others arrive through the
// exchange callback (an all-gather of K*4-byte elements).  The presence words travel with it
// (the callback's contract) but all vertices send, so nobody reads them.
static int sgd_exchange(gm_graph_t* g, float* x, int K, hipStream_t s) {
  if (!g->xfn) return GM_OK;
  g->run_stream = s;
  void* bits = nullptr;
  int rc;
  if ((rc = gm_graph_workspace(g, 2, ((size_t)(g->desc.ndevice + 31) / 32 + 2) * 4, &bits))) return rc;
  if (gm_graph_exchange(g, GM_XCHG_MESSAGES, x, (int64_t)K * 4, (uint32_t*)bits, nullptr) != 0) {
    set_error("gm_run_sgd: message exchange callback failed");
    return GM_ERR_INVALID;
  }
  return GM_OK;
}

// MODE 0: SGD messages into y (row stride K);  MODE 1: RMSE, squared errors summed into y1[row]
// One wave per row, kSgdTile (16) edges per step.  The step's x rows are fetched exactly once, as
// whole coalesced bursts (each load instruction covers 64/(K/4) complete rows), and parked in the
// wave's LDS tile; the next step's rows are already in flight while this step computes:
//   phase A  lane = edge: the ordered K-term dot product of its x row (LDS) with the row's own
//            vector (LDS, broadcast), then err = rating - estimate;
//   phase B  lane = K/64 components: the tile's edges in stored order, y += x_row * err, x from LDS.
constexpr int kSgdMulBlock = 128;  // 2 waves: 2 x (16 x (K+4) + K) floats of LDS = 17.9 KB at K = 128
constexpr int kSgdTile = 16;       // measured on 2e8 ratings: tile 8 / 16 / 32 / 64 -> 41.0 / 31.6 / 34.6 / 50.3 ms per iteration

template <int K, int MODE>
__global__ void __launch_bounds__(kSgdMulBlock)
k_sgd_multiply(gm_csr_t A, const float* __restrict__ x, const float* __restrict__ vp, float* __restrict__ y,
               const uint32_t* __restrict__ prev_bits, int accumulate,
               const int32_t* __restrict__ rows = nullptr /* only these rows (a block of the bipartite exchange), or all */, int nlist = 0) {
  static_assert(K % 64 == 0 && 256 % K == 0, "K must be 64, 128 or 256");
  constexpr int PER = K / 64;        // components per lane in phase B
  constexpr int LPR = K / 4;         // lanes that cover one x row with a float4 each
  constexpr int RPL = 64 / LPR;      // rows per load instruction
  constexpr int NLD = kSgdTile / RPL;  // load instructions per tile
  constexpr int XS = K + 4;          // LDS row stride in floats (keeps float4 alignment, staggers banks)
  constexpr int WPB = kSgdMulBlock / 64;
  __shared__ float s_v[WPB][K];
  __shared__ __attribute__((aligned(16))) float s_x[WPB][kSgdTile][XS];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int row = blockIdx.x * WPB + wv;
  if (rows != nullptr) {
    if (row >= nlist) return;
    row = rows[row];
  }
  if (row >= A.nrows) return;
  const int64_t e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
  if (e1 == e0) return;
  const float* vrow = vp + (int64_t)row * (K + 1);
#pragma unroll
  for (int j = 0; j < PER; j++) s_v[wv][lane + 64 * j] = vrow[lane + 64 * j];
  bool has = accumulate && ((prev_bits[row >> 5] >> (row & 31)) & 1u);
  float acc[PER];
  float sq = 0.f;
  if (MODE == 0) {
#pragma unroll
    for (int j = 0; j < PER; j++) acc[j] = has ? y[(int64_t)row * K + lane + 64 * j] : 0.f;
  } else if (has) {
    sq = y[row];
  }
  const int* vals = (const int*)A.vals;
  const int sub = lane / LPR, part = lane % LPR;  // which of the RPL rows of a load, which float4 of it

  int col = 0, rating = 0;
  float4 q[NLD];
  auto fetch = [&](int64_t base) {  // column ids + ratings of the tile, then its x rows
    const int n = (int)((e1 - base) < kSgdTile ? (e1 - base) : kSgdTile);
    col = 0;
    rating = 0;
    if (lane < n) {
      col = __builtin_nontemporal_load(A.colidx + base + lane);
      rating = __builtin_nontemporal_load(vals + base + lane);
    }
#pragma unroll
    for (int r = 0; r < NLD; r++) {
      const int e = r * RPL + sub;
      const int c = __shfl(col, e);
      q[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < n) q[r] = *reinterpret_cast<const float4*>(x + (int64_t)c * K + 4 * part);
    }
  };
  fetch(e0);
  for (int64_t base = e0; base < e1; base += kSgdTile) {
    const int n = (int)((e1 - base) < kSgdTile ? (e1 - base) : kSgdTile);
    const int my_rating = rating;
    __builtin_amdgcn_wave_barrier();  // earlier reads of the tile are done (LDS is in order per wave)
#pragma unroll
    for (int r = 0; r < NLD; r++) *reinterpret_cast<float4*>(&s_x[wv][r * RPL + sub][4 * part]) = q[r];
    __builtin_amdgcn_wave_barrier();
    if (base + kSgdTile < e1) fetch(base + kSgdTile);  // next tile's rows fly during the phases below
    // phase A
    float err = 0.f;
    if (lane < n) {
      float est = 0.f;
#pragma unroll 8
      for (int i = 0; i < K / 4; i++) {
        const float4 xq = *reinterpret_cast<const float4*>(&s_x[wv][lane][4 * i]);
        const float4 w = *reinterpret_cast<const float4*>(&s_v[wv][4 * i]);
        est += xq.x * w.x;
        est += xq.y * w.y;
        est += xq.z * w.z;
        est += xq.w * w.w;
      }
      err = (float)my_rating - est;
    }
    if (MODE == 1) {
      const float e2 = err * err;
      for (int e = 0; e < n; e++) {
        const float t = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(e2), e));
        if (has) sq += t; else { sq = t; has = true; }
      }
    } else {
      // phase B
      for (int e = 0; e < n; e++) {
        const float ee = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(err), e));
#pragma unroll
        for (int j = 0; j < PER; j++) {
          const float r = s_x[wv][e][lane + 64 * j] * ee;
          acc[j] = has ? acc[j] + r : r;
        }
        has = true;
      }
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int j = 0; j < PER; j++) y[(int64_t)row * K + lane + 64 * j] = acc[j];
  } else if (lane == 0) {
    y[row] = sq;
  }
}

// ---- the dot products on the matrix cores (option "sgd_mfma"; NOT the default, an opt-in MEASUREMENT form: the instruction sums
// its partial dots in its own order, so results deviate from the reference's sequential K-term dot by up to ~1e-5 of the vectors'
// scale -- outside north_star's 1e-6 bar and two orders of magnitude outside the reference's own fused-vs-unfused spread
// (tests/test_gpu_parity.py::test_sgd_k128_matrix_core_option_deviation_bound, tests/test_oracle_golden.py) ------------------------
// What a matrix core can do for this path at all: per edge the work is ONE K-term dot product of a gathered row with the
// row's own vector -- no operand is shared between the edges of different rows, and at a ratings density of 1e-4 a
// 32 x 32 (rows x columns) block of the matrix holds 0.1 ratings, so there is no dense tile to multiply.  The one
// mapping that wastes little is v_mfma_f32_4x4x1_16b_f32: 16 independent 4 x 4 outer products per instruction, lane
// l = 4 b + i supplying A_b[i] and B_b[i].  With A_b[i] = x_row(edge 4 b + i)[k] and B_b[j] = v[k] for every j, the
// accumulator D_b[i][j] of 128 consecutive instructions (k ascending: the reference's order of the terms) is the dot
// product of edge 4 b + i, four times over (j): 64 edges per wave advance one term per instruction, 1/4 of the unit's
// flops useful, against 16 of 64 lanes busy in the vector form of phase A.  That needs 64-edge tiles: one wave per
// workgroup with a 33 KB LDS tile (64 rows x (K + 1) floats: lane = row at a common k is bank-conflict free) and the next
// tile's 64 rows (32 float4 registers per lane) in flight meanwhile -- the same bytes in flight per CU as the vector form.
// Measured against it in profiles/r04_sgd_mfma.md.
constexpr int kSgdMTile = 64;
typedef float gm_f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ void __launch_bounds__(64)
k_sgd_multiply_mfma(gm_csr_t A, const float* __restrict__ x, const float* __restrict__ vp, float* __restrict__ y,
                    const uint32_t* __restrict__ prev_bits, int accumulate) {
  static_assert(K == 128, "K = 128");
  constexpr int PER = K / 64;        // components per lane in phase B
  constexpr int LPR = K / 4;         // lanes that cover one x row with a float4 each
  constexpr int RPL = 64 / LPR;      // rows per load instruction (2)
  constexpr int NLD = kSgdMTile / RPL;  // load instructions per tile (32)
  constexpr int XS = K + 1;
  __shared__ float s_v[K];
  __shared__ float s_x[kSgdMTile][XS];
  const int lane = threadIdx.x;
  const int row = blockIdx.x;
  if (row >= A.nrows) return;
  const int64_t e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
  if (e1 == e0) return;
  const float* vrow = vp + (int64_t)row * (K + 1);
#pragma unroll
  for (int j = 0; j < PER; j++) s_v[lane + 64 * j] = vrow[lane + 64 * j];
  bool has = accumulate && ((prev_bits[row >> 5] >> (row & 31)) & 1u);
  float acc[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) acc[j] = has ? y[(int64_t)row * K + lane + 64 * j] : 0.f;
  const int* vals = (const int*)A.vals;
  const int sub = lane / LPR, part = lane % LPR;
  int col = 0, rating = 0;
  float4 q[NLD];
  auto fetch = [&](int64_t base) {
    const int n = (int)((e1 - base) < kSgdMTile ? (e1 - base) : kSgdMTile);
    col = 0;
    rating = 0;
    if (lane < n) {
      col = __builtin_nontemporal_load(A.colidx + base + lane);
      rating = __builtin_nontemporal_load(vals + base + lane);
    }
#pragma unroll
    for (int r = 0; r < NLD; r++) {
      const int e = r * RPL + sub;
      const int c = __shfl(col, e);
      q[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < n) q[r] = *reinterpret_cast<const float4*>(x + (int64_t)c * K + 4 * part);
    }
  };
  fetch(e0);
  for (int64_t base = e0; base < e1; base += kSgdMTile) {
    const int n = (int)((e1 - base) < kSgdMTile ? (e1 - base) : kSgdMTile);
    const int my_rating = rating;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < NLD; r++) {
      float* d = &s_x[r * RPL + sub][4 * part];
      d[0] = q[r].x; d[1] = q[r].y; d[2] = q[r].z; d[3] = q[r].w;
    }
    __builtin_amdgcn_wave_barrier();
    if (base + kSgdMTile < e1) fetch(base + kSgdMTile);
    // phase A on the matrix cores: lane = edge; D_b[i][j] += x_{4b+i}[k] * v[k], k ascending
    gm_f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
    for (int k = 0; k < K; k++) d4 = __builtin_amdgcn_mfma_f32_4x4x1f32(s_x[lane][k], s_v[k], d4, 0, 0, 0);
    // lane 4 b + j holds D_b[0..3][j]: edge `lane` = 4 b + (lane & 3) is row lane & 3 of its block
    const int i3 = lane & 3;
    const float est = i3 == 0 ? d4.x : i3 == 1 ? d4.y : i3 == 2 ? d4.z : d4.w;
    const float err = lane < n ? (float)my_rating - est : 0.f;
    // phase B: lane = K/64 components, the tile's edges in stored order
    for (int e = 0; e < n; e++) {
      const float ee = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(err), e));
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const float r = s_x[e][lane + 64 * j] * ee;
        acc[j] = has ? acc[j] + r : r;
      }
      has = true;
    }
  }
#pragma unroll
  for (int j = 0; j < PER; j++) y[(int64_t)row * K + lane + 64 * j] = acc[j];
}

template <int K>
__global__ void __launch_bounds__(kSgdBlock)
k_sgd_apply(const float* __restrict__ y, const uint32_t* __restrict__ bits, float* __restrict__ vp, int n, float lambda,
            float step) {
  const int64_t i = (int64_t)blockIdx.x * kSgdBlock + threadIdx.x;
  if (i >= (int64_t)n * K) return;
  const int v = (int)(i / K), c = (int)(i % K);
  if (!((bits[v >> 5] >> (v & 31)) & 1u)) return;
  float* p = vp + (int64_t)v * (K + 1) + c;
  *p += step * (-lambda * *p + y[i]);  // src/SGD.cpp:111-115
}
__global__ void __launch_bounds__(kSgdBlock)
k_rmse_apply(const float* __restrict__ y1, const uint32_t* __restrict__ bits, float* __restrict__ vp, int n, int stride) {
  const int v = blockIdx.x * kSgdBlock + threadIdx.x;
  if (v < n && ((bits[v >> 5] >> (v & 31)) & 1u)) vp[(int64_t)v * stride + stride - 1] = y1[v];  // sqerr field
}

int g_sgd_mfma = 0;  // gm_set_option("sgd_mfma", 1): the dot products of K = 128 fp32 SGD on the matrix cores (k_sgd_multiply_mfma)

// fixed iteration count; sharded graphs exchange x once per iteration
template <int K>
int run_sgd_wide(gm_graph_t* g, float* d_latent, float lambda, float step, int iterations, int* iters_done, hipStream_t s) {
  const gm_graph_desc_t& d = g->desc;
  const int n = d.row_hi - d.row_lo;
  void *px = nullptr, *py = nullptr;
  int rc;
  if ((rc = gm_graph_workspace(g, 1, (size_t)d.ndevice * K * 4 + 64, &px))) return rc;
  if ((rc = gm_graph_workspace(g, 3, (size_t)n * K * 4 + 64, &py))) return rc;
  const int wgrid = (n + kSgdMulBlock / 64 - 1) / (kSgdMulBlock / 64);
  const int egrid = (int)(((int64_t)n * K + kSgdBlock - 1) / kSgdBlock);
  gm_run_stats_t st;
  memset(&st, 0, sizeof(st));
  hipEvent_t ev0, ev1;
  GM_TRY_HIP(hipEventCreate(&ev0));
  GM_TRY_HIP(hipEventCreate(&ev1));
  GM_TRY_HIP(hipEventRecord(ev0, s));
  for (int it = 0; it < iterations; it++) {
    hipLaunchKernelGGL((k_sgd_send<K>), dim3(egrid), dim3(kSgdBlock), 0, s, (const float*)d_latent,
                       (float*)px + (size_t)d.row_lo * K, n);
    if ((rc = sgd_exchange(g, (float*)px, K, s))) return rc;
    bool mfma = false;
    if constexpr (K == 128) {
      if (g_sgd_mfma) {
        hipLaunchKernelGGL((k_sgd_multiply_mfma<K>), dim3(n), dim3(64), 0, s, g->out.view, (const float*)px, (const float*)d_latent, (float*)py,
                           (const uint32_t*)nullptr, 0);
        hipLaunchKernelGGL((k_sgd_multiply_mfma<K>), dim3(n), dim3(64), 0, s, g->in.view, (const float*)px, (const float*)d_latent, (float*)py,
                           (const uint32_t*)g->out.rowbits, 1);
        mfma = true;
      }
    }
    if (!mfma) {
      hipLaunchKernelGGL((k_sgd_multiply<K, 0>), dim3(wgrid), dim3(kSgdMulBlock), 0, s, g->out.view, (const float*)px,
                         (const float*)d_latent, (float*)py, (const uint32_t*)nullptr, 0);
      hipLaunchKernelGGL((k_sgd_multiply<K, 0>), dim3(wgrid), dim3(kSgdMulBlock), 0, s, g->in.view, (const float*)px,
                         (const float*)d_latent, (float*)py, (const uint32_t*)g->out.rowbits, 1);
    }
    hipLaunchKernelGGL((k_sgd_apply<K>), dim3(egrid), dim3(kSgdBlock), 0, s, (const float*)py,
                       (const uint32_t*)g->rowbits_all, d_latent, n, lambda, step);
  }
  GM_TRY_HIP(hipEventRecord(ev1, s));
  GM_TRY_HIP(hipEventSynchronize(ev1));
  GM_TRY_HIP(hipEventElapsedTime(&st.total_ms, ev0, ev1));
  (void)hipEventDestroy(ev0);
  (void)hipEventDestroy(ev1);
  st.iterations = iterations;
  g->stats = st;
  if (iters_done) *iters_done = iterations;
  return GM_OK;
}
// ---- SGD on a bipartite ratings graph split by USERS: only the item side travels ----------------------------------
// Row sharding makes a shard's item rows read practically every user vector, so gm_run_sgd's exchange moves all of x
// (config 5: 11 M x 512 B = 5.6 GB per GPU and iteration).  The reference partitions 2-D and moves the smaller operand
// (include/GMDP/multinode/spmspv3.h:74-170).  Here rank k holds the edges of a contiguous NATIVE range of users (ranges
// ascending with the rank) as an ordinary single-GPU graph over all vertices, and every rank keeps a copy of the items:
//   * a user's fold (IN pass: rows = sources) sees all of its items locally;
//   * an item's fold over its users in ascending native order is the concatenation of the ranks' segments in rank
//     order, so its running K-vector travels rank 0 -> 1 -> ... -> N-1, each rank continuing the ordered fold where
//     the previous one stopped (k_sgd_multiply with the received values as y and their presence flags) -- in `blocks`
//     blocks of items, block b being on rank k at step k + b (a systolic ring: every rank busy after N - 1 steps);
//   * rank N-1 ends up with every item's complete sum and broadcasts them; every rank applies them to its copy of the
//     items (the same arithmetic everywhere, so the copies stay identical) and its own users' sums to its users.
// Per rank and iteration: nitems * K * 4 bytes received along the ring + the same again from the broadcast
// (config 5: 2 x 512 MB instead of 5.6 GB).  The bits are those of one GPU.
__global__ void __launch_bounds__(kSgdBlock)
k_rows_pack(const float* __restrict__ y, const uint32_t* __restrict__ bits_a, const uint32_t* __restrict__ bits_b, const int32_t* __restrict__ rows,
            int n, int K, float* __restrict__ out, int32_t* __restrict__ flags) {  // out[i][:] = y[rows[i]][:], flags[i] = presence
  const int64_t i = (int64_t)blockIdx.x * kSgdBlock + threadIdx.x;
  if (i >= (int64_t)n * K) return;
  const int r = (int)(i / K), c = (int)(i % K);
  const int row = rows[r];
  out[i] = y[(int64_t)row * K + c];
  if (c == 0) {
    uint32_t b = (bits_a[row >> 5] >> (row & 31)) & 1u;
    if (bits_b) b |= (bits_b[row >> 5] >> (row & 31)) & 1u;
    flags[r] = (int32_t)b;
  }
}
__global__ void __launch_bounds__(kSgdBlock)
k_rows_unpack(const float* __restrict__ in, const int32_t* __restrict__ flags, const int32_t* __restrict__ rows, int n, int K,
              float* __restrict__ y, uint32_t* __restrict__ bits) {  // y[rows[i]][:] = in[i][:], presence bit set from flags
  const int64_t i = (int64_t)blockIdx.x * kSgdBlock + threadIdx.x;
  if (i >= (int64_t)n * K) return;
  const int r = (int)(i / K), c = (int)(i % K);
  const int row = rows[r];
  y[(int64_t)row * K + c] = in[i];
  if (c == 0 && flags[r]) atomicOr(&bits[row >> 5], 1u << (row & 31));
}

__global__ void __launch_bounds__(kSgdBlock) k_rows_in_range(const int32_t* __restrict__ rows, int nlist, int n, unsigned int* __restrict__ bad) {
  const int i = blockIdx.x * kSgdBlock + threadIdx.x;
  if (i < nlist && (rows[i] < 0 || rows[i] >= n)) atomicAdd(bad, 1u);
}

template <int K>
int run_sgd_bipartite(gm_graph_t* g, float* d_latent, const int32_t* d_item_rows, int nitems, int blocks, float lambda, float step,
                      int iterations, int* iters_done, hipStream_t s) {
  const gm_graph_desc_t& d = g->desc;
  const int n = d.row_hi - d.row_lo;
  int rank = 0, nranks = 1;
  if (!dist_world(&rank, &nranks)) { rank = 0; nranks = 1; }
  if (blocks < 1) blocks = 1;
  if (blocks > nitems) blocks = nitems > 0 ? nitems : 1;
  void *px = nullptr, *py = nullptr, *pk = nullptr, *pb = nullptr;
  int rc;
  const size_t nw = (size_t)(n + 31) / 32 + 2;
  if ((rc = gm_graph_workspace(g, 1, (size_t)d.ndevice * K * 4 + 64, &px))) return rc;
  if ((rc = gm_graph_workspace(g, 3, (size_t)n * K * 4 + 64, &py))) return rc;
  if ((rc = gm_graph_workspace(g, 6, (size_t)nitems * (K + 1) * 4 + 256, &pk))) return rc;  // packed item sums + their flags
  if ((rc = gm_graph_workspace(g, 4, nw * 4 * 2 + 64, &pb))) return rc;                       // presence bits: received / applied (+ 64 bytes of scratch)
  float* pack = (float*)pk;
  int32_t* flags = (int32_t*)(pack + (size_t)nitems * K);
  uint32_t* got = (uint32_t*)pb;          // item rows that carry a value from the earlier ranks
  uint32_t* app = got + nw;               // rows apply() visits: this rank's users with ratings + items with any rating
  const int wgrid = (n + kSgdMulBlock / 64 - 1) / (kSgdMulBlock / 64);
  const int egrid = (int)(((int64_t)n * K + kSgdBlock - 1) / kSgdBlock);
  auto grid_items = [&](int cnt) { return (int)(((int64_t)cnt * K + kSgdBlock - 1) / kSgdBlock); };
  auto block_lo = [&](int b) { return (int)((int64_t)nitems * b / blocks); };
  gm_run_stats_t st;
  memset(&st, 0, sizeof(st));
  // the item rows must be rows of this graph, and the ranks must agree on the item count and the number of blocks: a rank that
  // left the ring early (or walked it with other block boundaries) would leave the others waiting in their next collective
  {
    unsigned int* d_chk = (unsigned int*)((char*)pb + nw * 4 * 2);  // (the 64 bytes behind the two bit vectors)
    GM_TRY_HIP(hipMemsetAsync(d_chk, 0, 16, s));
    hipLaunchKernelGGL(k_rows_in_range, dim3((nitems + kSgdBlock - 1) / kSgdBlock), dim3(kSgdBlock), 0, s, d_item_rows, nitems, n, d_chk);
    unsigned int bad = 0;
    GM_TRY_HIP(hipMemcpyAsync(&bad, d_chk, 4, hipMemcpyDeviceToHost, s));
    GM_TRY_HIP(hipStreamSynchronize(s));
    uint32_t agree[4] = {bad ? 1u : 0u, (uint32_t)nitems, (uint32_t)blocks, (uint32_t)iterations};
    if (nranks > 1) {
      // sum over the ranks: [0] = ranks with a bad row; [1..3] must be nranks x this rank's own value
      uint32_t* d_ag = (uint32_t*)d_chk + 8;
      GM_TRY_HIP(hipMemcpyAsync(d_ag, agree, 16, hipMemcpyHostToDevice, s));
      if ((rc = dist_all_reduce_sum_u32(d_ag, 4, s))) return rc;
      uint32_t tot[4];
      GM_TRY_HIP(hipMemcpyAsync(tot, d_ag, 16, hipMemcpyDeviceToHost, s));
      GM_TRY_HIP(hipStreamSynchronize(s));
      if (tot[1] != (uint32_t)nranks * agree[1] || tot[2] != (uint32_t)nranks * agree[2] || tot[3] != (uint32_t)nranks * agree[3]) {
        set_error("gm_run_sgd_bipartite: the ranks disagree on nitems / blocks / iterations (this rank: %d / %d / %d)", nitems, blocks, iterations);
        return GM_ERR_INVALID;
      }
      bad = tot[0];
    }
    if (bad) { set_error("gm_run_sgd_bipartite: d_item_rows holds a row outside [0, %d) (on this or another rank)", n); return GM_ERR_INVALID; }
  }
  struct Events {  // (destroyed on every exit path)
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  GM_TRY_HIP(hipEventCreate(&ev.a));
  GM_TRY_HIP(hipEventCreate(&ev.b));
  hipEvent_t& ev0 = ev.a;
  hipEvent_t& ev1 = ev.b;
  GM_TRY_HIP(hipEventRecord(ev0, s));
  unsigned long long moved = 0;
  for (int it = 0; it < iterations; it++) {
    hipLaunchKernelGGL((k_sgd_send<K>), dim3(egrid), dim3(kSgdBlock), 0, s, (const float*)d_latent, (float*)px, n);
    // users: rows of the by-source adjacency, all of their items are here
    hipLaunchKernelGGL((k_sgd_multiply<K, 0>), dim3(wgrid), dim3(kSgdMulBlock), 0, s, g->in.view, (const float*)px,
                       (const float*)d_latent, (float*)py, (const uint32_t*)nullptr, 0);
    // items: the ring.  Step t: this rank works on block t - rank.
    GM_TRY_HIP(hipMemsetAsync(got, 0, nw * 4, s));
    for (int t = 0; t < nranks + blocks - 1; t++) {
      const int b = t - rank;
      const bool mine = b >= 0 && b < blocks;
      const int lo = mine ? block_lo(b) : 0, cnt = mine ? block_lo(b + 1) - lo : 0;
      // block b arrives from the previous rank (which finished it in step t - 1) ...
      const size_t bytes_f = (size_t)cnt * K * 4, bytes_i = (size_t)cnt * 4;
      // ... while the block this rank finished in step t - 1 (block t - 1 - rank) moves on to the next rank
      const int bs = t - 1 - rank;
      const bool sending = bs >= 0 && bs < blocks && rank + 1 < nranks;
      const int slo = sending ? block_lo(bs) : 0, scnt = sending ? block_lo(bs + 1) - slo : 0;
      if (nranks > 1) {
        if ((rc = dist_ring_step(pack + (size_t)slo * K, (size_t)scnt * K * 4, pack + (size_t)lo * K, rank > 0 ? bytes_f : 0, s))) return rc;
        if ((rc = dist_ring_step(flags + slo, (size_t)scnt * 4, flags + lo, rank > 0 ? bytes_i : 0, s))) return rc;
        moved += (rank > 0 ? bytes_f + bytes_i : 0);
      }
      if (!mine || cnt == 0) continue;
      if (rank > 0)
        hipLaunchKernelGGL(k_rows_unpack, dim3(grid_items(cnt)), dim3(kSgdBlock), 0, s, (const float*)(pack + (size_t)lo * K),
                           (const int32_t*)(flags + lo), d_item_rows + lo, cnt, K, (float*)py, got);
      hipLaunchKernelGGL((k_sgd_multiply<K, 0>), dim3((cnt + kSgdMulBlock / 64 - 1) / (kSgdMulBlock / 64)), dim3(kSgdMulBlock), 0, s,
                         g->out.view, (const float*)px, (const float*)d_latent, (float*)py, (const uint32_t*)got, rank > 0 ? 1 : 0,
                         d_item_rows + lo, cnt);
      // the block as it leaves this rank (sent in the next step; on the last rank: the final sums)
      hipLaunchKernelGGL(k_rows_pack, dim3(grid_items(cnt)), dim3(kSgdBlock), 0, s, (const float*)py, (const uint32_t*)got,
                         (const uint32_t*)g->out.rowbits, d_item_rows + lo, cnt, K, pack + (size_t)lo * K, flags + lo);
    }
    if (nranks > 1) {
      // (the last block left rank N-2 in the final step above; rank N-1 now holds every item's complete sum)
      if ((rc = dist_broadcast(pack, (size_t)nitems * (K + 1) * 4, nranks - 1, s))) return rc;
      if (rank + 1 < nranks) moved += (unsigned long long)nitems * (K + 1) * 4;
    }
    // apply: this rank's users (rows with an out-edge here) and every item that received a rating anywhere
    GM_TRY_HIP(hipMemcpyAsync(app, g->in.rowbits, nw * 4, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_rows_unpack, dim3(grid_items(nitems)), dim3(kSgdBlock), 0, s, (const float*)pack, (const int32_t*)flags,
                       d_item_rows, nitems, K, (float*)py, app);
    hipLaunchKernelGGL((k_sgd_apply<K>), dim3(egrid), dim3(kSgdBlock), 0, s, (const float*)py, (const uint32_t*)app, d_latent, n,
                       lambda, step);
  }
  GM_TRY_HIP(hipEventRecord(ev1, s));
  GM_TRY_HIP(hipEventSynchronize(ev1));
  GM_TRY_HIP(hipEventElapsedTime(&st.total_ms, ev0, ev1));
  st.iterations = iterations;
  g->stats = st;
  g->note_val[1] = (int64_t)(iterations > 0 ? moved / (unsigned long long)iterations : 0ull);  // bytes received per iteration
  g->note_set[1] = 1;
  if (iters_done) *iters_done = iterations;
  return GM_OK;
}

template <int K>
int run_rmse_wide(gm_graph_t* g, float* d_latent, hipStream_t s) {
  const gm_graph_desc_t& d = g->desc;
  const int n = d.row_hi - d.row_lo;
  void *px = nullptr, *py = nullptr;
  int rc;
  if ((rc = gm_graph_workspace(g, 1, (size_t)d.ndevice * K * 4 + 64, &px))) return rc;
  if ((rc = gm_graph_workspace(g, 3, (size_t)n * K * 4 + 64, &py))) return rc;
  const int wgrid = (n + kSgdMulBlock / 64 - 1) / (kSgdMulBlock / 64);
  const int egrid = (int)(((int64_t)n * K + kSgdBlock - 1) / kSgdBlock);
  hipLaunchKernelGGL((k_sgd_send<K>), dim3(egrid), dim3(kSgdBlock), 0, s, (const float*)d_latent,
                     (float*)px + (size_t)d.row_lo * K, n);
  if ((rc = sgd_exchange(g, (float*)px, K, s))) return rc;
  hipLaunchKernelGGL((k_sgd_multiply<K, 1>), dim3(wgrid), dim3(kSgdMulBlock), 0, s, g->in.view, (const float*)px,
                     (const float*)d_latent, (float*)py, (const uint32_t*)nullptr, 0);
  hipLaunchKernelGGL(k_rmse_apply, dim3((n + kSgdBlock - 1) / kSgdBlock), dim3(kSgdBlock), 0, s, (const float*)py,
                     (const uint32_t*)g->in.rowbits, d_latent, n, K + 1);
  GM_TRY_HIP(hipStreamSynchronize(s));
  return GM_OK;
}
static bool wide_path_ok(const gm_graph_t* g) {
  const bool whole = g && g->desc.row_lo == 0 && g->desc.row_hi == g->desc.ndevice;
  return g && g->out.present && g->in.present && g->desc.val_bytes == 4 && g->out.vals && g->in.vals &&
         (whole || g->xfn != nullptr) && !g_force_ordered;
}

// run a program on caller-provided device state through the common engine
template <class P, class V>
int run_fixed(P& prog, gm_graph_t* g, V* d_vp, uint32_t* d_active, int iterations, int* iters_done, hipStream_t s) {
  typedef typename GraphMat::detail::types_of<P>::type PT;
  typedef typename PT::msg T;
  typedef typename PT::red U;
  if (!g || !d_vp) { set_error("gm_run_*: null graph or vertex state"); return GM_ERR_INVALID; }
  const gm_graph_desc_t& d = g->desc;
  const int rows = d.row_hi - d.row_lo;
  const auto order = prog.getOrder();
  if ((order != GraphMat::IN_EDGES && !g->out.present) || (order != GraphMat::OUT_EDGES && !g->in.present)) {
    set_error("gm_run_*: graph was built without the adjacency direction this program needs");
    return GM_ERR_INVALID;
  }
  void *p0, *p1, *p2, *p3, *p5;
  int rc;
  if ((rc = gm_graph_workspace(g, 1, (size_t)d.ndevice * sizeof(T) + 16, &p0))) return rc;
  if ((rc = gm_graph_workspace(g, 2, ((size_t)(d.ndevice + 31) / 32 + 2) * 4, &p1))) return rc;
  if ((rc = gm_graph_workspace(g, 3, (size_t)rows * sizeof(U) + 16, &p2))) return rc;
  if ((rc = gm_graph_workspace(g, 4, ((size_t)(rows + 31) / 32 + 2) * 4, &p3))) return rc;
  if (!d_active) {
    const size_t nw = (size_t)(rows + 31) / 32;
    if ((rc = gm_graph_workspace(g, 5, (nw + 2) * 4, &p5))) return rc;
    d_active = (uint32_t*)p5;
    hipLaunchKernelGGL(GraphMat::dev::k_fill_u32, dim3(GraphMat::detail::grid_for((int64_t)nw)),
                       dim3(GraphMat::dev::kBlock), 0, s, d_active, (int64_t)nw, 0xffffffffu);
    if (rows & 31) {
      uint32_t tail = (1u << (rows & 31)) - 1u;
      GM_TRY_HIP(hipMemcpyAsync(d_active + nw - 1, &tail, 4, hipMemcpyHostToDevice, s));
      GM_TRY_HIP(hipStreamSynchronize(s));
    }
  }
  int it = GraphMat::detail::run_on_device<P, T, U, V, int>(&prog, g, order, prog.getActivity(),
                                                             prog.getProcessMessageRequiresVertexprop(), d_vp, d_active,
                                                             (T*)p0, (uint32_t*)p1, (U*)p2, (uint32_t*)p3, iterations, s);
  if (iters_done) *iters_done = it;
  return GM_OK;
}

// The HIP runtime resolves a code object's kernel table at the first launch of one of its
// kernels (milliseconds for this translation unit's template instantiations): do it when the
// first graph is created, not inside the first run.
void warm_program_kernels(void* d_scratch256) {
  static bool done = false;
  if (done || !d_scratch256) return;
  done = true;
  hipLaunchKernelGGL(GraphMat::dev::k_fill_u32, dim3(1), dim3(GraphMat::dev::kBlock), 0, 0, (uint32_t*)d_scratch256, (int64_t)64, 0u);
  (void)hipDeviceSynchronize();
}

}  // namespace gm

extern "C" {

// ---- engine options (graphmat_hip.h: gm_engine_options_t): process defaults + per-graph overrides --------------------
namespace gm {
struct EngineKey { const char* name; int index; int def, lo, hi; };  // index = position of the field in gm_engine_options_t
static const EngineKey kEngineKeys[] = {
  {"debug_flags", 0, 0, 0, 0x7fffffff},
  {"wave16_form", 1, 2, 0, 31},
  {"rowwave_form", 2, 4, 0, 31},
  {"persist_per_cu", 3, 0, 0, 8},
  {"giant_maps", 4, 1, 0, 1},
  {"ordered_giant_two_pass", 5, 2, 0, 2},
  {"fuse_apply_send", 6, 1, 0, 1},
  {"untiled_pass_plain", 7, 1, 0, 1},
  {"last_rows_lanes", 8, 8, 8, 16},
  {"push_edge_permille", 9, 50, 0, 1000},
  {"bits_step_edges", 10, 2 << 20, 0, 0x7fffffff},
  {"sparse_step_edges", 11, 1 << 20, 0, 0x7fffffff},
  {"iteration_trace", 12, 0, 0, 1},
  {"ablate_cold_from", 13, 0, 0, 0x7fffffff},
  {"ablate_cold_short", 14, 0, 0, 0x7fffffff},
  {"two_stage_head_permille", 15, 900, 100, 990},
  {"giant_stream", 16, 1, 0, 2},
  {"sweep_form", 17, 0, 0, 511},
  {"blocked_form", 18, 2, 0, 31},
  {"guided_pull", 19, 1, 0, 2},
};
static_assert(offsetof(gm_engine_options_t, guided_pull) == 19 * sizeof(int32_t), "kEngineKeys follows the field order");
static bool engine_value_ok(const EngineKey& k, int v) {
  if (v < k.lo || v > k.hi) return false;
#ifndef GRAPHMAT_ABLATION
  // result-invalidating switches exist in -DGRAPHMAT_ABLATION builds only (build/ablation/libgraphmat_hip.so, tools/): the
  // product library rejects them -- the in-kernel ablation bits of debug_flags (1, 2, 4, 8, 16384) and the cold-column ablation
  if (!strcmp(k.name, "debug_flags") && (v & (15 | 16384)) != 0) return false;
  if (!strncmp(k.name, "ablate_", 7)) return false;
#endif
  if (!strcmp(k.name, "wave16_form")) return (v & 15) <= 5;
  if (!strcmp(k.name, "rowwave_form")) return (v & 15) <= 4;
  if (!strcmp(k.name, "last_rows_lanes")) return v == 8 || v == 16;
  return true;
}
static gm_engine_options_t engine_documented_defaults() {
  gm_engine_options_t o;
  memset(&o, 0, sizeof(o));
  for (const EngineKey& k : kEngineKeys) ((int32_t*)&o)[k.index] = k.def;
  const char* tr = getenv("GRAPHMAT_ITERATION_TRACE");
  if (tr && tr[0] == '1') o.iteration_trace = 1;
  // GRAPHMAT_OPTIONS="key=value,key=value": engine options for applications that cannot call gm_set_option themselves (the
  // reference's unchanged sources); unknown keys and values out of range are reported and ignored
  if (const char* env = getenv("GRAPHMAT_OPTIONS")) {
    std::string all(env);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find(',', pos);
      if (end == std::string::npos) end = all.size();
      const std::string kv = all.substr(pos, end - pos);
      pos = end + 1;
      const size_t eq = kv.find('=');
      bool ok = false;
      if (eq != std::string::npos) {
        const std::string key = kv.substr(0, eq);
        const int value = atoi(kv.c_str() + eq + 1);
        for (const EngineKey& k : kEngineKeys)
          if (key == k.name && engine_value_ok(k, value)) { ((int32_t*)&o)[k.index] = value; ok = true; }
      }
      if (!ok && !kv.empty()) fprintf(stderr, "GraphMat(HIP): GRAPHMAT_OPTIONS: ignoring '%s'\n", kv.c_str());
    }
  }
  return o;
}
static gm_engine_options_t& engine_defaults() {
  static gm_engine_options_t o = engine_documented_defaults();
  return o;
}
static const EngineKey* engine_key(const char* key) {
  if (!key) return nullptr;
  for (const EngineKey& k : kEngineKeys)
    if (!strcmp(k.name, key)) return &k;
  return nullptr;
}
}  // namespace gm

int gm_graph_engine_options(const gm_graph_t* g, gm_engine_options_t* out) {
  if (!out) { gm::set_error("gm_graph_engine_options: null argument"); return GM_ERR_INVALID; }
  *out = gm::engine_defaults();
  if (g)
    for (const gm::EngineKey& k : gm::kEngineKeys)
      if ((g->opt_set >> k.index) & 1u) ((int32_t*)out)[k.index] = ((const int32_t*)&g->opt)[k.index];
  return GM_OK;
}
int gm_graph_set_option(gm_graph_t* g, const char* key, int value) {
  const gm::EngineKey* k = gm::engine_key(key);
  if (!g || !k || !gm::engine_value_ok(*k, value)) { gm::set_error("gm_graph_set_option: unknown engine option or value out of range"); return GM_ERR_INVALID; }
  ((int32_t*)&g->opt)[k->index] = value;
  g->opt_set |= 1u << k->index;
  return GM_OK;
}
int gm_reset_options(void) {
  gm::engine_defaults() = gm::engine_documented_defaults();
  gm::g_force_ordered = 0;
  gm::g_sgd_mfma = 0;
  gm::g_short_row = GM_SHORT_ROW;
  gm::g_giant_row = 0;
  gm::g_rank_cap = 0;
  gm::g_rank_by = 0;
  gm::g_tile_min_row = GM_TILE_MIN_ROW;
  gm::g_tile_balance = 1;
  gm::g_long_mid = 0;
  gm::g_own_wave_row = 4096;
  gm::g_sort_tile_lists = 1;
  gm::g_sweep_slices = 1;
  gm::g_blocked_rows = 0;
  gm::g_sweep_acc_limit = GM_SWEEP_ACC_ROWS;
  gm::g_sweep_long_limit = GM_SWEEP_LONG_SLOTS;
  gm::g_sweep_long_row = 0;
  gm::g_sweep_fold_share = 50;
  gm::g_sweep_waves = 16;
  gm::g_sweep_stream = 1;
  gm::g_sweep_stream_weight = 300;
  gm::g_sweep_border_factor = 4;
  gm::g_col_tiles = 0;
  return GM_OK;
}
int gm_set_option(const char* key, int value) {
  if (const gm::EngineKey* k = gm::engine_key(key)) {
    if (!gm::engine_value_ok(*k, value)) { gm::set_error("gm_set_option: value out of range"); return GM_ERR_INVALID; }
    ((int32_t*)&gm::engine_defaults())[k->index] = value;
    return GM_OK;
  }
  if (key && !strcmp(key, "force_ordered")) { gm::g_force_ordered = value; return GM_OK; }
  if (key && !strcmp(key, "sgd_mfma") && (value == 0 || value == 1)) { gm::g_sgd_mfma = value; return GM_OK; }
  if (key && !strcmp(key, "short_row") && value >= 1 && value <= GM_BLOCK_NNZ) { gm::g_short_row = value; return GM_OK; }
  if (key && !strcmp(key, "giant_row") && value >= 64) { gm::g_giant_row = value; return GM_OK; }
  if (key && !strcmp(key, "rank_cap") && value >= 0) { gm::g_rank_cap = value; return GM_OK; }
  if (key && !strcmp(key, "rank_by") && value >= 0 && value <= 2) { gm::g_rank_by = value; return GM_OK; }
  // (0 = every row is tiled; otherwise at least the short-row limit: the untiled row-block kernel takes every row up to that limit)
  if (key && !strcmp(key, "long_mid") && value >= 0) { gm::g_long_mid = value; return GM_OK; }
  if (key && !strcmp(key, "tile_balance") && value >= 0 && value <= 100400) { gm::g_tile_balance = value; return GM_OK; }
  if (key && !strcmp(key, "tile_min_row") && (value == 0 || value >= gm::g_short_row)) { gm::g_tile_min_row = value; return GM_OK; }
  if (key && !strcmp(key, "col_tiles") && value >= 0 && value <= GM_MAX_TILES) { gm::g_col_tiles = value; return GM_OK; }
  if (key && !strcmp(key, "own_wave_row") && value >= 0) { gm::g_own_wave_row = value; return GM_OK; }
  if (key && !strcmp(key, "sort_tile_lists") && (value == 0 || value == 1)) { gm::g_sort_tile_lists = value; return GM_OK; }
  // (0: no slices, no sweep; 1: automatic slice count; 8 .. GM_MAX_SLICES: about that many slices)
  if (key && !strcmp(key, "sweep_slices") && (value == 0 || value == 1 || (value >= 8 && value <= GM_MAX_SLICES))) { gm::g_sweep_slices = value; return GM_OK; }
  if (key && !strcmp(key, "sweep_acc_rows") && value >= 1 && value <= GM_SWEEP_ACC_ROWS) { gm::g_sweep_acc_limit = value; return GM_OK; }
  if (key && !strcmp(key, "blocked_rows") && value >= -1 && value <= 1) { gm::g_blocked_rows = value; return GM_OK; }
  if (key && !strcmp(key, "sweep_long_row") && value >= 0 && value <= 8191) { gm::g_sweep_long_row = value; return GM_OK; }
  if (key && !strcmp(key, "sweep_border_factor") && value >= 1 && value <= 64) { gm::g_sweep_border_factor = value; return GM_OK; }
  if (key && !strcmp(key, "sweep_fold_share") && value >= 0 && value <= 100) { gm::g_sweep_fold_share = value; return GM_OK; }
  if (key && !strcmp(key, "sweep_stream") && (value == 0 || value == 1)) { gm::g_sweep_stream = value; return GM_OK; }
  if (key && !strcmp(key, "sweep_stream_weight") && value >= 50 && value <= 2000) { gm::g_sweep_stream_weight = value; return GM_OK; }
  if (key && !strcmp(key, "sweep_waves") && (value == 16 || value == 12 || value == 8)) { gm::g_sweep_waves = value; return GM_OK; }
  if (key && !strcmp(key, "sweep_long_slots") && value >= 1 && value <= GM_SWEEP_LONG_SLOTS) { gm::g_sweep_long_limit = value; return GM_OK; }
  gm::set_error("gm_set_option: unknown option");
  return GM_ERR_INVALID;
}

#ifdef GRAPHMAT_ABLATION
// ablation builds only (not in the header): the per-wave time stamps of the persistent kernels' last launch
extern "C" int gm_abl_wave_times(unsigned long long* out, size_t bytes) {
  const size_t n = sizeof(GraphMat::dev::g_abl_wave_times);
  GM_TRY_HIP(hipDeviceSynchronize());
  GM_TRY_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(GraphMat::dev::g_abl_wave_times), bytes < n ? bytes : n));
  return GM_OK;
}
#endif

int gm_debug_counters(int64_t out[4]) {
  unsigned long long h[4] = {0, 0, 0, 0};
  GM_TRY_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(GraphMat::dev::g_longrow_counters), sizeof(h)));
  for (int i = 0; i < 4; i++) out[i] = (int64_t)h[i];
  unsigned long long z[4] = {0, 0, 0, 0};
  GM_TRY_HIP(hipMemcpyToSymbol(HIP_SYMBOL(GraphMat::dev::g_longrow_counters), z, sizeof(z)));
  return GM_OK;
}

int gm_run_degree(gm_graph_t* g, gm_pr_t* d_vp, int iterations, int* iters_done, gm_stream_t stream) {
  gm::DegreeP p;
  return gm::run_fixed(p, g, (gm::PRv*)d_vp, (uint32_t*)nullptr, iterations, iters_done, (hipStream_t)stream);
}

int gm_run_pagerank(gm_graph_t* g, gm_pr_t* d_vp, float alpha, int iterations, int* iters_done, gm_stream_t stream) {
  if (gm::g_force_ordered) {
    gm::PageRankP<1> p(alpha);
    return gm::run_fixed(p, g, (gm::PRv*)d_vp, (uint32_t*)nullptr, iterations, iters_done, (hipStream_t)stream);
  }
  gm::PageRankP<0> p(alpha);
  return gm::run_fixed(p, g, (gm::PRv*)d_vp, (uint32_t*)nullptr, iterations, iters_done, (hipStream_t)stream);
}

int gm_run_bfs(gm_graph_t* g, gm_bfs_t* d_vp, uint32_t* d_active, int iterations, int* iters_done, gm_stream_t stream) {
  gm::BfsP p;
  return gm::run_fixed(p, g, (gm::BFSv*)d_vp, d_active, iterations, iters_done, (hipStream_t)stream);
}

int gm_run_sssp(gm_graph_t* g, uint32_t* d_dist, uint32_t* d_active, int iterations, int* iters_done,
                gm_stream_t stream) {
  gm::SsspP p;
  return gm::run_fixed(p, g, (gm::SSSPv*)d_dist, d_active, iterations, iters_done, (hipStream_t)stream);
}

int gm_run_sgd(gm_graph_t* g, void* d_latent, int K, int real_bytes, double lambda, double step, int iterations,
               int* iters_done, gm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (K == 20 && real_bytes == 8) {
    gm::SgdP<double, 20> p(lambda, step);
    return gm::run_fixed(p, g, (gm::Latent<double, 20>*)d_latent, (uint32_t*)nullptr, iterations, iters_done, s);
  }
  if (K == 20 && real_bytes == 4) {
    gm::SgdP<float, 20> p((float)lambda, (float)step);
    return gm::run_fixed(p, g, (gm::Latent<float, 20>*)d_latent, (uint32_t*)nullptr, iterations, iters_done, s);
  }
  if (K == 128 && real_bytes == 4 && iterations > 0 && gm::wide_path_ok(g) && d_latent)
    return gm::run_sgd_wide<128>(g, (float*)d_latent, (float)lambda, (float)step, iterations, iters_done, s);
  if (K == 128 && real_bytes == 4) {
    gm::SgdP<float, 128> p((float)lambda, (float)step);
    return gm::run_fixed(p, g, (gm::Latent<float, 128>*)d_latent, (uint32_t*)nullptr, iterations, iters_done, s);
  }
  gm::set_error("gm_run_sgd: (K=%d, real_bytes=%d) is not in the fixed menu", K, real_bytes);
  return GM_ERR_UNSUPPORTED;
}

int gm_run_sgd_bipartite(gm_graph_t* g, void* d_latent, int K, int real_bytes, const int32_t* d_item_rows, int nitems, int blocks,
                         double lambda, double step, int iterations, int* iters_done, gm_stream_t stream) {
  if (!g || !d_latent || !d_item_rows || nitems < 0 || iterations < 1) { gm::set_error("gm_run_sgd_bipartite: invalid argument"); return GM_ERR_INVALID; }
  if (K != 128 || real_bytes != 4) { gm::set_error("gm_run_sgd_bipartite: only K=128 fp32 (the dedicated kernels)"); return GM_ERR_UNSUPPORTED; }
  const gm_graph_desc_t& d = g->desc;
  if (d.row_lo != 0 || d.row_hi != d.ndevice || g->xfn != nullptr || !g->out.present || !g->in.present || d.val_bytes != 4 || !g->out.vals ||
      !g->in.vals) {
    gm::set_error("gm_run_sgd_bipartite: needs an unsharded graph of this rank's users' ratings, both directions, 4-byte values");
    return GM_ERR_INVALID;
  }
  return gm::run_sgd_bipartite<128>(g, (float*)d_latent, d_item_rows, nitems, blocks, (float)lambda, (float)step, iterations, iters_done,
                                    (hipStream_t)stream);
}

int gm_run_rmse(gm_graph_t* g, void* d_latent, int K, int real_bytes, gm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (K == 20 && real_bytes == 8) {
    gm::RmseP<double, 20> p;
    return gm::run_fixed(p, g, (gm::Latent<double, 20>*)d_latent, (uint32_t*)nullptr, 1, nullptr, s);
  }
  if (K == 20 && real_bytes == 4) {
    gm::RmseP<float, 20> p;
    return gm::run_fixed(p, g, (gm::Latent<float, 20>*)d_latent, (uint32_t*)nullptr, 1, nullptr, s);
  }
  if (K == 128 && real_bytes == 4 && gm::wide_path_ok(g) && d_latent) return gm::run_rmse_wide<128>(g, (float*)d_latent, s);
  if (K == 128 && real_bytes == 4) {
    gm::RmseP<float, 128> p;
    return gm::run_fixed(p, g, (gm::Latent<float, 128>*)d_latent, (uint32_t*)nullptr, 1, nullptr, s);
  }
  gm::set_error("gm_run_rmse: (K=%d, real_bytes=%d) is not in the fixed menu", K, real_bytes);
  return GM_ERR_UNSUPPORTED;
}

}  // extern "C"
