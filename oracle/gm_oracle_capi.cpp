// gm_oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY (see gm_oracle.hpp header).
//
// The vertex programs of the reference's example apps restated on top of the
// oracle driver, plus plain-C entry points so tests/ and bench.py can call them
// through ctypes.  Per-vertex arrays cross this API in ORIGINAL vertex order
// (index v-1 for the 1-based vertex id v of the .mtx file); the permutation to
// the reference's native order happens inside, as in include/Graph.h:312-364.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "gm_oracle.hpp"

using namespace gmo;

namespace {

// ---------------- PageRank: src/PageRank.cpp:34-112 ---------------------------
struct PR {
  float pagerank;
  int degree;
  PR() : pagerank(0.3), degree(0) {}  // :40-43 (0.3 narrowed to float)
};

struct DegreeProg {  // :53-79
  typedef int msg_t; typedef int red_t; typedef PR vp_t; typedef int edge_t;
  static const bool requires_vertexprop = false;
  EdgeDir order = IN_EDGES;
  Activity activity = ACTIVE_ONLY;
  bool send_message(const PR&, int& m) const { m = 1; return true; }
  void process_message(const int& m, const int, const PR&, int& r) const { r = m; }
  void reduce_function(int& a, const int& b) const { a += b; }
  void apply(const int& y, PR& vp) { vp.degree = y; }
  void do_every_iteration(int) {}
  // PR::operator!= (:44-46): |delta pagerank| > 1e-5, float subtract, compared in double
  static bool changed(PR& old, const PR& now) { return (double)fabsf(now.pagerank - old.pagerank) > 1e-5; }
};

struct PageRankProg {  // :81-112
  typedef float msg_t; typedef float red_t; typedef PR vp_t; typedef int edge_t;
  static const bool requires_vertexprop = false;
  EdgeDir order = OUT_EDGES;
  Activity activity = ALL_VERTICES;
  float alpha;
  explicit PageRankProg(float a) : alpha(a) {}
  bool send_message(const PR& vp, float& m) const {  // :100-107
    if (vp.degree == 0) m = 0.0; else m = vp.pagerank / (float)vp.degree;
    return true;
  }
  void process_message(const float& m, const int, const PR&, float& r) const { r = m; }
  void reduce_function(float& a, const float& b) const { a += b; }
  // :108-110: double arithmetic on float operands, narrowed on store.  Built
  // with -ffp-contract=off: mul then add, no fused multiply-add.
  void apply(const float& y, PR& vp) { vp.pagerank = alpha + (1.0 - alpha) * y; }
  void do_every_iteration(int) {}
  static bool changed(PR& old, const PR& now) { return (double)fabsf(now.pagerank - old.pagerank) > 1e-5; }
};

// ---------------- BFS: src/BFS.cpp:36-99 ------------------------------------------
const unsigned int MAX_DIST = std::numeric_limits<unsigned int>::max();
struct BFSD2 {
  unsigned int depth;
  unsigned long long parent;
  unsigned long long id;
  BFSD2() : depth(MAX_DIST), parent(-1), id(-1) {}
};
struct BfsProg {
  typedef unsigned long long msg_t; typedef unsigned long long red_t; typedef BFSD2 vp_t; typedef int edge_t;
  static const bool requires_vertexprop = false;
  EdgeDir order = OUT_EDGES;
  Activity activity = ACTIVE_ONLY;
  unsigned int current_depth = 1;
  bool send_message(const BFSD2& vp, msg_t& m) const { m = vp.id; return vp.depth == current_depth - 1; }
  void process_message(const msg_t& m, const int, const BFSD2&, red_t& r) const { r = m; }
  void reduce_function(red_t& a, const red_t& b) const { a = b; }  // :75-77 last writer wins
  void apply(const red_t& y, BFSD2& vp) {
    if (vp.depth == MAX_DIST) { vp.depth = current_depth; vp.parent = y; }
  }
  void do_every_iteration(int) { current_depth++; }
  static bool changed(BFSD2& old, const BFSD2& now) { return old.depth != now.depth; }
};

// ---------------- SSSP: src/SSSP.cpp:36-90 -----------------------------------------
struct SsspV { unsigned int distance; SsspV() : distance(MAX_DIST) {} };
struct SsspProg {
  typedef unsigned int msg_t; typedef unsigned int red_t; typedef SsspV vp_t; typedef int edge_t;
  static const bool requires_vertexprop = false;
  EdgeDir order = OUT_EDGES;
  Activity activity = ACTIVE_ONLY;
  bool send_message(const SsspV& vp, msg_t& m) const { m = vp.distance; return true; }
  void process_message(const msg_t& m, const int e, const SsspV&, red_t& r) const { r = m + e; }
  void reduce_function(red_t& a, const red_t& b) const { a = (a <= b) ? a : b; }
  void apply(const red_t& y, SsspV& vp) { vp.distance = std::min(vp.distance, y); }
  void do_every_iteration(int) {}
  static bool changed(SsspV& old, const SsspV& now) { return old.distance != now.distance; }
};

// ---------------- SGD / RMSE: src/SGD.cpp:36-156 -------------------------------------
// The reference is K=20, double.  Templated so the K=128 fp32 configuration of
// BASELINE.json has a checker of the same formula.
template <class R, int K>
struct Latent {
  R lv[K];
  R sqerr;
};
template <class R>
struct sgd_tol { static R abs_changed() { return (R)1e-7; } };

template <class R, int K>
struct SgdProg {  // :77-120
  typedef Latent<R, K> L;
  typedef L msg_t; typedef L red_t; typedef L vp_t; typedef int edge_t;
  static const bool requires_vertexprop = true;  // GraphProgram.h:56 default
  EdgeDir order = ALL_EDGES;
  Activity activity = ALL_VERTICES;
  R lambda, step;
  SgdProg(R l, R s) : lambda(l), step(s) {}
  bool send_message(const L& vp, L& m) const { m = vp; return true; }
  void process_message(const L& m, const int e, const L& vp, L& res) const {  // :93-104
    R estimate = 0;
    for (int i = 0; i < K; i++) estimate += m.lv[i] * vp.lv[i];
    R error = e - estimate;
    for (int i = 0; i < K; i++) res.lv[i] = m.lv[i] * error;
    res.sqerr = 0;  // the reference leaves this field indeterminate; it is never read
  }
  void reduce_function(L& v, const L& w) const { for (int i = 0; i < K; i++) v.lv[i] += w.lv[i]; }
  void apply(const L& y, L& vp) {  // :111-115
    for (int i = 0; i < K; i++) vp.lv[i] += step * (-lambda * vp.lv[i] + y.lv[i]);
  }
  void do_every_iteration(int) {}
  static bool changed(L& old, const L& now) {  // :48-56
    bool r = false;
    for (int i = 0; i < K; i++) if (std::fabs(now.lv[i] - old.lv[i]) > 1e-7) r = true;
    return r;
  }
};

template <class R, int K>
struct RmseProg {  // :122-156
  typedef Latent<R, K> L;
  typedef L msg_t; typedef R red_t; typedef L vp_t; typedef int edge_t;
  static const bool requires_vertexprop = true;
  EdgeDir order = IN_EDGES;
  Activity activity = ACTIVE_ONLY;
  bool send_message(const L& vp, L& m) const { m = vp; return true; }
  void process_message(const L& m, const int e, const L& vp, R& res) const {
    R est = 0;
    for (int i = 0; i < K; i++) est += m.lv[i] * vp.lv[i];
    R error = e - est;
    res = error * error;
  }
  void reduce_function(R& v, const R& w) const { v += w; }
  void apply(const R& y, L& vp) { vp.sqerr = y; }
  void do_every_iteration(int) {}
  static bool changed(L& old, const L& now) { return SgdProg<R, K>::changed(old, now); }
};

struct OGraph {
  Topo<int> topo;
  int ref_threads;
};

void copy_hist(const std::vector<int>& h, int* out, int cap) {
  if (!out) return;
  for (int i = 0; i < cap && i < (int)h.size(); i++) out[i] = h[i];
}

template <class R, int K>
int sgd_run(OGraph* og, double lambda, double step, int iterations, R* lv) {
  typedef Latent<R, K> L;
  Graph<L, int> g(&og->topo);
  for (int v = 1; v <= g.nvertices; v++) {
    L l;
    for (int j = 0; j < K; j++) l.lv[j] = lv[(size_t)(v - 1) * K + j];
    l.sqerr = 0;
    g.set_vertexproperty(v, l);
  }
  SgdProg<R, K> p((R)lambda, (R)step);
  g.set_all_active();
  int it = run_graph_program(p, g, iterations);
  for (int v = 1; v <= g.nvertices; v++) {
    L l = g.get_vertexproperty(v);
    for (int j = 0; j < K; j++) lv[(size_t)(v - 1) * K + j] = l.lv[j];
  }
  return it;
}

template <class R, int K>
double rmse_run(OGraph* og, const R* lv, R* sqerr_out) {
  typedef Latent<R, K> L;
  Graph<L, int> g(&og->topo);
  for (int v = 1; v <= g.nvertices; v++) {
    L l;
    for (int j = 0; j < K; j++) l.lv[j] = lv[(size_t)(v - 1) * K + j];
    l.sqerr = 0;
    g.set_vertexproperty(v, l);
  }
  RmseProg<R, K> p;
  g.set_all_active();
  run_graph_program(p, g, 1);
  R err = 0;  // src/SGD.cpp:189-190: applyReduceAllVertices(sum of sqerr), native order
  map_reduce(g.vp, &err, [](const L& v, R* o) { *o = v.sqerr; }, og->ref_threads);
  if (sqerr_out)
    for (int v = 1; v <= g.nvertices; v++) sqerr_out[v - 1] = g.get_vertexproperty(v).sqerr;
  return (double)err;
}

}  // namespace

extern "C" {

int gmo_vertex_to_native(int v1, int nparts, int len) { return vertex_to_native(v1, nparts, len); }
int gmo_native_to_vertex(int v1, int nparts, int len) { return native_to_vertex(v1, nparts, len); }
int gmo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
// STREAM-style triad a[i] = b[i] + 3 c[i] over three arrays of n doubles (first touched by the threads that walk them), `reps`
// passes: GB/s of the 24 n bytes a pass moves.  bench.py's cpu_baseline reports it next to the PageRank probe so that
// "more threads do not help" can be read against what the host's memory system delivers at that thread count.
static double gmo_now_s() {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
#endif
}
double gmo_stream_triad_gbs(long long n, int reps) {
  double* a = (double*)malloc((size_t)n * 8);
  double* b = (double*)malloc((size_t)n * 8);
  double* c = (double*)malloc((size_t)n * 8);
  if (!a || !b || !c) { free(a); free(b); free(c); return 0.0; }
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < n; i++) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
  double best = 0.0;
  for (int r = 0; r < reps; r++) {
    const double t0 = gmo_now_s();
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n; i++) a[i] = b[i] + 3.0 * c[i];
    const double dt = gmo_now_s() - t0;
    const double gbs = 24.0 * (double)n / dt * 1e-9;
    if (gbs > best) best = gbs;
  }
  volatile double sink = a[n / 2];
  (void)sink;
  free(a); free(b); free(c);
  return best;
}
void gmo_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

// ref_threads = the OMP_NUM_THREADS of the reference configuration being
// restated (it enters the id permutation and the partition count).
void* gmo_graph_create(int nv, long long ne, const int* src, const int* dst, const int* val, int ref_threads) {
  OGraph* g = new OGraph();
  g->ref_threads = ref_threads;
  g->topo.read_edgelist(nv, ne, src, dst, val, ref_threads);
  return g;
}
void gmo_graph_destroy(void* g) { delete (OGraph*)g; }

// Degree pass of src/PageRank.cpp:128-137: all active, one iteration, IN_EDGES
// (which yields OUT-degree).  degree_out[v-1].
int gmo_degree(void* gv, int* degree_out) {
  OGraph* og = (OGraph*)gv;
  Graph<PR, int> g(&og->topo);
  DegreeProg p;
  g.set_all_active();
  int it = run_graph_program(p, g, 1);
  for (int v = 1; v <= g.nvertices; v++) degree_out[v - 1] = g.get_vertexproperty(v).degree;
  return it;
}

// PageRank pass of src/PageRank.cpp:139-150.  pr is in/out; iterations<=0 means
// until convergence.  Returns iterations completed.
// seconds in send / multiply+reduce / apply since the last call with reset != 0 (gm_oracle.hpp: phase_seconds)
void gmo_phase_seconds(double out[3], int reset) {
  double* t = phase_seconds();
  for (int i = 0; i < 3; i++) { out[i] = t[i]; if (reset) t[i] = 0; }
}

int gmo_pagerank(void* gv, float alpha, int iterations, float* pr, const int* degree, int* changed_hist,
                 int hist_cap) {
  OGraph* og = (OGraph*)gv;
  Graph<PR, int> g(&og->topo);
  for (int v = 1; v <= g.nvertices; v++) {
    PR p;
    p.pagerank = pr[v - 1];
    p.degree = degree[v - 1];
    g.set_vertexproperty(v, p);
  }
  PageRankProg prog(alpha);
  g.set_all_active();
  std::vector<int> hist;
  int it = run_graph_program(prog, g, iterations, &hist);
  for (int v = 1; v <= g.nvertices; v++) pr[v - 1] = g.get_vertexproperty(v).pagerank;
  copy_hist(hist, changed_hist, hist_cap);
  return it;
}

// src/BFS.cpp:110-156: ids = 1..n, source depth 0 and active, until convergence.
// parent of the source / unreachable vertices stays (u64)-1.
int gmo_bfs(void* gv, int source, unsigned int* depth, unsigned long long* parent, int* changed_hist,
            int hist_cap) {
  OGraph* og = (OGraph*)gv;
  Graph<BFSD2, int> g(&og->topo);
  for (int v = 1; v <= g.nvertices; v++) {
    BFSD2 b = g.get_vertexproperty(v);
    b.id = v;
    g.set_vertexproperty(v, b);
  }
  BfsProg prog;
  g.set_all_inactive();
  BFSD2 s = g.get_vertexproperty(source);
  s.depth = 0;
  g.set_vertexproperty(source, s);
  g.set_active(source);
  std::vector<int> hist;
  int it = run_graph_program(prog, g, -1, &hist);
  for (int v = 1; v <= g.nvertices; v++) {
    BFSD2 b = g.get_vertexproperty(v);
    depth[v - 1] = b.depth;
    parent[v - 1] = b.parent;
  }
  copy_hist(hist, changed_hist, hist_cap);
  return it;
}

// src/SSSP.cpp:99-125
int gmo_sssp(void* gv, int source, unsigned int* dist) {
  OGraph* og = (OGraph*)gv;
  Graph<SsspV, int> g(&og->topo);
  SsspProg prog;
  g.set_all_inactive();
  SsspV z;
  z.distance = 0;
  g.set_vertexproperty(source, z);
  g.set_active(source);
  int it = run_graph_program(prog, g, -1);
  for (int v = 1; v <= g.nvertices; v++) dist[v - 1] = g.get_vertexproperty(v).distance;
  return it;
}

// src/SGD.cpp:163-224.  lv is [nv][K] row-major in original vertex order, in/out.
int gmo_sgd_f64(void* gv, int K, double lambda, double step, int iterations, double* lv) {
  OGraph* og = (OGraph*)gv;
  if (K == 20) return sgd_run<double, 20>(og, lambda, step, iterations, lv);
  if (K == 128) return sgd_run<double, 128>(og, lambda, step, iterations, lv);
  return -1;
}
int gmo_sgd_f32(void* gv, int K, double lambda, double step, int iterations, float* lv) {
  OGraph* og = (OGraph*)gv;
  if (K == 20) return sgd_run<float, 20>(og, lambda, step, iterations, lv);
  if (K == 128) return sgd_run<float, 128>(og, lambda, step, iterations, lv);
  return -1;
}
// returns the summed squared error (caller takes sqrt(err/nnz), SGD.cpp:191)
double gmo_rmse_f64(void* gv, int K, const double* lv, double* sqerr_out) {
  OGraph* og = (OGraph*)gv;
  if (K == 20) return rmse_run<double, 20>(og, lv, sqerr_out);
  if (K == 128) return rmse_run<double, 128>(og, lv, sqerr_out);
  return -1.0;
}
double gmo_rmse_f32(void* gv, int K, const float* lv, float* sqerr_out) {
  OGraph* og = (OGraph*)gv;
  if (K == 20) return rmse_run<float, 20>(og, lv, sqerr_out);
  if (K == 128) return rmse_run<float, 128>(og, lv, sqerr_out);
  return -1.0;
}

// Generic y = A (x) x for the (mul, add) semiring over doubles used by the
// reference's unit test test/test_spmv.cpp:38-81 (y = I*x == x).  x present
// where xmask!=0; ymask_out marks rows that received a message.
struct MulAddProg {
  typedef double msg_t; typedef double red_t; typedef double vp_t; typedef int edge_t;
  static const bool requires_vertexprop = false;
  void process_message(const double& m, const int e, const double&, double& r) const { r = m * e; }
  void reduce_function(double& a, const double& b) const { a += b; }
};
void gmo_spmv_f64(void* gv, int transpose, const double* x, const unsigned char* xmask, double* y,
                  unsigned char* ymask_out) {
  OGraph* og = (OGraph*)gv;
  int n = og->topo.nvertices, np = og->topo.nparts;
  Vec<double> xv(n), yv(n);
  for (int v = 1; v <= n; v++)
    if (xmask[v - 1]) { int i = vertex_to_native(v, np, n) - 1; xv.value[i] = x[v - 1]; bv_set(xv.bits, i); }
  std::vector<double> vp(n);
  MulAddProg p;
  spmspv(transpose ? og->topo.AT : og->topo.A, xv, vp, yv, p);
  for (int v = 1; v <= n; v++) {
    int i = vertex_to_native(v, np, n) - 1;
    ymask_out[v - 1] = bv_get(yv.bits, i);
    y[v - 1] = bv_get(yv.bits, i) ? yv.value[i] : 0.0;
  }
}


// MapReduce of test/test_reduce.cpp:37-38 (map: 2a, reduce: a+b) over an int vector with presence flags
int gmo_mapreduce_double_sum(const int* values, const unsigned char* present, int n, int nthreads, int init) {
  std::vector<int> v(values, values + n);
  std::vector<unsigned char> m(present, present + n);
  int res = init;
  map_reduce_present(v, m, &res, [](const int& a, int* b) { *b = 2 * a; }, nthreads > 0 ? nthreads : 1);
  return res;
}
}  // extern "C"
