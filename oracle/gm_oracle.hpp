// gm_oracle.hpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// CPU restatement of the reference's generalized-SpMV iteration
// (send_message -> process_message -> reduce -> apply), written from the
// algorithm description, used as the checker for the HIP path and as the
// "port" cpu_baseline in bench.py.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use anything in oracle/.
//
// Parity pin: the reference itself is NOT buildable in this image (it needs
// boost::serialization headers which are absent, and stand-ins are not
// allowed), so this restatement is pinned against
//   * the closed-form expectations in the reference's own tests
//     (test/test_bfs.cpp:97-236, test/test_spmv.cpp:38-81,
//      test/test_reduce.cpp:39-65, test/test_graph_basics.cpp:56-81), and
//   * the reference outputs G1..G3 recorded in SURVEY.md section 8c /
//     BASELINE.md section 2 on the data/*.bin.mtx fixtures
// (see tests/test_oracle_golden.py).
//
// Each function cites the reference lines it follows (paths relative to the
// reference tree).
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#include <parallel/algorithm>
#endif

namespace gmo {

// ---- bit vectors: bit (i & 31) of int32 word (i >> 5) ----------------------
// follows include/GMDP/utils/bitvector.h:36-64
inline void bv_set(std::vector<uint32_t>& bv, int i) { bv[i >> 5] |= (1u << (i & 31)); }
inline bool bv_get(const std::vector<uint32_t>& bv, int i) { return (bv[i >> 5] >> (i & 31)) & 1u; }

// ---- vertex id permutation --------------------------------------------------
// follows include/Graph.h:111-130 (1-based in, 1-based out);
// nparts = num_threads * 16 * nsegments, len = number of vertices.
inline int vertex_to_native(int vertex, int nparts, int len) {
  int v = vertex - 1;
  int height = len / nparts;
  int vmax = height * nparts;
  if (v >= vmax) return v + 1;
  int col = v % nparts;
  int row = v / nparts;
  return row + col * height + 1;
}
// follows include/Graph.h:132-150
inline int native_to_vertex(int vertex, int nparts, int len) {
  int v = vertex - 1;
  int height = len / nparts;
  int vmax = height * nparts;
  if (v >= vmax) return v + 1;
  int col = v / height;
  int row = v % height;
  return col + row * nparts + 1;
}

// ---- "dense segment": value array + presence bit vector ---------------------
// follows include/GMDP/vectors/DenseSegment.h:423-640 (value[] + bit_vector[])
template <class T>
struct Vec {
  int n = 0;
  std::vector<T> value;
  std::vector<uint32_t> bits;
  explicit Vec(int n_ = 0) : n(n_), value(n_), bits((n_ + 31) / 32, 0u) {}
  void clear() {  // (parallel: this is also the CPU baseline bench.py times; the result is that of a serial fill)
    const int64_t nw = (int64_t)bits.size();
#pragma omp parallel for
    for (int64_t w = 0; w < nw; w++) bits[w] = 0u;
  }
  void set_all(const T& v) {  // DenseSegment.h:617-640: exactly n bits set
    for (int i = 0; i < n; i++) { value[i] = v; bv_set(bits, i); }
  }
  int nnz() const {
    int c = 0;
    for (uint32_t w : bits) c += __builtin_popcount(w);
    return c;
  }
};

// ---- row-partitioned DCSC tile ----------------------------------------------
// follows include/GMDP/matrices/DCSCTile.h:185-220 (static_partition, round=32),
// :241-381 (constructor: partition id, sort by (partition, col, row), column
// compaction).  Only the iteration ORDER matters to results; the arrays are laid
// out so that spmspv() below walks them exactly like my_spmspv does.
template <class E>
struct Dcsc {
  int m = 0, n = 0;
  int64_t nnz = 0;
  int num_partitions = 0;
  std::vector<int> row_pointers;      // num_partitions+1
  std::vector<int64_t> edge_pointers; // num_partitions+1
  std::vector<int64_t> col_starts;    // num_partitions+1 (index into col_index/col_ptr)
  std::vector<int> col_index;         // per partition: distinct columns, ascending
  std::vector<int64_t> col_ptr;       // per partition: start of each column (+1 sentinel)
  std::vector<int> row_ind;           // nnz
  std::vector<E> vals;                // nnz

  struct TE { int row, col, part; E val; int64_t seq; };

  // rows/cols are 0-based native ids here.
  void build(const std::vector<int>& rows, const std::vector<int>& cols, const std::vector<E>& v,
             int m_, int n_, int nparts) {
    m = m_; n = n_; nnz = (int64_t)rows.size(); num_partitions = nparts;
    // static_partition(round = 32): DCSCTile.h:204-218
    const int round = 32;
    row_pointers.assign(nparts + 1, 0);
    {
      int n512 = std::max((m / round) / nparts, 1);
      int n_round = std::max(0, m / round - n512 * nparts);
      for (int p = 1; p < nparts; p++) {
        row_pointers[p] = row_pointers[p - 1] + ((n_round > 0) ? ((n512 + 1) * round) : (n512 * round));
        row_pointers[p] = std::min(row_pointers[p], m);
        if (n_round > 0) n_round--;
      }
      row_pointers[nparts] = m;
    }
    std::vector<TE> te(nnz);
#pragma omp parallel for
    for (int64_t i = 0; i < nnz; i++) {
      int r = rows[i];
      // partition id = the p with row_pointers[p] <= r < row_pointers[p+1]
      // (DCSCTile.h:262-279 finds it by search; with empty trailing partitions
      //  the first match from the top is the non-empty one, which upper_bound gives)
      int p = int(std::upper_bound(row_pointers.begin(), row_pointers.end(), r) - row_pointers.begin()) - 1;
      te[i] = TE{r, cols[i], p, v.empty() ? E() : v[i], i};
    }
    // sort key (partition, col, row): DCSCTile.h:41-58.  The reference's
    // parallel sort is not stable; ties (exact duplicate edges) are broken by
    // input order here so the restatement is deterministic.
    auto cmp = [](const TE& a, const TE& b) {
      if (a.part != b.part) return a.part < b.part;
      if (a.col != b.col) return a.col < b.col;
      if (a.row != b.row) return a.row < b.row;
      return a.seq < b.seq;
    };
#ifdef _OPENMP
    __gnu_parallel::sort(te.begin(), te.end(), cmp);
#else
    std::sort(te.begin(), te.end(), cmp);
#endif
    edge_pointers.assign(nparts + 1, nnz);
    {
      int p = 0;
      for (int64_t e = 0; e < nnz; e++)
        while (p <= te[e].part) edge_pointers[p++] = e;
      // remaining partitions point at nnz (DCSCTile.h:222-239)
    }
    edge_pointers[nparts] = nnz;
    col_starts.assign(nparts + 1, 0);
    row_ind.resize(nnz);
    vals.resize(nnz);
    col_index.clear();
    col_ptr.clear();
    for (int p = 0; p < nparts; p++) {
      col_starts[p] = (int64_t)col_index.size();
      int cur = -1;
      for (int64_t e = edge_pointers[p]; e < edge_pointers[p + 1]; e++) {
        row_ind[e] = te[e].row;
        vals[e] = te[e].val;
        if (cur < te[e].col) {
          cur = te[e].col;
          col_index.push_back(cur);
          col_ptr.push_back(e - edge_pointers[p]);
        }
      }
      col_index.push_back(n + 1);  // sentinel, DCSCTile.h:370
      col_ptr.push_back(edge_pointers[p + 1] - edge_pointers[p]);
    }
    col_starts[nparts] = (int64_t)col_index.size();
  }
};

// ---- y = A (x) x over a user semiring ----------------------------------------
// follows include/GMDP/singlenode/spmspv.h:39-86 (my_spmspv, 2-operand) and
// include/GMDP/singlenode/spmspv3.h:38-90 (my_spmspv3, +vertex property of the
// ROW), selected like include/SPMV.h:62-95 by requires_vertexprop.
// No additive identity: first touch of a row assigns, later touches reduce.
template <class P, class E, class T, class U, class V>
void spmspv(const Dcsc<E>& A, const Vec<T>& x, const std::vector<V>& vp, Vec<U>& y, const P& prog) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int p = 0; p < A.num_partitions; p++) {
    const int64_t cs = A.col_starts[p];
    const int64_t ncol = A.col_starts[p + 1] - cs - 1;
    const int64_t eb = A.edge_pointers[p];
    for (int64_t j = 0; j < ncol; j++) {
      int col = A.col_index[cs + j];
      if (!bv_get(x.bits, col)) continue;
      for (int64_t k = A.col_ptr[cs + j]; k < A.col_ptr[cs + j + 1]; k++) {
        int row = A.row_ind[eb + k];
        U tmp;
        if (P::requires_vertexprop) {
          prog.process_message(x.value[col], A.vals[eb + k], vp[row], tmp);
        } else {
          V dummy = V();  // SPMV.h:41-46 passes a default-constructed V
          prog.process_message(x.value[col], A.vals[eb + k], dummy, tmp);
        }
        if (bv_get(y.bits, row)) {
          U c = y.value[row];                // SPMV.h:54-59: c = a; reduce(c, b)
          prog.reduce_function(c, tmp);
          y.value[row] = c;
        } else {
          y.value[row] = tmp;
          y.bits[row >> 5] |= (1u << (row & 31));  // rows of a partition are private to it
        }
      }
    }
  }
}

enum EdgeDir { OUT_EDGES = 0, IN_EDGES = 1, ALL_EDGES = 2 };
enum Activity { ACTIVE_ONLY = 0, ALL_VERTICES = 1 };

// ---- graph container ----------------------------------------------------------
// follows include/Graph.h:58-107, :210-246 (ReadEdgelist: permute ids, build A
// with row=src col=dst, AT with row=dst col=src).  The adjacency (Topo) is kept
// apart from the vertex state so one build can serve several vertex-property
// types (the reference rebuilds per Graph<V,E>; results do not depend on that).
template <class E>
struct Topo {
  int nvertices = 0;
  int64_t nnz = 0;
  int nparts = 16;  // num_threads*16*nranks of the reference configuration being restated
  Dcsc<E> A, AT;
  // src/dst are 1-based original ids, as in the .mtx files
  void read_edgelist(int nv, int64_t ne, const int* src, const int* dst, const E* val, int ref_threads) {
    nvertices = nv; nnz = ne; nparts = ref_threads * 16;
    std::vector<int> r(ne), c(ne);
    std::vector<E> v(ne);
#pragma omp parallel for
    for (int64_t i = 0; i < ne; i++) {
      r[i] = vertex_to_native(src[i], nparts, nv) - 1;  // Graph.h:219-224
      c[i] = vertex_to_native(dst[i], nparts, nv) - 1;
      v[i] = val ? val[i] : E(1);
    }
    A.build(r, c, v, nv, nv, nparts);    // row = src, col = dst (Graph.h:226)
    AT.build(c, r, v, nv, nv, nparts);   // Graph.h:227, SpMat.h:422-443: transpose via edge list
  }
};

template <class V, class E>
struct Graph {
  const Topo<E>* topo = nullptr;
  int nvertices = 0;
  int64_t nnz = 0;
  int nparts = 16;
  std::vector<V> vp;            // native order, all present (Graph.h:232-234)
  std::vector<uint32_t> active; // bit vector (Graph.h:235-237: all clear)
  const Dcsc<E>& A() const { return topo->A; }
  const Dcsc<E>& AT() const { return topo->AT; }

  explicit Graph(const Topo<E>* t) : topo(t), nvertices(t->nvertices), nnz(t->nnz), nparts(t->nparts) {
    vp.assign(nvertices, V());
    active.assign((nvertices + 31) / 32, 0u);
  }
  int to_native0(int v1) const { return vertex_to_native(v1, nparts, nvertices) - 1; }
  void set_all_active() {  // Graph.h:263-266: exactly nvertices bits set (word-wise and parallel: also the timed CPU baseline)
    const int64_t nw = (int64_t)active.size();
#pragma omp parallel for
    for (int64_t w = 0; w < nw; w++) {
      const int64_t rem = (int64_t)nvertices - w * 32;
      active[w] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    }
  }
  void set_all_inactive() {  // Graph.h:268-280
    const int64_t nw = (int64_t)active.size();
#pragma omp parallel for
    for (int64_t w = 0; w < nw; w++) active[w] = 0u;
  }
  void set_active(int v1) { bv_set(active, to_native0(v1)); }               // Graph.h:283-286
  void set_vertexproperty(int v1, const V& val) { vp[to_native0(v1)] = val; }  // Graph.h:312-316
  V get_vertexproperty(int v1) const { return vp[to_native0(v1)]; }            // Graph.h:358-364
};

// ---- the iteration driver ------------------------------------------------------
// follows include/GraphMatRuntime.h:93-279 step by step.
// Program concept (static polymorphism instead of virtuals):
//   typedef T msg_t; U red_t; V vp_t; E edge_t;
//   static const bool requires_vertexprop;  EdgeDir order; Activity activity;
//   send_message(const V&, T&) const; process_message(const T&, E, const V&, U&) const;
//   reduce_function(U&, const U&) const; apply(const U&, V&); do_every_iteration(int);
//   static bool changed(V& old, const V& now)  -- the V::operator!= of the app
// Returns the number of iterations completed; per-iteration "changed" counts
// (the reference's `active->getNNZ()` print, :247) go to changed_hist if given.
// wall-clock seconds spent in send / multiply+reduce / apply since the last reset: the reference's __TIMING phases
// (:150-232), kept so that bench.py's cpu_baseline can report where the CPU's time goes
inline double* phase_seconds() {
  static double t[4] = {0, 0, 0, 0};
  return t;
}
template <class P, class V, class E>
int run_graph_program(P& prog, Graph<V, E>& g, int iterations, std::vector<int>* changed_hist = nullptr,
                      std::vector<int>* updated_hist = nullptr) {
  typedef typename P::msg_t T;
  typedef typename P::red_t U;
  const int n = g.nvertices;
  Vec<T> x(n);
  Vec<U> y(n);
  if (prog.activity == ALL_VERTICES) g.set_all_active();  // :121-123
  int it = 0;
  while (true) {
    const double t_0 = omp_get_wtime();
    x.clear();  // :139
    y.clear();  // :140
    // send: xbits = active & vpbits; x[i] = send_message(vp[i]); the bool result
    // is discarded (:79-85; singlenode/intersectreduce.h:43-65)
#pragma omp parallel for
    for (int w = 0; w < (int)x.bits.size(); w++) {
      uint32_t word = g.active[w];
      x.bits[w] = word;
      while (word) {
        int b = __builtin_ctz(word);
        int i = w * 32 + b;
        prog.send_message(g.vp[i], x.value[i]);
        word &= word - 1;
      }
    }
    const double t_1 = omp_get_wtime();
    // multiply + reduce (:160-176); OUT_EDGES -> AT, IN_EDGES -> A, ALL_EDGES ->
    // AT then A accumulating into the same y
    if (prog.order == OUT_EDGES) {
      spmspv(g.AT(), x, g.vp, y, prog);
    } else if (prog.order == IN_EDGES) {
      spmspv(g.A(), x, g.vp, y, prog);
    } else {
      spmspv(g.AT(), x, g.vp, y, prog);
      spmspv(g.A(), x, g.vp, y, prog);
    }
    const double t_2 = omp_get_wtime();
    g.set_all_inactive();  // :184
    // apply on set bits of y; changed => active, not converged (:195-225)
    int converged = 1;
#pragma omp parallel for reduction(& : converged)
    for (int w = 0; w < (int)y.bits.size(); w++) {
      uint32_t word = y.bits[w];
      while (word) {
        int b = __builtin_ctz(word);
        int i = w * 32 + b;
        V old = g.vp[i];
        prog.apply(y.value[i], g.vp[i]);
        if (P::changed(old, g.vp[i])) {
          g.active[w] |= (1u << b);
          converged = 0;
        }
        word &= word - 1;
      }
    }
    const double t_3 = omp_get_wtime();
    phase_seconds()[0] += t_1 - t_0;
    phase_seconds()[1] += t_2 - t_1;
    phase_seconds()[2] += t_3 - t_2;
    prog.do_every_iteration(it);  // :236
    if (updated_hist) updated_hist->push_back(y.nnz());
    if (changed_hist) {
      int c = 0;
      for (uint32_t w : g.active) c += __builtin_popcount(w);
      changed_hist->push_back(c);
    }
    if (prog.activity == ALL_VERTICES) g.set_all_active();  // :250-252
    it++;
    if (it == iterations) break;                  // :254-256
    if (iterations <= 0 && converged == 1) break;  // :257-259
  }
  return it;
}

// ---- map-reduce over vertex properties ------------------------------------------
// follows include/GMDP/singlenode/reduce.h:51-99 with nthreads chunks, then the
// serial combine (:87-95); `res` is the caller's initial value.
template <class V, class R, class Map>
void map_reduce(const std::vector<V>& vp, R* res, Map op_map, int nthreads) {
  int n = (int)vp.size();
  int per = (n + nthreads - 1) / nthreads;
  for (int p = 0; p < nthreads; p++) {
    int s = std::min(per * p, n), e = std::min(per * (p + 1), n);
    bool first = false;
    R local = R();
    for (int i = s; i < e; i++) {
      R t;
      op_map(vp[i], &t);
      if (first) local = local + t; else { local = t; first = true; }
    }
    if (first) *res = *res + local;
  }
}

// the same over a dense segment with presence bits: only the present entries take part
// (include/GMDP/singlenode/reduce.h:51-99 walks the set bits of the segment)
template <class V, class R, class Map>
void map_reduce_present(const std::vector<V>& vals, const std::vector<unsigned char>& present, R* res, Map op_map, int nthreads) {
  int n = (int)vals.size();
  int per = (n + nthreads - 1) / nthreads;
  for (int p = 0; p < nthreads; p++) {
    int s = std::min(per * p, n), e = std::min(per * (p + 1), n);
    bool first = false;
    R local = R();
    for (int i = s; i < e; i++) {
      if (!present[i]) continue;
      R t;
      op_map(vals[i], &t);
      if (first) local = local + t; else { local = t; first = true; }
    }
    if (first) *res = *res + local;
  }
}

}  // namespace gmo
