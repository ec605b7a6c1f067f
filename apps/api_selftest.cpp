// api_selftest.cpp -- exercises the Graph<V,E> / run_graph_program surface the way the
// reference's own unit tests do (test/test_graph_basics.cpp:56-161 get/set round trip and
// edge list in == out; test/test_apply_edges.cpp:39-112 argument order of applyToAllEdges;
// test/test_reduce.cpp:39-67 map-reduce sums; test/test_bfs.cpp BFS depths with a dummy
// message; test/test_spmv.cpp identity SpMV), against this repository's headers.
// Prints "SELFTEST PASS" and exits 0 when everything holds.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "GraphMatRuntime.h"

static int failures = 0;
#define CHECK(cond)                                                    \
  do {                                                                 \
    if (!(cond)) { printf("CHECK failed: %s (line %d)\n", #cond, __LINE__); failures++; } \
  } while (0)

struct IntProp {
  int v;
  IntProp() : v(0) {}
  bool operator!=(const IntProp& o) { return v != o.v; }
  friend std::ostream& operator<<(std::ostream& os, const IntProp& p) { return os << p.v; }
};

// y = A^T x over (mul, add): every vertex sends its value, products are summed
class SumIn : public GraphMat::GraphProgram<double, double, double> {
 public:
  SumIn() { this->process_message_requires_vertexprop = false; this->activity = GraphMat::ALL_VERTICES; }
  bool send_message(const double& v, double& m) const { m = v; return true; }
  void process_message(const double& m, const int e, const double&, double& r) const { r = m * e; }
  void reduce_function(double& a, const double& b) const { a += b; }
  void apply(const double& y, double& v) { v = y; }
};

struct Depth {
  unsigned int depth;
  Depth() : depth(0xffffffffu) {}
  bool operator!=(const Depth& o) { return depth != o.depth; }
  friend std::ostream& operator<<(std::ostream& os, const Depth& d) { return os << d.depth; }
};
class DepthBfs : public GraphMat::GraphProgram<unsigned long long, unsigned long long, Depth> {
 public:
  unsigned int current_depth;
  DepthBfs() : current_depth(1) { this->process_message_requires_vertexprop = false; }
  void reduce_function(unsigned long long& a, const unsigned long long& b) const { a = b; }
  void process_message(const unsigned long long& m, const int, const Depth&, unsigned long long& r) const { r = m; }
  bool send_message(const Depth& v, unsigned long long& m) const { m = 0; return v.depth == current_depth - 1; }
  void apply(const unsigned long long&, Depth& v) { if (v.depth == 0xffffffffu) v.depth = current_depth; }
  void do_every_iteration(int) { current_depth++; }
};

static void add_one(const IntProp& in, IntProp* out, void*) { out->v = in.v + 1; }
static void get_v(IntProp* p, int* out, void*) { *out = p->v; }
static void edge_fn(int* e, const IntProp& src, const IntProp& dst, void* param) { *e = src.v + (*(int*)param) * dst.v; }

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  const int n = 500;
  // random-ish graph: i -> (7i+3) mod n, i -> (i*i+1) mod n, plus a duplicate edge and a self loop
  std::vector<GraphMat::edge_t<int> > ed;
  for (int i = 0; i < n; i++) {
    ed.push_back(GraphMat::edge_t<int>(i + 1, (7 * i + 3) % n + 1, 1 + i % 5));
    ed.push_back(GraphMat::edge_t<int>(i + 1, (int)(((long long)i * i + 1) % n) + 1, 2));
  }
  ed.push_back(GraphMat::edge_t<int>(1, 4, 9));
  ed.push_back(GraphMat::edge_t<int>(5, 5, 3));
  GraphMat::edgelist_t<int> E(n, n, (int)ed.size());
  std::copy(ed.begin(), ed.end(), E.edges);

  {  // ---- get/set round trip through the id permutations; counts ----
    GraphMat::Graph<IntProp> G;
    G.ReadEdgelist(E);
    CHECK(G.getNumberOfVertices() == n);
    CHECK(G.nnz == (long long)ed.size());
    for (int v = 1; v <= n; v++) { IntProp p; p.v = 10 * v; G.setVertexproperty(v, p); }
    bool ok = true;
    for (int v = 1; v <= n; v++) ok &= (G.getVertexproperty(v).v == 10 * v);
    CHECK(ok);
    // ---- edge list in == out (as multisets) ----
    GraphMat::edgelist_t<int> out;
    G.getEdgelist(out);
    CHECK(out.nnz == (int)ed.size());
    auto key = [](const GraphMat::edge_t<int>& e) { return ((long long)e.src << 40) | ((long long)e.dst << 16) | e.val; };
    std::vector<long long> a, b;
    for (auto& e : ed) a.push_back(key(e));
    for (int i = 0; i < out.nnz; i++) b.push_back(key(out.edges[i]));
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    CHECK(a == b);
    out.clear();
    // ---- applyToAllVertices / applyReduceAllVertices ----
    G.applyToAllVertices(add_one);
    int sum = 0;
    G.applyReduceAllVertices(&sum, get_v);
    CHECK(sum == 10 * n * (n + 1) / 2 + n);
    // ---- applyToAllEdges: val = src + s*dst with src/dst the true endpoints ----
    int s = 1000;
    G.applyToAllEdges(edge_fn, &s);
    G.getEdgelist(out);
    ok = true;
    for (int i = 0; i < out.nnz; i++)
      ok &= (out.edges[i].val == (10 * out.edges[i].src + 1) + s * (10 * out.edges[i].dst + 1));
    CHECK(ok);
    out.clear();
    // ---- the same three operations on the device (functor forms): no host mirror involved ----
    {
      GraphMat::Graph<IntProp> H;
      H.ReadEdgelist(E);
      for (int v = 1; v <= n; v++) { IntProp p; p.v = 10 * v; H.setVertexproperty(v, p); }
      H.applyToAllVertices([](const IntProp& in, IntProp* out) { out->v = in.v + 1; });
      long long dsum = 5;  // combined with the caller's value, like the reference
      H.applyReduceAllVertices(&dsum, [](const IntProp& p, long long* out) { *out = p.v; });
      CHECK(dsum == 5 + 10LL * n * (n + 1) / 2 + n);
      int dmax = -1;
      H.applyReduceAllVertices(&dmax, [](const IntProp& p, int* out) { *out = p.v; },
                               [](const int& a, const int& b, int* c) { *c = a > b ? a : b; });
      CHECK(dmax == 10 * n + 1);
      const int s2 = 1000;
      H.applyToAllEdges([s2](int* e, const IntProp& src, const IntProp& dst) { *e = src.v + s2 * dst.v; });
      GraphMat::edgelist_t<int> out2;
      H.getEdgelist(out2);
      ok = out2.nnz == (int)ed.size();
      for (int i = 0; i < out2.nnz; i++)
        ok &= (out2.edges[i].val == (10 * out2.edges[i].src + 1) + s2 * (10 * out2.edges[i].dst + 1));
      CHECK(ok);
      out2.clear();
      ok = true;
      for (int v = 1; v <= n; v++) ok &= (H.getVertexproperty(v).v == 10 * v + 1);  // device result seen through the host API
      CHECK(ok);
    }
    // ---- activity bookkeeping ----
    G.setAllInactive();
    CHECK(G.active->getNNZ() == 0);
    G.setActive(7);
    G.setActive(n);
    CHECK(G.active->getNNZ() == 2);
    G.setInactive(7);
    CHECK(G.active->getNNZ() == 1);
    G.setAllActive();
    CHECK(G.active->getNNZ() == n);
  }
  {  // ---- y = A^T x on the device equals a host evaluation ----
    GraphMat::Graph<double> G;
    G.ReadEdgelist(E);
    std::vector<double> x(n + 1), want(n + 1, 0.0);
    std::vector<char> has(n + 1, 0);
    for (int v = 1; v <= n; v++) { x[v] = 0.5 * v; G.setVertexproperty(v, x[v]); }
    for (auto& e : ed) { want[e.dst] += x[e.src] * e.val; has[e.dst] = 1; }
    SumIn prog;
    G.setAllActive();
    GraphMat::run_graph_program(&prog, G, 1);
    bool ok = true;
    for (int v = 1; v <= n; v++) ok &= (G.getVertexproperty(v) == (has[v] ? want[v] : x[v]));  // small integers*0.5: exact
    CHECK(ok);
  }
  {  // ---- BFS depths on a circular chain (closed form of the reference's test) ----
    GraphMat::edgelist_t<int> C(n, n, n);
    for (int i = 0; i < n; i++) C.edges[i] = GraphMat::edge_t<int>(i + 1, (i + 1) % n + 1, 1);
    GraphMat::Graph<Depth> G;
    G.ReadEdgelist(C);
    C.clear();
    Depth d0;
    d0.depth = 0;
    G.setVertexproperty(n / 2, d0);
    G.setAllInactive();
    G.setActive(n / 2);
    DepthBfs prog;
    GraphMat::run_graph_program(&prog, G, GraphMat::UNTIL_CONVERGENCE);
    bool ok = true;
    for (int i = 1; i <= n; i++) {
      unsigned want = (i < n / 2) ? (unsigned)(n / 2 + i) : (unsigned)(i - n / 2);
      ok &= (G.getVertexproperty(i).depth == want);
    }
    CHECK(ok);
  }
  E.clear();
  printf(failures == 0 ? "SELFTEST PASS\n" : "SELFTEST FAIL (%d)\n", failures);
  MPI_Finalize();
  return failures == 0 ? 0 : 1;
}
