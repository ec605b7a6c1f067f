// mutating_program.cpp -- ALL_VERTICES programs whose do_every_iteration changes what send_message computes.
// The engine writes the next iteration's messages from inside the apply pass (kernels.hpp: k_apply_send) and may use
// them only if do_every_iteration left the program object unchanged; the reference evaluates send_message at the start
// of every iteration with the program as it is then (GraphMatRuntime.h:136-145,236).  Three programs over the same
// graph, several iterations in one run_graph_program call, each compared with a host evaluation of the reference's loop:
//   Steady    : the program never changes (the fused messages are always used)
//   EveryTime : `gain` changes after every iteration (never used)
//   Sometimes : `gain` changes after iterations 1 and 3 only (used, dropped, used, ...)
// and each once more with the fusion switched off (gm_set_option("fuse_apply_send", 0)): same bits.
// Integer-valued doubles kept small, so every fold order gives the same bits -- what is tested is WHICH messages the
// multiply reads.  Prints "MUTATING PASS" and exits 0.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "GraphMatRuntime.h"

static int failures = 0;
#define CHECK(cond)                                                    \
  do {                                                                 \
    if (!(cond)) { printf("CHECK failed: %s (line %d)\n", #cond, __LINE__); failures++; } \
  } while (0)

template <int MODE>  // 0 steady, 1 every time, 2 sometimes
class Scaled : public GraphMat::GraphProgram<double, double, double> {
 public:
  double gain;
  Scaled() : gain(2.0) { this->process_message_requires_vertexprop = false; this->activity = GraphMat::ALL_VERTICES; }
  bool send_message(const double& v, double& m) const { m = v * gain; return true; }
  void process_message(const double& m, const int e, const double&, double& r) const { r = m * e; }
  void reduce_function(double& a, const double& b) const { a += b; }
  void apply(const double& y, double& v) { v = (double)((long long)y % 1009 + 1); }
  void do_every_iteration(int it) {
    if (MODE == 1 || (MODE == 2 && (it == 1 || it == 3))) gain = (double)(it + 3);
  }
};

typedef std::vector<GraphMat::edge_t<int> > edges_t;

template <int MODE>
static bool run_and_compare(GraphMat::Graph<double>& G, const edges_t& ed, int n, int iters) {
  std::vector<double> v(n + 1), y(n + 1);
  std::vector<char> has(n + 1);
  for (int u = 1; u <= n; u++) v[u] = (double)(u % 97 + 1);
  for (int u = 1; u <= n; u++) G.setVertexproperty(u, v[u]);
  Scaled<MODE> host;  // the reference's loop on the host
  for (int it = 0; it < iters; it++) {
    std::fill(y.begin(), y.end(), 0.0);
    std::fill(has.begin(), has.end(), 0);
    for (auto& e : ed) {
      double m, r;
      host.send_message(v[e.src], m);
      host.process_message(m, e.val, 0.0, r);
      y[e.dst] += r;
      has[e.dst] = 1;
    }
    for (int u = 1; u <= n; u++)
      if (has[u]) host.apply(y[u], v[u]);
    host.do_every_iteration(it);
  }
  Scaled<MODE> prog;
  G.setAllActive();
  GraphMat::run_graph_program(&prog, G, iters);
  bool ok = true;
  int shown = 0;
  for (int u = 1; u <= n; u++) {
    if (!G.vertexNodeOwner(u)) continue;  // (several ranks: every vertex is checked by the rank that owns it)
    const double got = G.getVertexproperty(u);
    if (got != v[u]) {
      ok = false;
      if (shown++ < 3) printf("  mode %d vertex %d: %.1f, expected %.1f\n", MODE, u, got, v[u]);
    }
  }
  return ok;
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  const int n = 5000;
  edges_t ed;
  unsigned st = 12345u;
  auto next = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (int h = 0; h < 16; h++)
    for (int k = 0; k < 200 + 30 * h; k++) ed.push_back(GraphMat::edge_t<int>(1 + (int)(next() % n), 1 + h, 1 + (int)(next() % 3)));
  for (int i = 0; i < n; i++) ed.push_back(GraphMat::edge_t<int>(1 + i, 1 + (i + 1) % n, 1));
  for (int i = 0; i < 3 * n; i++) ed.push_back(GraphMat::edge_t<int>(1 + (int)(next() % n), 1 + (int)(next() % n), 1 + (int)(next() % 3)));
  // several ranks (one process per shard): every rank passes its part of the edge list; the sharded runs take the
  // two-stage schedule, whose sends happen BEFORE do_every_iteration and must be redone when it changes the program
  const int rank = GraphMat::get_global_myrank(), nranks = GraphMat::get_global_nrank();
  edges_t mine;
  for (size_t i = 0; i < ed.size(); i++)
    if ((int)(i % (size_t)nranks) == rank) mine.push_back(ed[i]);
  GraphMat::Graph<double> G;
  GraphMat::edgelist_t<int> E(n, n, (int)mine.size());
  std::copy(mine.begin(), mine.end(), E.edges);
  G.ReadEdgelist(E);
  E.clear();
  for (int fuse = 1; fuse >= 0; fuse--) {
    CHECK(gm_set_option("fuse_apply_send", fuse) == 0);
    const bool a = run_and_compare<0>(G, ed, n, 5), b = run_and_compare<1>(G, ed, n, 5), c = run_and_compare<2>(G, ed, n, 6);
    printf("fuse_apply_send=%d: steady %s, every-time %s, sometimes %s\n", fuse, a ? "ok" : "WRONG", b ? "ok" : "WRONG", c ? "ok" : "WRONG");
    CHECK(a); CHECK(b); CHECK(c);
  }
  if (failures == 0) printf("MUTATING PASS (rank %d of %d)\n", rank, nranks);
  else printf("MUTATING FAIL (%d) (rank %d of %d)\n", failures, rank, nranks);
  MPI_Finalize();
  return failures == 0 ? 0 : 1;
}
