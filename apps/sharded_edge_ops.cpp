// sharded_edge_ops.cpp -- Graph<V,E>::applyToAllEdges with SEVERAL RANKS (one process per shard): an edge's value is
// rewritten from BOTH endpoints' vertex properties, and the other endpoint may live on another shard (the reference
// moves the properties along its tile rows and columns, include/GMDP/multinode/applyedges.h:45-161; here they are
// all-gathered with the graph's message exchange).  Every rank contributes its part of the edge list, checks the edges
// of the rows it owns (val = src + s * dst, the closed form of the reference's test/test_apply_edges.cpp:39-112), then a
// weighted SpMV reads the new values.  Host function-pointer form and device-functor form.
// Prints "SHARDEDEDGES rank R ok E edges" per rank and exits 0.  Works with one rank too.
#include <cstdio>
#include <vector>

#include "GraphMatRuntime.h"

static int failures = 0;
#define CHECK(cond)                                                    \
  do {                                                                 \
    if (!(cond)) { printf("CHECK failed: %s (line %d)\n", #cond, __LINE__); failures++; } \
  } while (0)

class SumIn : public GraphMat::GraphProgram<double, double, double> {
 public:
  SumIn() { this->process_message_requires_vertexprop = false; this->activity = GraphMat::ALL_VERTICES; }
  bool send_message(const double& v, double& m) const { m = v; return true; }
  void process_message(const double& m, const int e, const double&, double& r) const { r = m * e; }
  void reduce_function(double& a, const double& b) const { a += b; }
  void apply(const double& y, double& v) { v = y; }
};

static void edge_fn(int* e, const double& src, const double& dst, void* param) { *e = (int)src + (*(int*)param) * (int)dst; }

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  const int rank = GraphMat::get_global_myrank(), nranks = GraphMat::get_global_nrank();
  const int n = 3000;
  // the whole edge list (every rank can evaluate the expected results) and this rank's part of it
  std::vector<GraphMat::edge_t<int> > all;
  unsigned st = 12345u;
  auto next = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (int h = 0; h < 12; h++)
    for (int k = 0; k < 300; k++) all.push_back(GraphMat::edge_t<int>(1 + (int)(next() % n), 1 + h * 7, 1));  // rows above the wave threshold
  for (int i = 0; i < 5 * n; i++) all.push_back(GraphMat::edge_t<int>(1 + (int)(next() % n), 1 + (int)(next() % n), 1));
  std::vector<GraphMat::edge_t<int> > mine;
  for (size_t i = 0; i < all.size(); i++)
    if ((int)(i % (size_t)nranks) == rank) mine.push_back(all[i]);
  std::vector<double> x(n + 1);
  for (int v = 1; v <= n; v++) x[v] = (double)(v % 89 + 1);

  long long checked = 0;
  for (int form = 0; form < 2; form++) {
    GraphMat::edgelist_t<int> E(n, n, (int)mine.size());
    std::copy(mine.begin(), mine.end(), E.edges);
    GraphMat::Graph<double> G;
    G.ReadEdgelist(E);
    E.clear();
    for (int v = 1; v <= n; v++) G.setVertexproperty(v, x[v]);  // (a rank sets the vertices it owns)
    int s = 3 + form;
    if (form == 0) G.applyToAllEdges(edge_fn, (void*)&s);
    else G.applyToAllEdges([s](int* e, const double& src, const double& dst) { *e = (int)src + s * (int)dst; });
    GraphMat::edgelist_t<int> out;
    G.getEdgelist(out);  // this rank's rows
    bool ok = true;
    for (int i = 0; i < out.nnz; i++) ok &= (out.edges[i].val == (int)x[out.edges[i].src] + s * (int)x[out.edges[i].dst]);
    CHECK(ok);
    if (form == 0) checked = out.nnz;
    out.clear();
    // the multiply reads the rewritten values
    std::vector<double> want(n + 1, 0.0);
    std::vector<char> has(n + 1, 0);
    for (auto& e : all) { want[e.dst] += x[e.src] * (double)((int)x[e.src] + s * (int)x[e.dst]); has[e.dst] = 1; }
    SumIn prog;
    G.setAllActive();
    GraphMat::run_graph_program(&prog, G, 1);
    ok = true;
    int owned = 0;
    for (int v = 1; v <= n; v++)
      if (G.vertexNodeOwner(v)) { owned++; ok &= (G.getVertexproperty(v) == (has[v] ? want[v] : x[v])); }
    CHECK(ok);
    CHECK(owned > 0);
  }
  printf("SHARDEDEDGES rank %d %s %lld edges of %zu\n", rank, failures == 0 ? "ok" : "FAIL", checked, all.size());
  MPI_Finalize();
  return failures == 0 ? 0 : 1;
}
