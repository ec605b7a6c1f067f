// speculated_float_sum.cpp -- giant rows of programs that DECLARE nothing about their reduce_function (no program_traits, no
// environment variables: what every unchanged application of the reference is).  The engine folds such a row strictly in stored
// order (SPMV.h:54-59: c = a; reduce(c, b)); for a row of tens of thousands of edges that is one long chain of dependent calls.
// When the function answers the runtime's questions like a float addition, the engine SPECULATES -- the exact parallel replay of
// the float sum -- and PROVES every 8192-product chunk of it with the program's own function (kernels.hpp: k_giant_verify_chunks);
// a row with a disagreeing chunk is folded again in order.  Checked here, bit for bit against a host fold in the reference's order
// (ascending native id of the source), with sums whose bits depend on the order:
//   (1) PlainSum: a += b                                   -- the speculation holds
//   (2) TrickySum: a += b, except that b == 1/64 adds 1 more -- answers every question like an addition, is none on this data:
//       the proof must fail for the rows that receive such a message and their results must still be the ordered fold's
//   (3) the smaller of a and b -- no addition; ANY function is speculated to be associative on the operands at hand (chunk totals
//       combined in order give candidates for the running values, every chunk is folded again from its candidate and compared:
//       k_giant_verify_chunks_any); holds for a minimum
//   (4) a = a / 2 + b -- neither: every proof fails, the rows are folded in order
// Prints "SPECULATED PASS" and exits 0.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "GraphMatRuntime.h"

static int failures = 0;
#define CHECK(cond)                                                    \
  do {                                                                 \
    if (!(cond)) { printf("CHECK failed: %s (line %d)\n", #cond, __LINE__); failures++; } \
  } while (0)

template <int KIND>
class Fold : public GraphMat::GraphProgram<float, float, float> {
 public:
  Fold() { this->process_message_requires_vertexprop = false; this->activity = GraphMat::ALL_VERTICES; }
  bool send_message(const float& v, float& m) const { m = v; return true; }
  void process_message(const float& m, const int e, const float&, float& r) const { r = m; }
  void reduce_function(float& a, const float& b) const {
    if (KIND == 0) a += b;
    else if (KIND == 1) { if (b == 0.015625f) a = a + b + 1.0f; else a += b; }
    else if (KIND == 2) a = b < a ? b : a;
    else a = a * 0.5f + b;
  }
  void apply(const float& y, float& v) { v = y; }
};

typedef std::vector<GraphMat::edge_t<int> > edges_t;

// how many giant rows the last multiply had to fold again (white box: the engine keeps the chunk boundaries and the per-row flags of
// k_giant_verify_chunks in workspace slot 15 of the graph: [ngchunk + 2] boundaries of 8 bytes, then one int per giant row); -1 = no speculation ran
static int rows_folded_again(GraphMat::Graph<float>& G) {
  gm_csr_t ca;
  if (gm_graph_csr(G.A, GM_DIR_OUT, &ca) != 0) return -1;
  void* ws = nullptr;
  size_t bytes = 0;
  int external = 0;
  if (gm_graph_workspace_info(G.A, 15, &ws, &bytes, &external) != 0 || ws == nullptr || bytes < ((size_t)ca.ngchunk + 2) * 8 + (size_t)ca.ngiant * 4) return -1;
  std::vector<int> redo(ca.ngiant);
  if (hipMemcpy(redo.data(), (const char*)ws + ((size_t)ca.ngchunk + 2) * 8, (size_t)ca.ngiant * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  int k = 0;
  for (int r : redo) k += r != 0;
  return k;
}

template <int KIND>
static int run_case(GraphMat::Graph<float>& G, const edges_t& ed, int n, const std::vector<float>& x, const char* what) {
  // host: every destination folds its messages in stored order (duplicate edges carry equal messages)
  std::vector<std::vector<int> > in(n + 1);
  for (auto& e : ed) in[e.dst].push_back(e.src);
  Fold<KIND> prog;
  std::vector<float> want(n + 1);
  // (the reference's stored order: ascending NATIVE id of the source -- Graph.h:111-150 renumbers the vertices)
  std::vector<int> nat(n + 1);
  for (int v = 1; v <= n; v++) nat[v] = G.vertexToNative(v, G.tiles_per_dim, G.nvertices);
  for (int v = 1; v <= n; v++) {
    std::sort(in[v].begin(), in[v].end(), [&](int a, int b) { return nat[a] < nat[b]; });
    bool has = false;
    float acc = 0.f;
    for (int s : in[v]) {
      if (has) prog.reduce_function(acc, x[s]); else { acc = x[s]; has = true; }
    }
    want[v] = has ? acc : x[v];
  }
  for (int v = 1; v <= n; v++) G.setVertexproperty(v, x[v]);
  G.setAllActive();
  GraphMat::run_graph_program(&prog, G, 1);
  int bad = 0, first = 0;
  for (int v = 1; v <= n; v++) {
    const float got = G.getVertexproperty(v);
    if (memcmp(&got, &want[v], 4) != 0) { if (!bad) first = v; bad++; }
  }
  printf("%s: %d of %d vertices differ from the host's ordered fold", what, bad, n);
  if (bad) printf(" (first: vertex %d, %d in-edges, got %.9g, want %.9g)", first, (int)in[first].size(), G.getVertexproperty(first), want[first]);
  printf("\n");
  return bad;
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  const int n = 60000;
  edges_t ed;
  unsigned st = 12345u;
  auto next = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  // hubs 1..4: 40000 / 20000 / 9000 / 8192 in-edges (five, three, two and one 8192-product chunks), 5..20: medium rows, the rest sparse
  const int hub_edges[4] = {40000, 20000, 9000, 8192};
  for (int h = 0; h < 4; h++)
    for (int k = 0; k < hub_edges[h]; k++) ed.push_back(GraphMat::edge_t<int>(1 + (int)(next() % n), 1 + h, 1));
  for (int h = 4; h < 20; h++)
    for (int k = 0; k < 100 + 40 * h; k++) ed.push_back(GraphMat::edge_t<int>(1 + (int)(next() % n), 1 + h, 1));
  for (int i = 0; i < n; i++) ed.push_back(GraphMat::edge_t<int>(1 + i, 1 + (i + 1) % n, 1));
  for (int i = 0; i < 3 * n; i++) ed.push_back(GraphMat::edge_t<int>(1 + (int)(next() % n), 1 + (int)(next() % n), 1));
  GraphMat::Graph<float> G;
  {
    GraphMat::edgelist_t<int> E(n, n, (int)ed.size());
    std::copy(ed.begin(), ed.end(), E.edges);
    G.ReadEdgelist(E);
    E.clear();
  }
  gm_csr_t ca;
  CHECK(gm_graph_csr(G.A, GM_DIR_OUT, &ca) == 0);
  printf("%d vertices, %d edges, %d giant rows in %d pieces\n", n, (int)ed.size(), ca.ngiant, ca.ngchunk);
  CHECK(ca.ngiant >= 3);
  // messages whose sums round at every step: the bits of a row's result depend on the order of its terms
  std::vector<float> x(n + 1);
  for (int v = 1; v <= n; v++) x[v] = 1.0f / (float)(v % 97 + 1) + (float)(v % 13) * 0.37f;
  CHECK(run_case<0>(G, ed, n, x, "a += b") == 0);
  CHECK(rows_folded_again(G) == 0);  // the speculation ran and every chunk was proven
  // every 1500th vertex sends exactly 1/64: TrickySum is not an addition on these rows
  std::vector<float> xt = x;
  for (int v = 7; v <= n; v += 1500) xt[v] = 0.015625f;
  CHECK(run_case<1>(G, ed, n, xt, "a += b, one more for b == 1/64") == 0);
  {
    const int again = rows_folded_again(G);
    printf("   giant rows whose proof failed and that were folded again in order: %d of %d\n", again, ca.ngiant);
    CHECK(again > 0);
  }
  CHECK(run_case<0>(G, ed, n, xt, "a += b on the same messages") == 0);
  CHECK(rows_folded_again(G) == 0);
  CHECK(run_case<2>(G, ed, n, x, "the smaller of a and b") == 0);
  CHECK(rows_folded_again(G) == 0);  // associative on any data: the chunk totals combine to the ordered fold's running values
  CHECK(run_case<3>(G, ed, n, x, "half of a, plus b") == 0);
  CHECK(rows_folded_again(G) > 0);   // neither an addition nor associative: every proof fails, the rows are folded in order
  // a second iteration on the results of the first (the replay's binade hints come from the pass before)
  CHECK(run_case<0>(G, ed, n, x, "a += b again") == 0);
  CHECK(rows_folded_again(G) == 0);  // (a failed proof stops the speculation for the rest of THAT run only)
  printf(failures == 0 ? "SPECULATED PASS\n" : "SPECULATED FAIL (%d)\n", failures);
  MPI_Finalize();
  return failures == 0 ? 0 : 1;
}
