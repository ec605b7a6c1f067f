// swept_edge_values.cpp -- tiled_edge_update.cpp's checks with 4-byte (float) messages, so that the multiply of the rows above 64
// edges runs through the row-stationary sweep WITH edge values (kernels.hpp: k_spmv_sell<..., HAS_VALS = true>; gm_sweep_t.sval /
// lval / gval): medium rows in groups, long rows staged through LDS, giant rows' products gathered by the sweep.  The Graph<V,E>
// operations that touch edge values or the device order, on a graph whose device order is sliced (GRAPHMAT_COL_TILES=4 forces
// what the library picks by itself for large graphs):
//   (1) applyToAllEdges, device-functor form and host function-pointer form (include/Graph.h of the reference:
//       :395-402 -> GMDP applyedges.h:38-78), then a weighted SpMV that reads the rewritten values through the tiles;
//   (2) shareVertexProperty (:300-305) between a tiled graph and a second graph whose vertices with edges are NOT a
//       subset of the first one's (the relayouted graph then cannot reuse the tiles and must still multiply correctly),
//       and one whose vertices are a subset (it reuses them).
// Every result is compared with a host evaluation; small integer-valued floats (every sum below 2^24), so any fold order gives the same bits --
// what is tested is WHICH values the multiply reads.  Prints "SWEPTEDGES PASS" and exits 0.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "GraphMatRuntime.h"

static int failures = 0;
#define CHECK(cond)                                                    \
  do {                                                                 \
    if (!(cond)) { printf("CHECK failed: %s (line %d)\n", #cond, __LINE__); failures++; } \
  } while (0)

// y = A^T x over (mul, add): every vertex sends its value, products are summed
class SumIn : public GraphMat::GraphProgram<float, float, float> {
 public:
  SumIn() { this->process_message_requires_vertexprop = false; this->activity = GraphMat::ALL_VERTICES; }
  bool send_message(const float& v, float& m) const { m = v; return true; }
  void process_message(const float& m, const int e, const float&, float& r) const { r = m * (float)e; }
  void reduce_function(float& a, const float& b) const { a += b; }
  void apply(const float& y, float& v) { v = y; }
};

static void edge_fn(int* e, const float& src, const float& dst, void* param) {
  *e = ((int)src % 5) + (*(int*)param) * ((int)dst % 3) + 1;
}

typedef std::vector<GraphMat::edge_t<int> > edges_t;

// hubs 1..nhub receive from many sources (rows far above the 64-edge tiling threshold), the rest is sparse;
// `lo..hi` bounds the vertex ids that take part at all
static edges_t make_edges(int lo, int hi, int nhub, unsigned seed) {
  edges_t ed;
  unsigned st = seed;
  auto next = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  const int span = hi - lo + 1;
  // (hub 0: 6000 in-edges -- a giant row at this size; hubs 1..3: 1500..2100 -- long rows of the sweep; the others 100..230: medium rows)
  for (int h = 0; h < nhub; h++)
    for (int k = 0; k < (h == 0 ? 6000 : h <= 3 ? 1200 + 300 * h : 70 + 7 * h); k++) ed.push_back(GraphMat::edge_t<int>(lo + (int)(next() % span), lo + h, 1 + (int)(next() % 4)));
  for (int i = 0; i < span; i++) ed.push_back(GraphMat::edge_t<int>(lo + i, lo + (i + 1) % span, 2));  // every vertex of the range has an edge
  for (int i = 0; i < 4 * span; i++) ed.push_back(GraphMat::edge_t<int>(lo + (int)(next() % span), lo + (int)(next() % span), 1 + (int)(next() % 4)));
  return ed;
}

static GraphMat::edgelist_t<int> as_list(const edges_t& ed, int n) {
  GraphMat::edgelist_t<int> E(n, n, (int)ed.size());
  std::copy(ed.begin(), ed.end(), E.edges);
  return E;
}

// one SumIn iteration on G against the host evaluation with edge values `val_of`
template <class F>
static bool spmv_matches(GraphMat::Graph<float>& G, const edges_t& ed, int n, const std::vector<float>& x, F val_of) {
  std::vector<float> want(n + 1, 0.0f);
  std::vector<char> has(n + 1, 0);
  for (auto& e : ed) { want[e.dst] += x[e.src] * val_of(e); has[e.dst] = 1; }
  for (int v = 1; v <= n; v++) G.setVertexproperty(v, x[v]);
  SumIn prog;
  G.setAllActive();
  GraphMat::run_graph_program(&prog, G, 1);
  bool ok = true;
  int shown = 0;
  std::vector<int> indeg(n + 1, 0);
  for (auto& e : ed) indeg[e.dst]++;
  for (int v = 1; v <= n; v++) {
    const float got = G.getVertexproperty(v), exp = has[v] ? want[v] : x[v];
    if (got != exp) { ok = false; if (shown++ < 8) printf("  vertex %d (in-degree %d): %g, expected %g\n", v, indeg[v], got, exp); }
  }
  return ok;
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  setenv("GRAPHMAT_COL_TILES", "4", 1);
  // "blocked": the rows of at most 64 edges also go through the column-blocked stream with their edge values (kernels.hpp: k_spmv_blocked<..., HAS_VALS = true>;
  // gm_blocked_t.eval / epos), which the library builds by itself only for large graphs without skew
  const bool blocked = argc > 1 && std::string(argv[1]) == "blocked";
  if (blocked) CHECK(gm_set_option("blocked_rows", 1) == 0);
  const int n = 6000;
  std::vector<float> x(n + 1);
  for (int v = 1; v <= n; v++) x[v] = (float)(v % 7 + 1);

  {  // ---- (1) edge values rewritten on a tiled graph ----
    edges_t ed = make_edges(1, n, 24, 7u);
    GraphMat::Graph<float> G;
    GraphMat::edgelist_t<int> E = as_list(ed, n);
    G.ReadEdgelist(E);
    E.clear();
    int nt = 1;
    gm_graph_tiles(G.A, GM_DIR_OUT, &nt);
    gm_sweep_t sw;
    CHECK(gm_graph_sweep(G.A, &sw) == 0);
    gm_csr_t ca;
    CHECK(gm_graph_csr(G.A, GM_DIR_OUT, &ca) == 0);
    printf("graph 1: %d column tiles, %d slices, sweep: %d rows (%d long), value bytes %d, %lld giant-row edges gathered by the sweep, %d giant rows\n", nt, sw.nslices, sw.nrows,
           sw.nrows_long, sw.val_bytes, (long long)sw.ngiant_edges, ca.ngiant);
    CHECK(nt > 1);
    gm_blocked_t bl;
    CHECK(gm_graph_blocked(G.A, &bl) == 0);
    if (blocked) { printf("graph 1: column-blocked stream: %d short rows, %lld entries, value bytes %d\n", bl.nrows, (long long)bl.nentries, bl.val_bytes); CHECK(bl.nrows > 0 && bl.val_bytes == 4 && bl.eval && bl.epos); }
    else CHECK(bl.nrows == 0);
    CHECK(sw.nrows > sw.nrows_long && sw.nrows_long > 0 && sw.val_bytes == 4 && ca.ngiant > 0 && sw.ngiant_edges > 0);
    CHECK(spmv_matches(G, ed, n, x, [](const GraphMat::edge_t<int>& e) { return (float)e.val; }));
    // device functor form: the whole-CSR values are rewritten in place, the tile copies must follow
    for (int v = 1; v <= n; v++) G.setVertexproperty(v, x[v]);
    const int s2 = 7;
    G.applyToAllEdges([s2](int* e, const float& src, const float& dst) { *e = ((int)src % 5) + s2 * ((int)dst % 3) + 1; });
    CHECK(spmv_matches(G, ed, n, x, [&](const GraphMat::edge_t<int>& e) { return (float)(((int)x[e.src] % 5) + s2 * ((int)x[e.dst] % 3) + 1); }));
    // host function-pointer form
    for (int v = 1; v <= n; v++) G.setVertexproperty(v, x[v]);
    int s3 = 11;
    G.applyToAllEdges(edge_fn, &s3);
    CHECK(spmv_matches(G, ed, n, x, [&](const GraphMat::edge_t<int>& e) { return (float)(((int)x[e.src] % 5) + s3 * ((int)x[e.dst] % 3) + 1); }));
  }
  {  // ---- (2) shareVertexProperty with a tiled graph ----
    // A uses the vertices 1..n/2 only; B1 uses all of them (NOT a subset of A's: its other half has no edges in A and
    // sits behind A's last tile in A's device order); B2 uses 1..n/4 (a subset).
    edges_t ea = make_edges(1, n / 2, 24, 11u), eb1 = make_edges(1, n, 24, 13u), eb2 = make_edges(1, n / 4, 24, 17u);
    GraphMat::Graph<float> A, B1, B2;
    GraphMat::edgelist_t<int> LA = as_list(ea, n), LB1 = as_list(eb1, n), LB2 = as_list(eb2, n);
    A.ReadEdgelist(LA);
    B1.ReadEdgelist(LB1);
    B2.ReadEdgelist(LB2);
    LA.clear(); LB1.clear(); LB2.clear();
    int nta = 1, nt1 = 1, nt2 = 1;
    gm_graph_tiles(A.A, GM_DIR_OUT, &nta);
    CHECK(nta > 1);
    B1.shareVertexProperty(A);
    B2.shareVertexProperty(A);
    gm_graph_tiles(B1.A, GM_DIR_OUT, &nt1);
    gm_graph_tiles(B2.A, GM_DIR_OUT, &nt2);
    printf("graph A: %d tiles; after shareVertexProperty: B1 (not a subset) %d, B2 (subset) %d\n", nta, nt1, nt2);
    CHECK(nt1 == 1);    // cannot adopt A's tiles
    CHECK(nt2 == nta);  // can
    auto plain = [](const GraphMat::edge_t<int>& e) { return (float)e.val; };
    CHECK(spmv_matches(B1, eb1, n, x, plain));
    CHECK(spmv_matches(B2, eb2, n, x, plain));
    CHECK(spmv_matches(A, ea, n, x, plain));
    // the vector really is shared: what B2's run left is what A sees
    bool same = true;
    for (int v = 1; v <= n; v++) same &= (A.getVertexproperty(v) == B2.getVertexproperty(v));
    CHECK(same);
    // edge values of a relayouted, tiled graph
    for (int v = 1; v <= n; v++) B2.setVertexproperty(v, x[v]);
    const int s4 = 5;
    B2.applyToAllEdges([s4](int* e, const float& src, const float& dst) { *e = ((int)src % 5) + s4 * ((int)dst % 3) + 1; });
    CHECK(spmv_matches(B2, eb2, n, x, [&](const GraphMat::edge_t<int>& e) { return (float)(((int)x[e.src] % 5) + s4 * ((int)x[e.dst] % 3) + 1); }));
  }
  printf(failures == 0 ? "SWEPTEDGES PASS\n" : "SWEPTEDGES FAIL (%d)\n", failures);
  MPI_Finalize();
  return failures == 0 ? 0 : 1;
}
