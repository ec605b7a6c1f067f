// reduce_probe_cases.cpp -- unannotated programs whose reduce_function must NOT be mistaken for float a+b.
//
// Programs without GraphMat::program_traits are probed (engine.hpp: probe_reduce_kind) and the inferred
// strategy is cross-checked on the device against the ordered fold (k_check_rows).  Four programs over the same
// graph (hub rows of ~49000 in-edges, wave rows, short rows; every vertex sends 1.0f or a random value):
//   PlainAdd   a += b                          -> probed as float add (strategy 3), check passes
//   SatAdd     a = fminf(a + b, 1e30f)         -> the probe's large operands expose it: ordered fold (0)
//   FloatMax   a = fmaxf(a, b)                 -> ordered fold (0)
//   QuirkAdd   a + b, except 20000.0f -> 20000.5f   -> indistinguishable from a+b on probe operands (3); the device
//                                                 cross-check sees the hub rows differ and falls back to the ordered fold
// Every result must equal a host fold in ascending native id order with the program's own function.
// Prints "<name> strategy-ok results-ok" per program and "PROBECASES PASS".
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "GraphMatRuntime.h"

struct FV {
  float val, sum;
  int got;
  FV() : val(0.f), sum(0.f), got(0) {}
  bool operator!=(const FV& o) { return got != o.got || sum != o.sum; }
  friend std::ostream& operator<<(std::ostream& os, const FV& v) { return os << v.sum; }
};
template <int KIND>
class Fold : public GraphMat::GraphProgram<float, float, FV> {
 public:
  Fold() {
    this->activity = GraphMat::ALL_VERTICES;
    this->process_message_requires_vertexprop = false;
  }
  bool send_message(const FV& v, float& m) const { m = v.val; return true; }
  void process_message(const float& m, const int, const FV&, float& r) const { r = m; }
  void reduce_function(float& a, const float& b) const {
    if (KIND == 0) a += b;
    else if (KIND == 1) a = fminf(a + b, 1e30f);
    else if (KIND == 2) a = fmaxf(a, b);
    else { float s = a + b; if (s == 20000.0f) s = 20000.5f; a = s; }
  }
  void apply(const float& y, FV& v) { v.sum = y; v.got = 1; }
};

static unsigned long long rng_state = 88172645463325252ull;
static unsigned int rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned int)(rng_state >> 11); }

template <int KIND>
static bool run_case(const char* name, GraphMat::edgelist_t<int>& E, const std::vector<GraphMat::edge_t<int> >& ed, int n, bool ones) {
  GraphMat::Graph<FV> G;
  G.ReadEdgelist(E);
  std::vector<float> val(n + 1);
  for (int v = 1; v <= n; v++) {
    FV p;
    p.val = ones ? 1.0f : (float)(1 + rnd() % 4096) * (1.0f / (float)(1u << (rnd() % 12)));
    val[v] = p.val;
    G.setVertexproperty(v, p);
  }
  Fold<KIND> prog;
  GraphMat::run_graph_program(&prog, G, 1);
  std::vector<std::vector<std::pair<int, int> > > in(n + 1);
  for (size_t i = 0; i < ed.size(); i++) in[ed[i].dst].push_back(std::make_pair(G.vertexToNative(ed[i].src, G.tiles_per_dim, n), (int)i));
  int bad = 0;
  for (int v = 1; v <= n; v++) {
    std::sort(in[v].begin(), in[v].end());
    FV got = G.getVertexproperty(v);
    if (in[v].empty()) { bad += (got.got != 0); continue; }
    float s = 0.f;
    for (size_t k = 0; k < in[v].size(); k++) {
      const float a = val[ed[in[v][k].second].src];
      if (k == 0) s = a; else prog.reduce_function(s, a);
    }
    if (got.got != 1 || memcmp(&got.sum, &s, 4) != 0) {
      if (bad < 3) printf("%s vertex %d: device %.9g host %.9g (%zu terms)\n", name, v, got.sum, s, in[v].size());
      bad++;
    }
  }
  printf("%s: %d mismatches => %s\n", name, bad, bad == 0 ? "results-ok" : "results-WRONG");
  return bad == 0;
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  const int n = 70000;
  std::vector<GraphMat::edge_t<int> > ed;
  for (int v = 1; v <= n; v++)
    for (int h = 1; h <= 3; h++)
      if (rnd() % 10 < 7) ed.push_back(GraphMat::edge_t<int>(v, h, 1));  // hubs: ~49000 in-edges each
  for (int d = 4; d <= 300; d++)
    for (int k = 0; k < 700; k++) ed.push_back(GraphMat::edge_t<int>(1 + rnd() % n, d, 1));  // wave rows
  for (int k = 0; k < 200000; k++) ed.push_back(GraphMat::edge_t<int>(1 + rnd() % n, 301 + rnd() % (n - 300), 1));  // short rows
  GraphMat::edgelist_t<int> E(n, n, (int)ed.size());
  std::copy(ed.begin(), ed.end(), E.edges);
  bool ok = true;
  printf("== PlainAdd\n");
  ok &= run_case<0>("PlainAdd", E, ed, n, false);
  printf("== SatAdd\n");
  ok &= run_case<1>("SatAdd", E, ed, n, false);
  printf("== FloatMax\n");
  ok &= run_case<2>("FloatMax", E, ed, n, false);
  printf("== QuirkAdd\n");
  ok &= run_case<3>("QuirkAdd", E, ed, n, true);  // all messages 1.0f: the hub rows pass through 20000.0f, deep inside the stretch the parallel replay synthesises
  E.clear();
  printf(ok ? "PROBECASES PASS\n" : "PROBECASES FAIL\n");
  MPI_Finalize();
  return ok ? 0 : 1;
}
