// active_float_sum.cpp -- exactness of the float-sum strategies with a SPARSE message vector.
//
// A user program (float messages, ACTIVE_ONLY, reduce a += b, declared REDUCE_F32_ADD) runs one
// iteration on a graph that has giant rows (hubs), wave rows and short rows, with only part of
// the vertices active.  The device result must equal, bit for bit, a host fold of each vertex's
// active in-neighbours in ascending NATIVE id order (the reference's reduction order), first
// present message assigning.  This covers the products-stream presence words of the giant-row
// passes, which the dense PageRank path never uses.  Prints "FLOATSUM PASS".
#include <algorithm>
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
class FloatSum : public GraphMat::GraphProgram<float, float, FV> {
 public:
  FloatSum() { this->process_message_requires_vertexprop = false; }
  bool send_message(const FV& v, float& m) const { m = v.val; return true; }
  void process_message(const float& m, const int, const FV&, float& r) const { r = m; }
  void reduce_function(float& a, const float& b) const { a += b; }
  void apply(const float& y, FV& v) { v.sum = y; v.got = 1; }
};
namespace GraphMat {
template <> struct program_traits<FloatSum> { static constexpr reduce_kind reduce = REDUCE_F32_ADD; };
}

static unsigned long long rng_state = 88172645463325252ull;
static unsigned int rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned int)(rng_state >> 11); }

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  const int n = 70000;
  std::vector<GraphMat::edge_t<int> > ed;
  for (int v = 1; v <= n; v++)
    for (int h = 1; h <= 3; h++)
      if (rnd() % 10 < 7) ed.push_back(GraphMat::edge_t<int>(v, h, 1));  // hubs: ~49000 in-edges each
  for (int d = 4; d <= 300; d++)
    for (int k = 0; k < 700; k++) ed.push_back(GraphMat::edge_t<int>(1 + rnd() % n, d, 1));  // wave rows (with duplicates)
  for (int k = 0; k < 200000; k++) ed.push_back(GraphMat::edge_t<int>(1 + rnd() % n, 301 + rnd() % (n - 300), 1));  // short rows
  GraphMat::edgelist_t<int> E(n, n, (int)ed.size());
  std::copy(ed.begin(), ed.end(), E.edges);

  GraphMat::Graph<FV> G;
  G.ReadEdgelist(E);
  std::vector<float> val(n + 1);
  std::vector<char> act(n + 1, 0);
  G.setAllInactive();
  for (int v = 1; v <= n; v++) {
    FV p;
    // magnitudes over many binades, some with few mantissa bits (round-to-even ties)
    p.val = (float)(1 + rnd() % 4096) * (1.0f / (float)(1u << (rnd() % 20)));
    if (rnd() % 5 == 0) p.val = (float)(1 + rnd() % 16);
    val[v] = p.val;
    G.setVertexproperty(v, p);
    if (rnd() % 10 < 6) { act[v] = 1; G.setActive(v); }
  }
  FloatSum prog;
  GraphMat::run_graph_program(&prog, G, 1);

  // host fold: per destination, active sources in ascending native id, duplicates in input order
  std::vector<std::vector<std::pair<int, int> > > in(n + 1);
  for (size_t i = 0; i < ed.size(); i++)
    if (act[ed[i].src]) in[ed[i].dst].push_back(std::make_pair(G.vertexToNative(ed[i].src, G.tiles_per_dim, n), (int)i));
  int bad = 0, giant = 0;
  for (int v = 1; v <= n; v++) {
    std::sort(in[v].begin(), in[v].end());
    FV got = G.getVertexproperty(v);
    if (in[v].empty()) { bad += (got.got != 0); continue; }
    float s = 0.f;
    bool has = false;
    for (size_t k = 0; k < in[v].size(); k++) {
      float a = val[ed[in[v][k].second].src];
      if (has) s += a; else { s = a; has = true; }
    }
    if (in[v].size() > 4096) giant++;
    if (got.got != 1 || memcmp(&got.sum, &s, 4) != 0) {
      if (bad < 5) printf("vertex %d: device %.9g host %.9g (%zu terms)\n", v, got.sum, s, in[v].size());
      bad++;
    }
  }
  E.clear();
  printf("%d mismatches; %d rows with more than 4096 present terms\n", bad, giant);
  printf(bad == 0 && giant >= 3 ? "FLOATSUM PASS\n" : "FLOATSUM FAIL\n");
  MPI_Finalize();
  return bad == 0 ? 0 : 1;
}
