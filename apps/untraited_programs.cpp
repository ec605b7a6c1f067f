// untraited_programs.cpp -- what an UNCHANGED GraphMat application gets by default: programs that declare no
// program_traits (so every reduce_function is folded strictly in stored order) on a graph large enough to have
// row-blocks, 16-rows-per-wave rows, one-wave-per-row rows and GIANT rows -- whose ordered fold runs in two passes
// (k_giant_terms spreads gathers and process_message over the chip, k_giant_fold_ordered folds the products stream):
//   * a breadth-first search with 8-byte messages and reduce a = b over a SPARSE message vector (presence words),
//   * single-source shortest paths (4-byte min over edge weights, sparse),
//   * a float sum over a DENSE message vector (PageRank's arithmetic, fixed iteration count).
// Results are printed per vertex for tests/test_dropin_apps.py to compare with the oracle.
//
//   untraited_programs graph.bin.mtx <source vertex> <pagerank iterations>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "GraphMatRuntime.h"

static const unsigned int kUnreached = UINT_MAX;

struct Visit {
  unsigned int depth;
  unsigned long long parent;
  unsigned long long self;
  Visit() : depth(kUnreached), parent(0), self(0) {}
  bool operator!=(const Visit& o) const { return depth != o.depth; }
  friend std::ostream& operator<<(std::ostream& os, const Visit& v) { return os << v.depth; }
};
class PlainBfs : public GraphMat::GraphProgram<unsigned long long, unsigned long long, Visit> {
 public:
  unsigned int level;
  PlainBfs() : level(1) {
    this->order = GraphMat::OUT_EDGES;
    this->process_message_requires_vertexprop = false;
  }
  bool send_message(const Visit& v, unsigned long long& m) const { m = v.self; return true; }
  void process_message(const unsigned long long& m, const int, const Visit&, unsigned long long& r) const { r = m; }
  void reduce_function(unsigned long long& a, const unsigned long long& b) const { a = b; }
  void apply(const unsigned long long& y, Visit& v) {
    if (v.depth == kUnreached) { v.depth = level; v.parent = y; }
  }
  void do_every_iteration(int) { level++; }
};

struct Dist {
  unsigned int d;
  Dist() : d(kUnreached) {}
  bool operator!=(const Dist& o) const { return d != o.d; }
  friend std::ostream& operator<<(std::ostream& os, const Dist& v) { return os << v.d; }
};
class PlainSssp : public GraphMat::GraphProgram<unsigned int, unsigned int, Dist> {
 public:
  PlainSssp() {
    this->order = GraphMat::OUT_EDGES;
    this->process_message_requires_vertexprop = false;
  }
  bool send_message(const Dist& v, unsigned int& m) const { m = v.d; return true; }
  void process_message(const unsigned int& m, const int e, const Dist&, unsigned int& r) const { r = m + (unsigned int)e; }
  void reduce_function(unsigned int& a, const unsigned int& b) const { a = a < b ? a : b; }
  void apply(const unsigned int& y, Dist& v) { if (y < v.d) v.d = y; }
};

struct Rank {
  float pr;
  int deg;
  Rank() : pr(0.3f), deg(0) {}
  bool operator!=(const Rank& o) const { return pr != o.pr; }
  friend std::ostream& operator<<(std::ostream& os, const Rank& v) { return os << v.pr; }
};
class CountOut : public GraphMat::GraphProgram<int, int, Rank> {
 public:
  CountOut() {
    this->order = GraphMat::IN_EDGES;
    this->activity = GraphMat::ALL_VERTICES;
    this->process_message_requires_vertexprop = false;
  }
  bool send_message(const Rank&, int& m) const { m = 1; return true; }
  void process_message(const int& m, const int, const Rank&, int& r) const { r = m; }
  void reduce_function(int& a, const int& b) const { a += b; }
  void apply(const int& y, Rank& v) { v.deg = y; }
};
class PlainRank : public GraphMat::GraphProgram<float, float, Rank> {
 public:
  float alpha;
  PlainRank() : alpha(0.3f) {
    this->order = GraphMat::OUT_EDGES;
    this->activity = GraphMat::ALL_VERTICES;
    this->process_message_requires_vertexprop = false;
  }
  bool send_message(const Rank& v, float& m) const { m = v.deg == 0 ? 0.0f : v.pr / (float)v.deg; return true; }
  void process_message(const float& m, const int, const Rank&, float& r) const { r = m; }
  void reduce_function(float& a, const float& b) const { a += b; }
  void apply(const float& y, Rank& v) { v.pr = alpha + (1.0 - alpha) * y; }  // (double arithmetic on float operands, narrowed on store: src/PageRank.cpp:108-110)
};

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  if (argc < 4) { printf("usage: %s graph.bin.mtx <source vertex> <pagerank iterations>\n", argv[0]); return 1; }
  const int source = atoi(argv[2]), iters = atoi(argv[3]);
  {
    GraphMat::Graph<Visit> G;
    G.ReadMTX(argv[1]);
    for (int v = 1; v <= G.getNumberOfVertices(); v++) {
      Visit x;
      x.self = (unsigned long long)v;
      if (v == source) x.depth = 0;
      G.setVertexproperty(v, x);
    }
    G.setAllInactive();
    G.setActive(source);
    PlainBfs prog;
    GraphMat::run_graph_program(&prog, G, GraphMat::UNTIL_CONVERGENCE);
    for (int v = 1; v <= G.getNumberOfVertices(); v++) {
      const Visit x = G.getVertexproperty(v);
      if (x.depth != kUnreached) printf("bfs %d %u %lld\n", v, x.depth, v == source ? -1LL : (long long)x.parent);
    }
  }
  {
    GraphMat::Graph<Dist> G;
    G.ReadMTX(argv[1]);
    Dist z;
    z.d = 0;
    G.setVertexproperty(source, z);
    G.setAllInactive();
    G.setActive(source);
    PlainSssp prog;
    GraphMat::run_graph_program(&prog, G, GraphMat::UNTIL_CONVERGENCE);
    for (int v = 1; v <= G.getNumberOfVertices(); v++) {
      const Dist x = G.getVertexproperty(v);
      if (x.d != kUnreached) printf("sssp %d %u\n", v, x.d);
    }
  }
  {
    GraphMat::Graph<Rank> G;
    G.ReadMTX(argv[1]);
    CountOut deg;
    GraphMat::run_graph_program(&deg, G, 1);
    PlainRank pr;
    GraphMat::run_graph_program(&pr, G, iters);
    for (int v = 1; v <= G.getNumberOfVertices(); v++) {
      const Rank x = G.getVertexproperty(v);
      unsigned int bits;
      memcpy(&bits, &x.pr, 4);
      printf("pr %d %d %08x\n", v, x.deg, bits);
    }
  }
  MPI_Finalize();
  return 0;
}
