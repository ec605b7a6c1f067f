// bfs_bottom_up.cpp -- breadth-first search written against the GraphMat surface (depth and
// parent per vertex, parent = the last message folded in, like the reference's BFS program)
// plus the ONE piece of knowledge the runtime cannot discover by itself: a vertex that already
// has a depth ignores further messages (program_row_filter).  With it the engine skips such
// rows in the multiply ("bottom-up" levels) without changing any result; without the trait the
// same program runs, only slower.  It also states its reduction strategy (a = b: the last message
// wins, program_traits<...>::reduce = REDUCE_LAST): a program that declares nothing gets the
// ordered fold -- always exact, and slower here (no backward scans, no top-down steps).
//
//   bfs_bottom_up graph.bin.mtx <source vertex>   -> "vertex <v> depth <d> parent <p>" per reached vertex
#include <climits>
#include <cstdio>
#include <cstdlib>

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

class LevelBfs : public GraphMat::GraphProgram<unsigned long long, unsigned long long, Visit> {
 public:
  unsigned int level;
  LevelBfs() : level(1) {
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

namespace GraphMat {
template <> struct program_traits<LevelBfs> { static constexpr reduce_kind reduce = REDUCE_LAST; };
template <> struct program_row_filter<LevelBfs> {
  static constexpr bool enabled = true;
  static bool wants(const LevelBfs&, const Visit& v) { return v.depth == kUnreached; }
};
}  // namespace GraphMat

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  if (argc < 3) { printf("usage: %s graph.bin.mtx <source vertex>\n", argv[0]); return 1; }
  const int source = atoi(argv[2]);
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
  LevelBfs prog;
  GraphMat::run_graph_program(&prog, G, GraphMat::UNTIL_CONVERGENCE);
  for (int v = 1; v <= G.getNumberOfVertices(); v++) {
    const Visit x = G.getVertexproperty(v);
    if (x.depth != kUnreached) printf("vertex %d depth %u parent %lld\n", v, x.depth, v == source ? -1LL : (long long)x.parent);
  }
  MPI_Finalize();
  return 0;
}
