// label_propagation.cpp -- a vertex program that is NOT in the library's fixed menu, written
// the way GraphMat applications are written (plain C++, no device annotations), to show the
// generic path: hipcc --hipstdpar compiles these methods into the engine's kernels.
//
// Connected components by minimum-label propagation over ALL_EDGES: every vertex starts with
// its own id as label, sends it along in- and out-edges, keeps the minimum it hears.
//
//   label_propagation graph.bin.mtx        -> prints "component <vertex> <label>" per vertex
#include <climits>
#include <cstdio>

#include "GraphMatRuntime.h"

struct Label {
  unsigned int label;
  Label() : label(UINT_MAX) {}
  bool operator!=(const Label& o) { return label != o.label; }
  friend std::ostream& operator<<(std::ostream& os, const Label& l) { return os << l.label; }
};

class MinLabel : public GraphMat::GraphProgram<unsigned int, unsigned int, Label> {
 public:
  MinLabel() {
    this->order = GraphMat::ALL_EDGES;
    this->process_message_requires_vertexprop = false;
  }
  bool send_message(const Label& v, unsigned int& m) const { m = v.label; return true; }
  void process_message(const unsigned int& m, const int, const Label&, unsigned int& r) const { r = m; }
  void reduce_function(unsigned int& a, const unsigned int& b) const { a = b < a ? b : a; }
  void apply(const unsigned int& y, Label& v) { if (y < v.label) v.label = y; }
};

// min is associative, commutative and exact: let the runtime pick any evaluation order
namespace GraphMat {
template <> struct program_traits<MinLabel> { static constexpr reduce_kind reduce = REDUCE_COMMUTATIVE; };
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  if (argc < 2) { printf("usage: %s graph.bin.mtx\n", argv[0]); return 1; }
  GraphMat::Graph<Label> G;
  G.ReadMTX(argv[1]);
  for (int v = 1; v <= G.getNumberOfVertices(); v++) {
    Label l;
    l.label = (unsigned int)v;
    G.setVertexproperty(v, l);
  }
  MinLabel prog;
  G.setAllActive();
  GraphMat::run_graph_program(&prog, G, GraphMat::UNTIL_CONVERGENCE);
  for (int v = 1; v <= G.getNumberOfVertices(); v++) printf("component %d %u\n", v, G.getVertexproperty(v).label);
  MPI_Finalize();
  return 0;
}
