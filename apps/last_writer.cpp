// last_writer.cpp -- an a=b ("last writer wins") program whose ACTIVE vertices can change again.
//
// BFS, the usual a=b program, never rewrites a vertex that has sent a message, which hides a hazard of the
// list-based top-down step: k_push_finish applies y to destinations while other lanes still fetch the
// messages of the (active) sources, so a message evaluated on demand from the sender's vertex property
// could already see this step's rewrite.  Here every vertex keeps the largest label it has seen
// (apply: label = max(label, y)), the message is the sender's label BEFORE the step, and the reduction
// keeps the message of the in-neighbour with the largest native id (GraphMat's order) -- so senders are
// rewritten in the very step they send in.  The device result (labels, update counts, iteration count)
// must equal a host restatement of the synchronous iteration.  Prints "LASTWRITER PASS".
#include <algorithm>
#include <cstdio>
#include <vector>

#include "GraphMatRuntime.h"

struct RV {
  int label, updates;
  RV() : label(0), updates(0) {}
  bool operator!=(const RV& o) { return label != o.label || updates != o.updates; }
  friend std::ostream& operator<<(std::ostream& os, const RV& v) { return os << v.label; }
};
class Relay : public GraphMat::GraphProgram<int, int, RV> {
 public:
  Relay() {
    this->order = GraphMat::OUT_EDGES;
    this->activity = GraphMat::ACTIVE_ONLY;
    this->process_message_requires_vertexprop = false;
  }
  bool send_message(const RV& v, int& m) const { m = v.label; return true; }
  void process_message(const int& m, const int, const RV&, int& r) const { r = m; }
  void reduce_function(int& a, const int& b) const { a = b; }
  void apply(const int& y, RV& v) {
    if (y > v.label) { v.label = y; v.updates++; }
  }
};

// counts the iterations the runtime performs (do_every_iteration is the host hook)
class CountedRelay : public Relay {
 public:
  int n;
  CountedRelay() : n(0) {}
  void do_every_iteration(int) { n++; }
};

static unsigned long long rng_state = 0x2545F4914F6CDD1Dull;
static unsigned int rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned int)(rng_state >> 11); }

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  const int n = 30000, ring = 6000;
  std::vector<GraphMat::edge_t<int> > ed;
  for (int v = 1; v <= ring; v++) {
    ed.push_back(GraphMat::edge_t<int>(v, v % ring + 1, 1));                 // the ring: everybody sends and receives
    ed.push_back(GraphMat::edge_t<int>(v, (v * 7 + 3) % ring + 1, 1));       // chords: several in-neighbours per vertex
    if (v % 3 == 0) ed.push_back(GraphMat::edge_t<int>(v, (v * 11) % ring + 1, 1));
  }
  for (int k = 0; k < 60000; k++) ed.push_back(GraphMat::edge_t<int>(1 + rnd() % ring, ring + 1 + rnd() % (n - ring), 1));  // fan-out
  for (int k = 0; k < 20000; k++) ed.push_back(GraphMat::edge_t<int>(ring + 1 + rnd() % (n - ring), 1 + rnd() % n, 1));       // and back
  GraphMat::edgelist_t<int> E(n, n, (int)ed.size());
  std::copy(ed.begin(), ed.end(), E.edges);

  GraphMat::Graph<RV> G;
  G.ReadEdgelist(E);
  std::vector<int> label(n + 1, 0), updates(n + 1, 0);
  std::vector<char> act(n + 1, 0);
  G.setAllInactive();
  for (int v = 1; v <= n; v++) {
    RV p;
    p.label = (v <= ring) ? (int)(1 + rnd() % 1000000) : 0;
    label[v] = p.label;
    G.setVertexproperty(v, p);
    if (v <= ring && v % 2 == 0) { act[v] = 1; G.setActive(v); }  // 3000 active vertices: list-based top-down steps
  }
  CountedRelay counted;
  GraphMat::run_graph_program(&counted, G, GraphMat::UNTIL_CONVERGENCE);

  // host restatement of the synchronous iteration
  std::vector<std::vector<std::pair<int, int> > > in(n + 1);  // (native id of the source, source)
  for (size_t i = 0; i < ed.size(); i++)
    in[ed[i].dst].push_back(std::make_pair(G.vertexToNative(ed[i].src, G.tiles_per_dim, n), ed[i].src));
  for (int v = 1; v <= n; v++) std::sort(in[v].begin(), in[v].end());
  int iters = 0;
  while (true) {
    std::vector<int> newlabel(label);
    std::vector<char> newact(n + 1, 0);
    bool any = false;
    for (int v = 1; v <= n; v++) {
      int y = 0;
      bool has = false;
      for (size_t k = 0; k < in[v].size(); k++)
        if (act[in[v][k].second]) { y = label[in[v][k].second]; has = true; }  // ascending native id: the last present one stays
      if (has && y > label[v]) { newlabel[v] = y; updates[v]++; newact[v] = 1; any = true; }
    }
    label.swap(newlabel);
    act.swap(newact);
    iters++;
    if (!any) break;
  }
  int bad = 0;
  for (int v = 1; v <= n; v++) {
    RV got = G.getVertexproperty(v);
    if (got.label != label[v] || got.updates != updates[v]) {
      if (bad < 5) printf("vertex %d: device (%d, %d updates) host (%d, %d updates)\n", v, got.label, got.updates, label[v], updates[v]);
      bad++;
    }
  }
  E.clear();
  printf("%d mismatches; %d iterations on the device, %d on the host\n", bad, counted.n, iters);
  printf(bad == 0 && counted.n == iters ? "LASTWRITER PASS\n" : "LASTWRITER FAIL\n");
  MPI_Finalize();
  return (bad == 0 && counted.n == iters) ? 0 : 1;
}
