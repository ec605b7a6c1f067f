// graph_converter -- convert / clean graph files (same command line as the reference's
// src/graph_converter.cpp: option names, defaults, value meanings and the order in which
// the clean-up steps are applied, :161-222 there), on this engine's edge-list I/O
// (gm_edgelist_read / gm_edgelist_write) and host-side transformations.
//
//   graph_converter [options] <input file prefix> <output file prefix>
//
// Files are named <prefix><rank>; this tool runs as one process, so <prefix>0 (an input
// without the suffix is accepted too, see graphmat/edgelist.h).
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "GraphMatRuntime.h"

namespace {

struct Settings {
  int selfloops = 0;          // 0 remove, 1 keep
  int duplicatededges = 0;    // 0 remove, 1 keep
  int uppertriangular = 0;    // orient u <= v
  int bidirectional = 0;      // add (v,u) for every (u,v)
  int inputformat = 1;        // 0 binary mtx, 1 text mtx, 2 graph file of this engine
  int outputformat = 0;
  int inputheader = 1;
  int outputheader = 1;
  int inputedgeweights = 1;
  int outputedgeweights = 1;  // 0 none, 1 keep, 2 unit, 3 random in [1, r)
  int edgeweighttype = 0;     // 0 unsigned int, 1 double, 2 float
  int nvertices = 0;          // only with --inputheader 0
  int random_range = 128;
  int randomizeID = 0;
  int nsplits = 1;
};

struct Field {
  const char* name;
  int Settings::*slot;
  bool takes_value;
  const char* help;
};
const Field kFields[] = {
    {"selfloops", &Settings::selfloops, true, "0: remove all self loops (default)  1: retain them"},
    {"duplicatededges", &Settings::duplicatededges, true, "0: remove duplicated edges (default)  1: retain them"},
    {"uppertriangular", &Settings::uppertriangular, false, "swap u and v of every edge (u,v) with u > v"},
    {"bidirectional", &Settings::bidirectional, false, "for every edge (u,v) also add (v,u)"},
    {"inputformat", &Settings::inputformat, true, "0: binary mtx  1: text mtx (default)  2: graph file written by --outputformat 2"},
    {"outputformat", &Settings::outputformat, true, "0: binary mtx (default)  1: text mtx  2: graph file (GraphMat-bin role)"},
    {"inputheader", &Settings::inputheader, true, "0: no header (see --nvertices)  1: \"n n nnz\" header (default)"},
    {"outputheader", &Settings::outputheader, true, "0: no header  1: header (default)"},
    {"inputedgeweights", &Settings::inputedgeweights, true, "0: no weights in the input  1: weights present (default)"},
    {"outputedgeweights", &Settings::outputedgeweights, true,
     "0: write no weights  1: write the weights (default)  2: unit weights  3: random weights in [1,r)"},
    {"edgeweighttype", &Settings::edgeweighttype, true, "0: int (default)  1: double  2: float"},
    {"r", &Settings::random_range, true, "range of the random weights of --outputedgeweights 3 (default 128)"},
    {"nvertices", &Settings::nvertices, true, "number of vertices (only with --inputheader 0)"},
    {"split", &Settings::nsplits, true, "deprecated"},
    {"randomizeID", &Settings::randomizeID, false, "relabel the vertices with a pseudo-random permutation"},
};
const int kNumFields = (int)(sizeof(kFields) / sizeof(kFields[0]));

void usage(const char* argv0) {
  printf("Usage: %s [options] <input mtx file prefix> <output mtx file prefix> \n", argv0);
  printf("Options:\n\t--help Print help message and exit.\n");
  for (int i = 0; i < kNumFields; i++) printf("\t--%s%s\n\t\t%s\n", kFields[i].name, kFields[i].takes_value ? " [number]" : "", kFields[i].help);
}

bool settings_ok(const Settings& o) {
  bool ok = true;
  if (o.selfloops != 0 && o.selfloops != 1) { printf("selfloops must be 0 or 1 \n"); ok = false; }
  if (o.uppertriangular == 1 && o.bidirectional == 1) { printf("Cannot be both uppertriangular and bidirectional\n"); ok = false; }
  if (o.inputedgeweights == 0 && o.outputedgeweights == 1) { printf("No input edge weights and want output edge weights\n"); ok = false; }
  if (o.nsplits != 1) { printf("Split functionality is deprecated.\n"); ok = false; }
  if (!ok) printf("Error in validating options\n");
  return ok;
}

template <typename W>
void convert(const char* in, const char* out, const Settings& o) {
  GraphMat::edgelist_t<W> el;
  if (o.inputformat == 0 || o.inputformat == 1) {
    GraphMat::load_edgelist<W>(in, &el, o.inputformat == 0, o.inputheader == 1, o.inputedgeweights == 1);
    const int side = el.m > el.n ? el.m : el.n;  // square
    el.m = el.n = side;
    if (o.nvertices > 0) {
      if (o.nvertices < side) { printf("--nvertices %d is smaller than the largest vertex id %d\n", o.nvertices, side); exit(1); }
      el.m = el.n = o.nvertices;
    }
  } else {
    GraphMat::Graph<int, W> g;
    g.ReadGraphMatBin(in);
    g.getEdgelist(el);
  }
  if (o.outputedgeweights == 3) GraphMat::random_edge_weights(&el, o.random_range);
  GraphMat::shuffle_edges(&el);
  if (o.selfloops == 0) GraphMat::remove_selfedges(&el);
  if (o.bidirectional == 1) GraphMat::create_bidirectional_edges(&el);
  if (o.uppertriangular == 1) GraphMat::convert_to_dag(&el);
  if (o.duplicatededges == 0) GraphMat::remove_duplicate_edges(&el);
  if (o.randomizeID == 1) GraphMat::randomize_edgelist_square(&el);
  if (o.outputformat == 0 || o.outputformat == 1) {
    GraphMat::write_edgelist<W>(out, el, o.outputformat == 0, o.outputheader == 1, o.outputedgeweights != 0);
  } else {
    GraphMat::Graph<int, W> g;
    g.ReadEdgelist(el);
    g.WriteGraphMatBin(out);
  }
  el.clear();
}

}  // namespace

int main(int argc, char* argv[]) {
  MPI_Init(&argc, &argv);
  Settings o;
  struct option longopts[kNumFields + 2];
  memset(longopts, 0, sizeof(longopts));
  for (int i = 0; i < kNumFields; i++) {
    longopts[i].name = kFields[i].name;
    longopts[i].has_arg = kFields[i].takes_value ? required_argument : no_argument;
    longopts[i].flag = nullptr;
    longopts[i].val = 1000 + i;
  }
  longopts[kNumFields].name = "help";
  longopts[kNumFields].val = 'h';
  for (;;) {
    int idx = 0;
    const int c = getopt_long(argc, argv, "h", longopts, &idx);
    if (c == -1) break;
    if (c == 'h') { usage(argv[0]); MPI_Finalize(); return 0; }
    if (c >= 1000 && c < 1000 + kNumFields) {
      const Field& f = kFields[c - 1000];
      o.*(f.slot) = f.takes_value ? (int)strtol(optarg, nullptr, 0) : 1;
    }
  }
  if (optind != argc - 2) { usage(argv[0]); MPI_Finalize(); return 0; }
  if (!settings_ok(o)) { MPI_Finalize(); return 1; }
  printf("Options -- \n");
  for (int i = 0; i < kNumFields; i++) printf("%s = %d \n", kFields[i].name, o.*(kFields[i].slot));
  const char* in = argv[optind];
  const char* out = argv[optind + 1];
  switch (o.edgeweighttype) {
    case 0: convert<unsigned int>(in, out, o); break;
    case 1: convert<double>(in, out, o); break;
    case 2: convert<float>(in, out, o); break;
    default: printf("Invalid edge type: %d\n", o.edgeweighttype); MPI_Finalize(); return 1;
  }
  MPI_Finalize();
  return 0;
}
