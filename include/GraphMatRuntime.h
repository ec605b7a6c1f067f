// GraphMatRuntime.h -- run_graph_program() of the MI355X GraphMat engine.
//
// Public names of the reference's include/GraphMatRuntime.h:51-94
// (UNTIL_CONVERGENCE, run_graph_program_temp_structure, graph_program_init,
// graph_program_clear, run_graph_program) so applications written against
// GraphMat compile unchanged -- with hipcc --hipstdpar, which compiles the
// application's un-annotated vertex-program methods for the device.  The
// iteration loop itself is include/graphmat/engine.hpp.
#ifndef GRAPHMAT_HIP_RUNTIME_H_
#define GRAPHMAT_HIP_RUNTIME_H_

#include <math.h>
#include <stdlib.h>
#include <sys/time.h>

#include <utility>
#include <vector>

#include "Graph.h"
#include "GraphProgram.h"
#include "SPMV.h"
#include "graphmat/device_globals.hpp"

namespace GraphMat {

const int UNTIL_CONVERGENCE = -1;

template <class T, class U, class V>
struct run_graph_program_temp_structure {
  GraphMat::SpVec<GraphMat::DenseSegment<T> >* px;
  GraphMat::SpVec<GraphMat::DenseSegment<U> >* py;
};

// scratch message vectors, device resident (reference: GraphMatRuntime.h:59-70)
template <class T, class U, class V, class E>
struct run_graph_program_temp_structure<T, U, V> graph_program_init(const GraphProgram<T, U, V, E>& gp,
                                                                    const Graph<V, E>& g) {
  struct run_graph_program_temp_structure<T, U, V> rgpts;
  // x holds every shard's slice (the whole device id space), y this rank's rows; one rank: both nvertices
  gm_graph_desc_t d;
  gm_graph_desc(g.A, &d);
  rgpts.px = new GraphMat::SpVec<GraphMat::DenseSegment<T> >(d.ndevice);
  rgpts.py = new GraphMat::SpVec<GraphMat::DenseSegment<U> >(d.row_hi - d.row_lo);
  return rgpts;
}

template <class T, class U, class V>
void graph_program_clear(struct run_graph_program_temp_structure<T, U, V>& rgpts) {
  delete rgpts.px;
  delete rgpts.py;
}

// iterations = -1 ==> until convergence.  `Prog` is the concrete program class (deduced
// from the pointer), which is what lets the kernels inline its methods.
template <class Prog, class V, class E>
void run_graph_program(
    Prog* gp, Graph<V, E>& g, int iterations = 1,
    struct run_graph_program_temp_structure<typename detail::types_of<Prog>::type::msg,
                                            typename detail::types_of<Prog>::type::red, V>* rgpts = NULL) {
  typedef typename detail::types_of<Prog>::type PT;
  typedef typename PT::msg T;
  typedef typename PT::red U;
  static_assert(std::is_same<typename PT::vprop, V>::value, "program and graph disagree on the vertex property type");
  static_assert(std::is_same<typename PT::edge, E>::value, "program and graph disagree on the edge type");

  struct timeval init_start, init_end;
  gettimeofday(&init_start, 0);
  gm_graph_desc_t d;
  gm_graph_desc(g.A, &d);
  const int rows = d.row_hi - d.row_lo;

  GraphMat::SpVec<GraphMat::DenseSegment<T> >* px = NULL;
  GraphMat::SpVec<GraphMat::DenseSegment<U> >* py = NULL;
  T* x;
  uint32_t* xbits;
  U* y;
  uint32_t* ybits;
  if (rgpts == NULL) {
    void *p0, *p1, *p2, *p3;
    gm_graph_workspace(g.A, 1, (size_t)d.ndevice * sizeof(T) + 16, &p0);
    gm_graph_workspace(g.A, 2, ((size_t)(d.ndevice + 31) / 32 + 2) * 4, &p1);
    gm_graph_workspace(g.A, 3, (size_t)rows * sizeof(U) + 16, &p2);
    gm_graph_workspace(g.A, 4, ((size_t)(rows + 31) / 32 + 2) * 4, &p3);
    x = (T*)p0; xbits = (uint32_t*)p1; y = (U*)p2; ybits = (uint32_t*)p3;
  } else {
    px = rgpts->px;
    py = rgpts->py;
    x = (T*)px->segment->value; xbits = px->segment->bit_vector;
    y = (U*)py->segment->value; ybits = py->segment->bit_vector;
  }
  g.vertexproperty->segment->need_device();
  g.active->segment->need_device();
  detail::refresh_device_globals();  // host namespace-scope variables the program reads (e.g. MAX_DIST)
  gettimeofday(&init_end, 0);
#ifdef __TIMING
  printf("Nvertices = %d \n", g.getNumberOfVertices());
  printf("GraphMat init time = %f ms \n",
         (init_end.tv_sec - init_start.tv_sec) * 1e3 + (init_end.tv_usec - init_start.tv_usec) * 1e-3);
#endif

#ifdef __TIMING
  gm_graph_set_option(g.A, "iteration_trace", 1);  // the reference's per-iteration lines (:150-248)
#endif
  int it = detail::run_on_device<Prog, T, U, V, E>(
      gp, g.A, gp->getOrder(), gp->getActivity(), gp->getProcessMessageRequiresVertexprop(),
      (V*)g.vertexproperty->segment->value, g.active->segment->bit_vector, x, xbits, y, ybits, iterations, 0);

  g.vertexproperty->segment->device_modified();
  g.active->segment->device_modified();
  if (px) { px->segment->device_modified(); py->segment->device_modified(); }
  printf("Completed %d iterations \n", it);
}

}  // namespace GraphMat
#endif
