// SPMV.h -- one multiply+reduce pass of a vertex program over the graph.
//
// Public names of the reference's include/SPMV.h:62-95: SpMSpV uses A (rows =
// sources: IN_EDGES programs), SpMTSpV uses AT (rows = destinations: OUT_EDGES
// programs); both pick the 3-operand kernel when the program asks for the
// destination vertex property.  Here a pass is one or two kernel launches
// (row-blocks + long rows) over the CSR in HBM instead of a walk over DCSC tiles.
#ifndef GRAPHMAT_HIP_SPMV_H_
#define GRAPHMAT_HIP_SPMV_H_
#include "Graph.h"
#include "GraphProgram.h"

namespace GraphMat {

namespace detail {
template <class T, class U, class V, class E>
struct program_types {
  typedef T msg;
  typedef U red;
  typedef V vprop;
  typedef E edge;
};
template <class T, class U, class V, class E>
program_types<T, U, V, E> deduce_program_types(const GraphProgram<T, U, V, E>*);
template <class Prog>
struct types_of {
  typedef decltype(deduce_program_types((Prog*)nullptr)) type;
};

template <class Prog, class V, class E, class XV, class YV>
void spmv_pass(Graph<V, E>& G, const Prog* gp, int dir, XV* x, YV* y, int accumulate) {
  typedef typename types_of<Prog>::type PT;
  typedef typename PT::msg T;
  typedef typename PT::red U;
  gm_csr_t c;
  if (gm_graph_csr(G.A, dir, &c) != GM_OK) { printf("%s\n", gm_last_error()); exit(1); }
  dev::ProgArg<Prog> pa = dev::make_prog_arg(gp);
  G.vertexproperty->segment->need_device();
  x->segment->need_device();
  y->segment->need_device();
  int launches = 0;
  const T* xv = (const T*)x->segment->value;
  U* yv = (U*)y->segment->value;
  gm_engine_options_t opt;
  gm_graph_engine_options(G.A, &opt);
  const Launch L{G.A, (hipStream_t)0, opt, &launches, nullptr, nullptr};
  launch_spmv_vp<Prog, T, U, V, E>(gp->getProcessMessageRequiresVertexprop(), L, pa, c, xv, (const uint32_t*)x->segment->bit_vector,
                                   (const V*)G.vertexproperty->segment->value, yv, y->segment->bit_vector, accumulate, reduce_kind_of<Prog, U>(gp));
  GM_HIP_OK(hipStreamSynchronize(0));
  y->segment->device_modified();
}
}  // namespace detail

// y (+)= A (x) x : messages travel destination -> source ("IN_EDGES").  y accumulates into
// entries already present, like the reference (DenseSegment::initialize only clears when
// uninitialised).
template <class Prog, class V, class E, class T, class U>
void SpMSpV(Graph<V, E>& G, const Prog* gp, SpVec<DenseSegment<T> >* x, SpVec<DenseSegment<U> >* y) {
  detail::spmv_pass(G, gp, GM_DIR_IN, x, y, 1);
}
// y (+)= A^T (x) x : messages travel source -> destination ("OUT_EDGES")
template <class Prog, class V, class E, class T, class U>
void SpMTSpV(Graph<V, E>& G, const Prog* gp, SpVec<DenseSegment<T> >* x, SpVec<DenseSegment<U> >* y) {
  detail::spmv_pass(G, gp, GM_DIR_OUT, x, y, 1);
}

}  // namespace GraphMat
#endif
