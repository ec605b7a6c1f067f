// GraphProgram.h -- vertex-program base class of the MI355X GraphMat engine.
//
// Same public surface as the reference's include/GraphProgram.h:34-100 (enums,
// protected flags, five virtuals, getters) so that existing programs compile
// unchanged.  Differences that matter here:
//   * the device kernels never go through the vtable: run_graph_program()
//     recovers the concrete program type and calls P::send_message etc.
//     directly (see include/graphmat/kernels.hpp), so the virtuals below only
//     serve host-side callers;
//   * process_message_requires_vertexprop selects the 2- vs 3-operand kernel
//     exactly as in the reference's include/SPMV.h:67-71.
#ifndef GRAPHMAT_HIP_GRAPHPROGRAM_H_
#define GRAPHMAT_HIP_GRAPHPROGRAM_H_
#include <cstdlib>
#include <iostream>

#include "graphmat/engine.hpp"  // edge_direction, activity_type

namespace GraphMat {

template <class T, class U, class V, class E = int>  // T message, U reduced message, V vertex property, E edge
class GraphProgram {
 protected:
  edge_direction order;
  activity_type activity;
  bool process_message_requires_edge_value;  // accepted for compatibility; never consulted (as in the reference)
  bool process_message_requires_vertexprop;

 public:
  typedef T message_type;
  typedef U message_reduction_type;
  typedef V vertex_property_type;
  typedef E edge_type;

  GraphProgram()
      : order(OUT_EDGES),
        activity(ACTIVE_ONLY),
        process_message_requires_edge_value(true),
        process_message_requires_vertexprop(true) {}

  edge_direction getOrder() const { return order; }
  activity_type getActivity() const { return activity; }
  bool getProcessMessageRequiresVertexprop() const { return process_message_requires_vertexprop; }

  // A program that reaches one of these did not override what the runtime
  // needs; the reference prints and exits (GraphProgram.h:73-96), so do we.
  virtual void reduce_function(U&, const U&) const { die("reduce_function"); }
  virtual void process_message(const T&, const E, const V&, U&) const { die("process_message"); }
  virtual bool send_message(const V&, T&) const { die("send_message"); return true; }
  virtual void apply(const U&, V&) { die("apply"); }
  virtual void do_every_iteration(int /*iteration_number*/) {}

 private:
  static void die(const char* what) {
    std::cout << "Trying to use default (null) " << what << std::endl;
    exit(1);
  }
};

}  // namespace GraphMat
#endif
