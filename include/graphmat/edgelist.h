// graphmat/edgelist.h -- host edge lists with the reference's public names
// (edge_t / edgelist_t: include/GMDP/utils/edgelist.h:38-78 of the reference) and
// the binary .mtx loader (ibid. :242-334) on top of the C-ABI reader.
#ifndef GRAPHMAT_HIP_EDGELIST_H_
#define GRAPHMAT_HIP_EDGELIST_H_
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <type_traits>

#include "../graphmat_hip.h"

namespace GraphMat {

inline int get_global_nrank();
inline int get_global_myrank();

template <typename T>
struct edge_t {
  edge_t() {}
  edge_t(int _src, int _dst, T _val) : src(_src), dst(_dst), val(_val) {}
  int src;
  int dst;
  T val;
};

template <typename T>
struct edgelist_t {
  edge_t<T>* edges;
  int m;
  int n;
  int nnz;
  edgelist_t() : edges(nullptr), m(0), n(0), nnz(0) {}
  edgelist_t(int _m, int _n, int _nnz) : edges(nullptr), m(_m), n(_n), nnz(_nnz) {
    if (nnz > 0) edges = static_cast<edge_t<T>*>(malloc((size_t)nnz * sizeof(edge_t<T>)));
  }
  edgelist_t(edge_t<T>* e, int _m, int _n, int _nnz) : edges(e), m(_m), n(_n), nnz(_nnz) {}
  void clear() {
    if (edges != nullptr) free(edges);
    edges = nullptr;
    nnz = 0;
    m = 0;
    n = 0;
  }
};

namespace detail {
template <typename T>
inline int edge_value_kind() {
  if (std::is_same<T, int>::value) return GM_VAL_I32;
  if (std::is_same<T, unsigned int>::value) return GM_VAL_U32;
  if (std::is_same<T, float>::value) return GM_VAL_F32;
  if (std::is_same<T, double>::value) return GM_VAL_F64;
  return GM_VAL_RAW((int)sizeof(T));  // opaque: binary files only
}
template <typename T>
inline typename std::enable_if<std::is_constructible<T, int>::value>::type assign_one(T& v) { v = T(1); }
template <typename T>
inline typename std::enable_if<!std::is_constructible<T, int>::value>::type assign_one(T&) {}
}  // namespace detail

// Edge-list files: binary or text, with or without the "m n nnz" header, with or without
// edge values (every combination the reference's loader takes).  File naming follows the
// reference: rank r reads <prefix><r>, <prefix><r+nranks>, ... until one is missing.  As a
// convenience a plain <prefix> (no suffix) is read when <prefix>0 does not exist.
template <typename T>
void load_edgelist(const char* dir, edgelist_t<T>* edgelist, bool binaryformat = true, bool header = true,
                   bool edgeweights = true) {
  edgelist->m = edgelist->n = edgelist->nnz = 0;
  edgelist->edges = nullptr;
  const int nrank = get_global_nrank(), myrank = get_global_myrank();
  for (int i = myrank;; i += nrank) {
    std::stringstream fname_ss;
    fname_ss << dir << i;
    std::string fname = fname_ss.str();
    FILE* probe = fopen(fname.c_str(), "rb");
    if (!probe) {
      if (i == 0 && myrank == 0 && (probe = fopen(dir, "rb")) != nullptr) {  // (rank 0 alone reads an unsplit file)
        fname = dir;
      } else {
        if (i == myrank) printf("Could not open file: %s\n", fname.c_str());
        break;
      }
    }
    fclose(probe);
    printf("Reading file: %s\n", fname.c_str());
    int fm = 0, fn = 0;
    int64_t nnz = 0;
    int32_t *s = nullptr, *d = nullptr;
    void* v = nullptr;
    if (gm_edgelist_read(fname.c_str(), binaryformat ? 1 : 0, header ? 1 : 0, edgeweights ? 1 : 0,
                         detail::edge_value_kind<T>(), &fm, &fn, &nnz, &s, &d, &v) != GM_OK) {
      printf("%s\n", gm_last_error());
      exit(1);
    }
    size_t old = (size_t)edgelist->nnz;
    edgelist->edges = static_cast<edge_t<T>*>(realloc(edgelist->edges, (old + (size_t)nnz + 1) * sizeof(edge_t<T>)));
    for (int64_t k = 0; k < nnz; k++) {
      edge_t<T>& e = edgelist->edges[old + k];
      e.src = s[k];
      e.dst = d[k];
      memcpy(&e.val, (const char*)v + (size_t)k * sizeof(T), sizeof(T));
      if (!edgeweights) detail::assign_one(e.val);  // unweighted files: every edge gets (T)1 (edgelist.h:186-188 of the reference)
    }
    gm_host_free(s);
    gm_host_free(d);
    gm_host_free(v);
    edgelist->nnz += (int)nnz;
    if (fm > edgelist->m) edgelist->m = fm;
    if (fn > edgelist->n) edgelist->n = fn;
    if (fname == dir) break;
  }
  printf("Got: %d by %d  vertices\n", edgelist->m, edgelist->n);
  printf("Got: %d edges\n", edgelist->nnz);
}

// one file per rank, <prefix><rank> (write_edgelist of the reference, edgelist.h:208-240)
template <typename T>
void write_edgelist(const char* dir, const edgelist_t<T>& edgelist, bool binaryformat = true, bool header = true,
                    bool edgeweights = true) {
  std::stringstream fname_ss;
  fname_ss << dir << get_global_myrank();
  printf("Writing file: %s\n", fname_ss.str().c_str());
  const size_t ne = (size_t)edgelist.nnz;
  int32_t* s = static_cast<int32_t*>(malloc(ne * 4 + 4));
  int32_t* d = static_cast<int32_t*>(malloc(ne * 4 + 4));
  char* v = static_cast<char*>(malloc(ne * sizeof(T) + 8));
  for (size_t k = 0; k < ne; k++) {
    s[k] = edgelist.edges[k].src;
    d[k] = edgelist.edges[k].dst;
    memcpy(v + k * sizeof(T), &edgelist.edges[k].val, sizeof(T));
  }
  int rc = gm_edgelist_write(fname_ss.str().c_str(), binaryformat ? 1 : 0, header ? 1 : 0, edgeweights ? 1 : 0,
                             detail::edge_value_kind<T>(), edgelist.m, edgelist.n, (int64_t)ne, s, d, v);
  free(s);
  free(d);
  free(v);
  if (rc != GM_OK) {
    printf("%s\n", gm_last_error());
    exit(1);
  }
}

// keep the edges for which the predicate holds (public name of the reference's
// include/GMDP/utils/edgelist_transformation.h:431-443; used by src/DeltaStepping.cpp)
template <typename T>
edgelist_t<T> filter_edges(edgelist_t<T>* edgelist, bool (*filter_function)(edge_t<T>, void*), void* param = NULL) {
  edgelist_t<T> kept(edgelist->m, edgelist->n, edgelist->nnz);
  int k = 0;
  for (int i = 0; i < edgelist->nnz; i++)
    if (filter_function(edgelist->edges[i], param)) kept.edges[k++] = edgelist->edges[i];
  kept.nnz = k;
  return kept;
}

}  // namespace GraphMat
#endif
