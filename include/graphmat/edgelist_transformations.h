// graphmat/edgelist_transformations.h -- host-side edge-list clean-up steps with the
// reference's public names (include/GMDP/utils/edgelist_transformation.h:37-443 and
// randomize_edgelist_square, include/GMDP/utils/edgelist.h:336-366): what
// src/graph_converter.cpp applies between reading and writing a graph file.
// Single-process: the reference's shuffle_edges moves edges between MPI ranks by source id,
// which is the identity on one rank.
#ifndef GRAPHMAT_HIP_EDGELIST_TRANSFORMATIONS_H_
#define GRAPHMAT_HIP_EDGELIST_TRANSFORMATIONS_H_
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "edgelist.h"

namespace GraphMat {

namespace detail {
// replace the list's storage by `kept` (first `count` entries are valid)
template <typename T>
void adopt_edges(edgelist_t<T>* el, edge_t<T>* kept, long long count) {
  const int m = el->m, n = el->n;
  if (el->edges) free(el->edges);
  el->edges = kept;
  el->m = m;
  el->n = n;
  el->nnz = (int)count;
}
template <typename T>
edge_t<T>* alloc_edges(size_t count) {
  return static_cast<edge_t<T>*>(malloc((count + 1) * sizeof(edge_t<T>)));
}
}  // namespace detail

// drop edges v -> v
template <typename T>
void remove_selfedges(edgelist_t<T>* edgelist) {
  edge_t<T>* kept = detail::alloc_edges<T>((size_t)edgelist->nnz);
  long long k = 0;
  for (int i = 0; i < edgelist->nnz; i++)
    if (edgelist->edges[i].src != edgelist->edges[i].dst) kept[k++] = edgelist->edges[i];
  detail::adopt_edges(edgelist, kept, k);
}

// one rank owns every edge already
template <typename T>
void shuffle_edges(edgelist_t<T>* edgelist) {
  if (get_global_nrank() != 1) {
    printf("GraphMat(HIP): shuffle_edges across ranks is not supported; run the converter as one process\n");
    exit(1);
  }
  printf("Rank %d: Before shuffle %d edges\n", get_global_myrank(), edgelist->nnz);
  printf("Rank %d: After shuffle %ld edges\n", get_global_myrank(), (long)edgelist->nnz);
}

// sort by (src, dst) and keep one edge per pair.  The reference sorts with an unstable
// parallel sort, so which duplicate's value survives is unspecified there; here the first
// in input order survives.
template <typename T>
void remove_duplicate_edges_local(edgelist_t<T>* edgelist) {
  if (edgelist->nnz <= 0) return;
  std::stable_sort(edgelist->edges, edgelist->edges + edgelist->nnz, [](const edge_t<T>& a, const edge_t<T>& b) {
    return a.src != b.src ? a.src < b.src : a.dst < b.dst;
  });
  edge_t<T>* kept = detail::alloc_edges<T>((size_t)edgelist->nnz);
  long long k = 0;
  for (int i = 0; i < edgelist->nnz; i++) {
    const edge_t<T>& e = edgelist->edges[i];
    if (k > 0 && kept[k - 1].src == e.src && kept[k - 1].dst == e.dst) continue;
    kept[k++] = e;
  }
  detail::adopt_edges(edgelist, kept, k);
}
template <typename T>
void remove_duplicate_edges(edgelist_t<T>* edgelist) {
  if (get_global_nrank() != 1) shuffle_edges(edgelist);
  remove_duplicate_edges_local(edgelist);
}

// swap the end points of each edge with probability 1/2 (libc rand(), like the reference)
template <typename T>
void randomize_edge_direction(edgelist_t<T>* edgelist) {
  for (int i = 0; i < edgelist->nnz; i++)
    if ((double)rand() / (double)RAND_MAX < 0.5) std::swap(edgelist->edges[i].src, edgelist->edges[i].dst);
}

// every edge followed by its reverse
template <typename T>
void create_bidirectional_edges(edgelist_t<T>* edgelist) {
  const size_t ne = (size_t)edgelist->nnz;
  edge_t<T>* both = detail::alloc_edges<T>(2 * ne);
  for (size_t i = 0; i < ne; i++) {
    const edge_t<T>& e = edgelist->edges[i];
    both[2 * i] = e;
    both[2 * i + 1] = edge_t<T>(e.dst, e.src, e.val);
  }
  detail::adopt_edges(edgelist, both, (long long)(2 * ne));
}

// orient every edge from the smaller to the larger id (upper triangular matrix)
template <typename T>
void convert_to_dag(edgelist_t<T>* edgelist) {
  for (int i = 0; i < edgelist->nnz; i++) {
    edge_t<T>& e = edgelist->edges[i];
    if (e.src > e.dst) std::swap(e.src, e.dst);
  }
}

// value := uniform in [1, random_range] from libc rand(), truncated to T
template <typename T>
void random_edge_weights(edgelist_t<T>* edgelist, int random_range) {
  for (int i = 0; i < edgelist->nnz; i++) {
    double t = (double)rand() / (double)RAND_MAX * (double)random_range;
    t = std::min(t, (double)random_range);
    t = std::max(t, 1.0);
    edgelist->edges[i].val = (T)t;
  }
}

// relabel the vertices of a square edge list by the reference's pseudo-random permutation:
// srand(5); draw r[i] = rand() % m for all i first; then, in order, swap slots i and r[i] of the
// identity map.  Same libc => same permutation as the reference's converter.
template <typename T>
void randomize_edgelist_square(edgelist_t<T>* edgelist) {
  const int m = edgelist->m;
  std::vector<unsigned int> map((size_t)m), pick((size_t)m);
  srand(5);
  for (int i = 0; i < m; i++) {
    map[i] = (unsigned int)i;
    pick[i] = (unsigned int)(rand() % m);
  }
  for (int i = 0; i < m; i++) std::swap(map[i], map[pick[i]]);
  for (int i = 0; i < edgelist->nnz; i++) {
    edge_t<T>& e = edgelist->edges[i];
    e.src = (int)map[e.src - 1] + 1;
    e.dst = (int)map[e.dst - 1] + 1;
  }
}

}  // namespace GraphMat
#endif
