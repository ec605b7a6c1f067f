// Graph.h -- Graph<V,E> of the MI355X GraphMat engine.
//
// Same public surface as the reference's include/Graph.h:58-107 (fields
// nvertices/nnz/..., ReadMTX, ReadEdgelist, set/get vertex properties, activity,
// applyTo*), with different internals:
//   * adjacency: one gm_graph_t (C-ABI, include/graphmat_hip.h) holding CSR by
//     destination and CSR by source in HBM -- the roles of the reference's AT and A
//     (SpMat<DCSCTile<E>>), built on the device;
//   * vertex properties and the active set are device arrays in native order with a
//     lazily synchronised host mirror, because applications touch single vertices
//     from host loops before/after a run (src/BFS.cpp:114-119, src/SGD.cpp:176-184
//     of the reference) while the iteration loop itself is device resident.
// Vertex ids at this API are 1-based, as in the reference.
#ifndef GRAPHMAT_HIP_GRAPH_H_
#define GRAPHMAT_HIP_GRAPH_H_

#include <sys/time.h>

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#if !defined(GRAPHMAT_NO_MPI) && defined(__has_include)
#if __has_include(<mpi.h>)
#include <mpi.h>
#define GRAPHMAT_HAVE_REAL_MPI 1
#endif
#endif
#ifndef GRAPHMAT_HAVE_REAL_MPI
#include "graphmat/mpi_single.h"
#endif

#include "graphmat/edgelist.h"
#include "graphmat/edgelist_transformations.h"
#include "graphmat/device_globals.hpp"
#include "graphmat/engine.hpp"

namespace GraphMat {

inline int get_global_nrank() {
  int n = 1;
  MPI_Comm_size(MPI_COMM_WORLD, &n);
  return n;
}
inline int get_global_myrank() {
  int r = 0;
  MPI_Comm_rank(MPI_COMM_WORLD, &r);
  return r;
}
inline double get_compression_threshold() { return 0.5; }

class Serializable {};  // tag of the reference (gmdp.h:80); such message types are host-only there and unsupported here

inline double sec(struct timeval start, struct timeval end) {
  return ((double)(((end.tv_sec * 1000000 + end.tv_usec) - (start.tv_sec * 1000000 + start.tv_usec)))) / 1.0e6;
}

template <class T>
void AddFn(const T& a, const T& b, T* c, void* vsp) {
  *c = a + b;
}

// Host mirrors live in pinned memory: an application typically touches every vertex on the host
// between runs (src/BFS.cpp:114-119 of the reference), so each run starts with a full upload,
// which from pageable memory costs more than the traversal itself (100 MB at RMAT-22: 2.5-7 ms).
template <class T>
struct PinnedAllocator {
  typedef T value_type;
  PinnedAllocator() {}
  template <class O>
  PinnedAllocator(const PinnedAllocator<O>&) {}
  T* allocate(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n * sizeof(T) > 0 ? n * sizeof(T) : 1, hipHostMallocDefault) != hipSuccess || !p) {
      printf("GraphMat(HIP): could not allocate %zu bytes of pinned host memory\n", n * sizeof(T));
      exit(1);
    }
    return static_cast<T*>(p);
  }
  void deallocate(T* p, size_t) { (void)hipHostFree(p); }
  template <class O>
  bool operator==(const PinnedAllocator<O>&) const { return true; }
  template <class O>
  bool operator!=(const PinnedAllocator<O>&) const { return false; }
};

// ---- device vector with the reference's "dense segment" semantics ---------------------
// value[capacity] + presence bits (bit i&31 of word i>>5), cf. the reference's
// include/GMDP/vectors/DenseSegment.h:423-640; here both live in HBM, with an
// optional host mirror.
template <class T>
class DenseSegment {
 public:
  typedef typename std::conditional<std::is_same<T, bool>::value, unsigned char, T>::type store_t;
  int capacity;
  int logical;           // entries [0, logical) exist (a shard's slice may end in padding slots: their bits stay clear)
  int num_ints;
  store_t* value;        // device; nullptr for bits-only vectors (the active set)
  uint32_t* bit_vector;  // device
  bool owns_device;
  bool mirrored;
  std::vector<store_t, PinnedAllocator<store_t> > hvalue;
  std::vector<uint32_t, PinnedAllocator<uint32_t> > hbits;
  bool host_valid, dev_valid;

  DenseSegment(int n, bool with_values)
      : capacity(n), logical(n), num_ints((n + 31) / 32), value(nullptr), bit_vector(nullptr), owns_device(true), mirrored(true),
        host_valid(true), dev_valid(false) {
    // one spare word so 64-wide kernels may write a full pair of words at the tail
    GM_HIP_OK(hipMalloc((void**)&bit_vector, (size_t)(num_ints + 2) * 4));
    GM_HIP_OK(hipMemset(bit_vector, 0, (size_t)(num_ints + 2) * 4));
    if (with_values) {
      GM_HIP_OK(hipMalloc((void**)&value, std::max<size_t>((size_t)n * sizeof(store_t), 16)));
      hvalue.resize(n);
    }
    hbits.assign(num_ints, 0u);
  }
  // borrow existing device arrays (C-ABI fixed-menu entry points): no host mirror
  DenseSegment(int n, store_t* d_value, uint32_t* d_bits)
      : capacity(n), logical(n), num_ints((n + 31) / 32), value(d_value), bit_vector(d_bits), owns_device(false), mirrored(false),
        host_valid(false), dev_valid(true) {}
  ~DenseSegment() {
    if (owns_device) {
      if (value) (void)hipFree(value);
      if (bit_vector) (void)hipFree(bit_vector);
    }
  }
  DenseSegment(const DenseSegment&) = delete;
  DenseSegment& operator=(const DenseSegment&) = delete;

  void need_host() {
    if (!mirrored) { printf("GraphMat(HIP): host access to a device-only vector\n"); exit(1); }
    if (host_valid) return;
    if (value) GM_HIP_OK(hipMemcpy(hvalue.data(), value, (size_t)capacity * sizeof(store_t), hipMemcpyDeviceToHost));
    GM_HIP_OK(hipMemcpy(hbits.data(), bit_vector, (size_t)num_ints * 4, hipMemcpyDeviceToHost));
    host_valid = true;
  }
  void need_device() {
    if (dev_valid) return;
    if (value && capacity) GM_HIP_OK(hipMemcpy(value, hvalue.data(), (size_t)capacity * sizeof(store_t), hipMemcpyHostToDevice));
    if (num_ints) GM_HIP_OK(hipMemcpy(bit_vector, hbits.data(), (size_t)num_ints * 4, hipMemcpyHostToDevice));
    dev_valid = true;
  }
  void host_modified() { need_host(); dev_valid = false; }
  void device_modified() { host_valid = false; dev_valid = true; }

  void setAllBits(bool on) {  // exactly `logical` bits (DenseSegment.h:617-633 sets exactly `capacity`)
    host_modified();
    std::fill(hbits.begin(), hbits.end(), 0u);
    if (!on) return;
    for (int w = 0; w < logical / 32; w++) hbits[w] = 0xffffffffu;
    if (logical & 31) hbits[logical / 32] = (1u << (logical & 31)) - 1u;
  }
};

template <class Segment>
class SpVec;

template <class T>
class SpVec<DenseSegment<T> > {
 public:
  int n;
  int nsegments;
  DenseSegment<T>* segment;
  SpVec(int _n, bool with_values = true) : n(_n), nsegments(1), segment(new DenseSegment<T>(_n, with_values)) {}
  SpVec(int _n, typename DenseSegment<T>::store_t* d_value, uint32_t* d_bits)
      : n(_n), nsegments(1), segment(new DenseSegment<T>(_n, d_value, d_bits)) {}
  ~SpVec() { delete segment; }
  void setAll(const T& v) {
    segment->setAllBits(true);
    for (auto& e : segment->hvalue) e = v;
  }
  void set(int idx, const T& v) {  // 1-based native index
    segment->host_modified();
    if (segment->value) segment->hvalue[idx - 1] = v;
    segment->hbits[(idx - 1) >> 5] |= (1u << ((idx - 1) & 31));
  }
  void unset(int idx) {
    segment->host_modified();
    segment->hbits[(idx - 1) >> 5] &= ~(1u << ((idx - 1) & 31));
  }
  void get(int idx, T* out) const {
    segment->need_host();
    *out = segment->hvalue[idx - 1];
  }
  int getNNZ() const {
    segment->need_host();
    int c = 0;
    for (uint32_t w : segment->hbits) c += __builtin_popcount(w);
    return c;
  }
  bool node_owner(int) const { return true; }
};

// ---- the graph ---------------------------------------------------------------------------
template <class V, class E = int>
class Graph {
 public:
  int nvertices;
  long long int nnz;
  bool vertexpropertyowner;
  int tiles_per_dim;
  int num_threads;  // layout parameter only: enters the id permutation like the reference's OpenMP thread count

  gm_graph_t* A;   // both directions live in one handle; A and AT name the same object
  gm_graph_t* AT;
  bool adjacencyowner;
  std::vector<int32_t> dev_of_native;  // device slot of each native id (host copy; identity if empty)
  // Several ranks (one process per GPU, see graphmat/mpi_single.h): this object holds shard `get_global_myrank()`
  // of a 1-D row-sharded graph -- device rows [row_lo, row_hi), the first valid_rows of which are vertices
  // (the rest pads the slice to a multiple of 64).  vertexproperty / active cover exactly these rows, like the
  // reference's per-rank segments; one rank: [0, nvertices).
  int row_lo, row_hi, valid_rows;
  SpVec<DenseSegment<V> >* vertexproperty;
  SpVec<DenseSegment<bool> >* active;

 public:
  Graph()
      : nvertices(0), nnz(0), vertexpropertyowner(true), tiles_per_dim(get_global_nrank()),
        num_threads(default_num_threads()), A(nullptr), AT(nullptr), adjacencyowner(true), row_lo(0), row_hi(0), valid_rows(0),
        vertexproperty(nullptr), active(nullptr) {}
  // Wrap an existing adjacency and device state (used by the C-ABI fixed-menu programs).
  Graph(gm_graph_t* handle, V* d_vp, uint32_t* d_active)
      : vertexpropertyowner(true), tiles_per_dim(1), num_threads(1), A(handle), AT(handle), adjacencyowner(false) {
    gm_graph_desc_t d;
    gm_graph_desc(handle, &d);
    nvertices = d.nvertices;
    row_lo = d.row_lo;
    row_hi = d.row_hi;
    valid_rows = d.row_hi - d.row_lo;
    gm_csr_t c;
    nnz = 0;
    if (gm_graph_csr(handle, GM_DIR_OUT, &c) == GM_OK) nnz = c.nnz;
    else if (gm_graph_csr(handle, GM_DIR_IN, &c) == GM_OK) nnz = c.nnz;
    int rows = d.row_hi - d.row_lo;
    vertexproperty = new SpVec<DenseSegment<V> >(rows, (typename DenseSegment<V>::store_t*)d_vp, nullptr);
    active = new SpVec<DenseSegment<bool> >(rows, nullptr, d_active);
  }

  void ReadEdgelist(GraphMat::edgelist_t<E> A_edges);
  void getVertexEdgelist(GraphMat::edgelist_t<V>& myedges);
  void getEdgelist(GraphMat::edgelist_t<E>& myedges);
  void ReadMTX(const char* filename);
  void ReadGraphMatBin(const char* filename);
  void WriteGraphMatBin(const char* filename);

  void setAllActive();
  void setAllInactive();
  void setActive(int v);
  void setInactive(int v);

  void setAllVertexproperty(const V& val);
  void setVertexproperty(int v, const V& val);
  V getVertexproperty(int v) const;
  bool vertexNodeOwner(const int v) const;
  void saveVertexproperty(std::string fname, bool includeHeader = true) const;
  void reset();
  void shareVertexProperty(Graph<V, E>& g);
  int getNumberOfVertices() const;
  void applyToAllVertices(void (*ApplyFn)(const V&, V*, void*), void* param = nullptr);
  template <class T>
  void applyReduceAllVertices(T* val, void (*ApplyFn)(V*, T*, void*),
                              void (*ReduceFn)(const T&, const T&, T*, void*) = AddFn<T>, void* param = nullptr);
  void applyToAllEdges(void (*ApplyFn)(E*, const V&, const V&, void*), void* param = nullptr);
  // The same three operations ON THE DEVICE, for callables the compiler can see (function objects / lambdas
  // compiled with --hipstdpar; captures take the place of the void* parameter): vertex state and edge values
  // stay in HBM, nothing is copied to the host mirror.  The function-pointer forms above necessarily run on the
  // host mirror -- a host function pointer cannot be called from a kernel -- like the reference's OpenMP loops
  // (include/GMDP/singlenode/apply.h, reduce.h:51-99, applyedges.h:38-78).
  //   applyToAllVertices(f)              f(const V& in, V* out)
  //   applyReduceAllVertices(&t, map)    map(const V& v, T* out), summed with operator+ (AddFn)
  //   applyReduceAllVertices(&t, map, r) r(const T& a, const T& b, T* c): must be associative and commutative
  //                                      (blocks are combined in a tree, not in the reference's chunk order)
  //   applyToAllEdges(f)                 f(E* value, const V& src, const V& dst)
  template <class F, class = decltype(std::declval<F&>()(std::declval<const V&>(), (V*)nullptr))>
  void applyToAllVertices(F f);
  template <class T, class Map, class = decltype(std::declval<Map&>()(std::declval<const V&>(), (T*)nullptr))>
  void applyReduceAllVertices(T* val, Map map);
  template <class T, class Map, class Reduce, class = decltype(std::declval<Map&>()(std::declval<const V&>(), (T*)nullptr)),
            class = decltype(std::declval<Reduce&>()(std::declval<const T&>(), std::declval<const T&>(), (T*)nullptr))>
  void applyReduceAllVertices(T* val, Map map, Reduce reduce);
  template <class F, class = decltype(std::declval<F&>()((E*)nullptr, std::declval<const V&>(), std::declval<const V&>()))>
  void applyToAllEdges(F f);
  ~Graph();

  int vertexToNative(int vertex, int nsegments, int len) const {
    return gm_vertex_to_native(vertex, num_threads * 16 * nsegments, len);
  }
  int nativeToVertex(int vertex, int nsegments, int len) const {
    return gm_native_to_vertex(vertex, num_threads * 16 * nsegments, len);
  }
  // 1-based slot in the device-ordered vectors of the vertex with 1-based id `vertex`
  int vertexToSlot(int vertex) const {
    int nat = vertexToNative(vertex, tiles_per_dim, nvertices);
    return dev_of_native.empty() ? nat : dev_of_native[nat - 1] + 1;
  }
  // 1-based index into this rank's vertexproperty / active vectors, 0 when another rank owns the vertex
  int localSlot(int vertex) const {
    if (vertex < 1 || vertex > nvertices) return 0;
    const int s = vertexToSlot(vertex) - 1 - row_lo;
    return (s >= 0 && s < row_hi - row_lo) ? s + 1 : 0;
  }
  bool sharded() const { return tiles_per_dim > 1; }
  // several ranks: a device array over ALL device ids holding every shard's vertex properties (all-gathered with the
  // graph's message exchange; owned by the graph's workspace, valid until the next call)
  const V* gathered_vertexproperty();

 private:
  static int default_num_threads() {
    const char* e = getenv("GRAPHMAT_NUM_THREADS");
    if (!e) e = getenv("OMP_NUM_THREADS");
    int t = e ? atoi(e) : 1;
    return t > 0 ? t : 1;
  }
};

template <class V, class E>
void Graph<V, E>::ReadEdgelist(GraphMat::edgelist_t<E> A_edges) {
  struct timeval start, end;
  gettimeofday(&start, 0);
  tiles_per_dim = GraphMat::get_global_nrank();
  const int myrank = GraphMat::get_global_myrank();
  size_t ne = (size_t)A_edges.nnz;
  std::vector<int32_t> src(ne), dst(ne);
  std::vector<E> val(ne);
  for (size_t i = 0; i < ne; i++) {
    src[i] = A_edges.edges[i].src;
    dst[i] = A_edges.edges[i].dst;
    val[i] = A_edges.edges[i].val;
  }
  long long ne_global = (long long)ne;
  if (tiles_per_dim > 1) {
    // Several ranks: every rank holds the edges it read (the reference's loader gives rank r the files
    // <prefix>r, <prefix>r+nranks, ...; GMDP then shuffles edges to their tiles, SpMat.h:422-443).  Here the
    // library does the shuffle (gm_graph_desc_t.edges_local): degree counts are all-reduced, each edge travels to
    // the shard owning its row, duplicates keep the order "rank, then position".  Only m, n and the edge count are
    // combined here -- the reference's MPI_Allreduce(MAX) over m and n (edgelist.h:279-284).
    static_assert(std::is_trivially_copyable<E>::value, "edge values travel between ranks as bytes");
    std::vector<int64_t> counts((size_t)tiles_per_dim);
    const long long mine[3] = {A_edges.m, A_edges.n, (long long)ne};
    void* all = nullptr;
    if (gm_dist_allgatherv_host(mine, (int64_t)sizeof(mine), &all, counts.data()) != GM_OK) { printf("GraphMat(HIP): %s\n", gm_last_error()); exit(1); }
    ne_global = 0;
    for (int r = 0; r < tiles_per_dim; r++) {
      const long long* o = (const long long*)all + 3 * r;
      A_edges.m = std::max(A_edges.m, (int)o[0]);
      A_edges.n = std::max(A_edges.n, (int)o[1]);
      ne_global += o[2];
    }
    gm_host_free(all);
    A_edges.m = A_edges.n = std::max(A_edges.m, A_edges.n);
  }
  gm_graph_desc_t d;
  memset(&d, 0, sizeof(d));
  d.nvertices = A_edges.m;
  d.nparts = num_threads * 16 * tiles_per_dim;  // Graph.h:117 of the reference
  d.row_lo = 0;
  d.row_hi = A_edges.m;
  d.directions = GM_DIR_OUT | GM_DIR_IN;
  d.val_bytes = (int)sizeof(E);
  const char* lay = getenv("GRAPHMAT_LAYOUT");
  d.layout = (lay && !strcmp(lay, "native")) ? GM_LAYOUT_NATIVE : GM_LAYOUT_DEGREE;
  d.nshards = tiles_per_dim;
  d.shard = myrank;
  if (tiles_per_dim > 1) {
    d.layout = GM_LAYOUT_DEGREE;  // (shards are slices of the degree-ranked order)
    d.edges_local = 1;
  }
  if (A && adjacencyowner) gm_graph_destroy(A);
  A = AT = nullptr;
  if (gm_graph_create(&A, &d, (int64_t)ne, src.data(), dst.data(), val.data(), nullptr) != GM_OK) {
    printf("GraphMat(HIP): graph construction failed: %s\n", gm_last_error());
    exit(1);
  }
  AT = A;
  adjacencyowner = true;
  dev_of_native.assign((size_t)A_edges.m, 0);
  gm_graph_maps_to_host(A, dev_of_native.data(), nullptr);
  nvertices = A_edges.m;
  nnz = ne_global;
  gm_graph_desc(A, &d);
  row_lo = d.row_lo;
  row_hi = d.row_hi;
  valid_rows = row_hi - row_lo;
  if (tiles_per_dim > 1) {
    // rank k of the degree ranking lives in shard k % nranks at position k / nranks: the vertices of a slice
    // are a prefix of it, the rest is padding
    valid_rows = (nvertices - myrank + tiles_per_dim - 1) / tiles_per_dim;
    if (valid_rows < 0) valid_rows = 0;
    if (gm_graph_use_rccl(A) != GM_OK) { printf("GraphMat(HIP): %s\n", gm_last_error()); exit(1); }
  }
  const int rows = row_hi - row_lo;
  if (vertexproperty && vertexpropertyowner) delete vertexproperty;
  if (active) delete active;
  vertexproperty = new SpVec<DenseSegment<V> >(rows, true);
  vertexproperty->segment->logical = valid_rows;
  V* __v = new V;
  vertexproperty->setAll(*__v);
  delete __v;
  active = new SpVec<DenseSegment<bool> >(rows, false);
  active->segment->logical = valid_rows;
  vertexpropertyowner = true;
  detail::warm_code_object();  // one-time code-object load belongs to construction, not to the first run
  {
    // scratch the runs will ask for (messages / reduced values of up to 8 bytes, presence words,
    // the push step's bid and list arrays): allocated here so the first run does not pay hipMalloc
    void* p = nullptr;
    const size_t n = (size_t)d.ndevice, words = ((n + 31) / 32 + 2) * 4;
    const size_t want[GM_WS_SLOTS] = {4096, n * 8 + 16, words, n * 8 + 16, words, 0, n * 8 + 64, n * 4 + 64};
    for (int slot = 0; slot < GM_WS_SLOTS; slot++)
      if (want[slot]) (void)gm_graph_workspace(A, slot, want[slot], &p);
    // the first launch of one of the engine's kernels makes the HIP runtime resolve this
    // executable's kernel table (~8 ms measured with the reference's BFS.cpp): pay it here
    if (gm_graph_workspace(A, 0, 4096, &p) == GM_OK) {
      hipLaunchKernelGGL(dev::k_fill_u32, dim3(1), dim3(dev::kBlock), 0, 0, (uint32_t*)p, (int64_t)64, 0u);
      (void)hipDeviceSynchronize();
    }
  }
  gettimeofday(&end, 0);
  std::cout << "Finished GraphMat read + construction, time: " << sec(start, end) << std::endl;
}

template <class V, class E>
void Graph<V, E>::ReadMTX(const char* filename) {
  GraphMat::edgelist_t<E> A_edges;
  GraphMat::load_edgelist(filename, &A_edges, true, true, true);  // binary format with header and edge weights
  if (A_edges.m != A_edges.n) {
    int maxn = std::max(A_edges.m, A_edges.n);
    A_edges.m = maxn;
    A_edges.n = maxn;
  }
  ReadEdgelist(A_edges);
  A_edges.clear();
}

// GraphMat-bin: the reference archives its partitioned matrices with boost::serialization
// (Graph.h:152-208), a format tied to its in-memory classes.  This engine's file of the same
// role holds the edge list in original vertex ids (the device layout is rebuilt on read, so a
// file does not depend on the layout parameters): 8-byte magic, int32 sizeof(E), int32
// nvertices, int64 nnz, int32 num_threads at the time of writing, int32 0, then src[nnz],
// dst[nnz] (int32 each) and val[nnz].  Like the reference: one file per rank (<name><rank>),
// vertex properties and activity are not stored and come back default-initialised.
namespace detail {
static const char kGraphBinMagic[8] = {'G', 'M', 'H', 'I', 'P', 'B', 'N', '1'};
}
template <class V, class E>
void Graph<V, E>::ReadGraphMatBin(const char* filename) {
  std::stringstream fname_ss;
  fname_ss << filename << GraphMat::get_global_myrank();
  std::cout << "Reading file " << fname_ss.str() << std::endl;
  FILE* fp = fopen(fname_ss.str().c_str(), "rb");
  if (!fp) { std::cout << "Could not open file: " << fname_ss.str() << std::endl; exit(1); }
  char magic[8];
  int32_t head[2], tail[2];
  int64_t count = 0;
  bool ok = fread(magic, 1, 8, fp) == 8 && memcmp(magic, detail::kGraphBinMagic, 8) == 0 && fread(head, 4, 2, fp) == 2 &&
            fread(&count, 8, 1, fp) == 1 && fread(tail, 4, 2, fp) == 2;
  if (!ok || head[0] != (int32_t)sizeof(E) || head[1] <= 0 || count < 0) {
    std::cout << "Error reading file - not a GraphMat(HIP) graph file for this edge type" << std::endl;
    exit(1);
  }
  GraphMat::edgelist_t<E> el(head[1], head[1], (int)count);
  std::vector<int32_t> s((size_t)count), d((size_t)count);
  std::vector<E> v((size_t)count);
  ok = fread(s.data(), 4, (size_t)count, fp) == (size_t)count && fread(d.data(), 4, (size_t)count, fp) == (size_t)count &&
       fread((void*)v.data(), sizeof(E), (size_t)count, fp) == (size_t)count;
  fclose(fp);
  if (!ok) { std::cout << "Error reading file - truncated" << std::endl; exit(1); }
  for (int64_t k = 0; k < count; k++) el.edges[k] = edge_t<E>(s[k], d[k], v[k]);
  ReadEdgelist(el);
  el.clear();
}
template <class V, class E>
void Graph<V, E>::WriteGraphMatBin(const char* filename) {
  std::stringstream fname_ss;
  fname_ss << filename << GraphMat::get_global_myrank();
  std::cout << "Writing file " << fname_ss.str() << std::endl;
  GraphMat::edgelist_t<E> el;
  getEdgelist(el);
  const size_t count = (size_t)el.nnz;
  std::vector<int32_t> s(count), d(count);
  std::vector<E> v(count);
  for (size_t k = 0; k < count; k++) { s[k] = el.edges[k].src; d[k] = el.edges[k].dst; v[k] = el.edges[k].val; }
  el.clear();
  FILE* fp = fopen(fname_ss.str().c_str(), "wb");
  if (!fp) { std::cout << "Could not open file for writing: " << fname_ss.str() << std::endl; exit(1); }
  const int32_t head[2] = {(int32_t)sizeof(E), (int32_t)nvertices}, tail[2] = {(int32_t)num_threads, 0};
  const int64_t cnt = (int64_t)count;
  bool ok = fwrite(detail::kGraphBinMagic, 1, 8, fp) == 8 && fwrite(head, 4, 2, fp) == 2 && fwrite(&cnt, 8, 1, fp) == 1 &&
            fwrite(tail, 4, 2, fp) == 2 && fwrite(s.data(), 4, count, fp) == count && fwrite(d.data(), 4, count, fp) == count &&
            fwrite((const void*)v.data(), sizeof(E), count, fp) == count;
  if (fclose(fp) != 0) ok = false;
  if (!ok) { std::cout << "Error writing file " << fname_ss.str() << std::endl; exit(1); }
}

template <class V, class E>
void Graph<V, E>::setAllActive() { active->segment->setAllBits(true); }
template <class V, class E>
void Graph<V, E>::setAllInactive() { active->segment->setAllBits(false); }
// (per-vertex setters act on the owning rank only, like the reference's SpVec::set on a non-owner)
template <class V, class E>
void Graph<V, E>::setActive(int v) { if (int ls = localSlot(v)) active->set(ls, true); }
template <class V, class E>
void Graph<V, E>::setInactive(int v) { if (int ls = localSlot(v)) active->unset(ls); }

template <class V, class E>
void Graph<V, E>::reset() {
  setAllInactive();
  V v;
  vertexproperty->setAll(v);
}

template <class V, class E>
void Graph<V, E>::shareVertexProperty(Graph<V, E>& g) {
  // the shared vector is indexed in g's device order: bring this graph's adjacency into it
  // (with several ranks this is a collective: every rank's part of the edges travels to the shard that owns its row in
  // g's order -- gm_graph_relayout_like; the reference's two graphs share one 2-D tile distribution instead)
  if (A != nullptr && g.A != nullptr && A != g.A) {
    if (gm_graph_relayout_like(A, g.A, nullptr) != GM_OK) {
      printf("GraphMat(HIP): shareVertexProperty: %s\n", gm_last_error());
      exit(1);
    }
    dev_of_native = g.dev_of_native;
  }
  if (vertexproperty != nullptr && vertexpropertyowner) delete vertexproperty;
  vertexproperty = g.vertexproperty;
  vertexpropertyowner = false;
}

template <class V, class E>
void Graph<V, E>::setAllVertexproperty(const V& val) { vertexproperty->setAll(val); }
template <class V, class E>
void Graph<V, E>::setVertexproperty(int v, const V& val) {
  if (int ls = localSlot(v)) vertexproperty->set(ls, val);
}
template <class V, class E>
V Graph<V, E>::getVertexproperty(const int v) const {
  V vp;
  if (int ls = localSlot(v)) vertexproperty->get(ls, &vp);  // (a non-owner gets a default-constructed value, as in the reference)
  return vp;
}
template <class V, class E>
bool Graph<V, E>::vertexNodeOwner(const int v) const { return localSlot(v) != 0; }
template <class V, class E>
int Graph<V, E>::getNumberOfVertices() const { return nvertices; }

template <class V, class E>
void Graph<V, E>::getVertexEdgelist(GraphMat::edgelist_t<V>& myedges) {
  vertexproperty->segment->need_host();
  myedges = edgelist_t<V>(nvertices, 1, nvertices);
  for (int v = 1; v <= nvertices; v++) {
    myedges.edges[v - 1].src = v;
    myedges.edges[v - 1].dst = 1;
    if (int ls = localSlot(v)) myedges.edges[v - 1].val = vertexproperty->segment->hvalue[ls - 1];
    else myedges.edges[v - 1].val = V();
  }
}

template <class V, class E>
void Graph<V, E>::getEdgelist(GraphMat::edgelist_t<E>& myedges) {
  gm_csr_t c;
  if (gm_graph_csr(A, GM_DIR_IN, &c) != GM_OK) { printf("%s\n", gm_last_error()); exit(1); }
  std::vector<int64_t> rp(c.nrows + 1);
  std::vector<int32_t> ci((size_t)c.nnz);
  std::vector<E> vv((size_t)c.nnz);
  gm_graph_csr_to_host(A, GM_DIR_IN, rp.data(), ci.data(), vv.data());
  myedges = edgelist_t<E>(nvertices, nvertices, (int)c.nnz);
  std::vector<int32_t> nod((size_t)c.ncols);
  gm_graph_maps_to_host(A, nullptr, nod.data());
  size_t k = 0;
  for (int r = 0; r < c.nrows; r++)  // (the rows of this rank's shard)
    for (int64_t e = rp[r]; e < rp[r + 1]; e++, k++) {
      myedges.edges[k].src = nativeToVertex(nod[c.row_base + r] + 1, tiles_per_dim, nvertices);
      myedges.edges[k].dst = nativeToVertex(nod[ci[e]] + 1, tiles_per_dim, nvertices);
      myedges.edges[k].val = vv[e];
    }
}

// text dump "vertex value" per line, vertices in id order (reference: Graph.h:337-350 writes one
// file per rank via DenseSegment::save; single rank here, file name gets the rank suffix 0)
template <class V, class E>
void Graph<V, E>::saveVertexproperty(std::string fname, bool includeHeader) const {
  vertexproperty->segment->need_host();
  std::ofstream f((fname + std::to_string(get_global_myrank())).c_str());
  if (includeHeader) f << nvertices << " " << 1 << " " << nvertices << std::endl;
  for (int v = 1; v <= nvertices; v++)  // (each rank writes the vertices it owns into its own file, like the reference)
    if (int ls = localSlot(v)) f << v << " " << vertexproperty->segment->hvalue[ls - 1] << std::endl;
}

// several ranks: every rank reduces the vertices it owns (in native order), rank 0's result is then combined with
// the other ranks' in rank order and everybody gets it -- multinode/reduce.h:38-70 of the reference, including its
// quirk that every rank's partial result starts from the caller's initial value
template <class T, class R>
static void combine_over_ranks(T* val, R reduce) {
  static_assert(std::is_trivially_copyable<T>::value, "map-reduce results travel between ranks as bytes");
  const int nr = get_global_nrank();
  std::vector<int64_t> counts((size_t)nr);
  void* all = nullptr;
  if (gm_dist_allgatherv_host((const void*)val, (int64_t)sizeof(T), &all, counts.data()) != GM_OK) { printf("GraphMat(HIP): %s\n", gm_last_error()); exit(1); }
  T res;
  memcpy((void*)&res, all, sizeof(T));
  for (int r = 1; r < nr; r++) {
    T other, t = res;
    memcpy((void*)&other, (const char*)all + (size_t)r * sizeof(T), sizeof(T));
    reduce(t, other, &res);
  }
  gm_host_free(all);
  *val = res;
}

// Host-side element-wise helpers.  The callbacks are host function pointers, so they run
// on the host mirror (the reference runs them under OpenMP: include/GMDP/singlenode/apply.h,
// reduce.h:51-99).
template <class V, class E>
void Graph<V, E>::applyToAllVertices(void (*ApplyFn)(const V&, V*, void*), void* param) {
  vertexproperty->segment->host_modified();
  for (auto& v : vertexproperty->segment->hvalue) ApplyFn(v, &v, param);
}

template <class V, class E>
template <class T>
void Graph<V, E>::applyReduceAllVertices(T* val, void (*ApplyFn)(V*, T*, void*),
                                         void (*ReduceFn)(const T&, const T&, T*, void*), void* param) {
  vertexproperty->segment->need_host();
  auto& h = vertexproperty->segment->hvalue;
  const int n = nvertices, rows = row_hi - row_lo;
  const int nthreads = num_threads;  // chunking of reduce.h:57-66
  const int per = (n + nthreads - 1) / nthreads;
  for (int p = 0; p < nthreads; p++) {
    int s = std::min(per * p, n), e = std::min(per * (p + 1), n);
    bool first = false;
    T local;
    for (int i = s; i < e; i++) {  // native order, like the reference's segment walk; the vertices this rank owns
      const int slot = (dev_of_native.empty() ? i : dev_of_native[i]) - row_lo;
      if (slot < 0 || slot >= rows) continue;
      T t2;
      ApplyFn(&h[slot], &t2, param);
      if (first) { T t = local; ReduceFn(t, t2, &local, param); }
      else { local = t2; first = true; }
    }
    if (first) { T t = *val; ReduceFn(t, local, val, param); }
  }
  if (sharded()) combine_over_ranks(val, [&](const T& a, const T& b, T* c) { ReduceFn(a, b, c, param); });
}


template <class V, class E>
const V* Graph<V, E>::gathered_vertexproperty() {
  gm_graph_desc_t d;
  gm_graph_desc(A, &d);
  void* buf = nullptr;
  if (gm_graph_workspace(A, 13, (size_t)d.ndevice * sizeof(V) + 64, &buf) != GM_OK) { printf("GraphMat(HIP): %s\n", gm_last_error()); exit(1); }
  const int rows = d.row_hi - d.row_lo;
  vertexproperty->segment->need_device();
  GM_HIP_OK(hipMemcpy((char*)buf + (size_t)d.row_lo * sizeof(V), vertexproperty->segment->value, (size_t)rows * sizeof(V), hipMemcpyDeviceToDevice));
  gm_graph_set_run_stream(A, nullptr);
  if (gm_graph_exchange(A, GM_XCHG_MESSAGES, buf, (int64_t)sizeof(V), nullptr, nullptr) != GM_OK) { printf("GraphMat(HIP): vertex property exchange failed: %s\n", gm_last_error()); exit(1); }
  GM_HIP_OK(hipDeviceSynchronize());
  return (const V*)buf;
}

template <class V, class E>
void Graph<V, E>::applyToAllEdges(void (*ApplyFn)(E*, const V&, const V&, void*), void* param) {
  // (several ranks: the other endpoint may live on another shard -- the reference moves the properties along the tile
  // rows and columns, GMDP/multinode/applyedges.h:45-161; here every shard's properties are all-gathered once, with the
  // exchange that carries the message vector, and read through a host copy indexed by device id)
  std::vector<V> hall;
  if (sharded()) {
    gm_graph_desc_t d;
    gm_graph_desc(A, &d);
    vertexproperty->segment->need_device();
    const V* dv = gathered_vertexproperty();
    hall.resize((size_t)d.ndevice);
    GM_HIP_OK(hipMemcpy((void*)hall.data(), dv, (size_t)d.ndevice * sizeof(V), hipMemcpyDeviceToHost));
  } else {
    vertexproperty->segment->need_host();
  }
  const V* h = sharded() ? hall.data() : vertexproperty->segment->hvalue.data();
  const int rbase = sharded() ? row_lo : 0;
  for (int dir : {GM_DIR_OUT, GM_DIR_IN}) {
    gm_csr_t c;
    if (gm_graph_csr(A, dir, &c) != GM_OK) continue;
    std::vector<int64_t> rp(c.nrows + 1);
    std::vector<int32_t> ci((size_t)c.nnz);
    std::vector<E> vv((size_t)c.nnz);
    gm_graph_csr_to_host(A, dir, rp.data(), ci.data(), vv.data());
    for (int r = 0; r < c.nrows; r++)
      for (int64_t e = rp[r]; e < rp[r + 1]; e++) {
        // OUT: row = destination, col = source;  IN: row = source, col = destination
        if (dir == GM_DIR_OUT) ApplyFn(&vv[e], h[ci[e]], h[rbase + r], param);
        else ApplyFn(&vv[e], h[rbase + r], h[ci[e]], param);
      }
    if (gm_graph_set_vals(A, dir, vv.data()) != GM_OK) { printf("GraphMat(HIP): %s\n", gm_last_error()); exit(1); }
  }
}

// ---- device forms (functors) -------------------------------------------------------------------
namespace dev {
template <class V, class F>
__global__ void __launch_bounds__(kBlock) k_vertices_apply(V* __restrict__ vp, const uint32_t* __restrict__ bits, int n, F f) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n || !bit_get(bits, i)) return;
  const V in = vp[i];
  V out = in;
  f(in, &out);
  vp[i] = out;
}
// map + reduce over the present vertices: private folds (grid-stride), then a tree per workgroup in LDS;
// one partial (and "has" flag) per workgroup, finished on the host
template <class V, class T, class Map, class Reduce>
__global__ void __launch_bounds__(kBlock) k_vertices_reduce(const V* __restrict__ vp, const uint32_t* __restrict__ bits, int n, Map map,
                                                            Reduce reduce, T* __restrict__ partial, int* __restrict__ partial_has) {
  static_assert(sizeof(T) <= 128, "applyReduceAllVertices on the device: reduction type of at most 128 bytes");
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kBlock * sizeof(T)];
  __shared__ int s_has[kBlock];
  T* s_val = reinterpret_cast<T*>(s_raw);
  T acc;
  bool has = false;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (!bit_get(bits, (int)i)) continue;
    T t;
    map(vp[i], &t);
    if (has) { T a = acc; reduce(a, t, &acc); } else { acc = t; has = true; }
  }
  s_has[threadIdx.x] = has;
  if (has) s_val[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s && s_has[threadIdx.x + s]) {
      if (s_has[threadIdx.x]) { T a = s_val[threadIdx.x]; reduce(a, s_val[threadIdx.x + s], &s_val[threadIdx.x]); }
      else { s_val[threadIdx.x] = s_val[threadIdx.x + s]; s_has[threadIdx.x] = 1; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial_has[blockIdx.x] = s_has[0];
    if (s_has[0]) partial[blockIdx.x] = s_val[0];
  }
}
// one wave per row of one direction's CSR; rows_are_dst: row = destination, column = source
template <class V, class E, class F>
// (vp is indexed by device id: for a shard, the all-gathered copy of every shard's properties; rows are local)
__global__ void __launch_bounds__(kBlock) k_edges_apply(gm_csr_t A, E* __restrict__ vals, const V* __restrict__ vp, int rows_are_dst, F f) {
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= A.nrows) return;
  const V vr = vp[A.row_base + row];
  for (int64_t e = A.rowptr[row] + (threadIdx.x & 63); e < A.rowptr[row + 1]; e += 64) {
    const V vc = vp[A.colidx[e]];
    E v = vals[e];
    if (rows_are_dst) f(&v, vc, vr); else f(&v, vr, vc);
    vals[e] = v;
  }
}
}  // namespace dev

template <class V, class E>
template <class F, class>
void Graph<V, E>::applyToAllVertices(F f) {
  auto* seg = vertexproperty->segment;
  seg->need_device();
  const int n = seg->capacity;
  if (n > 0) hipLaunchKernelGGL((dev::k_vertices_apply<V, F>), dim3(detail::grid_for(n)), dim3(dev::kBlock), 0, 0, (V*)seg->value, (const uint32_t*)seg->bit_vector, n, f);
  GM_HIP_OK(hipDeviceSynchronize());
  seg->device_modified();
}

namespace detail {
template <class T>
struct device_add {
  __host__ __device__ void operator()(const T& a, const T& b, T* c) const { *c = a + b; }
};
}  // namespace detail

template <class V, class E>
template <class T, class Map, class>
void Graph<V, E>::applyReduceAllVertices(T* val, Map map) {
  applyReduceAllVertices(val, map, detail::device_add<T>());
}

template <class V, class E>
template <class T, class Map, class Reduce, class, class>
void Graph<V, E>::applyReduceAllVertices(T* val, Map map, Reduce reduce) {
  auto* seg = vertexproperty->segment;
  seg->need_device();
  const int n = seg->capacity;
  if (n <= 0) return;
  const int grid = detail::grid_for(n) < 1024 ? detail::grid_for(n) : 1024;
  T* d_partial = nullptr;
  int* d_has = nullptr;
  GM_HIP_OK(hipMalloc((void**)&d_partial, (size_t)grid * sizeof(T)));
  GM_HIP_OK(hipMalloc((void**)&d_has, (size_t)grid * sizeof(int)));
  hipLaunchKernelGGL((dev::k_vertices_reduce<V, T, Map, Reduce>), dim3(grid), dim3(dev::kBlock), 0, 0, (const V*)seg->value,
                     (const uint32_t*)seg->bit_vector, n, map, reduce, d_partial, d_has);
  std::vector<unsigned char> raw((size_t)grid * sizeof(T));
  std::vector<int> has((size_t)grid);
  GM_HIP_OK(hipMemcpy(raw.data(), d_partial, raw.size(), hipMemcpyDeviceToHost));
  GM_HIP_OK(hipMemcpy(has.data(), d_has, (size_t)grid * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(d_partial);
  (void)hipFree(d_has);
  bool first = false;
  T total;
  for (int b = 0; b < grid; b++) {
    if (!has[b]) continue;
    T t;
    memcpy((void*)&t, raw.data() + (size_t)b * sizeof(T), sizeof(T));
    if (first) { T a = total; reduce(a, t, &total); } else { total = t; first = true; }
  }
  if (first) { T a = *val; reduce(a, total, val); }  // like reduce.h:93-96: combined with the caller's value
  if (sharded()) combine_over_ranks(val, reduce);
}

template <class V, class E>
template <class F, class>
void Graph<V, E>::applyToAllEdges(F f) {
  auto* seg = vertexproperty->segment;
  seg->need_device();
  const V* vp_all = sharded() ? gathered_vertexproperty() : (const V*)seg->value;  // an edge needs both endpoints' properties
  for (int dir : {GM_DIR_OUT, GM_DIR_IN}) {
    gm_csr_t c;
    if (gm_graph_csr(A, dir, &c) != GM_OK || c.vals == nullptr || c.nrows == 0) continue;
    if (!sharded()) c.row_base = 0;
    hipLaunchKernelGGL((dev::k_edges_apply<V, E, F>), dim3((c.nrows + dev::kBlock / 64 - 1) / (dev::kBlock / 64)), dim3(dev::kBlock), 0, 0, c,
                       (E*)const_cast<void*>(c.vals) /* library-owned edge values, rewritten in place */, vp_all,
                       dir == GM_DIR_OUT ? 1 : 0, f);
  }
  GM_HIP_OK(hipDeviceSynchronize());
  if (gm_graph_sync_tile_vals(A, nullptr) != GM_OK) { printf("GraphMat(HIP): %s\n", gm_last_error()); exit(1); }  // (column tiles hold copies)
}

template <class V, class E>
Graph<V, E>::~Graph() {
  if (A != nullptr && adjacencyowner) gm_graph_destroy(A);
  A = AT = nullptr;
  if (vertexpropertyowner && vertexproperty != nullptr) delete vertexproperty;
  vertexproperty = nullptr;
  if (active != nullptr) delete active;
  active = nullptr;
}

}  // namespace GraphMat
#endif
