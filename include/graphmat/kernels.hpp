// graphmat/kernels.hpp -- gfx950 kernel templates of the generalized-SpMV iteration.
//
// One iteration of GraphMat's run_graph_program (reference:
// include/GraphMatRuntime.h:136-261) is three device phases:
//   send    x[i] = send_message(vp[i]) for active i      (singlenode/intersectreduce.h:39-66)
//   multiply+reduce  y = A (x) x over the user semiring  (singlenode/spmspv.h:39-86, spmspv3.h:38-90)
//   apply   apply(y[i], vp[i]); changed -> active        (GraphMatRuntime.h:195-225)
//
// Design (MI355X-first, not a translation of the reference's DCSC column walk):
//  * adjacency is CSR by row with columns ascending inside a row, which is the
//    order in which the reference reduces a row's messages; a row is always
//    folded in that order, so non-commutative / floating-point reductions give
//    the reference's bits.
//  * rows are grouped into row-blocks of < 2*GM_BLOCK_NNZ edges: a 256-thread
//    workgroup streams the block's column ids coalesced, gathers the messages
//    in parallel into LDS, then one lane per row folds its segment from LDS.
//  * rows of GM_SHORT_ROW+1..GM_GIANT_ROW edges get one wave each (k_spmv_wave): the
//    products of 64 edges are folded in order straight out of the lanes' registers.
//  * rows longer than GM_GIANT_ROW get a workgroup each (k_spmv_giant) with a
//    reduction strategy chosen by program_traits<P>::reduce.
//  * presence bit vectors keep the reference layout (bit i&31 of word i>>5).
//  * the vertex program is passed by value as raw bytes and its methods are
//    called qualified (p.P::f(...)), i.e. never through the host vtable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "../graphmat_hip.h"

namespace GraphMat {

// How a program's reduce_function may be evaluated.  The default is always safe.
enum reduce_kind {
  REDUCE_AUTO = -1,        // let the runtime probe reduce_function on the host (engine.hpp)
  REDUCE_ORDERED = 0,      // fold strictly in ascending column order (any functor)
  REDUCE_COMMUTATIVE = 1,  // associative+commutative and exact (integer +, min, max): any order
  REDUCE_LAST = 2,         // reduce(a,b) is a=b: the result is the last present message
  REDUCE_F32_ADD = 3       // float a+=b: ordered result, reproduced bit-exactly in parallel
};

// Optional, per-program knowledge the runtime may exploit; specialise for your
// program type.  Nothing here changes results, only how they are computed.
template <class P>
struct program_traits {
  static constexpr reduce_kind reduce = REDUCE_AUTO;
};

// Optional row filter: a program may declare that apply(y, vp) is a no-op whenever
// wants(prog, vp) is false (BFS: a vertex that already has a depth ignores further messages).
// Such rows then skip the multiply altogether -- the "bottom-up" optimisation -- without
// changing any result.  Specialise with enabled = true and a __host__ __device__ wants().
// wants() must depend on the vertex property alone (not on program members that
// do_every_iteration changes): the engine caches it as one bit per row, recomputed at the start
// of a run and whenever apply() touches the row.
template <class P>
struct program_row_filter {
  static constexpr bool enabled = false;
  template <class V>
  __host__ __device__ static bool wants(const P&, const V&) { return true; }
};

namespace dev {

// Which rows a multiply pass works on.  `want` is a bitmap over the rows: the program's row filter
// (program_row_filter; the engine keeps it as one bit per row, k_want_init / k_apply).
// `want` == nullptr evaluates the program's filter directly (every row if it has none).
template <class P, class V>
__device__ __forceinline__ bool row_wanted(const P& p, const V* __restrict__ vp, const uint32_t* __restrict__ want, int row) {
  if (want != nullptr) return (want[row >> 5] >> (row & 31)) & 1u;
  if constexpr (program_row_filter<P>::enabled) return program_row_filter<P>::wants(p, vp[row]);
  return true;
}


// The message of vertex c.  a=b programs consume one message per row, so (unsharded) they do
// not need the send pass at all: with x == nullptr the message is evaluated on demand from the
// sender's vertex property -- send_message is const and vertex properties do not change between
// the send and the multiply phase, so this is the value k_send would have stored.
template <class P, class T, class V>
__device__ __forceinline__ T message_of(const P& p, const T* __restrict__ x, const V* __restrict__ vp, int c) {
  if (x != nullptr) return x[c];
  T m;
  p.P::send_message(vp[c], m);
  return m;
}

// [0] chunks accepted by the exact fp32 replay, [1] chunks folded serially (long rows)
static __device__ unsigned long long g_longrow_counters[4];

constexpr int kBlock = 256;               // threads per workgroup (4 wave64)
constexpr int kStage = 2 * GM_BLOCK_NNZ;  // LDS slots of a row-block

template <class P>
struct ProgArg {
  alignas(16) unsigned char b[sizeof(P)];
};
template <class P>
inline ProgArg<P> make_prog_arg(const P* p) {
  ProgArg<P> a;
  memcpy(a.b, (const void*)p, sizeof(P));
  return a;
}

// column ids (and other read-once streams) are loaded non-temporally: 4 B/edge of streaming
// data would otherwise keep evicting the re-used lines of the message vector from the L2s
__device__ __forceinline__ int stream_load(const int32_t* __restrict__ p) { return __builtin_nontemporal_load(p); }

#ifdef GRAPHMAT_ABLATION
// ablation builds only: "cold" columns are not gathered.  A.cold_from > 0: columns whose position inside the adjacency's
// column slice (c - A.hot_base) is at least that; < 0: columns whose position inside THEIR column tile is at least
// -A.cold_from (the untiled short-row pass of a tiled graph: tile bases from g_abl_tile_base).
static __device__ int g_abl_tile_base[GM_MAX_TILES + 1];
static __device__ int g_abl_ntiles;
__device__ __forceinline__ bool abl_cold(int c, int cold_from, int base) {
  if (cold_from > 0) return c - base >= cold_from;
  if (cold_from < 0) {
    int b = 0;
    for (int t = 0; t < g_abl_ntiles; t++) if (c >= g_abl_tile_base[t]) b = g_abl_tile_base[t];
    return c - b >= -cold_from;
  }
  return false;
}
// per-wave time stamps of the persistent kernels' LAST launch (tools/wave_times_probe.py): [kernel 0 = k_spmv_wave16p,
// 1 = k_spmv_rowwave][wave of the grid][0 start, 1 hot set loaded, 2 done, 3 work items (steps / 64-row groups)], 100 MHz ticks
static __device__ unsigned long long g_abl_wave_times[2][8192][4];
static __global__ void k_abl_count(const int32_t* __restrict__ colidx, int64_t nnz, int cold_from, int base, unsigned long long* out) {
  unsigned long long n = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) n += abl_cold(colidx[i], cold_from, base) ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}
#define GM_ABL_COLD(c, A) abl_cold((c), (A).cold_from, (A).hot_base)
#else
#define GM_ABL_COLD(c, A) false
#endif

__device__ __forceinline__ bool bit_get(const uint32_t* __restrict__ bits, int i) {
  return (bits[i >> 5] >> (i & 31)) & 1u;
}

// raw storage type used to stage small trivially-copyable messages in LDS
template <int N> struct raw_of { typedef void type; };
template <> struct raw_of<1> { typedef uint8_t type; };
template <> struct raw_of<2> { typedef uint16_t type; };
template <> struct raw_of<4> { typedef uint32_t type; };
template <> struct raw_of<8> { typedef uint64_t type; };
template <class T>
struct stageable {
  static constexpr bool value = (sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4 || sizeof(T) == 8);
};

// ------------------------------------------------------------------------------------
// send: x[row_base+i] = send_message(vp[i]) where active; x presence word = active word.
// The bool returned by send_message is ignored, as in GraphMatRuntime.h:79-85.
template <class P, class T, class V>
__global__ void __launch_bounds__(kBlock)
k_send(ProgArg<P> pa, const V* __restrict__ vp, const uint32_t* __restrict__ active, T* __restrict__ x,
       uint32_t* __restrict__ xbits, int n, int row_base) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  uint32_t w = active ? active[i >> 5] : 0xffffffffu;
  if ((w >> (i & 31)) & 1u) {
    T m;
    p.P::send_message(vp[i], m);
    x[(size_t)row_base + i] = m;
  }
  if ((i & 31) == 0) {
    int rem = n - i;
    if (rem < 32) w &= (1u << rem) - 1u;
    xbits[(row_base + i) >> 5] = w;
  }
}

// ------------------------------------------------------------------------------------
// apply of iteration i and send of iteration i+1 in one pass over the vertices, for programs whose every vertex
// sends every iteration (ALL_VERTICES) on one GPU: the plain lean k_apply, then x[i] = send_message(vp[i]) from the
// property just written -- vp is read once instead of twice and one launch goes away.  The engine uses the messages
// only if do_every_iteration leaves the program object unchanged (engine.hpp: fused apply + send).
template <class P, class T, class U, class V>
__global__ void __launch_bounds__(kBlock)
k_apply_send(ProgArg<P> pa, const U* __restrict__ y, const uint32_t* __restrict__ ybits, V* __restrict__ vp,
             uint32_t* __restrict__ active, int n, int* __restrict__ changed_flag, uint32_t* __restrict__ want,
             T* __restrict__ x, uint32_t* __restrict__ xbits, int row_base) {
  bool any = false;
  for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
    const int i = (int)base + threadIdx.x;
    bool changed = false, applied = false, still = false;
    if (i < n) {
      ProgArg<P> local = pa;  // apply() is non-const in the API: give it a private copy
      P& p = *reinterpret_cast<P*>(local.b);
      V cur = vp[i];
      if (bit_get(ybits, i)) {
        V old_prop = cur;  // (operator!= of user types is not const)
        p.P::apply(y[i], cur);
        vp[i] = cur;
        if (old_prop != cur) changed = true;
        applied = true;
        if constexpr (program_row_filter<P>::enabled) still = program_row_filter<P>::wants(p, cur);
      }
      const P& ps = *reinterpret_cast<const P*>(pa.b);  // (the send of the NEXT iteration: the program as captured, not apply's copy)
      T m;
      ps.P::send_message(cur, m);
      x[(size_t)row_base + i] = m;
    }
    if constexpr (program_row_filter<P>::enabled) {
      if (want != nullptr) {  // the wave owns the two words of its 64 rows
        const unsigned long long ma = __ballot(applied), mw = __ballot(still);
        if ((threadIdx.x & 63) == 0 && i < n && ma != 0ull) {
          want[i >> 5] = (want[i >> 5] & ~(uint32_t)ma) | (uint32_t)mw;
          if (i + 32 < n) want[(i >> 5) + 1] = (want[(i >> 5) + 1] & ~(uint32_t)(ma >> 32)) | (uint32_t)(mw >> 32);
        }
      }
    }
    const unsigned long long m = __ballot(changed);
    const unsigned long long in_range = __ballot(i < n);
    if ((threadIdx.x & 63) == 0 && i < n) {
      active[i >> 5] = (uint32_t)m;
      if (i + 32 < n) active[(i >> 5) + 1] = (uint32_t)(m >> 32);
      if (m) any = true;
      xbits[(row_base + i) >> 5] = (uint32_t)in_range;  // every vertex sent (k_send with no active vector)
      if (i + 32 < n) xbits[((row_base + i) >> 5) + 1] = (uint32_t)(in_range >> 32);
    }
  }
  if (any) *changed_flag = 1;
}

// ------------------------------------------------------------------------------------
// Building a compact list from a grid-stride loop.  Atomics with a result on ONE global counter cost
// ~10 ns each on this chip, so a wave-level append of tens of thousands of entries takes
// milliseconds.  A workgroup therefore collects its entries in LDS (LDS atomics) over all its loop
// iterations and reserves global space once per kListBuf entries.  Every thread of the workgroup must
// call add() the same number of times and finish() once.  `cap` > 0: the workgroup stops adding
// once one of its reservations starts at or beyond `cap` entries (a list of at most `cap` entries
// in total is always complete).
constexpr int kListBuf = 1024;
struct BlockList {
  int32_t* buf;         // LDS, kListBuf entries
  unsigned int* fill;   // LDS: [0] entries in buf, [1] global base of the last flush, [2] stopped
  __device__ __forceinline__ void init() {
    if (threadIdx.x == 0) { fill[0] = 0; fill[2] = 0; }
    __syncthreads();
  }
  __device__ __forceinline__ void flush(int32_t* __restrict__ list, unsigned int* __restrict__ count, unsigned int cap) {
    __syncthreads();
    const unsigned int k = fill[0];
    if (threadIdx.x == 0 && k) fill[1] = atomicAdd(count, k);
    __syncthreads();
    const unsigned int base = fill[1];
    for (unsigned int j = threadIdx.x; j < k; j += kBlock) list[base + j] = buf[j];
    __syncthreads();
    if (threadIdx.x == 0) {
      fill[0] = 0;
      if (cap && k && base >= cap) fill[2] = 1;
    }
    __syncthreads();
  }
  __device__ __forceinline__ void add(bool mine, int value, int32_t* __restrict__ list, unsigned int* __restrict__ count, unsigned int cap) {
    if (fill[0] + kBlock > kListBuf) flush(list, count, cap);  // block-uniform: room for one entry per thread
    if (fill[2]) return;
    const unsigned long long m = __ballot(mine);
    if (m) {
      const int lane = threadIdx.x & 63;
      unsigned int start = 0;
      if (lane == 0) start = atomicAdd(&fill[0], (unsigned int)__popcll(m));
      start = (unsigned int)__shfl((int)start, 0, 64);
      if (mine) buf[start + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = value;
    }
    __syncthreads();
  }
  __device__ __forceinline__ void finish(int32_t* __restrict__ list, unsigned int* __restrict__ count, unsigned int cap) {
    flush(list, count, cap);
  }
};

// The same per wave, without workgroup barriers: right when only a few waves will have anything
// to add (k_apply lists the changed vertices only when the next active set is bound to be small), so
// the number of global reservations stays small although every wave makes its own.
constexpr int kWaveListBuf = 256;
struct WaveList {
  int32_t* buf;       // this wave's LDS buffer, kWaveListBuf entries
  unsigned int fill;  // wave-uniform
  bool stopped;       // wave-uniform
  __device__ __forceinline__ void init(int32_t* lds_of_block) {
    buf = lds_of_block + (threadIdx.x >> 6) * kWaveListBuf;
    fill = 0;
    stopped = false;
  }
  __device__ __forceinline__ void flush(int32_t* __restrict__ list, unsigned int* __restrict__ count, unsigned int cap) {
    if (fill == 0) return;
    const int lane = threadIdx.x & 63;
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(count, fill);
    base = (unsigned int)__shfl((int)base, 0, 64);
    for (unsigned int j = lane; j < fill; j += 64) list[base + j] = buf[j];
    if (cap && base >= cap) stopped = true;
    fill = 0;
  }
  __device__ __forceinline__ void add(bool mine, int value, int32_t* __restrict__ list, unsigned int* __restrict__ count, unsigned int cap) {
    if (stopped) return;
    const unsigned long long m = __ballot(mine);
    if (m == 0ull) return;
    const unsigned int k = (unsigned int)__popcll(m);
    if (fill + k > (unsigned int)kWaveListBuf) flush(list, count, cap);
    if (stopped) return;
    const int lane = threadIdx.x & 63;
    if (mine) buf[fill + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = value;
    fill += k;
  }
};

// ------------------------------------------------------------------------------------
// apply on rows whose y bit is set; a changed vertex (V::operator!=) becomes active and
// raises the changed flag (zeroed by the host before the launch).  The active vector is fully rewritten (the reference clears
// it right before, GraphMatRuntime.h:184).
// row-filter bits of all rows (program_row_filter), computed once per run; k_apply keeps them current
template <class P, class V>
__global__ void __launch_bounds__(kBlock)
k_want_init(ProgArg<P> pa, const V* __restrict__ vp, int n, uint32_t* __restrict__ want) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int i = blockIdx.x * kBlock + threadIdx.x;
  bool w = false;
  if constexpr (program_row_filter<P>::enabled) w = i < n && program_row_filter<P>::wants(p, vp[i]);
  const unsigned long long m = __ballot(w);
  if ((threadIdx.x & 63) == 0 && i < n) {
    want[i >> 5] = (uint32_t)m;
    want[(i >> 5) + 1] = (uint32_t)(m >> 32);
  }
}

// With `stats` (programs that may take top-down steps) the kernel also sizes the next active set:
// vertices, their out-edges, largest out-degree.  The kernel is grid-stride (64-row groups stay
// with one wave, so the active words are written whole); a workgroup adds its totals with one
// set of atomics at the end, into one of kStatSlots counter triples that the host sums.
constexpr int kStatSlots = 64;
constexpr int kApplyMaxBlocks = 4096;
constexpr int kSparseListCap = 65536;  // top-down steps are taken for active sets of at most this many vertices (2^20: slower, the list kernels live on global atomics)
// STEER: the variant used by programs that may take top-down steps (statistics, list, LDS buffer);
// plain fixed-count programs (PageRank) run the lean one.
template <class P, class U, class V, bool STEER>
__global__ void __launch_bounds__(kBlock)
k_apply(ProgArg<P> pa, const U* __restrict__ y, const uint32_t* __restrict__ ybits, V* __restrict__ vp,
        uint32_t* __restrict__ active, int n, int* __restrict__ changed_flag, const int64_t* __restrict__ src_rowptr,
        unsigned long long* __restrict__ stats /* kStatSlots x {vertices, out-edges, max out-degree, -} or null */,
        uint32_t* __restrict__ want /* row-filter bits to keep up to date, or null */,
        int32_t* __restrict__ next_list = nullptr /* with stats: the changed vertices, while they are few */,
        unsigned int* __restrict__ next_count = nullptr) {
  __shared__ unsigned long long s_c[STEER ? kBlock / 64 : 1], s_e[STEER ? kBlock / 64 : 1], s_m[STEER ? kBlock / 64 : 1];
  __shared__ int32_t s_lbuf[STEER ? (kBlock / 64) * kWaveListBuf : 1];
  WaveList wlist;
  wlist.init(s_lbuf);
  if constexpr (!STEER) { stats = nullptr; next_list = nullptr; }
  unsigned long long cnt = 0, edges = 0, mx = 0;
  bool any = false;
  for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
    const int i = (int)base + threadIdx.x;
    bool changed = false, applied = false, still = false;
    if (i < n && bit_get(ybits, i)) {
      ProgArg<P> local = pa;  // apply() is non-const in the API: give it a private copy
      P& p = *reinterpret_cast<P*>(local.b);
      V old_prop = vp[i];
      V cur = old_prop;
      p.P::apply(y[i], cur);
      vp[i] = cur;
      if (old_prop != cur) changed = true;
      applied = true;
      if constexpr (program_row_filter<P>::enabled) still = program_row_filter<P>::wants(p, cur);
    }
    if constexpr (program_row_filter<P>::enabled) {
      if (want != nullptr) {  // the wave owns the two words of its 64 rows
        const unsigned long long ma = __ballot(applied), mw = __ballot(still);
        if ((threadIdx.x & 63) == 0 && i < n && ma != 0ull) {
          want[i >> 5] = (want[i >> 5] & ~(uint32_t)ma) | (uint32_t)mw;
          if (i + 32 < n) want[(i >> 5) + 1] = (want[(i >> 5) + 1] & ~(uint32_t)(ma >> 32)) | (uint32_t)(mw >> 32);
        }
      }
    }
    const unsigned long long m = __ballot(changed);
    if ((threadIdx.x & 63) == 0 && i < n) {
      active[i >> 5] = (uint32_t)m;
      if (i + 32 < n) active[(i >> 5) + 1] = (uint32_t)(m >> 32);
      if (m) any = true;
    }
    if (stats != nullptr && changed) {
      const unsigned long long d = src_rowptr ? (unsigned long long)(src_rowptr[i + 1] - src_rowptr[i]) : 0ull;
      cnt++;
      edges += d;
      mx = d > mx ? d : mx;
    }
    // compact list of the changed vertices for a following top-down step (while they are few)
    if (next_list != nullptr) wlist.add(changed, i, next_list, next_count, (unsigned int)kSparseListCap);
  }
  if (next_list != nullptr) wlist.flush(next_list, next_count, (unsigned int)kSparseListCap);
  if (any) *changed_flag = 1;
  if (stats == nullptr) return;
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_down(cnt, off, 64);
    edges += __shfl_down(edges, off, 64);
    const unsigned long long o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_c[wv] = cnt; s_e[wv] = edges; s_m[wv] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; w++) { cnt += s_c[w]; edges += s_e[w]; mx = s_m[w] > mx ? s_m[w] : mx; }
    if (cnt) {
      unsigned long long* slot = stats + 4 * (blockIdx.x % kStatSlots);
      atomicAdd(&slot[0], cnt);
      atomicAdd(&slot[1], edges);
      atomicMax(&slot[2], mx);
    }
  }
}

// ------------------------------------------------------------------------------------
// fold helper: acc (+)= process_message(msg, edge, vp_row)
template <class P, class T, class U, class V, class E>
__device__ __forceinline__ void fold_one(const P& p, const T& m, const E& ev, const V& vprow, U& acc, bool& has) {
  U res;
  p.P::process_message(m, ev, vprow, res);
  if (has) {
    p.P::reduce_function(acc, res);  // SPMV.h:54-59: c = a; reduce(c, b)
  } else {
    acc = res;  // no additive identity: first message assigns (spmspv.h:73-77)
    has = true;
  }
}

template <class E>
__device__ __forceinline__ E edge_at(const void* __restrict__ vals, int64_t k) {
  return vals ? reinterpret_cast<const E*>(vals)[k] : E();
}

// ------------------------------------------------------------------------------------
// debug/ablation switches (gm_set_option("debug_flags", ...)); 0 in production
// `accumulate` argument of the multiply kernels: ACC_READ_PREV = y may already hold results of an
// earlier pass (their presence is read from ybits); ACC_STATIC_BITS = do not write presence
// bits (every x entry is present, so y's presence equals the graph's static row bits)
enum { ACC_READ_PREV = 1, ACC_STATIC_BITS = 2 };
// The first four and the last are tested INSIDE the multiply kernels (skip the fold, skip the gathers ...): they exist for ablation
// builds only (-DGRAPHMAT_ABLATION, graphmat_amd/build.py with GRAPHMAT_ABLATION=1).  In a production build they are 0,
// so every `dbg & DBG_*` in a kernel is a compile-time false and the tests are not in the code.  The others pick
// between exact strategies on the host side of a launch and are always available.
#ifdef GRAPHMAT_ABLATION
#define GM_ABL(bit) (bit)
#define GM_DBG_PARAM , const int dbg
#define GM_DBG_ARG(flags) , (int)(flags)
#define GM_DBG_PASS , dbg
#else
#define GM_ABL(bit) 0
// a production build's kernels have no such parameter at all: `dbg` inside them names this constant
#define GM_DBG_PARAM
#define GM_DBG_ARG(flags)
#define GM_DBG_PASS
static constexpr int dbg = 0;
#endif
enum { DBG_SKIP_FOLD = GM_ABL(1), DBG_SKIP_GATHER = GM_ABL(2), DBG_FIRST_CHUNK_ONLY = GM_ABL(4), DBG_NO_REPLAY = GM_ABL(8), DBG_NO_OVERLAP = 16, DBG_NO_PUSH = 32, DBG_NO_GROUPED = 64, DBG_NO_PIPELINE = 128, DBG_NO_LAZY_SEND = 256, DBG_NO_WAVE16 = 512, DBG_NO_TILES = 1024, DBG_NO_SPARSE_XCHG = 2048, DBG_LATE_GIANTS = 4096, DBG_LONG_ON_MAIN = 8192, DBG_PLAIN_WAVE16 = GM_ABL(16384) };

// presence bits of a wave's 64 consecutive rows: one atomicOr per 32-row word (not per row:
// same-word atomics from 32 lanes serialise in the L2); nothing when the bits are static
__device__ __forceinline__ void publish_row_bits(bool wrote, int row, uint32_t* __restrict__ ybits, int accumulate) {
  const unsigned long long hm = __ballot(wrote);
  const int lane = threadIdx.x & 63;
  if (hm != 0ull && !(accumulate & 2 /* ACC_STATIC_BITS */) && (lane == 0 || (row & 31) == 0)) {
    const int in_word = 32 - (row & 31);
    const int left = 64 - lane;
    const int cnt = in_word < left ? in_word : left;
    const unsigned long long mask = (cnt >= 64) ? ~0ull : ((1ull << cnt) - 1ull);
    const uint32_t bits = (uint32_t)(((hm >> lane) & mask) << (row & 31));
    if (bits) atomicOr(&ybits[row >> 5], bits);
  }
}

// ------------------------------------------------------------------------------------
// Short rows of REDUCE_LAST programs (reduce is a=b: a row needs only its LAST present message).
// No row-blocks, no staging: one lane per row of the whole row range; a lane whose row is short
// (1..short_row edges; longer rows belong to the wave / giant kernels) and still wanted walks it
// backwards, eight edges per step (independent loads), and stops at the first present
// in-neighbour.  Rows that are filtered out cost one bit of a coalesced word; short rows average
// a handful of edges, so most finish in one step, and in the levels where the frontier is large
// the scan ends after an edge or two instead of testing every edge.
template <class P, class T, class U, class V, class E, bool USE_VP>
__global__ void __launch_bounds__(kBlock)
k_spmv_short_last(ProgArg<P> pa, gm_csr_t A, const T* __restrict__ x, const uint32_t* __restrict__ xbits,
                  const V* __restrict__ vp, U* __restrict__ y, uint32_t* __restrict__ ybits, int accumulate GM_DBG_PARAM,
                  const uint32_t* __restrict__ want,
                  const uint32_t* __restrict__ xsum /* 1 bit per 64 x entries "any present", or null (k_bits_summary) */) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int row = blockIdx.x * kBlock + threadIdx.x;
  const bool dense = (xbits == nullptr);
  bool wrote = false;
  if (row < A.nrows && row_wanted(p, vp, want, row)) {
    const int64_t rp0 = A.rowptr[row], rp1 = A.rowptr[row + 1];
    if (rp1 > rp0 && rp1 - rp0 <= (int64_t)A.short_row) {
      constexpr int B = 8;
      int64_t kk = -1;
      int cc = -1;
      for (int64_t hi = rp1; hi > rp0 && cc < 0; hi -= B) {
        int c[B];
#pragma unroll
        for (int j = 0; j < B; j++) c[j] = (hi - 1 - j >= rp0) ? A.colidx[hi - 1 - j] : -1;
        bool pr[B];
#pragma unroll
        for (int j = 0; j < B; j++)
          pr[j] = c[j] >= 0 && (dense || ((dbg & DBG_SKIP_GATHER) ? (c[j] == 0x7fffffff)
                                                                   : ((xsum == nullptr || bit_get(xsum, c[j] >> 6)) && bit_get(xbits, c[j]))));
#pragma unroll
        for (int j = B - 1; j >= 0; j--)
          if (pr[j]) { cc = c[j]; kk = hi - 1 - j; }  // ends with the smallest j = the edge nearest the row's end
      }
      if (cc >= 0 && !(dbg & DBG_SKIP_FOLD)) {
        V vprow;
        if constexpr (USE_VP) vprow = vp[row];
        T m = message_of(p, x, vp, cc);
        U res;
        p.P::process_message(m, edge_at<E>(A.vals, kk), vprow, res);
        y[row] = res;
        wrote = true;
      }
    }
  }
  publish_row_bits(wrote, row, ybits, accumulate);
}

// ------------------------------------------------------------------------------------
// multiply+reduce over row-blocks (rows of at most GM_SHORT_ROW edges).
//   USE_VP : 3-operand form, process_message sees vp[row] (spmspv3.h:70)
//   xbits == nullptr : every x entry present (ALL_VERTICES programs)
//   accumulate : y may already hold partial results (second pass of ALL_EDGES)
// Phase 1 keeps many independent loads in flight per lane: all column ids of the lane's
// slots first, then all gathers, then the LDS stores.
template <class P, class T, class U, class V, class E, bool USE_VP, bool DENSE, int RK>
__global__ void __launch_bounds__(kBlock)
k_spmv_rowblock(ProgArg<P> pa, gm_csr_t A, const T* __restrict__ x, const uint32_t* __restrict__ xbits,
                const V* __restrict__ vp, U* __restrict__ y, uint32_t* __restrict__ ybits, int accumulate GM_DBG_PARAM,
                const uint32_t* __restrict__ want) {
  // a=b needs one message per row (the last present one): gather it in phase 2 instead of all of them
  constexpr bool STAGE = stageable<T>::value && RK != REDUCE_LAST;
  constexpr int PER = kStage / kBlock;  // 8 slots per lane
  typedef typename raw_of<STAGE ? (int)sizeof(T) : 1>::type raw_t;
  // slot(k) = k + k/32: consecutive rows of EQUAL length d (the degree-ranked device order
  // produces long runs of them) start d*(1+1/32) slots apart, so the lane-per-row reads of
  // phase 2 do not pile onto one LDS bank when d is a multiple of 32, 16, 8 ...
  constexpr int kPad = kStage + kStage / 32;
  __shared__ int s_col[kPad];
  __shared__ raw_t s_msg[STAGE ? kPad : 1];
#define GM_SLOT(k) ((k) + ((k) >> 5))

  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int sg = A.blk_seg[blockIdx.x];
  const int r0 = A.seg_row[sg], r1 = A.seg_row[sg + 1];
  const int row = r0 + threadIdx.x;
  int64_t rp0 = 0, rp1 = 0;
  if (row < r1) { rp0 = A.rowptr[row]; rp1 = A.rowptr[row + 1]; }
  const int64_t e0 = A.rowptr[r0], e1 = A.rowptr[r1];
  const int n = (int)(e1 - e0);
  if (n == 0 || n > kStage) return;  // cannot happen for a row-block (see gm_csr_t)
  constexpr bool dense = DENSE;  // every x entry present (xbits == nullptr)
  bool wanted = true;
  if (program_row_filter<P>::enabled || want != nullptr) {  // (uniform over the launch)
    wanted = row < r1 && rp1 > rp0 && row_wanted(p, vp, want, row);
    if (!__syncthreads_or(wanted)) return;  // no row of this block would use a message
  }

  int c[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    int k = threadIdx.x + j * kBlock;
    c[j] = (k < n) ? stream_load(&A.colidx[e0 + k]) : -1;
  }
  if (!dense) {
#pragma unroll
    for (int j = 0; j < PER; j++)
      if (c[j] >= 0 && !bit_get(xbits, c[j])) c[j] = -1;
  }
  if constexpr (STAGE) {
    raw_t m[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
      m[j] = raw_t();
      if (c[j] >= 0 && !(dbg & DBG_SKIP_GATHER) && !GM_ABL_COLD(c[j], A)) m[j] = reinterpret_cast<const raw_t*>(x)[c[j]];
    }
#pragma unroll
    for (int j = 0; j < PER; j++) {
      int k = threadIdx.x + j * kBlock;
      if (k < n) { s_msg[GM_SLOT(k)] = m[j]; if (!dense) s_col[GM_SLOT(k)] = c[j]; }
    }
  } else {
#pragma unroll
    for (int j = 0; j < PER; j++) {
      int k = threadIdx.x + j * kBlock;
      if (k < n) s_col[GM_SLOT(k)] = c[j];
    }
  }
  __syncthreads();
  if (dbg & DBG_SKIP_FOLD) return;

  // phase 2: one lane per row folds its segment in ascending column order
  bool wrote = false;
  if (row < r1 && rp1 > rp0 && wanted) {
    const int kb = (int)(rp0 - e0), ke = (int)(rp1 - e0);
    bool has = (accumulate & ACC_READ_PREV) && bit_get(ybits, row);
    U acc;
    if (has) acc = y[row];
    V vprow;
    if constexpr (USE_VP) vprow = vp[row];
    if constexpr (RK == REDUCE_LAST) {
      // reduce is a=b: the last present message of the segment wins; walk it backwards
      for (int k = ke - 1; k >= kb; k--) {
        T m;
        if constexpr (STAGE) {
          if (!dense && s_col[GM_SLOT(k)] < 0) continue;
          raw_t r = s_msg[GM_SLOT(k)];
          memcpy(&m, &r, sizeof(T));
        } else {
          int cc = s_col[GM_SLOT(k)];
          if (cc < 0) continue;
          m = x[cc];
        }
        p.P::process_message(m, edge_at<E>(A.vals, e0 + k), vprow, acc);
        has = true;
        break;
      }
    } else if constexpr (STAGE && DENSE) {
      // all messages present and staged: LDS reads four at a time, then the ordered folds
      int k = kb;
      if (!has) {
        T m;
        raw_t r = s_msg[GM_SLOT(k)];
        memcpy(&m, &r, sizeof(T));
        p.P::process_message(m, edge_at<E>(A.vals, e0 + k), vprow, acc);
        has = true;
        k++;
      }
      for (; k + 4 <= ke; k += 4) {
        raw_t r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = s_msg[GM_SLOT(k + u)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          T m;
          memcpy(&m, &r[u], sizeof(T));
          U res;
          p.P::process_message(m, edge_at<E>(A.vals, e0 + k + u), vprow, res);
          p.P::reduce_function(acc, res);
        }
      }
      for (; k < ke; k++) {
        T m;
        raw_t r = s_msg[GM_SLOT(k)];
        memcpy(&m, &r, sizeof(T));
        U res;
        p.P::process_message(m, edge_at<E>(A.vals, e0 + k), vprow, res);
        p.P::reduce_function(acc, res);
      }
    } else {
      for (int k = kb; k < ke; k++) {
        T m;
        if constexpr (STAGE) {
          if (s_col[GM_SLOT(k)] < 0) continue;
          raw_t r = s_msg[GM_SLOT(k)];
          memcpy(&m, &r, sizeof(T));
        } else {
          int cc = s_col[GM_SLOT(k)];
          if (cc < 0) continue;
          m = x[cc];
        }
        fold_one<P, T, U, V, E>(p, m, edge_at<E>(A.vals, e0 + k), vprow, acc, has);
      }
    }
    if (has) {
      y[row] = acc;
      wrote = true;
    }
  }
  publish_row_bits(wrote, row, ybits, accumulate);
#undef GM_SLOT
}

// ------------------------------------------------------------------------------------
// wave-level helpers for values of any small trivially-copyable type
template <class U>
__device__ __forceinline__ U wave_bcast(const U& v, int srclane) {  // srclane is wave-uniform
  constexpr int NW = (int)((sizeof(U) + 3) / 4);
  uint32_t w[NW] = {};
  memcpy(w, &v, sizeof(U));
#pragma unroll
  for (int i = 0; i < NW; i++) w[i] = (uint32_t)__builtin_amdgcn_readlane((int)w[i], srclane);
  U r;
  memcpy(&r, w, sizeof(U));
  return r;
}
template <class U>
__device__ __forceinline__ U wave_shfl_down(const U& v, int delta) {
  constexpr int NW = (int)((sizeof(U) + 3) / 4);
  uint32_t w[NW] = {};
  memcpy(w, &v, sizeof(U));
#pragma unroll
  for (int i = 0; i < NW; i++) w[i] = (uint32_t)__shfl_down((int)w[i], delta, 64);
  U r;
  memcpy(&r, w, sizeof(U));
  return r;
}

// The hottest x entries of the adjacency's column slice, kept in LDS by the wave kernels (device order is
// degree-ranked: they are the first ones of the slice; a column tile's columns are the slice
// [hot_base, hot_base + hot_len) of the device order, busiest first).  Sharded graphs: the degree ranking is
// dealt over the NS slices of x, so the hot set is the first HOT / NS entries of every slice
// (s_hot[q * per + i] = x[q * stride + i]).
template <class T>
struct HotSet {
  const T* s_hot;
  int base, nhot, NS, per, stride;
  int shift;  // log2(stride) when stride is a power of two, else -1
  float inv_stride;
  int cold_from;  // (ablation builds: columns from here on are read from LDS instead of being gathered)
  __device__ __forceinline__ T get(const T* __restrict__ x, int c) const {
#ifdef GRAPHMAT_ABLATION
    if (abl_cold(c, cold_from, base)) return s_hot[c & 4095];
#endif
    if (NS == 1) {
      const unsigned rel = (unsigned)(c - base);
      return rel < (unsigned)nhot ? s_hot[rel] : x[c];
    }
    // slice q of the column and its position in it: a shift and a mask when the slices are a power of two long (2^s
    // vertices over 2^k shards: the benchmark's case), else a float estimate of the quotient, fixed up exactly
    int q, pos;
    if (shift >= 0) {
      q = c >> shift;
      pos = c & (stride - 1);
    } else {
      q = (int)((float)c * inv_stride);
      q = q >= NS ? NS - 1 : q;
      pos = c - q * stride;
      if (pos < 0) { q--; pos += stride; } else if (pos >= stride) { q++; pos -= stride; }
    }
    return pos < per ? s_hot[q * per + pos] : x[c];
  }
};
// fills s_hot (all threads of the workgroup; the caller synchronises) and describes it
template <class T, int HOT, int BLOCK>
__device__ __forceinline__ HotSet<T> hot_load(const gm_csr_t& A, const T* __restrict__ x, T* s_hot) {
  HotSet<T> h;
  h.s_hot = s_hot;
  h.cold_from = A.cold_from;
  h.base = A.hot_base;
  h.NS = A.hot_slices > 1 ? A.hot_slices : 1;
  h.stride = A.hot_stride;
  h.per = HOT > 1 ? ((A.hot_len < HOT / h.NS ? A.hot_len : HOT / h.NS)) : 0;  // hot entries per slice
  h.nhot = h.per * h.NS;
  h.inv_stride = h.NS > 1 ? 1.0f / (float)A.hot_stride : 0.f;
  h.shift = (h.NS > 1 && A.hot_stride > 0 && (A.hot_stride & (A.hot_stride - 1)) == 0) ? (31 - __clz(A.hot_stride)) : -1;
  if constexpr (HOT > 1) {
    if (h.NS == 1) {
      const T* __restrict__ xhot = x + A.hot_base;
      for (int i = threadIdx.x; i < h.nhot; i += BLOCK) s_hot[i] = xhot[i];
    } else {
      for (int i = threadIdx.x; i < h.nhot; i += BLOCK) s_hot[i] = x[(size_t)(i / h.per) * A.hot_stride + (i % h.per)];
    }
  }
  return h;
}

// ------------------------------------------------------------------------------------
// multiply+reduce, one wave per row (rows of GM_SHORT_ROW+1 .. GM_GIANT_ROW edges; also
// giant rows of programs without a faster strategy).  64 edges at a time: coalesced
// column ids, parallel gathers and products, then -- for ordered reductions -- the 64
// products are folded in order by broadcasting them one by one out of the lanes'
// registers (v_readlane), every lane carrying the same running value: no LDS, no
// barriers, a few cycles per edge.  Loads run two chunks ahead of the fold.
// wave_row: the work of one wave on one row [e0, e1)
template <class P, class T, class U, class V, class E, bool USE_VP, int RK>
__device__ __forceinline__ void wave_row(const P& p, const gm_csr_t& A, const int row, const int64_t e0, const int64_t e1,
                                         const int lane, const T* __restrict__ x, const uint32_t* __restrict__ xbits,
                                         const V* __restrict__ vp, U* __restrict__ y, uint32_t* __restrict__ ybits,
                                         const int accumulate GM_DBG_PARAM, U* out_acc = nullptr, bool* out_has = nullptr,
                                         const uint32_t* __restrict__ xsum = nullptr) {
  // (out_acc / out_has, ordered kind only: hand the result back instead of storing it -- k_check_rows)
  // (xsum, REDUCE_LAST only: 1 bit per 64 x entries "any present" (k_bits_summary), tested before the presence bit)
  const bool dense = (xbits == nullptr);
  V vprow;
  if constexpr (USE_VP) vprow = vp[row];

  if constexpr (RK == REDUCE_LAST) {
    // reduce is a=b: the last present edge of the row wins; scan backwards.  Most rows that find
    // a message find it in their last chunk, so the scan starts one chunk at a time; a row that
    // keeps failing (early bottom-up levels: nearly every row scans all its edges) doubles the
    // number of chunks it has in flight, up to four, instead of paying two dependent memory
    // latencies per 64 edges.
    constexpr int DMAX = 4;
    int depth = 1;
    for (int64_t hi = e1; hi > e0;) {
      int c[DMAX];
      bool pres[DMAX];
#pragma unroll
      for (int u = 0; u < DMAX; u++) {
        const int64_t k = hi - 64 * (u + 1) + lane;
        c[u] = (u < depth && k >= e0) ? stream_load(&A.colidx[k]) : -1;
      }
#pragma unroll
      for (int u = 0; u < DMAX; u++) pres[u] = c[u] >= 0 && (dense || ((xsum == nullptr || bit_get(xsum, c[u] >> 6)) && bit_get(xbits, c[u])));
#pragma unroll
      for (int u = 0; u < DMAX; u++) {
        const unsigned long long mask = __ballot(pres[u]);
        if (mask) {
          if (lane == 63 - __clzll(mask)) {
            const int64_t k = hi - 64 * (u + 1) + lane;
            T m = message_of(p, x, vp, c[u]);
            U res;
            p.P::process_message(m, edge_at<E>(A.vals, k), vprow, res);
            y[row] = res;
            if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
          }
          return;
        }
      }
      hi -= 64 * depth;
      depth = depth < DMAX ? depth * 2 : DMAX;
    }
    return;  // nothing present: with accumulate an earlier pass's value simply stays
  } else if constexpr (RK == REDUCE_COMMUTATIVE) {
    // any order: private strided folds, then a shuffle tree with the user's reduce_function
    bool has = false;
    U acc;
    for (int64_t k = e0 + lane; k < e1; k += 64) {
      int c = A.colidx[k];
      if (!dense && !bit_get(xbits, c)) continue;
      T m = x[c];
      fold_one<P, T, U, V, E>(p, m, edge_at<E>(A.vals, k), vprow, acc, has);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      U o = wave_shfl_down(acc, off);
      int oh = __shfl_down((int)has, off, 64);
      if (oh) { if (has) p.P::reduce_function(acc, o); else { acc = o; has = true; } }
    }
    if (lane == 0) {
      if ((accumulate & ACC_READ_PREV) && bit_get(ybits, row)) {
        U prev = y[row];
        if (has) { U t = acc; acc = prev; p.P::reduce_function(acc, t); } else { acc = prev; has = true; }
      }
      if (has) {
        y[row] = acc;
        if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
      }
    }
    return;
  } else {
    // ordered: every lane carries the same running value
    bool has = false;
    U acc;
    if ((accumulate & ACC_READ_PREV) && bit_get(ybits, row)) { acc = y[row]; has = true; }
    // software pipeline, D chunks of 64 edges deep: slot u holds the gathered messages of chunk
    // i+u and the column ids of chunk i+u+D, so a gather has D-1 folds to complete under
    constexpr int D = 4;
    auto load_col = [&](int64_t k) -> int { return (k < e1) ? stream_load(&A.colidx[k]) : -1; };
    int c[D], cn[D];
    T m[D];
#pragma unroll
    for (int u = 0; u < D; u++) c[u] = load_col(e0 + 64 * u + lane);
#pragma unroll
    for (int u = 0; u < D; u++) {
      if (c[u] >= 0 && !dense && !bit_get(xbits, c[u])) c[u] = -1;
      if (c[u] >= 0 && !(dbg & DBG_SKIP_GATHER) && !GM_ABL_COLD(c[u], A)) m[u] = x[c[u]];
    }
#pragma unroll
    for (int u = 0; u < D; u++) cn[u] = load_col(e0 + 64 * (D + u) + lane);
    for (int64_t base = e0; base < e1; base += 64 * D) {
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int64_t cb = base + 64 * u;
        if (cb < e1) {
          // products of this chunk, then the ordered fold out of the lanes' registers
          const bool pres = c[u] >= 0;
          U term;
          if (pres) p.P::process_message(m[u], edge_at<E>(A.vals, cb + lane), vprow, term);
          unsigned long long mask = __ballot(pres);
          if (!(dbg & DBG_SKIP_FOLD)) {
            if (!has && mask) {  // first message of the row assigns
              const int i = __ffsll((long long)mask) - 1;
              acc = wave_bcast(term, i);
              has = true;
              mask &= mask - 1;
            }
            if (mask == ~0ull) {
#pragma unroll
              for (int i = 0; i < 64; i++) { U t = wave_bcast(term, i); p.P::reduce_function(acc, t); }
            } else {
              while (mask) {
                const int i = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                U t = wave_bcast(term, i);
                p.P::reduce_function(acc, t);
              }
            }
          }
        }
        // refill slot u: gathers of chunk i+D, column ids of chunk i+2D
        c[u] = cn[u];
        if (c[u] >= 0 && !dense && !bit_get(xbits, c[u])) c[u] = -1;
        if (c[u] >= 0 && !(dbg & DBG_SKIP_GATHER) && !GM_ABL_COLD(c[u], A)) m[u] = x[c[u]];
        cn[u] = load_col(cb + 64 * 2 * D + lane);
      }
    }
    if (out_acc != nullptr) {
      if (has) *out_acc = acc;
      *out_has = has;
    } else if (lane == 0 && has) {
      y[row] = acc;
      if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
    }
  }
}

// Cross-check of a reduction strategy that was chosen by PROBING the program's reduce_function (engine.hpp:
// probe_reduce_kind): a sample of rows is folded again, strictly in order with the program's own functions,
// and compared bit for bit with what the fast strategy left in y.  One wave per sampled row; rows come from
// a list or, with rows == nullptr, are every `stride`-th row.
template <class P, class T, class U, class V, class E, bool USE_VP>
__global__ void __launch_bounds__(kBlock)
k_check_rows(ProgArg<P> pa, gm_csr_t A, const int32_t* __restrict__ rows, int nlist, int stride, const T* __restrict__ x,
             const uint32_t* __restrict__ xbits, const V* __restrict__ vp, const U* __restrict__ y, const uint32_t* __restrict__ ybits,
             const uint32_t* __restrict__ want, unsigned int* __restrict__ mismatches) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (w >= nlist) return;
  const int row = rows ? rows[w] : w * stride;
  if (row >= A.nrows || !row_wanted(p, vp, want, row)) return;
  const int64_t e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
  if (e1 == e0) return;
  U acc;
  bool has = false;
  wave_row<P, T, U, V, E, USE_VP, REDUCE_ORDERED>(p, A, row, e0, e1, threadIdx.x & 63, x, xbits, vp, (U*)nullptr, (uint32_t*)nullptr, 0 GM_DBG_ARG(0),
                                                  &acc, &has);
  if ((threadIdx.x & 63) == 0) {
    const bool present = bit_get(ybits, row);
    bool bad = has != present;
    if (!bad && has) {
      const U got = y[row];
      const unsigned char *pa_ = reinterpret_cast<const unsigned char*>(&got), *pb_ = reinterpret_cast<const unsigned char*>(&acc);
      for (size_t i = 0; i < sizeof(U); i++) bad |= pa_[i] != pb_[i];
    }
    if (bad) atomicAdd(mismatches, 1u);
  }
}


template <class P, class T, class U, class V, class E, bool USE_VP, int RK>
__global__ void __launch_bounds__(kBlock)
k_spmv_wave(ProgArg<P> pa, gm_csr_t A, const int32_t* __restrict__ rows, int nlist, const T* __restrict__ x,
            const uint32_t* __restrict__ xbits, const V* __restrict__ vp, U* __restrict__ y,
            uint32_t* __restrict__ ybits, int accumulate GM_DBG_PARAM, const uint32_t* __restrict__ want) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (w >= nlist) return;
  const int row = rows[w];
  if (!row_wanted(p, vp, want, row)) return;
  wave_row<P, T, U, V, E, USE_VP, RK>(p, A, row, A.rowptr[row], A.rowptr[row + 1], threadIdx.x & 63, x, xbits, vp, y, ybits,
                                      accumulate GM_DBG_PASS);
}

// Ordered folds of wave rows without paying 64 serial broadcasts per 64 edges: a wave takes kWaveRows
// (16) consecutive entries of the row list at once.  Per step it fetches the next 64-edge chunk of
// each of them (16 coalesced index loads and 16 gathers per lane in flight, the next step's already
// issued), writes the 16 x 64 products to its LDS tile, and then lanes 0..15 fold one row each,
// in stored order, straight out of LDS -- 16 folds run side by side instead of every lane carrying
// the same running value.  The list is in device (degree-ranked) order, so the 16 rows have
// similar lengths.  2-operand programs with a reduction type of 4 or 8 bytes; the others keep
// k_spmv_wave.
constexpr int kWaveRows = 16;
constexpr int kHotEntries = 8192;
constexpr int kWave16Block = 256;  // (512 threads sharing one hot set: no difference; 1024: slower)

// one wave folds the kWaveRows list entries [first, first + kWaveRows) (s_t / s_mask: the wave's LDS tile)
template <class P, class T, class U, class V, class E, int KSTRIDE>
__device__ __forceinline__ void wave16_group(const P& p, const gm_csr_t& A, const int32_t* __restrict__ rows, const int nlist,
                                             const int first, const int lane, const T* __restrict__ x,
                                             const uint32_t* __restrict__ xbits, const V* __restrict__ vp, U* __restrict__ y,
                                             uint32_t* __restrict__ ybits, const int accumulate GM_DBG_PARAM,
                                             const uint32_t* __restrict__ want, const HotSet<T>& hot, U (*s_t)[KSTRIDE],
                                             unsigned long long* s_mask) {
  constexpr int G = kWaveRows;
  const bool dense = (xbits == nullptr);
  // lane r < G owns list entry first + r
  int my_row = -1;
  int64_t my_e0 = 0, my_e1 = 0;
  if (lane < G && first + lane < nlist) {
    my_row = rows[first + lane];
    if (row_wanted(p, vp, want, my_row)) { my_e0 = A.rowptr[my_row]; my_e1 = A.rowptr[my_row + 1]; }
  }
  int64_t longest = my_e1 - my_e0;
  for (int off = 8; off > 0; off >>= 1) {
    const int64_t o = __shfl_xor(longest, off, 64);
    longest = o > longest ? o : longest;
  }
  longest = __shfl(longest, 0, 64);
  const int nsteps = (int)((longest + 63) / 64);
  bool has = false;
  U acc;
  if (lane < G && my_row >= 0 && (accumulate & ACC_READ_PREV) && bit_get(ybits, my_row)) { acc = y[my_row]; has = true; }
  V no_vp;  // 2-operand programs ignore the vertex property argument

  int c[G];
  T m[G];
  auto fetch = [&](int step) {  // column ids and messages of chunk `step` of every row
#pragma unroll
    for (int r = 0; r < G; r++) {
      const int64_t k = wave_bcast(my_e0, r) + (int64_t)step * 64 + lane;
      c[r] = (k < wave_bcast(my_e1, r)) ? stream_load(&A.colidx[k]) : -1;
    }
#pragma unroll
    for (int r = 0; r < G; r++) {
      if (c[r] >= 0 && !dense && !bit_get(xbits, c[r])) c[r] = -1;
      if (c[r] >= 0) m[r] = hot.get(x, c[r]);
    }
  };
  if (nsteps > 0) fetch(0);
  for (int step = 0; step < nsteps; step++) {
    // products of this step into the tile
#pragma unroll
    for (int r = 0; r < G; r++) {
      const bool pres = c[r] >= 0;
      U term;
      if (pres) {
        const int64_t k = wave_bcast(my_e0, r) + (int64_t)step * 64 + lane;
        p.P::process_message(m[r], edge_at<E>(A.vals, k), no_vp, term);
        s_t[r][lane] = term;
      }
      const unsigned long long mask = __ballot(pres);
      if (lane == 0) s_mask[r] = mask;
    }
    __builtin_amdgcn_wave_barrier();
    if (step + 1 < nsteps) fetch(step + 1);  // in flight during the folds below
    if (lane < G && !(dbg & DBG_SKIP_FOLD)) {
      unsigned long long mask = s_mask[lane];
      const U* t = s_t[lane];
      if (mask == ~0ull) {
        int k = 0;
        if (!has) { acc = t[0]; has = true; k = 1; }
        for (; k < 64; k++) { U v = t[k]; p.P::reduce_function(acc, v); }
      } else if ((mask & (mask + 1)) == 0) {
        // the present products are the first n of the chunk (dense x: the tail of a row).  A counted loop whose LDS reads run
        // four elements ahead of the folds: the bit-scan loop below pays an LDS round trip per product, and nearly every
        // step has a row in its last chunk (RMAT-26 with every gather ablated: 235 -> 215 us per launch of this kernel)
        const int n = __popcll(mask);
        int k = 0;
        if (!has && n > 0) { acc = t[0]; has = true; k = 1; }
        U nx[4];
        if (k + 4 <= n) {
#pragma unroll
          for (int u = 0; u < 4; u++) nx[u] = t[k + u];
        }
        while (k + 4 <= n) {
          U cur[4];
#pragma unroll
          for (int u = 0; u < 4; u++) cur[u] = nx[u];
          k += 4;
          if (k + 4 <= n) {
#pragma unroll
            for (int u = 0; u < 4; u++) nx[u] = t[k + u];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) p.P::reduce_function(acc, cur[u]);
        }
        for (; k < n; k++) { U v = t[k]; p.P::reduce_function(acc, v); }
      } else {
        while (mask) {
          const int k = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          U v = t[k];
          if (has) p.P::reduce_function(acc, v); else { acc = v; has = true; }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < G && my_row >= 0 && has) {
    y[my_row] = acc;
    if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[my_row >> 5], 1u << (my_row & 31));
  }
}

// ---- the same fold for the case the headline runs: every x entry present, no row filter, 2-operand program, edge
// positions that fit 32 bits, one slice of x (single GPU).  The general form above spends ~50 wave instructions per
// 64-edge chunk (64-bit positions re-broadcast per step, a branch around every load -- which also makes the compiler wait
// for ALL outstanding loads at every use, presence masks through LDS) and the kernel is bound by instruction issue, not
// by the vector-memory path: with every gather replaced by an LDS read it still took 1.9 of its 2.6 ms per RMAT-26 iteration
// (profiles/r04_instruction_diet.md).  Here the step body is branch-free -- loads are unconditional (positions clamped
// into the row, lanes that need no gather re-read the slice's first entry, an L1 hit), so the compiler counts the
// outstanding loads exactly -- and software-pipelined three deep: column ids of step s+2 and messages of step s+1 are in
// flight while step s is folded; a row's metadata (row id, CSR range, running value) is loaded while the PREVIOUS group
// is worked on (W16Meta).  Same products in the same order: bit-identical results.
template <class U>
struct W16Meta {
  int row, e0, len;
  bool has;
  U acc;
};
// the row id of list entry first + lane (lanes >= G and entries past the list: -1)
__device__ __forceinline__ int w16_row(const int32_t* __restrict__ rows, int nlist, int first, int lane) {
  const int i = first + lane;
  const bool ok = lane < kWaveRows && i < nlist && first >= 0;
  const int v = rows[ok ? i : 0];
  return ok ? v : -1;
}
template <class U>
__device__ __forceinline__ W16Meta<U> w16_meta(const gm_csr_t& A, int row, const U* __restrict__ y, const uint32_t* __restrict__ ybits, int accumulate) {
  W16Meta<U> mt;
  const int r = row >= 0 ? row : 0;
  const int64_t a = A.rowptr[r], b = A.rowptr[r + 1];
  mt.row = row;
  mt.e0 = (int)a;
  mt.len = row >= 0 ? (int)(b - a) : 0;
  mt.has = false;
  if (accumulate & ACC_READ_PREV) {
    const uint32_t w = ybits[r >> 5];
    mt.acc = y[r];
    mt.has = row >= 0 && ((w >> (r & 31)) & 1u);
  }
  return mt;
}
template <class P, class T, class U, class V, class E, int KSTRIDE>
__device__ __forceinline__ void wave16_dense(const P& p, const gm_csr_t& A, const W16Meta<U>& mt, const int lane, const T* __restrict__ x,
                                             U* __restrict__ y, uint32_t* __restrict__ ybits, const int accumulate GM_DBG_PARAM,
                                             const HotSet<T>& hot, U (*s_t)[KSTRIDE]) {
  constexpr int G = kWaveRows;
  int longest = mt.len;
  for (int off = 8; off > 0; off >>= 1) {
    const int o = __shfl_xor(longest, off, 64);
    longest = o > longest ? o : longest;
  }
  longest = __builtin_amdgcn_readfirstlane(longest);
  const int nsteps = (longest + 63) >> 6;
  if (nsteps == 0) return;
  bool has = mt.has;
  U acc = mt.acc;
  V no_vp;
  const int last_edge = (int)(A.nnz - 1);
  const T* __restrict__ xdummy = x + hot.base;  // (lanes without a gather of their own all read this entry: one L1-resident line)
  int c[G];
  T m[G];
  auto cols = [&](int step) {  // column ids of chunk `step` of every row (-1 past the row's end)
    const int pos = step * 64 + lane;
#pragma unroll
    for (int r = 0; r < G; r++) {
      const int lr = __builtin_amdgcn_readlane(mt.len, r);
      int k = __builtin_amdgcn_readlane(mt.e0, r) + (pos < lr ? pos : lr - 1);
      k = k < 0 ? 0 : (k > last_edge ? last_edge : k);
      const int v = stream_load(&A.colidx[k]);
      c[r] = pos < lr ? v : -1;
    }
  };
  auto msgs = [&]() {  // messages of the chunk whose column ids are in c
#pragma unroll
    for (int r = 0; r < G; r++) {
      const int cc = c[r];
      const unsigned rel = (unsigned)(cc - hot.base);
      bool in_lds = rel < (unsigned)hot.nhot;
      unsigned slot = rel;
#ifdef GRAPHMAT_ABLATION
      if (cc >= 0 && abl_cold(cc, hot.cold_from, hot.base)) { in_lds = true; slot = (unsigned)cc & 4095u; }
      if (dbg & DBG_SKIP_GATHER) { in_lds = true; slot = 0; }
#endif
      const T mh = hot.s_hot[in_lds ? slot : 0u];
      const T* __restrict__ ga = (in_lds || cc < 0) ? xdummy : x + cc;
      const T mg = *ga;
      m[r] = in_lds ? mh : mg;
    }
  };
  cols(0);
  msgs();
  if (nsteps > 1) cols(1);
  for (int step = 0; step < nsteps; step++) {
    // products of this step into the tile (every lane writes: a lane past its row's end writes a value nobody folds)
#pragma unroll
    for (int r = 0; r < G; r++) {
      U term;
      // (only programs that read edge values use it; a lane past its row's end -- its product is never folded -- must not read
      // past the end of the value array: the position is clamped like the column positions)
      const int kraw = __builtin_amdgcn_readlane(mt.e0, r) + step * 64 + lane;
      const int k = (int64_t)kraw < A.nnz ? kraw : (int)(A.nnz - 1);
      p.P::process_message(m[r], edge_at<E>(A.vals, k), no_vp, term);
      s_t[r][lane] = term;
    }
    __builtin_amdgcn_wave_barrier();
    if (step + 1 < nsteps) {
      msgs();                            // (their column ids were requested a whole step ago)
      if (step + 2 < nsteps) cols(step + 2);
    }
    if (lane < G && !(dbg & DBG_SKIP_FOLD)) {
      int n = mt.len - step * 64;
      n = n > 64 ? 64 : n;
      const U* t = s_t[lane];
      if (n == 64) {
        int k = 0;
        if (!has) { acc = t[0]; has = true; k = 1; }
        for (; k < 64; k++) { U v = t[k]; p.P::reduce_function(acc, v); }
      } else if (n > 0) {
        int k = 0;
        if (!has) { acc = t[0]; has = true; k = 1; }
        U nx[4];
        if (k + 4 <= n) {
#pragma unroll
          for (int u = 0; u < 4; u++) nx[u] = t[k + u];
        }
        while (k + 4 <= n) {
          U cur[4];
#pragma unroll
          for (int u = 0; u < 4; u++) cur[u] = nx[u];
          k += 4;
          if (k + 4 <= n) {
#pragma unroll
            for (int u = 0; u < 4; u++) nx[u] = t[k + u];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) p.P::reduce_function(acc, cur[u]);
        }
        for (; k < n; k++) { U v = t[k]; p.P::reduce_function(acc, v); }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < G && mt.row >= 0 && has) {
    y[mt.row] = acc;
    if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[mt.row >> 5], 1u << (mt.row & 31));
  }
}
// may the lean form run?  (wave-uniform)
template <class P, class T>
__device__ __forceinline__ bool w16_dense_ok(const gm_csr_t& A, const uint32_t* xbits, const uint32_t* want, const HotSet<T>& hot) {
  return xbits == nullptr && want == nullptr && !program_row_filter<P>::enabled && hot.NS == 1 && A.nnz < ((int64_t)1 << 31) && A.nnz > 0;
}

template <class P, class T, class U, class V, class E>
__global__ void __launch_bounds__(kWave16Block)
k_spmv_wave16(ProgArg<P> pa, gm_csr_t A, const int32_t* __restrict__ rows, int nlist, const T* __restrict__ x,
              const uint32_t* __restrict__ xbits, const V* __restrict__ vp, U* __restrict__ y,
              uint32_t* __restrict__ ybits, int accumulate GM_DBG_PARAM, const uint32_t* __restrict__ want) {
  static_assert(sizeof(U) == 4 || sizeof(U) == 8, "LDS tile of 4- or 8-byte products");
  constexpr int G = kWaveRows;
  constexpr int kStride = 64 + 16 / (int)sizeof(U);  // padded row of the tile (keeps 16-byte alignment, staggers banks)
  __shared__ __attribute__((aligned(16))) U s_t[kWave16Block / 64][G][kStride];
  __shared__ unsigned long long s_mask[kWave16Block / 64][G];
  // the hottest x entries live in LDS: at RMAT-26 the first 8192 vertices are the source of 14 % of the edges
  // (34 % at RMAT-22), and a gather served from LDS is one request less for the L2 (RMAT-26 wave rows 6.10 ->
  // 5.87 ms; 4096 / 8192 / 12288 / 16384 entries: 5.94 / 5.87 / 5.90 / 7.52 ms -- the last leaves one
  // workgroup per CU; the persistent form below is how a larger set pays)
  constexpr int kHot = sizeof(T) == 4 ? kHotEntries : 1;
  __shared__ T s_hot[kHot];
  const HotSet<T> hot = hot_load<T, kHot, kWave16Block>(A, x, s_hot);
  __syncthreads();
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int first = (blockIdx.x * (kWave16Block / 64) + wv) * G;
  if (first >= nlist) return;
  wave16_group<P, T, U, V, E, kStride>(p, A, rows, nlist, first, lane, x, xbits, vp, y, ybits, accumulate GM_DBG_PASS, want, hot, s_t[wv], s_mask[wv]);
}

// Persistent form with a LARGE hot set: the grid is a few workgroups per CU (not one per 64 rows), each
// loads HOT entries of the slice's busiest x values into LDS once and then its waves walk the row list,
// group after group (group g goes to wave g mod #waves: the list is degree-ranked, so consecutive groups
// cost about the same and the interleaving balances the waves).  With the set loaded once per workgroup
// instead of once per 64 rows it can take most of the CU's 160 KB.
template <class P, class T, class U, class V, class E, int BLOCK, int HOT>
__global__ void __launch_bounds__(BLOCK)
k_spmv_wave16p(ProgArg<P> pa, gm_csr_t A, const int32_t* __restrict__ rows, int nlist, const T* __restrict__ x,
               const uint32_t* __restrict__ xbits, const V* __restrict__ vp, U* __restrict__ y,
               uint32_t* __restrict__ ybits, int accumulate GM_DBG_PARAM, const uint32_t* __restrict__ want) {
  static_assert(sizeof(U) == 4 || sizeof(U) == 8, "LDS tile of 4- or 8-byte products");
  static_assert(sizeof(T) == 4, "4-byte messages");
  constexpr int G = kWaveRows;
  constexpr int kStride = 64 + 16 / (int)sizeof(U);
  __shared__ __attribute__((aligned(16))) U s_t[BLOCK / 64][G][kStride];
  __shared__ unsigned long long s_mask[BLOCK / 64][G];
  __shared__ T s_hot[HOT];
#ifdef GRAPHMAT_ABLATION
  const unsigned long long abl_t0 = wall_clock64();
#endif
  const HotSet<T> hot = hot_load<T, HOT, BLOCK>(A, x, s_hot);
  __syncthreads();
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ngroups = (nlist + G - 1) / G;
  const int nwaves = gridDim.x * (BLOCK / 64);
#ifdef GRAPHMAT_ABLATION
  const unsigned long long abl_t1 = wall_clock64();
  struct AblStamp {
    unsigned long long t0, t1;
    int w, lane;
    __device__ ~AblStamp() {
      if (lane == 0 && w < 8192) {
        g_abl_wave_times[0][w][0] = t0; g_abl_wave_times[0][w][1] = t1; g_abl_wave_times[0][w][2] = wall_clock64();
      }
    }
  } abl_stamp{abl_t0, abl_t1, (int)(blockIdx.x * (BLOCK / 64) + wv), lane};
#endif
  // wave w of the grid = workgroup (w mod gridDim.x): neighbouring groups -- the list is degree-ranked, so they cost
  // about the same -- go to different CUs, and the interleaving balances the waves.  (Handing the groups out off a
  // global work counter instead was measured: 26 000 same-address atomics per launch, 6.6 -> 7.5 ms per iteration.)
  if (w16_dense_ok<P, T>(A, xbits, want, hot) && !(dbg & DBG_PLAIN_WAVE16)) {  // (debug_flags 16384: the general form, for A/B runs)
    // lean form: the next group's row ids are loaded two groups ahead, its CSR range and running value one group ahead
    int g = wv * gridDim.x + blockIdx.x;
    int row_n = w16_row(rows, nlist, g + nwaves < ngroups ? (g + nwaves) * G : -1, lane);
    W16Meta<U> cur = w16_meta<U>(A, w16_row(rows, nlist, g < ngroups ? g * G : -1, lane), y, ybits, accumulate);
    while (g < ngroups) {
      const int gn = g + nwaves;
      const int row_nn = w16_row(rows, nlist, gn + nwaves < ngroups ? (gn + nwaves) * G : -1, lane);
      const W16Meta<U> nxt = w16_meta<U>(A, row_n, y, ybits, accumulate);
      wave16_dense<P, T, U, V, E, kStride>(p, A, cur, lane, x, y, ybits, accumulate GM_DBG_PASS, hot, s_t[wv]);
      cur = nxt;
      row_n = row_nn;
      g = gn;
    }
    return;
  }
  for (int g = wv * gridDim.x + blockIdx.x; g < ngroups; g += nwaves)
    wave16_group<P, T, U, V, E, kStride>(p, A, rows, nlist, g * G, lane, x, xbits, vp, y, ybits, accumulate GM_DBG_PASS, want, hot, s_t[wv], s_mask[wv]);
}

// Row-blocks by WAVES of persistent workgroups that share a large LDS hot set.  k_spmv_rowblock is bound by the
// per-CU vector-memory path (TA busy ~100 %: every gather is a request to the L2) and has no room for a hot set -- a
// workgroup only lives for ~1500 edges.  Here a workgroup stays (a few per CU), loads the busiest HOT x entries of the
// adjacency's column slice once, and each of its waves takes row-blocks by itself: 64 rows at a time (one lane per
// row), their edges in steps of 512 -- coalesced column ids, gathers served from LDS where the column is hot, messages
// staged in the wave's own LDS strip in edge order, then every lane folds the part of its row inside the step, in
// stored order, carrying its running value in registers.  No workgroup barrier after the hot set is loaded.
// Dense x, 2-operand programs, 4-byte messages; everything else keeps k_spmv_rowblock.
template <class P, class T, class U, class V, class E, int BLOCK, int HOT>
__global__ void __launch_bounds__(BLOCK)
k_spmv_rowwave(ProgArg<P> pa, gm_csr_t A, const T* __restrict__ x, U* __restrict__ y, uint32_t* __restrict__ ybits, int accumulate) {
  static_assert(sizeof(T) == 4, "4-byte messages");
  constexpr int W = BLOCK / 64;
  constexpr int CH = 512, PER = CH / 64;
  constexpr int kPadw = CH + CH / 32;
  __shared__ T s_hot[HOT > 0 ? HOT : 1];
  __shared__ T s_msg[W][kPadw];
#define GM_WSLOT(k) ((k) + ((k) >> 5))
#ifdef GRAPHMAT_ABLATION
  const unsigned long long abl_t0 = wall_clock64();
#endif
  const HotSet<T> hot = hot_load<T, HOT, BLOCK>(A, x, s_hot);
  __syncthreads();
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef GRAPHMAT_ABLATION
  const unsigned long long abl_t1 = wall_clock64();
  struct AblStamp {
    unsigned long long t0, t1;
    int w, lane;
    __device__ ~AblStamp() {
      if (lane == 0 && w < 8192) {
        g_abl_wave_times[1][w][0] = t0; g_abl_wave_times[1][w][1] = t1; g_abl_wave_times[1][w][2] = wall_clock64();
      }
    }
  } abl_stamp{abl_t0, abl_t1, (int)(blockIdx.x * (BLOCK / 64) + wv), lane};
#endif
  T* sm = s_msg[wv];
  const int nwaves = gridDim.x * W;
  V no_vp;
  for (int b = wv * gridDim.x + blockIdx.x; b < A.nblk; b += nwaves) {
    const int sg = A.blk_seg[b];
    const int r0 = A.seg_row[sg], r1 = A.seg_row[sg + 1];
    for (int rg = r0; rg < r1; rg += 64) {
      const int row = rg + lane;
      int64_t rp0 = 0, rp1 = 0;
      if (row < r1) { rp0 = A.rowptr[row]; rp1 = A.rowptr[row + 1]; }
      const int lastl = (r1 - rg - 1) < 63 ? (r1 - rg - 1) : 63;
      const int64_t g0 = wave_bcast(rp0, 0), g1 = wave_bcast(rp1, lastl);
      bool has = false;
      U acc;
      if (rp1 > rp0 && (accumulate & ACC_READ_PREV) && bit_get(ybits, row)) { acc = y[row]; has = true; }
      for (int64_t c0 = g0; c0 < g1; c0 += CH) {
        const int n = (int)((g1 - c0) < CH ? (g1 - c0) : CH);
        int c[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
          const int k = lane + 64 * j;
          c[j] = (k < n) ? stream_load(&A.colidx[c0 + k]) : -1;
        }
        T m[PER];
#pragma unroll
        for (int j = 0; j < PER; j++)
          if (c[j] >= 0) m[j] = hot.get(x, c[j]);
#pragma unroll
        for (int j = 0; j < PER; j++) {
          const int k = lane + 64 * j;
          if (k < n) sm[GM_WSLOT(k)] = m[j];
        }
        __builtin_amdgcn_wave_barrier();
        // this lane's row: the part of its segment that lies inside the step, in stored order
        const int64_t ka = rp0 > c0 ? rp0 : c0, kb = rp1 < c0 + n ? rp1 : c0 + n;
        if (ka < kb) {
          int k = (int)(ka - c0);
          const int ke = (int)(kb - c0);
          if (!has) {
            p.P::process_message(sm[GM_WSLOT(k)], edge_at<E>(A.vals, c0 + k), no_vp, acc);
            has = true;
            k++;
          }
          for (; k + 4 <= ke; k += 4) {
            T r[4];
#pragma unroll
            for (int u = 0; u < 4; u++) r[u] = sm[GM_WSLOT(k + u)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              U res;
              p.P::process_message(r[u], edge_at<E>(A.vals, c0 + k + u), no_vp, res);
              p.P::reduce_function(acc, res);
            }
          }
          for (; k < ke; k++) {
            U res;
            p.P::process_message(sm[GM_WSLOT(k)], edge_at<E>(A.vals, c0 + k), no_vp, res);
            p.P::reduce_function(acc, res);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      bool wrote = false;
      if (rp1 > rp0 && has) {
        y[row] = acc;
        wrote = true;
      }
      publish_row_bits(wrote, row, ybits, accumulate);
    }
  }
#undef GM_WSLOT
}

// ------------------------------------------------------------------------------------
// The rows of more than GM_SHORT_ROW edges that are not giant, in ONE launch per set: the row-stationary sweep over a
// sliced-ELLPACK layout (graphmat_hip.h: gm_sweep_t; built by gm_graph.hip: build_sweep; prototype and measurements:
// tools/sell_bench.hip, profiles/r05_sell_prototype*.txt -- RMAT-26: 876 M edges in 2.05 ms against 2.25 ms for 679 M of them
// in round 4's CSR-ordered sweep plus 1.2 ms of one-wave-per-row tile passes for the rest).
// Workgroup w (one per CU, 16 waves) owns the rows of length rank r % 256 == w and keeps their running values in LDS (medium
// rows) or in registers (long rows) from the first slice to the last; slices are native column ranges taken in ascending
// order and a (row, slice) piece holds its edges in ascending native column order, so every row is folded exactly as the
// reference folds it.  All workgroups walk the slices in the same order from the same start: at any time the chip gathers from
// one or two slices of x (~1.3 MB each: L2-resident), and each workgroup has the slice's busiest entries in LDS.
//  * medium rows: a wave streams its contiguous range of the block's groups as ONE sequence of 64-entry rows -- row k of a
//    group hands every lane the k-th edge of ITS piece (lane = piece, transposed storage): one coalesced load of byte offsets,
//    a branch-free gather (LDS hot set or L2), process_message, and the fold under the "entry is not padding" mask; U rows per
//    batch, the next batch's entries requested before this batch's messages are waited for; a group starts with a META row
//    (slots, first-piece flags, width) inside the same stream, at which the lanes' running values go back to LDS and the next
//    group's come out -- no other loads, nothing that would drain the prefetch;
//  * long rows (few, uneven: no 64-wide groups): all waves gather the block's long-row products into the LDS stage (coalesced
//    entries in (slot, column) order), then the last GM_SWEEP_LONG_SLOTS threads fold one piece each out of LDS while the other
//    waves start on their groups; blocks larger than the stage take several rounds.
// Dense x, 2-operand programs, 4-byte messages and reductions; edge values: none, or 4 bytes in the entries' positions.
// (ABL: measurement forms instantiated by tools/sweep_lib_bench.hip only -- 1: no gathers at all, 2: every gather served from
// LDS, 4: no long-row phase; their results are wrong by construction; 8: per-wave time per phase, 100 MHz ticks, into
// g_sell_phase_ticks[wave of the grid][phase])
#ifdef GM_SELL_PHASE_TIMES
static __device__ unsigned long long g_sell_phase_ticks[4096][8];
#define GM_SELL_TICK(k) do { if constexpr (ABL & 8) { const unsigned long long t_ = wall_clock64(); if (lane == 0) g_sell_phase_ticks[wg * 16 + wv][k] += t_ - tphase; tphase = t_; } } while (0)
#else
#define GM_SELL_TICK(k) do { } while (0)
#endif
// SHARDED (graphmat_hip.h: gm_sweep_t.nsub > 1; round 6): the rows are a shard's, the message vector is made of nsub owners' ranges of
// `stride` entries, slice_base holds positions inside a range -- a slice is the same sub-range of every owner's range, its hot set the
// first hq = min(slice length, hot_words / nsub) entries of each (LDS word q * hq + j) -- and the entries say at build time whether
// their message is in LDS (GM_SWEEP_HOT | LDS byte offset) or in the message vector (byte offset): the gather decodes, nothing else changes.
// SPARSE (round 6; ACTIVE_ONLY programs: x has presence bits): an entry only counts when its column's bit is set -- the message of an absent
// column is not even gathered --, the FIRST PRESENT message of a row assigns (the build's first-piece flags say nothing), so every accumulator
// slot has a "has a value" bit in LDS next to it and the long rows' stage a presence bit per product; rows that received nothing leave y and its
// presence bits alone.  Single-shard structures, two batches deep.
template <class P, class T, class U, class V, class E, bool HAS_VALS, int ABL, int UBATCH, int PIPE, int POOLW, bool SHARDED, int BLOCKT = 1024, bool SPARSE = false>
__device__ __forceinline__ void
sell_body(const ProgArg<P>& pa, int set, int stage_words, int nslices, int nrows_long, const int32_t* __restrict__ slice_base, const uint32_t* __restrict__ scol,
          const uint32_t* __restrict__ sval, const uint32_t* __restrict__ wrow, const int32_t* __restrict__ row_of_slot, const uint32_t* __restrict__ lcol,
          const uint32_t* __restrict__ lval, const uint32_t* __restrict__ lps, const int32_t* __restrict__ lrow_of_slot,
          const uint32_t* __restrict__ gcol, const uint32_t* __restrict__ gval, const uint32_t* __restrict__ gdst, const uint32_t* __restrict__ gslice,
          U* __restrict__ gterms /* products stream of the giant rows, or null: they gather for themselves */, const T* __restrict__ x, U* __restrict__ y,
          int nsub, int stride, int hot_words, const uint32_t* __restrict__ xbits = nullptr, uint32_t* __restrict__ ybits = nullptr,
          U* __restrict__ sterms = nullptr /* products stream of the short rows' STREAM groups, or null: nothing is stored for them */) {
  static_assert(sizeof(T) == 4 && sizeof(U) == 4, "4-byte messages and reductions");
  static_assert(!SPARSE || (PIPE == 2 && !SHARDED && ABL == 0), "the sparse form: single shard, two batches deep");
  constexpr int BLOCK = BLOCKT, W = BLOCK / 64, UB = UBATCH;
  constexpr int WS = 16;  // (wrow keeps WS + 1 entries per block whatever the workgroup's size: gm_sweep_t.waves)
  constexpr int ACC = GM_SWEEP_ACC_ROWS, NLP = GM_SWEEP_LONG_SLOTS;
  // entries of a staging round per thread (a smaller workgroup stages in rounds of at most 12 per thread: the engine caps its stage)
  constexpr int KMAX = BLOCK == 1024 ? GM_SWEEP_MAX_STAGE / BLOCK : 12;
  static_assert(BLOCK >= NLP && BLOCK % 64 == 0, "the last GM_SWEEP_LONG_SLOTS threads fold the long rows");
  static_assert((POOLW + GM_SWEEP_ACC_ROWS) * 4 <= 160 * 1024, "k_spmv_sell: the pool and the accumulators must fit gfx950's 160 KB of LDS per workgroup");
  __shared__ uint32_t s_pool[POOLW];  // [hot entries of the slice | stage of the long rows' products]
  __shared__ uint32_t s_acc[ACC];
  __shared__ uint32_t s_hasbits[SPARSE ? (ACC + 31) / 32 : 1];             // SPARSE: slot holds a value
  __shared__ uint32_t s_stagebits[SPARSE ? GM_SWEEP_MAX_STAGE / 32 : 1];   // SPARSE: staged product is present
  if constexpr (SPARSE) {
    for (int i = threadIdx.x; i < (ACC + 31) / 32; i += BLOCKT) s_hasbits[i] = 0u;  // (the first slice's barrier orders this before any use)
  }
  auto present = [&](uint32_t c4) { return ((xbits[c4 >> 7] >> ((c4 >> 2) & 31u)) & 1u) != 0u; };
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int wg = blockIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int HOT = POOLW - stage_words;
  uint32_t* const s_stage = s_pool + HOT;
  const char* __restrict__ xb = (const char*)x;
  V no_vp;
  const size_t vw = (size_t)set * 256 + wg;
  const int lj = (BLOCK - 1) - (int)threadIdx.x < NLP ? (BLOCK - 1) - (int)threadIdx.x : -1;  // long-row slot of this thread (slot 0 = the last thread: the last waves fold)
  bool lhas = false;
  U lacc;
  auto as_t = [](uint32_t raw) { T t; __builtin_memcpy(&t, &raw, 4); return t; };
  auto as_u = [](uint32_t raw) { U u; __builtin_memcpy(&u, &raw, 4); return u; };
  auto raw_u = [](const U& u) { uint32_t r; __builtin_memcpy(&r, &u, 4); return r; };
  auto as_e = [](uint32_t raw) { E e; if constexpr (HAS_VALS) __builtin_memcpy(&e, &raw, 4); else e = E(); return e; };
  // What a slice needs from global memory besides x is requested one slice AHEAD, before the barrier that ends the previous
  // slice: the wave's row range, its first batch of entries, the long rows' piece bounds and the first staging round's
  // entries -- after a workgroup barrier every wave would otherwise start with two or three exposed load latencies
  // (96 slices x ~3 us).  pc / pe: first batch, lc / le: staging round, (pr, prend): row range, (pl0, pl1, pps, ppe): long bounds.
  constexpr int GK = 2;  // giant-row entries per thread requested ahead (a share of more than GK * 1024 takes the loop below)
  uint32_t pc[UB], pe[UB], lc[KMAX], le[KMAX], gc[GK], gd[GK], ge[GK];
  uint32_t pr = 0, prend = 0, pl0 = 0, pl1 = 0, pps = 0, ppe = 0, pga = 0, pgb = 0;
  auto prefetch = [&](int sl) {
    const size_t blk = vw * (size_t)nslices + (size_t)sl;
    const uint32_t* __restrict__ wr = wrow + blk * (WS + 1);
    pr = __builtin_amdgcn_readfirstlane(wr[wv]);
    prend = __builtin_amdgcn_readfirstlane(wr[wv + 1]);
#pragma unroll
    for (int j = 0; j < UB; j++) {
      const uint32_t rr = pr + j < prend ? pr + j : (prend > 0 ? prend - 1 : 0);
      pc[j] = __builtin_nontemporal_load(&scol[(size_t)rr * 64 + lane]);
      if constexpr (HAS_VALS) pe[j] = __builtin_nontemporal_load(&sval[(size_t)rr * 64 + lane]); else pe[j] = 0u;
    }
    if (gterms != nullptr) {  // this workgroup's share of the slice's giant-row edges: the first GK per thread
      const uint32_t g0 = gslice[sl], gn = gslice[sl + 1] - g0;
      pga = g0 + (uint32_t)((unsigned long long)gn * (unsigned)wg / 256ull);
      pgb = g0 + (uint32_t)((unsigned long long)gn * (unsigned)(wg + 1) / 256ull);
#pragma unroll
      for (int j = 0; j < GK; j++) {
        const uint32_t i = pga + (uint32_t)(j * BLOCK) + threadIdx.x;
        if (i < pgb) {
          gc[j] = __builtin_nontemporal_load(&gcol[i]);
          gd[j] = __builtin_nontemporal_load(&gdst[i]);
          if constexpr (HAS_VALS) ge[j] = __builtin_nontemporal_load(&gval[i]); else ge[j] = 0u;
        }
      }
    }
    if (nrows_long > 0 && !(ABL & 4)) {
      const size_t eb = blk * NLP;
      pl0 = lps[eb];
      pl1 = lps[eb + NLP];
      if (lj >= 0) { pps = lps[eb + lj]; ppe = lps[eb + lj + 1]; }
      const uint32_t n = pl1 - pl0 < (uint32_t)stage_words ? pl1 - pl0 : (uint32_t)stage_words;
#pragma unroll
      for (int j = 0; j < KMAX; j++) {
        if ((uint32_t)(j * BLOCK) < n) {
          const uint32_t i = (uint32_t)(j * BLOCK) + threadIdx.x;
          const uint32_t ii = pl0 + (i < n ? i : n - 1);
          lc[j] = __builtin_nontemporal_load(&lcol[ii]);
          if constexpr (HAS_VALS) le[j] = __builtin_nontemporal_load(&lval[ii]); else le[j] = 0u;
        }
      }
    }
  };
  prefetch(0);
  [[maybe_unused]] unsigned long long tphase = 0;
  if constexpr (ABL & 8) tphase = wall_clock64();
  for (int sl = 0; sl < nslices; sl++) {
    const int base = slice_base[sl];
    const int slen = slice_base[sl + 1] - base;
    int hq = 0;  // (sharded: hot entries per owner)
    if constexpr (SHARDED) { const int cap = (hot_words < HOT ? hot_words : HOT) / nsub; hq = slen < cap ? slen : cap; }
    const int nhot = SHARDED ? hq * nsub : (slen < HOT ? slen : HOT);
    const uint32_t base4 = (uint32_t)base << 2, nhot4 = (uint32_t)nhot << 2;
    const uint32_t idle4 = SHARDED ? GM_SWEEP_HOT : base4;  // what a lane without an edge gathers (an LDS word / the slice's first entry)
    __syncthreads();  // the previous slice's folds are done: its hot set and stage may go, the running values are in s_acc
    GM_SELL_TICK(0);  // waiting for the other waves at the end of a slice
    // (requesting the hot entries before the barrier as well -- 12 of them per thread in registers -- was measured: 2.19 against
    // 2.09 ms, the kernel then needs all 128 VGPRs; the cold parts of the first staging round's and the giant rows' gathers
    // requested before the barriers: 2.15 against 2.10, with the giant rows 2.32 against 2.23)
    if constexpr (SHARDED) {
      for (int q = 0; q < nsub; q++) {
        const uint32_t* __restrict__ xq_ = (const uint32_t*)x + (size_t)q * (size_t)stride + (size_t)base;
        for (int i = threadIdx.x; i < hq; i += BLOCK) s_pool[q * hq + i] = xq_[i];
      }
    } else {
      for (int i = threadIdx.x; i < nhot; i += BLOCK) s_pool[i] = ((const uint32_t*)x)[base + i];
    }
    __syncthreads();
    GM_SELL_TICK(1);  // hot set
    auto gather = [&](uint32_t c4) {  // (branch-free: lanes whose column is in LDS re-read the slice's first entry, an L1 hit)
      if constexpr (SHARDED) {
        const bool h = (c4 & GM_SWEEP_HOT) != 0u;
        const uint32_t mh = *(const uint32_t*)((const char*)s_pool + (h ? (c4 & 0x3ffffffcu) : 0u));
        const uint32_t mg = *(const uint32_t*)(xb + (h ? base4 : c4));
        return h ? mh : mg;
      } else {
      const uint32_t rel4 = c4 - base4;
      if constexpr ((ABL & 3) == 1) return c4;
      if constexpr ((ABL & 3) == 2) return *(const uint32_t*)((const char*)s_pool + (nhot4 ? (rel4 % nhot4) & ~3u : 0u));
      const bool h = rel4 < nhot4;
      const uint32_t mh = *(const uint32_t*)((const char*)s_pool + (h ? rel4 : 0u));
      const uint32_t mg = *(const uint32_t*)(xb + (h ? base4 : c4));
      return h ? mh : mg;
      }
    };
    // the giant rows' share of this slice: their messages are requested here, next to the staging round's, and their products
    // go to the stream the giant rows' fold passes read (after the staging phase: by then they have arrived)
    uint32_t gmsg[GK];
    const uint32_t ga = pga, gb = pgb;
    if (gterms != nullptr) {
#pragma unroll
      for (int j = 0; j < GK; j++)
        if (ga + (uint32_t)(j * BLOCK) + threadIdx.x < gb) gmsg[j] = gather(gc[j]);
    }
    if (nrows_long > 0 && !(ABL & 4)) {
      const uint32_t l0 = pl0, l1 = pl1, ps = pps, pe_ = ppe;
      for (uint32_t c0 = l0; c0 < l1; c0 += (uint32_t)stage_words) {
        const uint32_t n = l1 - c0 < (uint32_t)stage_words ? l1 - c0 : (uint32_t)stage_words;
        if (c0 != l0) {
          __syncthreads();  // the previous round is folded: the stage may be overwritten
#pragma unroll
          for (int j = 0; j < KMAX; j++) {  // (later rounds: their entries are requested here)
            if ((uint32_t)(j * BLOCK) < n) {
              const uint32_t i = (uint32_t)(j * BLOCK) + threadIdx.x;
              const uint32_t ii = c0 + (i < n ? i : n - 1);
              lc[j] = __builtin_nontemporal_load(&lcol[ii]);
              if constexpr (HAS_VALS) le[j] = __builtin_nontemporal_load(&lval[ii]); else le[j] = 0u;
            }
          }
        }
        uint32_t m[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; j++)
          if ((uint32_t)(j * BLOCK) < n) {
            if constexpr (SPARSE) {
              const uint32_t i = (uint32_t)(j * BLOCK) + threadIdx.x;
              const bool pr = i < n && present(lc[j]);
              m[j] = gather(pr ? lc[j] : idle4);
              const unsigned long long pm = __ballot(pr);  // the wave's 64 consecutive stage entries
              if (lane == 0) s_stagebits[(i >> 5)] = (uint32_t)pm;
              if (lane == 32) s_stagebits[(i >> 5)] = (uint32_t)(pm >> 32);
            } else {
              m[j] = gather(lc[j]);
            }
          }
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
          const uint32_t i = (uint32_t)(j * BLOCK) + threadIdx.x;
          if (i < n) {
            U res;
            p.P::process_message(as_t(m[j]), as_e(le[j]), no_vp, res);
            s_stage[i] = raw_u(res);
          }
        }
        __syncthreads();
        GM_SELL_TICK(2);  // staging
        if (lj >= 0) {
          uint32_t k = ps > c0 ? ps : c0;
          const uint32_t ke = pe_ < c0 + n ? pe_ : c0 + n;
          if constexpr (SPARSE) {
            for (; k < ke; k++) {
              const uint32_t i = k - c0;
              if ((s_stagebits[i >> 5] >> (i & 31u)) & 1u) {
                if (lhas) p.P::reduce_function(lacc, as_u(s_stage[i])); else { lacc = as_u(s_stage[i]); lhas = true; }
              }
            }
          }
          if (k < ke && !lhas) { lacc = as_u(s_stage[k - c0]); lhas = true; k++; }
          for (; k + 4 <= ke; k += 4) {
            uint32_t r[4];
#pragma unroll
            for (int u = 0; u < 4; u++) r[u] = s_stage[k - c0 + u];
#pragma unroll
            for (int u = 0; u < 4; u++) p.P::reduce_function(lacc, as_u(r[u]));
          }
          for (; k < ke; k++) p.P::reduce_function(lacc, as_u(s_stage[k - c0]));
        }
      }
    }
    if (gterms != nullptr) {
#pragma unroll
      for (int j = 0; j < GK; j++) {
        if (ga + (uint32_t)(j * BLOCK) + threadIdx.x < gb) {
          U res;
          p.P::process_message(as_t(gmsg[j]), as_e(ge[j]), no_vp, res);
          gterms[gd[j]] = res;
        }
      }
      for (uint32_t i = ga + (uint32_t)(GK * BLOCK) + threadIdx.x; i < gb; i += BLOCK) {  // (a slice heavy in giant-row edges)
        const uint32_t c = __builtin_nontemporal_load(&gcol[i]), d = __builtin_nontemporal_load(&gdst[i]);
        uint32_t e = 0u;
        if constexpr (HAS_VALS) e = __builtin_nontemporal_load(&gval[i]);
        U res;
        p.P::process_message(as_t(gather(c)), as_e(e), no_vp, res);
        gterms[d] = res;
      }
    }
    // The wave's part of the block: rows [r, rend) of 64 entries each.  A group = one META row (bit 31 set in every lane;
    // bits 0-14 the lane's accumulator slot or 0x7fff, bit 15 "the row's first piece", bits 16-28 the group's width) followed by
    // `width` rows of column entries: the stream describes itself, the only loads are the entries, one batch ahead.
    GM_SELL_TICK(3);  // long folds (the last two waves)
    uint32_t r = pr;
    const uint32_t rend = prend;
    uint32_t left = 0;  // rows left in the current group (0: the next row is a meta row)
    int slot = 0x7fff;
    U acc;
    bool has = false;
    // STREAM groups (graphmat_hip.h: gm_sweep_t.nstream): bit 30 of the meta row; no lane has a slot, the products of row k of the group go
    // to sterms[(first + k) * 64 + lane], `first` spelled by bit 15 of lanes 0..31 -- the short rows' edges ride the sweep for their gathers
    // and are folded from that stream afterwards (k_short_fold)
    bool streaming = false;
    uint32_t srow = 0;
    if constexpr (PIPE == 2) {
      // Two batches deep: while batch k is folded, the messages of batch k + 1 and the entries of batch k + 2 are in flight (one
      // wait per iteration -- for the entries requested last, which the earlier gathers precede -- instead of entries, then
      // messages).  Every load is unconditional (rows past the stream's end re-read its last row, their gathers the slice's
      // first entry: cache hits), so the compiler counts outstanding loads exactly.
      uint32_t cA[UB], eA[UB], mA[UB], cB[UB], eB[UB];
#pragma unroll
      for (int j = 0; j < UB; j++) { cA[j] = pc[j]; eA[j] = pe[j]; }
      auto load_entries = [&](uint32_t r0, uint32_t (&cx)[UB], uint32_t (&ex)[UB]) {
#pragma unroll
        for (int j = 0; j < UB; j++) {
          const uint32_t rr = r0 + j < rend ? r0 + j : rend - 1;
          cx[j] = __builtin_nontemporal_load(&scol[(size_t)rr * 64 + lane]);
          if constexpr (HAS_VALS) ex[j] = __builtin_nontemporal_load(&sval[(size_t)rr * 64 + lane]); else ex[j] = 0u;
        }
      };
      // meta rows of a batch (wave-uniform; advances `left`); rows past the end count as set bits of `skip` only
      auto scan = [&](uint32_t r0, const uint32_t (&cx)[UB], uint32_t& skip) {
        uint32_t mask = 0;
        skip = 0;
#pragma unroll
        for (int j = 0; j < UB; j++) {
          if (r0 + j < rend) {
            if (left == 0) { mask |= 1u << j; left = ((uint32_t)__builtin_amdgcn_readfirstlane((int)cx[j]) >> 16) & 0x1fffu; }
            else left--;
          } else skip |= 1u << j;
        }
        skip |= mask;
        return mask;
      };
      auto gathers = [&](const uint32_t (&cx)[UB], uint32_t skip, uint32_t (&mx)[UB], uint32_t& px) {
        px = 0u;  // SPARSE: bit j = the lane's entry of row j is a real edge whose column is present
#pragma unroll
        for (int j = 0; j < UB; j++) {
          if constexpr (SPARSE) {
            const bool real = !((skip >> j) & 1u) && (int32_t)cx[j] >= 0;
            const bool pr = real && present(cx[j]);
            px |= pr ? (1u << j) : 0u;
            mx[j] = gather(pr ? cx[j] : idle4);
          } else {
            mx[j] = gather(((skip >> j) & 1u) ? idle4 : (cx[j] & 0x7fffffffu));
          }
        }
      };
      if (r < rend) {
        uint32_t skipA = 0, maskA = scan(r, cA, skipA), pA = 0u;
        gathers(cA, skipA, mA, pA);
        load_entries(r + UB, cB, eB);
        while (true) {
          const uint32_t rn = r + UB;
          uint32_t mB[UB], cC[UB], eC[UB];
          uint32_t skipB = 0;
          const uint32_t maskB = scan(rn, cB, skipB);
          uint32_t pB = 0u;
          gathers(cB, skipB, mB, pB);
          load_entries(rn + UB, cC, eC);
#pragma unroll
          for (int j = 0; j < UB; j++) {
            if (r + j < rend) {
              if ((maskA >> j) & 1u) {  // a new group: the previous group's running values go back to LDS, this one's come out
                if (slot != 0x7fff) {
                  s_acc[slot] = raw_u(acc);
                  if constexpr (SPARSE) { if (has) atomicOr(&s_hasbits[slot >> 5], 1u << (slot & 31)); }
                }
                slot = (int)(cA[j] & 0x7fffu);
                streaming = ((uint32_t)__builtin_amdgcn_readfirstlane((int)cA[j]) & 0x40000000u) != 0u;
                if (streaming) srow = (uint32_t)__ballot(((cA[j] >> 15) & 1u) != 0u);
                if constexpr (SPARSE) has = slot != 0x7fff && ((s_hasbits[slot >> 5] >> (slot & 31)) & 1u) != 0u;
                else has = !(cA[j] & 0x8000u);
                acc = as_u(slot != 0x7fff ? s_acc[slot] : 0u);
              } else if (streaming) {
                if (!SPARSE && sterms != nullptr && (int32_t)cA[j] >= 0) {
                  U res;
                  p.P::process_message(as_t(mA[j]), as_e(eA[j]), no_vp, res);
                  __builtin_nontemporal_store(raw_u(res), reinterpret_cast<uint32_t*>(sterms) + ((size_t)srow * 64 + lane));  // (streamed: the slice's part of x stays in L2)
                }
                srow++;
              } else if (SPARSE ? ((pA >> j) & 1u) != 0u : (int32_t)cA[j] >= 0) {
                U res;
                p.P::process_message(as_t(mA[j]), as_e(eA[j]), no_vp, res);
                if (has) p.P::reduce_function(acc, res);
                else acc = res;
                has = true;
              }
            }
          }
          r = rn;
          if (r >= rend) break;
          maskA = maskB;
          pA = pB;
#pragma unroll
          for (int j = 0; j < UB; j++) { cA[j] = cB[j]; eA[j] = eB[j]; mA[j] = mB[j]; cB[j] = cC[j]; eB[j] = eC[j]; }
        }
      }
    } else {
    uint32_t c[UB], ev[UB];
  #pragma unroll
      for (int j = 0; j < UB; j++) { c[j] = pc[j]; ev[j] = pe[j]; }
      auto entries = [&](uint32_t r0) {
  #pragma unroll
        for (int j = 0; j < UB; j++) {
          const uint32_t rr = r0 + j < rend ? r0 + j : rend - 1;
          c[j] = __builtin_nontemporal_load(&scol[(size_t)rr * 64 + lane]);
          if constexpr (HAS_VALS) ev[j] = __builtin_nontemporal_load(&sval[(size_t)rr * 64 + lane]); else ev[j] = 0u;
        }
      };
      while (r < rend) {
        uint32_t m[UB], cc[UB], e2[UB];
        // which rows of the batch are meta rows (wave-uniform): they are not gathered
        uint32_t metamask = 0, l = left;
  #pragma unroll
        for (int j = 0; j < UB; j++) {
          if (r + j < rend) {
            if (l == 0) { metamask |= 1u << j; l = ((uint32_t)__builtin_amdgcn_readfirstlane((int)c[j]) >> 16) & 0x1fffu; }
            else l--;
          }
        }
  #pragma unroll
        for (int j = 0; j < UB; j++) {
          cc[j] = c[j];
          e2[j] = ev[j];
          m[j] = 0u;
          if (!((metamask >> j) & 1u)) m[j] = gather(c[j] & 0x7fffffffu);
        }
        const uint32_t r0 = r;
        r += UB;
        if (r < rend) entries(r);
  #pragma unroll
        for (int j = 0; j < UB; j++) {
          if (r0 + j < rend) {
            if ((metamask >> j) & 1u) {  // a new group: the previous group's running values go back to LDS, this one's come out
              if (slot != 0x7fff) s_acc[slot] = raw_u(acc);
              slot = (int)(cc[j] & 0x7fffu);
              has = !(cc[j] & 0x8000u);
              acc = as_u(slot != 0x7fff ? s_acc[slot] : 0u);
            } else if ((int32_t)cc[j] >= 0) {
              U res;
              p.P::process_message(as_t(m[j]), as_e(e2[j]), no_vp, res);
              if (has) p.P::reduce_function(acc, res);  // SPMV.h:54-59: c = a; reduce(c, b)
              else acc = res;                           // no additive identity: the first message assigns (spmspv.h:73-77)
              has = true;
            }
          }
        }
        left = l;
      }
    }
    if (slot != 0x7fff) {
      s_acc[slot] = raw_u(acc);
      if constexpr (SPARSE) { if (has) atomicOr(&s_hasbits[slot >> 5], 1u << (slot & 31)); }
    }
    GM_SELL_TICK(4);  // groups
    if (sl + 1 < nslices) prefetch(sl + 1);  // (in flight while this wave waits for the others at the barrier)
    GM_SELL_TICK(5);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ACC; i += BLOCK) {
    const int row = row_of_slot[vw * ACC + i];  // (-1: no row in this slot; every row of the sweep has edges)
    if constexpr (SPARSE) {
      if (row >= 0 && ((s_hasbits[i >> 5] >> (i & 31)) & 1u)) { y[row] = as_u(s_acc[i]); atomicOr(&ybits[row >> 5], 1u << (row & 31)); }
    } else {
      if (row >= 0) y[row] = as_u(s_acc[i]);
    }
  }
  if (lj >= 0 && nrows_long > 0) {
    const int row = lrow_of_slot[vw * NLP + lj];
    if (row >= 0 && lhas) {
      y[row] = lacc;
      if constexpr (SPARSE) atomicOr(&ybits[row >> 5], 1u << (row & 31));
    }
  }
}
template <class P, class T, class U, class V, class E, bool HAS_VALS, int ABL = 0, int UBATCH = 7, int PIPE = 2, int POOLW = GM_SWEEP_POOL>
__global__ void __launch_bounds__(1024)
k_spmv_sell(ProgArg<P> pa, int set, int stage_words, int nslices, int nrows_long, const int32_t* __restrict__ slice_base, const uint32_t* __restrict__ scol,
            const uint32_t* __restrict__ sval, const uint32_t* __restrict__ wrow, const int32_t* __restrict__ row_of_slot, const uint32_t* __restrict__ lcol,
            const uint32_t* __restrict__ lval, const uint32_t* __restrict__ lps, const int32_t* __restrict__ lrow_of_slot,
            const uint32_t* __restrict__ gcol, const uint32_t* __restrict__ gval, const uint32_t* __restrict__ gdst, const uint32_t* __restrict__ gslice,
            U* __restrict__ gterms, const T* __restrict__ x, U* __restrict__ y) {
  sell_body<P, T, U, V, E, HAS_VALS, ABL, UBATCH, PIPE, POOLW, false>(pa, set, stage_words, nslices, nrows_long, slice_base, scol, sval, wrow, row_of_slot, lcol, lval, lps,
                                                                       lrow_of_slot, gcol, gval, gdst, gslice, gterms, x, y, 1, 0, 0);
}
// ... with the short rows' STREAM groups stored (gm_sweep_t.nstream; the first launch holds them all)
template <class P, class T, class U, class V, class E, bool HAS_VALS>
__global__ void __launch_bounds__(1024)
k_spmv_sell_stream(ProgArg<P> pa, int set, int stage_words, int nslices, int nrows_long, const int32_t* __restrict__ slice_base, const uint32_t* __restrict__ scol,
                   const uint32_t* __restrict__ sval, const uint32_t* __restrict__ wrow, const int32_t* __restrict__ row_of_slot, const uint32_t* __restrict__ lcol,
                   const uint32_t* __restrict__ lval, const uint32_t* __restrict__ lps, const int32_t* __restrict__ lrow_of_slot,
                   const uint32_t* __restrict__ gcol, const uint32_t* __restrict__ gval, const uint32_t* __restrict__ gdst, const uint32_t* __restrict__ gslice,
                   U* __restrict__ gterms, const T* __restrict__ x, U* __restrict__ y, U* __restrict__ sterms) {
  sell_body<P, T, U, V, E, HAS_VALS, 0, 7, 2, GM_SWEEP_POOL, false>(pa, set, stage_words, nslices, nrows_long, slice_base, scol, sval, wrow, row_of_slot, lcol, lval, lps,
                                                                     lrow_of_slot, gcol, gval, gdst, gslice, gterms, x, y, 1, 0, 0, nullptr, nullptr, sterms);
}
// The short rows' fold behind the sweep (gm_sweep_t.nstream): workgroup = one BIN of consecutive short rows (at most bin_cap + 63 products).
// The sweep left the bin's products in nslices chunks of the stream, (row, column) order inside a chunk; sinv says where each belongs in the
// bin's CSR order, the products are put there in LDS -- all chunks' entries dealt flat over the threads -- and one thread per row folds its run
// in ascending native column order: the first message assigns (spmspv.h:73-77), like every other kernel of the path.  ybits = nullptr: the
// presence bits are the graph's static ones.
template <class P, class U>
__global__ void __launch_bounds__(512, 6)
k_short_fold(ProgArg<P> pa, const U* __restrict__ sterms, const uint16_t* __restrict__ sinv, const uint32_t* __restrict__ schunk, int nslices, int cap,
             const uint32_t* __restrict__ sbin_row, const uint32_t* __restrict__ soff, const int32_t* __restrict__ srow_id, U* __restrict__ y,
             uint32_t* __restrict__ ybits) {
  static_assert(sizeof(U) == 4, "4-byte reductions");
  // (a bin's rows are degree-ranked: neighbouring threads fold rows of the SAME length L, i.e. read LDS L words apart -- one pad word per 32 keeps
  // L = 16 / 32 / 64 off a single bank)
  __shared__ uint32_t s_prod[(GM_STREAM_BIN + 64) + (GM_STREAM_BIN + 64) / 32 + 1];
  auto at = [](uint32_t q) { return q + (q >> 5); };
  __shared__ uint32_t s_cpos[GM_MAX_SLICES], s_cpre[GM_MAX_SLICES + 1];
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int b = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t* __restrict__ raw = reinterpret_cast<const uint32_t*>(sterms);
  const uint32_t i0 = sbin_row[b], i1 = sbin_row[b + 1], base = (uint32_t)b * (uint32_t)cap;
  // this thread's first rows are requested while the products arrive
  constexpr int RK = 4;
  uint32_t ro0[RK], ro1[RK];
  int rid[RK];
#pragma unroll
  for (int j = 0; j < RK; j++) {
    const uint32_t i = i0 + (uint32_t)(j * 512) + threadIdx.x;
    if (i < i1) { ro0[j] = soff[i]; ro1[j] = soff[i + 1]; rid[j] = srow_id[i]; }
  }
  if ((int)threadIdx.x < nslices) {  // where the bin's products of every slice are, and how many
    const uint2 pn = *reinterpret_cast<const uint2*>(&schunk[((size_t)b * nslices + threadIdx.x) * 2]);
    s_cpos[threadIdx.x] = pn.x;
    s_cpre[threadIdx.x] = pn.y;
  }
  __syncthreads();
  // wave w takes the chunks w, w + 8, ...: four chunks' first 256 products are requested before any of them is awaited
  constexpr int CB = 4, CK = 4;
  for (int c0 = wv; c0 < nslices; c0 += 8 * CB) {
    uint32_t vv[CB][CK];
    uint16_t qq[CB][CK];
    uint32_t cp[CB], cn[CB];
#pragma unroll
    for (int c = 0; c < CB; c++) {
      const int sl = c0 + 8 * c;
      cp[c] = sl < nslices ? s_cpos[sl] : 0u;
      cn[c] = sl < nslices ? s_cpre[sl] : 0u;
    }
#pragma unroll
    for (int c = 0; c < CB; c++) {
#pragma unroll
      for (int k = 0; k < CK; k++) {
        const uint32_t i = (uint32_t)(k * 64 + lane);
        if (i < cn[c]) {
          vv[c][k] = __builtin_nontemporal_load(&raw[cp[c] + i]);
          qq[c][k] = __builtin_nontemporal_load(&sinv[cp[c] + i]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CB; c++) {
#pragma unroll
      for (int k = 0; k < CK; k++)
        if ((uint32_t)(k * 64 + lane) < cn[c]) s_prod[at(qq[c][k])] = vv[c][k];
      for (uint32_t i = (uint32_t)(CK * 64 + lane); i < cn[c]; i += 64u) s_prod[at(sinv[cp[c] + i])] = raw[cp[c] + i];  // (a chunk of more than 256 products)
    }
  }
  __syncthreads();
  auto as_u = [](uint32_t r) { U u; __builtin_memcpy(&u, &r, 4); return u; };
  for (uint32_t ii = i0 + (uint32_t)(wv * 64), jj = 0; ii < i1; ii += 512u, jj++) {
    const uint32_t i = ii + (uint32_t)lane;
    const bool valid = i < i1;
    int row = 0;
    if (valid) {
      uint32_t o0, o1;
      if (jj < RK) {
        o0 = ro0[0]; o1 = ro1[0]; row = rid[0];
#pragma unroll
        for (int j = 1; j < RK; j++) if ((int)jj == j) { o0 = ro0[j]; o1 = ro1[j]; row = rid[j]; }
      } else { o0 = soff[i]; o1 = soff[i + 1]; row = srow_id[i]; }
      o0 -= base; o1 -= base;
      U acc = as_u(s_prod[at(o0)]);
      for (uint32_t k = o0 + 1; k < o1; k++) p.P::reduce_function(acc, as_u(s_prod[at(k)]));
      y[row] = acc;
    }
    if (ybits != nullptr) {  // (rows of a run are consecutive device ids: three words per wave instead of 64 atomics)
      const int r0 = __builtin_amdgcn_readfirstlane(row);
      if (__all(!valid || row == r0 + lane)) {
        const unsigned long long hb = __ballot(valid);
        const int sh = r0 & 31;
        const unsigned long long lo = hb << sh, hi = sh ? (hb >> (64 - sh)) : 0ull;
        const uint32_t w = lane == 0 ? (uint32_t)lo : lane == 1 ? (uint32_t)(lo >> 32) : lane == 2 ? (uint32_t)hi : 0u;
        if (lane < 3 && w != 0u) atomicOr(&ybits[(r0 >> 5) + lane], w);
      } else if (valid) {
        atomicOr(&ybits[row >> 5], 1u << (row & 31));
      }
    }
  }
}
// The giant rows' gathers in a kernel of their own (engine option sweep_form bit 4): gm_sweep_t.gcol is sorted by slice, so a grid-stride
// walk keeps all workgroups inside about one slice of the message vector at a time (L2-resident) without hot sets.  Runs on the auxiliary
// stream behind the sweep, next to the short rows' kernel, in front of the giant rows' fold passes: the sweep is then free of their 36 M
// (RMAT-26) gathers and stores.  Single-shard structures only (a shard's entries carry LDS offsets).
template <class P, class T, class U, class V, class E, bool HAS_VALS>
__global__ void __launch_bounds__(kBlock)
k_giant_gather_sliced(ProgArg<P> pa, const uint32_t* __restrict__ gcol, const uint32_t* __restrict__ gval, const uint32_t* __restrict__ gdst, int64_t n,
                      const T* __restrict__ x, U* __restrict__ gterms) {
  static_assert(sizeof(T) == 4 && sizeof(U) == 4, "4-byte messages and reductions");
  const P& p = *reinterpret_cast<const P*>(pa.b);
  V no_vp;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint32_t c4 = __builtin_nontemporal_load(&gcol[i]), d = __builtin_nontemporal_load(&gdst[i]);
    E ev = E();
    if constexpr (HAS_VALS) { const uint32_t raw = __builtin_nontemporal_load(&gval[i]); __builtin_memcpy(&ev, &raw, 4); }
    const T m = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(x) + c4);
    U res;
    p.P::process_message(m, ev, no_vp, res);
    gterms[d] = res;
  }
}

// The sweep in 768-thread workgroups (gm_sweep_t.waves = 12) with a smaller LDS pool: 12 waves x 128 VGPRs and ~124 KB of LDS leave every CU a
// quarter of its register file and 36 KB of LDS, so that the short rows' row-block kernel can run BESIDE the sweep (experiment of round 6)
template <class P, class T, class U, class V, class E, bool HAS_VALS>
__global__ void __launch_bounds__(768)
k_spmv_sell_w12(ProgArg<P> pa, int set, int stage_words, int nslices, int nrows_long, const int32_t* __restrict__ slice_base, const uint32_t* __restrict__ scol,
                const uint32_t* __restrict__ sval, const uint32_t* __restrict__ wrow, const int32_t* __restrict__ row_of_slot, const uint32_t* __restrict__ lcol,
                const uint32_t* __restrict__ lval, const uint32_t* __restrict__ lps, const int32_t* __restrict__ lrow_of_slot,
                const uint32_t* __restrict__ gcol, const uint32_t* __restrict__ gval, const uint32_t* __restrict__ gdst, const uint32_t* __restrict__ gslice,
                U* __restrict__ gterms, const T* __restrict__ x, U* __restrict__ y) {
  sell_body<P, T, U, V, E, HAS_VALS, 0, 7, 2, GM_SWEEP_POOL_W12, false, 768>(pa, set, stage_words, nslices, nrows_long, slice_base, scol, sval, wrow, row_of_slot, lcol, lval, lps,
                                                                             lrow_of_slot, gcol, gval, gdst, gslice, gterms, x, y, 1, 0, 0);
}

// the sweep over a SPARSE message vector (ACTIVE_ONLY programs; single shard): xbits = the presence bits of x, ybits = those of y (OR-ed in)
template <class P, class T, class U, class V, class E, bool HAS_VALS>
__global__ void __launch_bounds__(1024)
k_spmv_sell_sparse(ProgArg<P> pa, int set, int stage_words, int nslices, int nrows_long, const int32_t* __restrict__ slice_base, const uint32_t* __restrict__ scol,
                   const uint32_t* __restrict__ sval, const uint32_t* __restrict__ wrow, const int32_t* __restrict__ row_of_slot, const uint32_t* __restrict__ lcol,
                   const uint32_t* __restrict__ lval, const uint32_t* __restrict__ lps, const int32_t* __restrict__ lrow_of_slot, const T* __restrict__ x,
                   const uint32_t* __restrict__ xbits, U* __restrict__ y, uint32_t* __restrict__ ybits) {
  sell_body<P, T, U, V, E, HAS_VALS, 0, 7, 2, GM_SWEEP_POOL_SPARSE, false, 1024, true>(pa, set, stage_words, nslices, nrows_long, slice_base, scol, sval, wrow, row_of_slot, lcol, lval,
                                                                                        lps, lrow_of_slot, nullptr, nullptr, nullptr, nullptr, (U*)nullptr, x, y, 1, 0, 0, xbits, ybits);
}

// the same sweep over a shard's rows (gm_sweep_t.nsub > 1)
template <class P, class T, class U, class V, class E, bool HAS_VALS>
__global__ void __launch_bounds__(1024)
k_spmv_sell_sharded(ProgArg<P> pa, int set, int stage_words, int nslices, int nrows_long, const int32_t* __restrict__ slice_base, const uint32_t* __restrict__ scol,
                    const uint32_t* __restrict__ sval, const uint32_t* __restrict__ wrow, const int32_t* __restrict__ row_of_slot, const uint32_t* __restrict__ lcol,
                    const uint32_t* __restrict__ lval, const uint32_t* __restrict__ lps, const int32_t* __restrict__ lrow_of_slot,
                    const uint32_t* __restrict__ gcol, const uint32_t* __restrict__ gval, const uint32_t* __restrict__ gdst, const uint32_t* __restrict__ gslice,
                    U* __restrict__ gterms, const T* __restrict__ x, U* __restrict__ y, int nsub, int stride, int hot_words, U* __restrict__ sterms) {
  sell_body<P, T, U, V, E, HAS_VALS, 0, 7, 2, GM_SWEEP_POOL, true>(pa, set, stage_words, nslices, nrows_long, slice_base, scol, sval, wrow, row_of_slot, lcol, lval, lps,
                                                                    lrow_of_slot, gcol, gval, gdst, gslice, gterms, x, y, nsub, stride, hot_words, nullptr, nullptr, sterms);
}

// ------------------------------------------------------------------------------------
// The short rows of a graph WITHOUT skew as a column-blocked stream (graphmat_hip.h: gm_blocked_t; built by gm_graph.hip:
// build_blocked; prototype and measurements: tools/blocked_bench.hip, profiles/r05_short_rows_blocked_stream_prototype.md --
// uniform 16-out-regular 2^26: 9.0-9.5 ms against 21 ms for the row-blocks, whose every gather misses).
// Workgroup b of a pass owns GM_BLOCKED_ROWS short rows and keeps their running values in LDS from the first slice to the last; the
// grid's workgroups take the blocks of a pass side by side and walk the slices TOGETHER: a workgroup that has finished step (pass,
// slice) adds itself to its XCD's counter for that step (workgroups are dealt round-robin over the 8 XCDs), and nobody starts the next
// slice before all workgroups of the XCD have finished step + 1 - window -- the XCD's L2 then holds window + 1 slices of x whatever
// the workgroups' pace.  The wait is bounded: a workgroup that never arrives costs locality, not progress.  A (block, slice)
// segment's entries are in (row, CSR order) order; the 16 waves take equal shares cut at row borders (woff), so inside a slice no
// two waves touch the same running value; lanes holding the same row form a run whose head lane reads the running value, folds
// the run's products one after the other and writes it back: ascending native column order, the reference's.
// Dense x, 2-operand programs, 4-byte messages and reductions; edge values: none, or 4 bytes in the entries' positions (eval).
template <class P, class T, class U, class V, class E, bool HAS_VALS, int UB = 2>
__global__ void __launch_bounds__(1024)
k_spmv_blocked(ProgArg<P> pa, const uint32_t* __restrict__ ecol, const uint16_t* __restrict__ erow, const uint32_t* __restrict__ eval, const uint32_t* __restrict__ woff, int nslices, int nblk,
               const int32_t* __restrict__ row_of, const T* __restrict__ x, U* __restrict__ y, unsigned int* __restrict__ step_count, int nsteps, int window) {
  static_assert(sizeof(T) == 4 && sizeof(U) == 4, "4-byte messages and reductions");
  extern __shared__ uint32_t s_bacc[];  // GM_BLOCKED_ROWS running values
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned int nwg = (gridDim.x + 7 - (blockIdx.x & 7)) >> 3;  // workgroups of this XCD
  unsigned int* const mycnt = step_count + (size_t)(blockIdx.x & 7) * nsteps;
  const int npass = (nblk + (int)gridDim.x - 1) / (int)gridDim.x;
  V no_vp;
  bool gave_up = false;  // (thread 0: a wait for the XCD's other workgroups timed out once -- no more waiting in this launch)
  auto as_u = [](uint32_t raw) { U u; __builtin_memcpy(&u, &raw, 4); return u; };
  auto raw_u = [](const U& u) { uint32_t r; __builtin_memcpy(&r, &u, 4); return r; };
  auto as_e = [](uint32_t raw) { E e; if constexpr (HAS_VALS) __builtin_memcpy(&e, &raw, 4); else e = E(); return e; };
  for (int pass = 0; pass < npass; pass++) {
    const int blk = pass * (int)gridDim.x + (int)blockIdx.x;
    const bool has_blk = blk < nblk;  // (a workgroup without a block in the last pass still reports its steps)
    uint32_t ws = 0, we = 0;
    if (has_blk) { const uint32_t* wo = woff + ((size_t)blk * nslices) * 17 + wave; ws = wo[0]; we = wo[1]; }
    uint32_t c[UB], r[UB], ev[UB];
    uint32_t pre = ws;
#pragma unroll
    for (int u = 0; u < UB; u++) { c[u] = ecol[pre + u * 64 + lane]; r[u] = erow[pre + u * 64 + lane]; if constexpr (HAS_VALS) ev[u] = eval[pre + u * 64 + lane]; else ev[u] = 0u; }  // (both arrays are padded behind the last entry)
    for (int sl = 0; sl < nslices; sl++) {
      uint32_t nws = 0, nwe = 0;
      if (has_blk && sl + 1 < nslices) { const uint32_t* wo = woff + ((size_t)blk * nslices + sl + 1) * 17 + wave; nws = wo[0]; nwe = wo[1]; }  // (the next slice's range: requested now)
      if (pre != ws && ws < we) {  // (an empty range in between: the batch requested ahead is not this slice's first)
        pre = ws;
#pragma unroll
        for (int u = 0; u < UB; u++) { c[u] = ecol[pre + u * 64 + lane]; r[u] = erow[pre + u * 64 + lane]; if constexpr (HAS_VALS) ev[u] = eval[pre + u * 64 + lane]; else ev[u] = 0u; }
      }
      for (uint32_t p0 = ws; p0 < we;) {
        T m[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) m[u] = x[c[u]];
        // the next batch -- of this slice or the next -- is requested before this one is folded
        const uint32_t np0 = p0 + 64 * UB;
        const uint32_t nxt = np0 < we ? np0 : nws;
        uint32_t nc[UB], nr[UB], ne[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) { nc[u] = ecol[nxt + u * 64 + lane]; nr[u] = erow[nxt + u * 64 + lane]; if constexpr (HAS_VALS) ne[u] = eval[nxt + u * 64 + lane]; else ne[u] = 0u; }
#pragma unroll
        for (int u = 0; u < UB; u++) {
          const bool valid = p0 + u * 64 + lane < we;
          const uint32_t id = valid ? (r[u] & 0x7fffu) : (0x10000u + lane);
          const uint32_t prev = __shfl_up(id, 1);
          const bool head = valid && (lane == 0 || id != prev);
          const unsigned long long H = __ballot(head) | __ballot(!valid);  // (an entry past the range ends a run as a head would)
          const unsigned long long above = lane == 63 ? 0ull : (H >> (lane + 1));
          const int runlen = above ? (__builtin_ctzll(above) + 1) : (64 - lane);
          U res;
          __builtin_memset(&res, 0, 4);
          if (valid) p.P::process_message(m[u], as_e(ev[u]), no_vp, res);
          const uint32_t rraw = raw_u(res);
          U acc = res;
          if (head && !(r[u] & 0x8000u)) { acc = as_u(s_bacc[id]); p.P::reduce_function(acc, res); }  // SPMV.h:54-59: c = a; reduce(c, b)
          for (int k = 1; __ballot(head && k < runlen); k++) {
            const U t = as_u(__shfl(rraw, (lane + k) & 63));
            if (head && k < runlen) p.P::reduce_function(acc, t);
          }
          if (head) s_bacc[id] = raw_u(acc);
        }
#pragma unroll
        for (int u = 0; u < UB; u++) { c[u] = nc[u]; r[u] = nr[u]; ev[u] = ne[u]; }
        pre = nxt;
        p0 = np0;
      }
      __syncthreads();  // the slice is folded: the next one's shares cut the rows differently
      if (window > 0) {
        // ONE lane of the workgroup reports and waits, with a pause between two looks (all waves of an XCD polling one line of the
        // L2 keep the reports themselves from getting through); the barrier hands the result to the other waves
        const int step = pass * nslices + sl;
        if (threadIdx.x == 0) {
          __hip_atomic_fetch_add(&mycnt[step], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int need = step + 1 - window;
          // (the wait is bounded -- ~1000 looks, a few hundred microseconds, several times a step's duration -- and STICKY: a workgroup
          // that once waited in vain (its XCD's workgroups are not all resident: a masked or shared device) stops waiting for the rest
          // of the launch instead of paying the timeout at every step; it keeps reporting, so the others are not held up by it)
          if (need >= 0 && !gave_up) {
            int tries = 0;
            for (; tries < 1000; tries++) {
              if (__hip_atomic_load(&mycnt[need], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nwg) break;
              __builtin_amdgcn_s_sleep(4);
            }
            if (tries == 1000) gave_up = true;
          }
        }
        __syncthreads();
      }
      ws = nws; we = nwe;
    }
    if (has_blk) {
      const int32_t* __restrict__ ro = row_of + (size_t)blk * GM_BLOCKED_ROWS;
      for (int i = threadIdx.x; i < GM_BLOCKED_ROWS; i += 1024) {
        const int row = ro[i];  // (-1: past the last short row; every short row has an edge: its value was assigned)
        if (row >= 0) y[row] = as_u(s_bacc[i]);
      }
    }
    __syncthreads();  // the next pass's first messages are assigned into the same words
  }
}

// a=b programs, the wanted rows among a wave's 64 list entries: 64 / LPR rows at a time, LPR lanes each.
// A bottom-up level is a chain of dependent loads per row (row pointers, column ids, summary bit, presence bit, the
// winner's message) and most rows end in their last few edges, so what counts is how many rows a wave has in flight:
// every LPR-lane group scans its row backwards -- 2 LPR edges first, doubling up to 8 LPR in flight while nothing is
// found -- and takes the next wanted row as soon as it is done.  lane l holds list entry l (row, e0, e1; `todo` =
// wanted lanes).  (RMAT-26, first bottom-up level: one row per wave at a time 1.14 ms, LPR 16: 0.73 ms; whole
// traversals with LPR 8 another 0.1-0.25 ms faster than with 16.)
template <class P, class T, class U, class V, class E, bool USE_VP, int LPR>
__device__ __forceinline__ void group_rows_last(const P& p, const gm_csr_t& A, const int row, const int64_t e0, const int64_t e1,
                                                const unsigned long long todo, const int lane, const T* __restrict__ x,
                                                const uint32_t* __restrict__ xbits, const V* __restrict__ vp, U* __restrict__ y,
                                                uint32_t* __restrict__ ybits, const int accumulate,
                                                const uint32_t* __restrict__ xsum, unsigned char* s_idx /* 64 per wave */) {
  static_assert(LPR == 8 || LPR == 16 || LPR == 32, "lanes per row");
  const bool dense = (xbits == nullptr);
  const int nw = __popcll(todo);
  if ((todo >> lane) & 1ull) s_idx[__popcll(todo & ((1ull << lane) - 1ull))] = (unsigned char)lane;
  __builtin_amdgcn_wave_barrier();
  const int g = lane / LPR, sub = lane % LPR;
  const int e0l = (int)(uint32_t)e0, e0h = (int)(e0 >> 32), e1l = (int)(uint32_t)e1, e1h = (int)(e1 >> 32);
  int cur = 0, depth = 2, next = 0;
  int64_t lo = 0, hi = 0;
  bool active = false;
  constexpr int DMAX = 8;
  while (true) {
    // idle groups take the next wanted rows (in list order)
    const unsigned long long idle = __ballot(!active && sub == 0);
    const int k = next + __popcll(idle & ((1ull << (g * LPR)) - 1ull));
    const bool take = !active && k < nw;
    const int src = take ? (int)s_idx[k] : 0;
    const int r2 = __shfl(row, src, 64);
    const int a_l = __shfl(e0l, src, 64), a_h = __shfl(e0h, src, 64), b_l = __shfl(e1l, src, 64), b_h = __shfl(e1h, src, 64);
    if (take) {
      cur = r2;
      lo = ((int64_t)a_h << 32) | (uint32_t)a_l;
      hi = ((int64_t)b_h << 32) | (uint32_t)b_l;
      depth = 2;
      active = true;
    }
    next += __popcll(idle);
    if (__ballot(active) == 0ull) break;
    int c[DMAX];
    bool pres[DMAX];
#pragma unroll
    for (int u = 0; u < DMAX; u++) {
      const int64_t kk = hi - LPR * (u + 1) + sub;
      c[u] = (active && u < depth && kk >= lo) ? stream_load(&A.colidx[kk]) : -1;
    }
#pragma unroll
    for (int u = 0; u < DMAX; u++)
      pres[u] = c[u] >= 0 && (dense || ((xsum == nullptr || bit_get(xsum, c[u] >> 6)) && bit_get(xbits, c[u])));
    bool found = false;
#pragma unroll
    for (int u = 0; u < DMAX; u++) {
      const unsigned seg = (unsigned)((__ballot(pres[u]) >> (g * LPR)) & ((1ull << LPR) - 1ull));
      if (!found && seg) {
        found = true;
        if (sub == 31 - __clz((int)seg)) {  // the present edge nearest the row's end
          const int64_t kk = hi - LPR * (u + 1) + sub;
          V vprow;
          if constexpr (USE_VP) vprow = vp[cur];
          T m = message_of(p, x, vp, c[u]);
          U res;
          p.P::process_message(m, edge_at<E>(A.vals, kk), vprow, res);
          y[cur] = res;
          if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[cur >> 5], 1u << (cur & 31));
        }
      }
    }
    if (active) {
      if (found) active = false;
      else {
        hi -= LPR * depth;
        if (hi <= lo) active = false;  // nothing present: an earlier pass's value (accumulate) simply stays
        depth = depth < DMAX ? depth * 2 : DMAX;
      }
    }
  }
}

// The same for programs with a row filter once most rows have dropped out: a wave takes 64
// entries of the row list, tests their filter bits with one lane each, and then works through
// the rows that are still wanted one after the other -- a level of BFS in which few rows are
// unvisited costs a 64th of the waves (launching a wave per row just to find it filtered out was
// most of the time of such levels).
template <class P, class T, class U, class V, class E, bool USE_VP, int RK, int LAST_LPR = 8>
__global__ void __launch_bounds__(kBlock)
k_spmv_wave_grouped(ProgArg<P> pa, gm_csr_t A, const int32_t* __restrict__ rows, int nlist, const T* __restrict__ x,
                    const uint32_t* __restrict__ xbits, const V* __restrict__ vp, U* __restrict__ y,
                    uint32_t* __restrict__ ybits, int accumulate GM_DBG_PARAM, const uint32_t* __restrict__ want,
                    const uint32_t* __restrict__ xsum = nullptr) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int lane = threadIdx.x & 63;
  const int idx = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64 + lane;
  int row = 0;
  int64_t e0 = 0, e1 = 0;
  bool wanted = false;
  if (idx < nlist) {
    row = rows[idx];
    wanted = row_wanted(p, vp, want, row);
    if (wanted) { e0 = A.rowptr[row]; e1 = A.rowptr[row + 1]; }
  }
  unsigned long long todo = __ballot(wanted);
  if constexpr (RK == REDUCE_LAST) {
    __shared__ unsigned char s_idx[kBlock / 64][64];
    group_rows_last<P, T, U, V, E, USE_VP, LAST_LPR>(p, A, row, e0, e1, todo, lane, x, xbits, vp, y, ybits, accumulate, xsum, s_idx[threadIdx.x >> 6]);
    return;
  }
  while (todo) {
    const int r = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int rr = __builtin_amdgcn_readlane(row, r);
    const int64_t a = wave_bcast(e0, r), b = wave_bcast(e1, r);
    wave_row<P, T, U, V, E, USE_VP, RK>(p, A, rr, a, b, lane, x, xbits, vp, y, ybits, accumulate GM_DBG_PASS, nullptr, nullptr, xsum);
  }
}

// ------------------------------------------------------------------------------------
// Exact parallel replay of a sequential float accumulation (REDUCE_F32_ADD).
//
// The reference adds a row's messages one by one in fp32.  While the running sum S
// stays inside one binade [2^e, 2^(e+1)) every step is S <- S + round_to_multiple(a, u)
// with u = ulp(S) = 2^(e-23): in units of u, S is an integer in [2^23, 2^24), a = q + f
// with integer q and fraction f, and round-to-nearest-even adds q, plus 1 when f > 1/2,
// plus (parity of S+q) when f == 1/2.  So each term is a map on (integer S), depending on
// S only through its parity in the tie case; such maps compose associatively as a pair
// (delta if S even, delta if S odd).  Each lane composes the maps of its consecutive
// terms, an ordered block-wide scan composes the lanes, and the longest prefix whose
// terms are all finite, non-negative and below 2^e and which keeps S below 2^24 (no
// binade crossing; S is monotone) is accepted in one step.  The lane group where the
// prefix stops is folded serially (this is where S changes binade), then the replay
// resumes in the new binade.  Results are bit-identical to the serial loop.
struct ulp_map {
  uint32_t de, dod;  // ulps added when the incoming S is even / odd
};
// Deltas are non-negative and a prefix is only accepted while S stays below 2^24, so
// partial results saturate at 2^24 (no 32-bit wrap; a saturated value forces a stop).
constexpr uint32_t kUlpSat = 0x1000000u;
__device__ __forceinline__ ulp_map ulp_compose(ulp_map a, ulp_map b) {  // a first, then b
  ulp_map r;
  r.de = a.de + ((a.de & 1u) ? b.dod : b.de);
  r.dod = a.dod + (((a.dod + 1u) & 1u) ? b.dod : b.de);
  r.de = r.de > kUlpSat ? kUlpSat : r.de;
  r.dod = r.dod > kUlpSat ? kUlpSat : r.dod;
  return r;
}
// map of one term `a` (float bits) against binade exponent field eS (biased, of S).
// returns false when the term cannot be handled in this binade (a >= 2^e, negative, nan/inf).
__device__ __forceinline__ bool ulp_term(uint32_t abits, int eS, ulp_map& out) {
  if (abits == 0u) { out.de = 0; out.dod = 0; return true; }
  if (abits >> 31) return false;  // negative (or -0: fold serially)
  int ea = (int)(abits >> 23);
  if (ea == 255) return false;
  uint32_t ma = abits & 0x7fffffu;
  if (ea == 0) ea = 1; else ma |= 0x800000u;  // subnormal: no implicit one, exponent 1
  int sh = eS - ea;                            // a = ma * 2^(ea-150), u = 2^(eS-150)
  if (sh < 1) return false;
  if (sh > 25) { out.de = 0; out.dod = 0; return true; }  // f < 1/2, q = 0
  uint32_t q = ma >> sh;
  uint32_t rem = ma & ((1u << sh) - 1u);
  uint32_t half = 1u << (sh - 1);
  uint32_t up = rem > half ? 1u : 0u;
  uint32_t tie = rem == half ? 1u : 0u;
  // incoming S even: S+q parity = q&1; incoming odd: parity = (q+1)&1
  out.de = q + up + (tie & (q & 1u));
  out.dod = q + up + (tie & ((q + 1u) & 1u));
  return true;
}

// ---- the exact replay spread over MANY workgroups (REDUCE_F32_ADD, every x entry present) -----------
// One workgroup per row walking 8192-product chunks is a serial chain as long as the row (the 854 K-edge hub row of
// RMAT-26: 104 chunks, ~1.1 ms -- as long as a whole iteration of a shard of 8).  The ulp-maps of the replay compose
// associatively while the running sum S stays inside one binade, and a piece's composed map depends on S only
// through S's binade.  So every 4096-product piece keeps a small record: k_spmv_giant leaves, as a HINT for the next
// pass over the same rows (the next iteration: a row's partial sums move slowly), the binade S was in throughout the
// piece's chunk; k_giant_terms -- one workgroup per piece, the whole chip, the products in its registers anyway --
// composes the piece's exact map against the hinted binade; and k_spmv_giant merely APPLIES the maps of a chunk when
// the exact S it has reached really is in that binade and really stays there (S + delta < 2^24 ulps; S only grows:
// terms are non-negative or the piece has no map) -- otherwise, and around every binade crossing, it replays the chunk
// itself as before.  The hint decides only WHICH pieces get a map, never a value: the bits are the serial loop's.
// (round 6: one record per SUB-PIECE of kGiantSub = 512 products, 8 per piece, and the replay is k_giant_replay_maps: one WAVE per row
// walks the records 64 at a time -- a wave scan composes the maps and finds the first one that does not apply -- and replays only that
// sub-piece, with wave scans and no barrier.  On a shard of 8 of RMAT-26 the hub row's chain fell from ~0.45 ms to ~0.1 ms.)
constexpr int kGiantSub = 512;
// Which binade will S be in?  (Rounds 2-5 took the binade of the PREVIOUS pass as a hint; while a run's values still move -- the first
// ten PageRank iterations -- the places where S crosses into the next binade move by many sub-pieces from one pass to the next, every
// sub-piece in between saw S off its hint, and the replay of the hub row took 0.3-2.6 ms instead of 0.2.)  Round 6 PREDICTS it from
// this pass's own products: k_giant_sums adds up every sub-piece in double, k_giant_maps adds the sums of the row's earlier sub-pieces -- the double
// prefix differs from the float sum S by accumulated rounding only, ~1e-4 relative at worst -- and a sub-piece whose predicted S has the
// same binade at its start and at its end gets its map composed for that binade (k_giant_maps).  A wrong prediction (S within ~1e-4 of a
// power of two) costs the replay of one sub-piece, never a bit: the replay applies a map only when the EXACT S is in the map's binade
// and stays there.  No state survives a pass.
struct gchunk_state {
  double sum;         // k_giant_sums: the sub-piece's products added up in double
  int32_t e_map;      // k_giant_maps: the binade S is expected to have throughout the sub-piece and the map was composed for (0: none)
  uint32_t de, dod;   // k_giant_maps: ulps the sub-piece adds when the incoming S is even / odd
  uint32_t pad_[3];
};
static_assert(sizeof(gchunk_state) == 32, "gchunk_state: 32-byte records (gm_graph.hip allocates them)");

// ------------------------------------------------------------------------------------
// giant rows, pass 1 (REDUCE_F32_ADD): the gathers and products of a giant row are spread
// over many workgroups (one per GM_GIANT_CHUNK edges) because a single CU can only issue
// about one gather per 2.7 cycles; the products go to a scratch stream in edge order
// (plus presence words when x is sparse) that the ordered fold of pass 2 merely streams.
template <class P, class T, class U, class V, class E, bool USE_VP>
__global__ void __launch_bounds__(kBlock)
k_giant_terms(ProgArg<P> pa, gm_csr_t A, const T* __restrict__ x, const uint32_t* __restrict__ xbits,
              const V* __restrict__ vp, U* __restrict__ terms, unsigned long long* __restrict__ tpres GM_DBG_PARAM,
              gchunk_state* __restrict__ state /* unused since round 6 (kept so that the call sites keep their shape) */) {
  constexpr int PER = GM_GIANT_CHUNK / kBlock;
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int gi = A.gchunk_row[blockIdx.x];
  const int row = A.giant_row[gi];
  const int64_t eb = A.gchunk_edge[blockIdx.x];
  const int64_t e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
  const int n = (int)((e1 - eb) < GM_GIANT_CHUNK ? (e1 - eb) : GM_GIANT_CHUNK);
  const int64_t out0 = A.gterm_off[gi] + (eb - e0);  // multiple of 64
  V vprow;
  if constexpr (USE_VP) vprow = vp[row];
  int c[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    int k = threadIdx.x + j * kBlock;
    c[j] = (k < n) ? stream_load(&A.colidx[eb + k]) : -1;
  }
  if (xbits != nullptr) {
#pragma unroll
    for (int j = 0; j < PER; j++)
      if (c[j] >= 0 && !bit_get(xbits, c[j])) c[j] = -1;
  }
  T m[PER];
#pragma unroll
  for (int j = 0; j < PER; j++)
    if (c[j] >= 0) { if ((dbg & DBG_SKIP_GATHER) || GM_ABL_COLD(c[j], A)) memset(&m[j], 0, sizeof(T)); else m[j] = x[c[j]]; }
  (void)state;  // (the sub-pieces' ulp-maps are composed by k_giant_sums / k_giant_maps since round 6)
#pragma unroll
  for (int j = 0; j < PER; j++) {
    int k = threadIdx.x + j * kBlock;
    if (c[j] >= 0) {
      U t;
      p.P::process_message(m[j], edge_at<E>(A.vals, eb + k), vprow, t);
      terms[out0 + k] = t;
    }
    if (tpres != nullptr) {
      unsigned long long w = __ballot(c[j] >= 0);
      if ((threadIdx.x & 63) == 0 && (k & ~63) < n) tpres[(out0 + (k & ~63)) >> 6] = w;
    }
  }
}

__device__ __forceinline__ int lower_piece(const int32_t* __restrict__ gchunk_row, int n, int gi) {  // first piece of giant row gi
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (gchunk_row[mid] < gi) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// ---- the sub-pieces' maps of a pass (float sums over a dense x; see gchunk_state) ------------------------------------------------------
// one workgroup per 4096-product piece, a sub-piece = 32 consecutive threads x 16 consecutive products
// (1) the products of every sub-piece added up in double
__global__ void __launch_bounds__(kBlock)
k_giant_sums(gm_csr_t A, const float* __restrict__ terms, gchunk_state* __restrict__ state) {
  constexpr int PER = GM_GIANT_CHUNK / kBlock, kSubs = GM_GIANT_CHUNK / kGiantSub, kLanesPerSub = kGiantSub / PER;
  static_assert(kLanesPerSub == 32 && kSubs * kLanesPerSub == kBlock, "a sub-piece is half a wave's products");
  const int gi = A.gchunk_row[blockIdx.x];
  const int row = A.giant_row[gi];
  const int64_t eb = A.gchunk_edge[blockIdx.x], e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
  const int n = (int)((e1 - eb) < GM_GIANT_CHUNK ? (e1 - eb) : GM_GIANT_CHUNK);
  const float* __restrict__ t = terms + A.gterm_off[gi] + (eb - e0);
  const int k0 = threadIdx.x * PER;
  double acc = 0.0;
  if (k0 < n) {  // (16 consecutive products as four 16-byte loads: a row's slots are padded to multiples of 64, so the last ones may lie past n but not past the row)
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(t + k0);
    float4 q[PER / 4];
#pragma unroll
    for (int j = 0; j < PER / 4; j++) q[j] = t4[j];
#pragma unroll
    for (int j = 0; j < PER / 4; j++) {
      const float f[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (k0 + 4 * j + i < n) acc += (double)f[i];
    }
  }
#pragma unroll
  for (int off = kLanesPerSub / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & (kLanesPerSub - 1)) == 0) state[(size_t)blockIdx.x * kSubs + threadIdx.x / kLanesPerSub].sum = acc;
}
// (2) which binade is S expected to have?  the double prefix of the sums along the row (it differs from the float sum S by accumulated
// rounding only)
__device__ __forceinline__ int f32_binade_of(double d) {
  if (!(d > 0.0)) return 0;
  const float f = (float)d;
  const uint32_t b = __float_as_uint(f);
  const int e = (int)((b >> 23) & 0xff);
  return (e > 0 && e < 255) ? e : 0;
}
// (3) the map of every sub-piece whose S is expected to start and end in the same binade of a positive normal number.  The workgroup of a
// piece adds up the sums of the row's EARLIER sub-pieces itself (the hub row's last piece: 1664 doubles, seven loads per thread) instead of
// waiting for a scan kernel in between: a separate one-wave-per-row scan took 55 us for the hub row.
__global__ void __launch_bounds__(kBlock)
k_giant_maps(gm_csr_t A, const float* __restrict__ terms, gchunk_state* __restrict__ state, const float* __restrict__ y, const uint32_t* __restrict__ ybits,
             int accumulate) {
  constexpr int PER = GM_GIANT_CHUNK / kBlock, kSubs = GM_GIANT_CHUNK / kGiantSub, kLanesPerSub = kGiantSub / PER;
  const int gi = A.gchunk_row[blockIdx.x];
  const int row = A.giant_row[gi];
  const int64_t eb = A.gchunk_edge[blockIdx.x], e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
  const int n = (int)((e1 - eb) < GM_GIANT_CHUNK ? (e1 - eb) : GM_GIANT_CHUNK);
  const float* __restrict__ t = terms + A.gterm_off[gi] + (eb - e0);
  const int k0 = threadIdx.x * PER, lane = threadIdx.x & 63, sub = threadIdx.x / kLanesPerSub;
  gchunk_state* const rec = state + (size_t)blockIdx.x * kSubs + sub;
  // the expected S at this piece's start: what y holds (a pass that continues a fold) + the sums of the row's earlier sub-pieces
  __shared__ double s_part[kBlock / 64];
  {
    const int nbefore = (int)((eb - e0) / kGiantSub);  // (pieces of a row are consecutive: its first piece's records start nbefore records back)
    const gchunk_state* __restrict__ first = state + (size_t)blockIdx.x * kSubs - nbefore;
    double part = 0.0;
    for (int i = threadIdx.x; i < nbefore; i += kBlock) part += first[i].sum;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lane == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
  }
  double s_start = ((accumulate & ACC_READ_PREV) && bit_get(ybits, row)) ? (double)y[row] : 0.0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; w++) s_start += s_part[w];
  for (int u = 0; u < sub; u++) s_start += state[(size_t)blockIdx.x * kSubs + u].sum;
  const double s_end = s_start + rec->sum;
  const int ea = f32_binade_of(s_start), ee = f32_binade_of(s_end);
  const int e = (ea > 0 && ea == ee) ? ea : 0;  // (the same for the 32 threads of a sub-piece)
  ulp_map mine = {0u, 0u};
  bool ok = true;
  if (e > 0 && k0 < n) {
    const float4* __restrict__ t4 = reinterpret_cast<const float4*>(t + k0);
    float4 q[PER / 4];
#pragma unroll
    for (int j = 0; j < PER / 4; j++) q[j] = t4[j];
#pragma unroll
    for (int j = 0; j < PER / 4; j++) {
      const float f[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (k0 + 4 * j + i < n) {
          ulp_map m;
          if (!ulp_term(__float_as_uint(f[i]), e, m)) ok = false; else mine = ulp_compose(mine, m);
        }
      }
    }
  }
  ulp_map v = mine;
#pragma unroll
  for (int off = 1; off < kLanesPerSub; off <<= 1) {
    ulp_map o;
    o.de = __shfl_up(v.de, off, 64);
    o.dod = __shfl_up(v.dod, off, 64);
    if ((lane & (kLanesPerSub - 1)) >= off) v = ulp_compose(o, v);
  }
  const unsigned long long badm = __ballot(!ok);
  const bool bad = ((badm >> (lane & ~(kLanesPerSub - 1))) & 0xffffffffull) != 0ull;
  if ((lane & (kLanesPerSub - 1)) == kLanesPerSub - 1) {
    rec->e_map = (e > 0 && !bad && sub * kGiantSub < n) ? e : 0;
    rec->de = v.de;
    rec->dod = v.dod;
  }
}

// ------------------------------------------------------------------------------------
// giant rows, pass 2 for plain REDUCE_ORDERED programs (any reduce_function: the default for a program that declares no
// trait, i.e. every unchanged application): the products k_giant_terms left in the scratch stream are folded strictly
// in stored order, one wave per row.  What is left to the single wave is only the chain of reduce_function calls:
// the gathers and process_message calls were spread over the chip by pass 1 (one wave walking the 200 K edges of
// RMAT-22's hub row itself is bound by the gathers it can keep in flight: ~1.5 ms per pass), the products arrive as a
// dense, coalesced stream that is loaded one chunk ahead, staged in the wave's LDS strip, and lane 0 folds the chunk out
// of LDS -- the LDS reads pipeline under the dependent reduce calls, so an edge costs one reduce_function issue
// (~4-5 cycles) instead of a v_readlane plus the call in wave_row.
// the ordered fold of `deg` products that lie at terms[t0 ...] (t0 a multiple of 64 when presence words come with them) by one
// wave: lane 0 carries acc / has in and out, st is the wave's LDS strip of 512 products
template <class P, class U>
__device__ __forceinline__ void fold_products_ordered(const P& p, const U* __restrict__ terms, const unsigned long long* __restrict__ tpres,
                                                      const int64_t t0, const int64_t deg, const int lane, U* __restrict__ st, U& acc, bool& has) {
  constexpr int PER = 8, CH = PER * 64;
  U cur[PER], nxt[PER];
  unsigned long long pm[PER], pmn[PER];
  auto load = [&](int64_t base, U (&r)[PER], unsigned long long (&m)[PER]) {
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const int64_t k0 = base + (int64_t)j * 64;  // wave-uniform
      const int64_t left = deg - k0;
      unsigned long long mask = left >= 64 ? ~0ull : (left > 0 ? ((1ull << left) - 1ull) : 0ull);
      if (tpres != nullptr && mask) mask &= tpres[(t0 + k0) >> 6];
      m[j] = mask;
      if (k0 + lane < deg) r[j] = terms[t0 + k0 + lane];
    }
  };
  load(0, cur, pm);
  for (int64_t base = 0; base < deg; base += CH) {
#pragma unroll
    for (int j = 0; j < PER; j++) st[j * 64 + lane] = cur[j];
    __builtin_amdgcn_wave_barrier();
    if (base + CH < deg) load(base + CH, nxt, pmn);  // in flight during the fold below
    // A chunk whose 512 products are all present and whose row already carries a value (every chunk but a row's first and
    // last, with a dense x) takes a branch-free path: wave-uniform test, then lane 0 folds 16 sub-blocks of 32 out of two
    // register sets, the LDS reads of sub-block b + 1 issued before the 32 dependent reduce calls of sub-block b.  (With
    // the presence tests inside the loop the compiler sinks the reads into the branches and every sub-block waits for its
    // own reads: 15-23 cycles per edge on RMAT-22's 160 K-edge hub row.  Reading the products through a wave-uniform pointer
    // instead -- s_load_dwordx16 into scalar registers, no LDS at all, 32 registers a set -- was also built and measured:
    // 19 cycles per edge, scalar loads return out of order so every wait drains the prefetch as well.)
    bool dense_chunk = has;
#pragma unroll
    for (int j = 0; j < PER; j++) dense_chunk = dense_chunk && pm[j] == ~0ull;
    dense_chunk = __builtin_amdgcn_readfirstlane((int)dense_chunk) != 0;
    if (dense_chunk) {
      if (lane == 0) {
        constexpr int SB = 32, NSB = CH / SB;
        U ra[SB], rb[SB];
#pragma unroll
        for (int u = 0; u < SB; u++) ra[u] = st[u];
#pragma unroll
        for (int b = 0; b < NSB; b += 2) {
#pragma unroll
          for (int u = 0; u < SB; u++) rb[u] = st[(b + 1) * SB + u];
#pragma unroll
          for (int u = 0; u < SB; u++) p.P::reduce_function(acc, ra[u]);
          if (b + 2 < NSB) {
#pragma unroll
            for (int u = 0; u < SB; u++) ra[u] = st[(b + 2) * SB + u];
          }
#pragma unroll
          for (int u = 0; u < SB; u++) p.P::reduce_function(acc, rb[u]);
        }
      }
    } else if (lane == 0) {
#pragma unroll
      for (int j = 0; j < PER; j++) {
        unsigned long long mask = pm[j];
        const U* t = st + j * 64;
        while (mask) {  // (a row's first or last chunk, or a sparse x: by position)
          const int k = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          U v = t[k];
          if (has) p.P::reduce_function(acc, v); else { acc = v; has = true; }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < PER; j++) { cur[j] = nxt[j]; pm[j] = pmn[j]; }
  }
}

template <class P, class U, class V>
__global__ void __launch_bounds__(kBlock)
k_giant_fold_ordered(ProgArg<P> pa, gm_csr_t A, const V* __restrict__ vp, U* __restrict__ y, uint32_t* __restrict__ ybits, int accumulate,
                     const U* __restrict__ terms, const unsigned long long* __restrict__ tpres, const uint32_t* __restrict__ want,
                     const int32_t* __restrict__ only = nullptr /* per giant row: fold it (non-zero) or leave what is in y; null = every row */,
                     const int32_t* __restrict__ spec_off = nullptr /* non-zero: no speculation ran this pass, every row is folded here */) {
  static_assert(stageable<U>::value, "products of at most 8 bytes");
  __shared__ __attribute__((aligned(16))) U s_t[kBlock / 64][512];
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= A.ngiant) return;
  if (only != nullptr && only[w] == 0 && !(spec_off != nullptr && spec_off[0] != 0)) return;
  const int row = A.giant_row[w];
  if (!row_wanted(p, vp, want, row)) return;
  const int64_t deg = A.rowptr[row + 1] - A.rowptr[row];
  bool has = false;
  U acc;
  if (lane == 0 && (accumulate & ACC_READ_PREV) && bit_get(ybits, row)) { acc = y[row]; has = true; }
  fold_products_ordered<P, U>(p, terms, tpres, A.gterm_off[w], deg, lane, s_t[threadIdx.x >> 6], acc, has);
  if (lane == 0 && has) {
    y[row] = acc;
    if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
  }
}

// ------------------------------------------------------------------------------------
// multiply+reduce for one giant row per workgroup (REDUCE_COMMUTATIVE, REDUCE_LAST and
// REDUCE_F32_ADD; giant rows of plain REDUCE_ORDERED programs go to k_spmv_wave).
constexpr int kGiant = 512;                     // threads per workgroup of k_spmv_giant
constexpr int kLongPer = 16;                    // consecutive edges per lane and chunk
constexpr int kLongChunk = kLongPer * kGiant;   // 8192 edges per chunk


template <class P, class T, class U, class V, class E, bool USE_VP, int RK>
__global__ void __launch_bounds__(kGiant)
k_spmv_giant(ProgArg<P> pa, gm_csr_t A, const T* __restrict__ x, const uint32_t* __restrict__ xbits,
               const V* __restrict__ vp, U* __restrict__ y, uint32_t* __restrict__ ybits, int accumulate GM_DBG_PARAM,
               const U* __restrict__ terms, const unsigned long long* __restrict__ tpres, const uint32_t* __restrict__ want,
               gchunk_state* __restrict__ maps = nullptr /* per-piece maps (k_giant_terms) in, binade hints out; or null */,
               unsigned long long* __restrict__ bounds = nullptr /* REDUCE_F32_ADD, out: per piece that starts an 8192-product chunk, the running sum
                                                                    when the chunk starts (bits | has << 32): k_giant_verify_chunks */,
               const int32_t* __restrict__ spec_off = nullptr /* non-zero: an earlier pass of this run saw the speculation fail; do nothing */) {
  constexpr bool SMALL_U = sizeof(U) <= 8;
  constexpr int CH = kLongChunk, PER = kLongPer;
  // ordered kinds stage the per-edge products (U) of a chunk in LDS for the serial fold
  __shared__ __attribute__((aligned(16))) unsigned char s_term_raw[RK == REDUCE_F32_ADD ? CH * 4 : (RK == REDUCE_COMMUTATIVE && SMALL_U) ? kGiant * sizeof(U) : 16];
  __shared__ uint32_t s_pres[CH / 32];
  __shared__ int s_has[kGiant];
  __shared__ ulp_map s_wave[kGiant / 64];
  __shared__ int s_flag, s_next, s_fail;
  __shared__ uint32_t s_Sbits;

  const P& p = *reinterpret_cast<const P*>(pa.b);
  if (spec_off != nullptr && spec_off[0] != 0) return;
  const int row = A.giant_row[blockIdx.x];
  if (!row_wanted(p, vp, want, row)) return;
  const int64_t e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
  const int tid = threadIdx.x;
  V vprow;
  if constexpr (USE_VP) vprow = vp[row];

  if constexpr (RK == REDUCE_COMMUTATIVE) {
    // any order: strided private folds, then an LDS tree with the user's reduce_function
    static_assert(SMALL_U, "REDUCE_COMMUTATIVE is for small reduction types");
    U* s_res = reinterpret_cast<U*>(s_term_raw);
    bool has = false;
    U acc;
    for (int64_t k = e0 + tid; k < e1; k += kGiant) {
      int c = A.colidx[k];
      if (xbits != nullptr && !bit_get(xbits, c)) continue;
      T m = x[c];
      fold_one<P, T, U, V, E>(p, m, edge_at<E>(A.vals, k), vprow, acc, has);
    }
    if (tid == 0 && (accumulate & ACC_READ_PREV) && bit_get(ybits, row)) {
      U prev = y[row];
      if (has) { U t = acc; acc = prev; p.P::reduce_function(acc, t); } else { acc = prev; has = true; }
    }
    s_has[tid] = has;
    if (has) s_res[tid] = acc;
    __syncthreads();
    for (int s = kGiant / 2; s > 0; s >>= 1) {
      if (tid < s && s_has[tid + s]) {
        if (s_has[tid]) { U a = s_res[tid]; p.P::reduce_function(a, s_res[tid + s]); s_res[tid] = a; }
        else { s_res[tid] = s_res[tid + s]; s_has[tid] = 1; }
      }
      __syncthreads();
    }
    if (tid == 0 && s_has[0]) {
      y[row] = s_res[0];
      if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
    }
    return;
  } else if constexpr (RK == REDUCE_LAST) {
    // reduce is a=b: the answer is the last present edge of the row; scan backwards
    if (tid == 0) s_flag = -1;
    __syncthreads();
    for (int64_t hi = e1; hi > e0; hi -= CH) {
      int64_t lo = hi - CH < e0 ? e0 : hi - CH;
      int best = -1;
      for (int64_t k = lo + tid; k < hi; k += kGiant) {
        int c = A.colidx[k];
        if (xbits == nullptr || bit_get(xbits, c)) best = (int)(k - lo);
      }
      if (best >= 0) atomicMax(&s_flag, best);
      __syncthreads();
      int f = s_flag;
      if (f >= 0) {
        if (tid == 0) {
          int64_t k = lo + f;
          T m = message_of(p, x, vp, A.colidx[k]);
          U res;
          p.P::process_message(m, edge_at<E>(A.vals, k), vprow, res);
          y[row] = res;
          if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
        }
        return;
      }
    }
    return;  // no present message: with accumulate an earlier pass's value simply stays
  } else {
    // ---- REDUCE_F32_ADD: chunked exact replay; lane t owns PER consecutive edges ------------
    static_assert(RK == REDUCE_F32_ADD && sizeof(U) == 4 && SMALL_U, "k_spmv_giant: unsupported reduction kind");
    float* s_term = reinterpret_cast<float*>(s_term_raw);
    const int lane = tid & 63;
    if (tid == 0) {
      uint32_t sb = 0;
      bool h = (accumulate & ACC_READ_PREV) && bit_get(ybits, row);
      if (h) { U prev = y[row]; memcpy(&sb, &prev, 4); }
      s_Sbits = sb;
      s_has[0] = h;
    }
    // pass 2: stream this row's products (written by k_giant_terms) one chunk ahead of the fold.
    // Global loads are coalesced (lane t takes elements j*kGiant+t); the chunk is laid out
    // linearly in LDS, from which every lane then reads its PER consecutive products.
    const int k0 = tid * PER;
    const int64_t t0 = A.gterm_off[blockIdx.x];  // multiple of 64
    const int64_t deg = e1 - e0;
    auto load_chunk = [&](int64_t rel, float (&tt)[PER]) {
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const int64_t k = rel + j * kGiant + tid;
        tt[j] = 0.f;
        if (k < deg) { U u = terms[t0 + k]; memcpy(&tt[j], &u, 4); }
      }
    };
    float term[PER], pre[PER];
    // (the per-sub-piece ulp-maps of float sums over a dense x are walked by k_giant_replay_maps since round 6: this kernel is the replay
    // WITHOUT maps -- sparse x, a row filter, giant_maps = 0)
    (void)maps;
    int piece0 = 0;
    if (bounds != nullptr) piece0 = lower_piece(A.gchunk_row, A.ngchunk, (int)blockIdx.x);
    load_chunk(0, pre);
    for (int64_t base = e0; base < e1; base += CH) {
      __syncthreads();  // previous chunk fully consumed (and, first time round, s_maps / s_Sbits / s_has written)
      if (bounds != nullptr && tid == 0) {
        bounds[piece0 + 2 * (int)((base - e0) / CH)] = (unsigned long long)s_Sbits | ((unsigned long long)(s_has[0] != 0 ? 1 : 0) << 32);
      }
      const int n = (int)((e1 - base) < CH ? (e1 - base) : CH);
      const int ngroups = (n + PER - 1) / PER;
      const int64_t rel = base - e0;
#pragma unroll
      for (int j = 0; j < PER; j++) s_term[j * kGiant + tid] = pre[j];  // linear: slot(k) = k
      load_chunk(rel + CH, pre);
      uint32_t presmask = 0;
      if (k0 < n) {
        const int cnt = (n - k0) < PER ? (n - k0) : PER;
        presmask = cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u);
        if (tpres != nullptr) {
          const int64_t bit = t0 + rel + k0;  // multiple of 16
          presmask &= (uint32_t)((tpres[bit >> 6] >> (bit & 63)) & 0xffffu);
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < PER; j++) term[j] = s_term[k0 + j];
      if ((tid & 1) == 0) s_pres[tid >> 1] = 0;
      if (tid == 0) { s_next = 0; s_flag = 1; }  // s_flag: groups to fold serially after a stop
      __syncthreads();
      if (presmask) atomicOr(&s_pres[tid >> 1], presmask << ((tid & 1) * 16));
      __syncthreads();
      if (dbg & DBG_SKIP_FOLD) {
        if (dbg & DBG_FIRST_CHUNK_ONLY) break;
        continue;
      }

      while (true) {
        const int g0 = s_next;
        if (g0 >= ngroups) break;
        const bool has0 = s_has[0] != 0;
        const uint32_t sb = s_Sbits;
        const int span = s_flag;
        const int eS = (int)((sb >> 23) & 0xff);
        const bool s_ok = has0 && !(sb >> 31) && eS > 0 && eS < 255 && !(dbg & DBG_NO_REPLAY);
        int fail = g0;  // first group that cannot be taken in this binade
        if (s_ok) {
          ulp_map mine = {0u, 0u};
          bool ok = true;
          if (tid >= g0 && tid < ngroups) {
#pragma unroll
            for (int j = 0; j < PER; j++)
              if ((presmask >> j) & 1u) {
                ulp_map t;
                if (!ulp_term(__float_as_uint(term[j]), eS, t)) ok = false; else mine = ulp_compose(mine, t);
              }
          }
          // ordered inclusive scan over lanes, then over waves
          ulp_map v = mine;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            ulp_map o;
            o.de = __shfl_up(v.de, off, 64);
            o.dod = __shfl_up(v.dod, off, 64);
            if (lane >= off) v = ulp_compose(o, v);
          }
          if (lane == 63) s_wave[tid >> 6] = v;
          if (tid == 0) s_fail = ngroups;
          __syncthreads();
          ulp_map pre = {0u, 0u};
          for (int w = 0; w < (tid >> 6); w++) pre = ulp_compose(pre, s_wave[w]);
          v = ulp_compose(pre, v);  // maps of groups g0..tid composed
          const uint32_t Sint = (sb & 0x7fffffu) | 0x800000u;  // S in ulps, in [2^23, 2^24)
          const uint32_t add = (Sint & 1u) ? v.dod : v.de;
          const uint32_t Safter = Sint + add;  // add saturates at 2^24: no wrap
          const bool bad = !ok || Safter >= 0x1000000u;
          if (tid >= g0 && tid < ngroups && bad) atomicMin(&s_fail, tid);
          __syncthreads();
          fail = s_fail;
          if (fail > g0 && tid == fail - 1) {  // accept groups [g0, fail)
            s_Sbits = (sb & 0xff800000u) | (Safter & 0x7fffffu);
            atomicAdd(&g_longrow_counters[0], (unsigned long long)(fail - g0));
          }
        }
        __syncthreads();
        if (tid < 64) {
          // wave 0 folds `span` groups from `fail` on, in order, out of registers; the span
          // doubles while the replay makes no progress and resets once it does
          int nspan = (fail - g0 >= 8) ? 1 : span;
          if (!has0) nspan = nspan < 64 ? 64 : nspan;  // row start: S grows fastest here
          const int gend = fail + nspan < ngroups ? fail + nspan : ngroups;
          if (fail < ngroups) {
            uint32_t sb2 = s_Sbits;
            bool h = s_has[0] != 0;
            U S;
            memcpy(&S, &sb2, 4);
            const int kend = gend * PER < n ? gend * PER : n;
            for (int kb = fail * PER; kb < kend; kb += 64) {
              const int k = kb + lane;
              bool pr = false;
              U a;
              if (k < kend && ((s_pres[k >> 5] >> (k & 31)) & 1u)) {
                pr = true;
                memcpy(&a, &s_term[k], 4);
              }
              unsigned long long mask = __ballot(pr);
              if (!h && mask) {
                const int i = __ffsll((long long)mask) - 1;
                S = wave_bcast(a, i);
                h = true;
                mask &= mask - 1;
              }
              while (mask) {
                const int i = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                U t = wave_bcast(a, i);
                p.P::reduce_function(S, t);
              }
            }
            if (lane == 0) {
              memcpy(&sb2, &S, 4);
              s_Sbits = sb2;
              s_has[0] = h;
              s_next = gend;
              s_flag = (fail - g0 >= 8) ? 1 : (nspan * 2 > 256 ? 256 : nspan * 2);
              atomicAdd(&g_longrow_counters[1], (unsigned long long)(gend - fail));
            }
          } else if (lane == 0) {
            s_next = ngroups;
          }
        }
        __syncthreads();
      }
      if (dbg & DBG_FIRST_CHUNK_ONLY) break;
    }
    __syncthreads();
    if (tid == 0 && s_has[0]) {
      uint32_t sb = s_Sbits;
      U r;
      memcpy(&r, &sb, 4);
      y[row] = r;
      if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
    }
  }
}

// ------------------------------------------------------------------------------------
// The exact replay of a giant row's float sum from its products stream and its sub-pieces' ulp-maps (gchunk_state; REDUCE_F32_ADD,
// every x entry present): ONE WAVE per row, no LDS, no barrier.
//   * walk: the records of the next 64 sub-pieces, one per lane; a record applies when it was composed for the binade S is in and
//     S stays in it (S + delta < 2^24 ulps; S only grows: a map exists only for non-negative terms).  Records that do not apply
//     count as saturating maps, an ordered wave scan composes them, and the first lane whose composed delta leaves the binade ends
//     the accepted run -- up to 64 x 512 products per step;
//   * replay of the sub-piece where the run stopped (a binade crossing, a sub-piece without a map, the row's start): lane l holds 8
//     consecutive products; the same ulp-map scan over the lanes accepts the longest prefix that stays in S's binade, the lane where
//     it stops is folded serially with the program's own reduce_function (that is where S changes binade), and the scan resumes.
// Every accepted step adds exactly what the serial loop would have added (kernels.hpp: ulp_term / ulp_compose): the bits are the serial
// loop's.  bounds (speculating runs): the running sum at the start of every 8192-product chunk, for k_giant_verify_chunks.
template <class P, class U>
__global__ void __launch_bounds__(64)
k_giant_replay_maps(ProgArg<P> pa, gm_csr_t A, U* __restrict__ y, uint32_t* __restrict__ ybits, int accumulate, const U* __restrict__ terms,
                    const gchunk_state* __restrict__ maps, unsigned long long* __restrict__ bounds, const int32_t* __restrict__ spec_off) {
  static_assert(std::is_same<U, float>::value, "k_giant_replay_maps: float sums");
  constexpr int PER = kGiantSub / 64;                    // products per lane of a replayed sub-piece
  constexpr int kSubs = GM_GIANT_CHUNK / kGiantSub;      // records per piece
  constexpr int kSubsPerChunk = 8192 / kGiantSub;        // (bounds are kept per 8192-product chunk = two pieces)
  const P& p = *reinterpret_cast<const P*>(pa.b);
  if (spec_off != nullptr && spec_off[0] != 0) return;
  const int row = A.giant_row[blockIdx.x];
  const int lane = threadIdx.x;
  const int64_t deg = A.rowptr[row + 1] - A.rowptr[row];
  const int64_t t0 = A.gterm_off[blockIdx.x];
  const int piece0 = lower_piece(A.gchunk_row, A.ngchunk, (int)blockIdx.x);
  const gchunk_state* const rec0 = maps + (size_t)piece0 * kSubs;
  const int nsub = (int)((deg + kGiantSub - 1) / kGiantSub);
  bool has = (accumulate & ACC_READ_PREV) && bit_get(ybits, row);
  uint32_t sb = 0;  // bits of S (wave-uniform)
  if (has) sb = __float_as_uint(y[row]);
  unsigned long long n_skip = 0, n_par = 0, n_ser = 0;
  int sub = 0, pre_sub = -1, rbase = -(1 << 30);
  float pre[kGiantSub / 64];
  int rc_e = 0;  // record rbase + lane of the window: its binade and map
  uint32_t rc_de = 0u, rc_dod = 0u;
  while (sub < nsub) {
    if (has) {
      // ---- walk: which of the next 64 records apply, one after the other, to the S we have?
      const int j = sub + lane;
      const int eS = (int)((sb >> 23) & 0xff);
      // (the records are read 64 at a time and kept: this pass never changes a record's map, and after every replayed sub-piece the
      // walk resumes inside the window it already holds -- no load, and no scan either when the very next record does not apply)
      if (sub < rbase || sub >= rbase + 64) {
        rbase = sub;
        rc_e = 0; rc_de = 0u; rc_dod = 0u;
        if (rbase + lane < nsub) { const gchunk_state* r = rec0 + rbase + lane; rc_e = r->e_map; rc_de = r->de; rc_dod = r->dod; }
      }
      const int src = lane + (sub - rbase);  // the lane of the window that holds record j
      const int e_j = __shfl(rc_e, src & 63, 64);
      const uint32_t de_j = (uint32_t)__shfl((int)rc_de, src & 63, 64), dod_j = (uint32_t)__shfl((int)rc_dod, src & 63, 64);
      const bool applies = src < 64 && j < nsub && !(sb >> 31) && eS > 0 && eS < 255 && e_j == eS;
      if (!(__ballot(applies) & 1ull)) goto replay;  // (the next record is for another binade, or there is none: replay the sub-piece)
      ulp_map m = {kUlpSat, kUlpSat};
      if (applies) { m.de = de_j; m.dod = dod_j; }
      ulp_map v = m;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        ulp_map o;
        o.de = __shfl_up(v.de, off, 64);
        o.dod = __shfl_up(v.dod, off, 64);
        if (lane >= off) v = ulp_compose(o, v);
      }
      const uint32_t Sint = (sb & 0x7fffffu) | 0x800000u;
      const uint32_t Safter = Sint + ((Sint & 1u) ? v.dod : v.de);  // (deltas saturate at 2^24: no wrap)
      const unsigned long long stop = __ballot(!(Safter < 0x1000000u));
      const int cnt = stop ? (int)__ffsll((long long)stop) - 1 : 64;
      if (cnt > 0) {
        if (bounds != nullptr) {  // (wave-uniform) the running sum at the start of every 8192-product chunk inside the accepted run
          uint32_t pde = (uint32_t)__shfl_up(v.de, 1, 64), pdod = (uint32_t)__shfl_up(v.dod, 1, 64);
          if (lane == 0) { pde = 0u; pdod = 0u; }
          const uint32_t Sbefore = Sint + ((Sint & 1u) ? pdod : pde);
          if (lane < cnt && (j % kSubsPerChunk) == 0) bounds[piece0 + j / kSubs] = (unsigned long long)((sb & 0xff800000u) | (Sbefore & 0x7fffffu)) | (1ull << 32);
        }
        const uint32_t Slast = (uint32_t)__shfl((int)Safter, cnt - 1, 64);
        sb = (sb & 0xff800000u) | (Slast & 0x7fffffu);  // (same binade: the hints of the skipped sub-pieces stay what they are)
        sub += cnt;
        n_skip += (unsigned long long)cnt;
        continue;
      }
    }
  replay:
    // ---- replay sub-piece `sub`
    if (bounds != nullptr && lane == 0 && (sub % kSubsPerChunk) == 0) bounds[piece0 + sub / kSubs] = (unsigned long long)sb | ((unsigned long long)(has ? 1 : 0) << 32);
    const int64_t base = (int64_t)sub * kGiantSub;
    const int n = (int)((deg - base) < kGiantSub ? (deg - base) : kGiantSub);
    const int ngroups = (n + PER - 1) / PER;
    const int k0 = lane * PER;
    float term[PER];
    if (pre_sub == sub) {
#pragma unroll
      for (int i = 0; i < PER; i++) term[i] = pre[i];
    } else {
#pragma unroll
      for (int i = 0; i < PER; i++) term[i] = (k0 + i < n) ? terms[t0 + base + k0 + i] : 0.f;
    }
    if (sub + 1 < nsub) {  // (the next sub-piece's products are requested now: replays come in runs -- a row's start, a pass without hints)
      const int64_t nb = base + kGiantSub;
#pragma unroll
      for (int i = 0; i < PER; i++) pre[i] = (nb + k0 + i < deg) ? terms[t0 + nb + k0 + i] : 0.f;
      pre_sub = sub + 1;
    }
    int g0 = 0, span = 1;
    while (g0 < ngroups) {
      const int eS = (int)((sb >> 23) & 0xff);
      const bool s_ok = has && !(sb >> 31) && eS > 0 && eS < 255;
      int fail = g0;
      if (s_ok) {
        ulp_map mine = {0u, 0u};
        bool ok = true;
        if (lane >= g0 && lane < ngroups) {
#pragma unroll
          for (int i = 0; i < PER; i++)
            if (k0 + i < n) {
              ulp_map t;
              if (!ulp_term(__float_as_uint(term[i]), eS, t)) ok = false; else mine = ulp_compose(mine, t);
            }
        }
        ulp_map v = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          ulp_map o;
          o.de = __shfl_up(v.de, off, 64);
          o.dod = __shfl_up(v.dod, off, 64);
          if (lane >= off) v = ulp_compose(o, v);
        }
        const uint32_t Sint = (sb & 0x7fffffu) | 0x800000u;
        const uint32_t Safter = Sint + ((Sint & 1u) ? v.dod : v.de);
        // (a lane that cannot be taken poisons every later prefix: take the first such lane)
        const unsigned long long badm = __ballot(lane >= g0 && lane < ngroups && (!ok || Safter >= 0x1000000u));
        fail = badm ? (int)__ffsll((long long)badm) - 1 : ngroups;
        if (fail > g0) {
          const uint32_t Slast = (uint32_t)__shfl((int)Safter, fail - 1, 64);
          sb = (sb & 0xff800000u) | (Slast & 0x7fffffu);
          n_par += (unsigned long long)(fail - g0);
        }
      }
      if (fail < ngroups) {
        // `span` lanes from `fail` on are folded in order with the program's own function; the span doubles while the scan makes no
        // progress (a row's start, terms as large as S, negative terms) and goes back to one lane when it does
        int nspan = (fail > g0) ? 1 : span;
        if (!has && nspan < 4) nspan = 4;
        const int gend = fail + nspan < ngroups ? fail + nspan : ngroups;
        U S = __uint_as_float(sb);
        for (int l = fail; l < gend; l++) {
#pragma unroll
          for (int i = 0; i < PER; i++) {
            const U t = wave_bcast(term[i], l);
            if (l * PER + i < n) {
              if (!has) { S = t; has = true; } else p.P::reduce_function(S, t);
            }
          }
        }
        sb = __float_as_uint(S);
        n_ser += (unsigned long long)(gend - fail);
        span = (fail > g0) ? 1 : (nspan * 2 > 64 ? 64 : nspan * 2);
        g0 = gend;
      } else {
        g0 = ngroups;
      }
    }
    sub++;
  }
  if (lane == 0) {
    // (counters in 16-product groups, as the workgroup replay counts them)
    if (n_par) atomicAdd(&g_longrow_counters[0], n_par * PER / 16);
    if (n_ser) atomicAdd(&g_longrow_counters[1], (n_ser * PER + 15) / 16);
    if (n_skip) atomicAdd(&g_longrow_counters[2], n_skip * (kGiantSub / 16));
    if (has) {
      y[row] = __uint_as_float(sb);
      if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
    }
  }
}

// ------------------------------------------------------------------------------------
// Giant rows of a program that DECLARES nothing (REDUCE_ORDERED: the reference's `c = a; reduce(c, b)` in stored order,
// SPMV.h:54-59) whose reduce_function nevertheless answers like a float addition when the host asks it (engine.hpp:
// probe_reduce_guess).  The serial chain of such a row is the longest thing in the iteration (RMAT-26's hub: 854 K
// dependent calls), and an answer to a finite set of questions is no proof, so the guess is only used to SPECULATE:
// k_spmv_giant<REDUCE_F32_ADD> replays the row as float additions and leaves the running sum at every 8192-product chunk
// boundary; here one wave per chunk -- all chunks of all rows at once -- folds its chunk strictly in order with the PROGRAM'S
// OWN reduce_function, starting from the value the replay had at the chunk's start, and compares the bits with what the
// replay had at its end (the next boundary, or the row's result).  When every chunk of a row agrees, the row's result IS
// the ordered fold's, by induction over its chunks, whatever the function is; a row with a disagreeing chunk is flagged and
// folded again by k_giant_fold_ordered.  The serial part of a row shrinks from its length to one chunk.
template <class P, class U>
__global__ void __launch_bounds__(kBlock)
k_giant_verify_chunks(ProgArg<P> pa, gm_csr_t A, const U* __restrict__ terms, const unsigned long long* __restrict__ bounds, const U* __restrict__ y,
                      int32_t* __restrict__ redo /* per giant row, zeroed before: set when a chunk of the row disagrees */,
                      int32_t* __restrict__ spec_off /* set as well then: the following passes of this run do not speculate */) {
  static_assert(sizeof(U) == 4, "float sums");
  __shared__ __attribute__((aligned(16))) U s_t[kBlock / 64][512];
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= A.ngchunk || spec_off[0] != 0) return;
  const int gi = A.gchunk_row[q];
  const int piece0 = lower_piece(A.gchunk_row, A.ngchunk, gi);
  if ((q - piece0) & 1) return;  // (a chunk of the replay = two pieces)
  const int row = A.giant_row[gi];
  const int64_t deg = A.rowptr[row + 1] - A.rowptr[row];
  const int64_t rel = (int64_t)(q - piece0) * GM_GIANT_CHUNK;
  const int64_t n = deg - rel < (int64_t)kLongChunk ? deg - rel : (int64_t)kLongChunk;
  const unsigned long long b0 = bounds[q];
  const uint32_t bits0 = (uint32_t)b0;
  U acc;
  memcpy(&acc, &bits0, 4);
  bool has = ((b0 >> 32) & 1ull) != 0;
  fold_products_ordered<P, U>(p, terms, nullptr, A.gterm_off[gi] + rel, n, lane, s_t[threadIdx.x >> 6], acc, has);
  if (lane == 0) {
    uint32_t got, want_bits;
    memcpy(&got, &acc, 4);
    bool ok = has;
    if (rel + (int64_t)kLongChunk < deg) {
      const unsigned long long b1 = bounds[q + 2];
      want_bits = (uint32_t)b1;
      ok = ok && ((b1 >> 32) & 1ull) != 0;
    } else {
      const U fin = y[row];
      memcpy(&want_bits, &fin, 4);
    }
    if (!ok || got != want_bits) { redo[gi] = 1; spec_off[0] = 1; }
  }
}

// The same idea for ANY reduce_function of a program that declares nothing, without asking it anything: speculate that the function is
// ASSOCIATIVE on the operands at hand (min, max, integer sums, a = b ... are; a float sum is not, its rows end up folded again).
// k_giant_chunk_totals folds every 8192-product chunk by itself (all chunks of all rows at once, present messages only, in order);
// k_giant_chunk_scan combines a row's chunk totals in order -- a few dozen calls -- which gives a candidate for the running value at
// every chunk boundary and for the row's result; k_giant_verify_chunks_any then folds every chunk once more, in order, STARTING from the
// candidate at its start, and compares with the candidate at its end.  Agreement of all chunks of a row proves, by induction, that
// the candidates are the running values of the one ordered fold (SPMV.h:54-59) -- for this function on these operands, associative
// in general or not.  The serial part of a row: two chunks and the scan instead of its length.
template <class P, class U, class V>
__global__ void __launch_bounds__(kBlock)
k_giant_chunk_totals(ProgArg<P> pa, gm_csr_t A, const V* __restrict__ vp, const uint32_t* __restrict__ want, const U* __restrict__ terms,
                     const unsigned long long* __restrict__ tpres, U* __restrict__ tval, int32_t* __restrict__ thas, const int32_t* __restrict__ spec_off) {
  static_assert(stageable<U>::value, "products of at most 8 bytes");
  __shared__ __attribute__((aligned(16))) U s_t[kBlock / 64][512];
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= A.ngchunk || spec_off[0] != 0) return;
  const int gi = A.gchunk_row[q];
  const int piece0 = lower_piece(A.gchunk_row, A.ngchunk, gi);
  if ((q - piece0) & 1) return;
  const int row = A.giant_row[gi];
  if (!row_wanted(p, vp, want, row)) return;
  const int64_t deg = A.rowptr[row + 1] - A.rowptr[row];
  const int64_t rel = (int64_t)(q - piece0) * GM_GIANT_CHUNK;
  const int64_t n = deg - rel < (int64_t)kLongChunk ? deg - rel : (int64_t)kLongChunk;
  U acc;
  bool has = false;
  fold_products_ordered<P, U>(p, terms, tpres, A.gterm_off[gi] + rel, n, lane, s_t[threadIdx.x >> 6], acc, has);
  if (lane == 0) {
    thas[q] = has ? 1 : 0;
    if (has) tval[q] = acc;
  }
}

// one wave per giant row: the chunk totals combined in order; bval / bhas[first piece of chunk c] = the candidate running value when
// chunk c starts, fin / finhas[row] = the candidate result, which is also stored as the row's result
template <class P, class U, class V>
__global__ void __launch_bounds__(kBlock)
k_giant_chunk_scan(ProgArg<P> pa, gm_csr_t A, const V* __restrict__ vp, const uint32_t* __restrict__ want, U* __restrict__ y, uint32_t* __restrict__ ybits,
                   int accumulate, const U* __restrict__ tval, const int32_t* __restrict__ thas, U* __restrict__ bval, int32_t* __restrict__ bhas,
                   U* __restrict__ fin, int32_t* __restrict__ finhas, const int32_t* __restrict__ spec_off) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int gi = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (gi >= A.ngiant || spec_off[0] != 0) return;
  const int row = A.giant_row[gi];
  if (!row_wanted(p, vp, want, row)) return;
  const int64_t deg = A.rowptr[row + 1] - A.rowptr[row];
  const int nchunks = (int)((deg + kLongChunk - 1) / kLongChunk);
  const int piece0 = lower_piece(A.gchunk_row, A.ngchunk, gi);
  U B;
  bool h = false;  // (wave-uniform)
  for (int c0 = 0; c0 < nchunks; c0 += 64) {
    const int c = c0 + lane;
    U tv;
    int th = 0;
    if (c < nchunks) {
      th = thas[piece0 + 2 * c];
      if (th) tv = tval[piece0 + 2 * c];
    }
    const int m = nchunks - c0 < 64 ? nchunks - c0 : 64;
    for (int i = 0; i < m; i++) {
      if (lane == 0) {
        bhas[piece0 + 2 * (c0 + i)] = h ? 1 : 0;
        if (h) bval[piece0 + 2 * (c0 + i)] = B;
      }
      if (__builtin_amdgcn_readlane(th, i)) {
        U t = wave_bcast(tv, i);
        if (h) p.P::reduce_function(B, t); else { B = t; h = true; }
      }
    }
  }
  if (lane == 0) {
    finhas[gi] = h ? 1 : 0;
    if (h) {
      fin[gi] = B;
      y[row] = B;
      if (!(accumulate & ACC_STATIC_BITS)) atomicOr(&ybits[row >> 5], 1u << (row & 31));
    }
  }
}

template <class P, class U, class V>
__global__ void __launch_bounds__(kBlock)
k_giant_verify_chunks_any(ProgArg<P> pa, gm_csr_t A, const V* __restrict__ vp, const uint32_t* __restrict__ want, const U* __restrict__ terms,
                          const unsigned long long* __restrict__ tpres, const U* __restrict__ bval, const int32_t* __restrict__ bhas,
                          const U* __restrict__ fin, const int32_t* __restrict__ finhas, int32_t* __restrict__ redo, int32_t* __restrict__ spec_off) {
  static_assert(stageable<U>::value, "products of at most 8 bytes");
  __shared__ __attribute__((aligned(16))) U s_t[kBlock / 64][512];
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int q = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= A.ngchunk || spec_off[0] != 0) return;
  const int gi = A.gchunk_row[q];
  const int piece0 = lower_piece(A.gchunk_row, A.ngchunk, gi);
  if ((q - piece0) & 1) return;
  const int row = A.giant_row[gi];
  if (!row_wanted(p, vp, want, row)) return;
  const int64_t deg = A.rowptr[row + 1] - A.rowptr[row];
  const int64_t rel = (int64_t)(q - piece0) * GM_GIANT_CHUNK;
  const int64_t n = deg - rel < (int64_t)kLongChunk ? deg - rel : (int64_t)kLongChunk;
  bool has = bhas[q] != 0;
  U acc;
  if (has) acc = bval[q];
  fold_products_ordered<P, U>(p, terms, tpres, A.gterm_off[gi] + rel, n, lane, s_t[threadIdx.x >> 6], acc, has);
  if (lane == 0) {
    const bool last = rel + (int64_t)kLongChunk >= deg;
    const bool ehas = (last ? finhas[gi] : bhas[q + 2]) != 0;
    bool ok = has == ehas;
    if (ok && has) {
      const U e = last ? fin[gi] : bval[q + 2];
      const unsigned char *a_ = reinterpret_cast<const unsigned char*>(&acc), *b_ = reinterpret_cast<const unsigned char*>(&e);
      for (size_t i = 0; i < sizeof(U); i++) ok = ok && a_[i] == b_[i];
    }
    if (!ok) { redo[gi] = 1; spec_off[0] = 1; }
  }
}

// ------------------------------------------------------------------------------------
// Top-down ("push") step for REDUCE_LAST programs with a tiny active set (first and last BFS
// levels): instead of every row scanning its in-edges for a present message, every active
// source walks its out-edges (the GM_DIR_IN adjacency: rows = sources) and bids for each
// destination with atomicMax on (native id of the source + 1) << 32 | edge position.  The
// maximum is exactly the message the ordered pull would have kept -- the present in-neighbour
// with the largest native id, latest duplicate edge -- so results are identical.  A resolve
// pass then evaluates process_message once per reached destination.

// frontier statistics: number of active vertices, of their out-edges, and the largest out-degree.
// Grid-stride with one set of atomics per workgroup (launch <= 2048 workgroups).
__global__ void __launch_bounds__(kBlock)
k_frontier_stats(const uint32_t* __restrict__ active, const int64_t* __restrict__ src_rowptr, int n,
                 unsigned long long* __restrict__ stats /* [0] vertices, [1] out-edges, [2] max out-degree */,
                 int32_t* __restrict__ list /* also list the active vertices while they are few, or null */,
                 unsigned int* __restrict__ count) {
  __shared__ unsigned long long s_c[kBlock / 64], s_e[kBlock / 64], s_m[kBlock / 64];
  __shared__ int32_t s_lbuf[kListBuf];
  __shared__ unsigned int s_lfill[4];
  BlockList blist{s_lbuf, s_lfill};
  if (list != nullptr) blist.init();
  unsigned long long cnt = 0, edges = 0, mx = 0;
  for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
    const int64_t i = base + threadIdx.x;
    const bool act = i < n && bit_get(active, (int)i);
    if (act) {
      unsigned long long d = src_rowptr ? (unsigned long long)(src_rowptr[i + 1] - src_rowptr[i]) : 0ull;
      cnt++;
      edges += d;
      mx = d > mx ? d : mx;
    }
    if (list != nullptr) blist.add(act, (int)i, list, count, (unsigned int)kSparseListCap);
  }
  if (list != nullptr) blist.finish(list, count, (unsigned int)kSparseListCap);
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_down(cnt, off, 64);
    edges += __shfl_down(edges, off, 64);
    unsigned long long o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_c[wv] = cnt; s_e[wv] = edges; s_m[wv] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; w++) { cnt += s_c[w]; edges += s_e[w]; mx = s_m[w] > mx ? s_m[w] : mx; }
    if (cnt) {
      atomicAdd(&stats[0], cnt);
      atomicAdd(&stats[1], edges);
      atomicMax(&stats[2], mx);
    }
  }
}

// compact list of the active vertices (order irrelevant)
__global__ void __launch_bounds__(kBlock)
k_frontier_list(const uint32_t* __restrict__ active, int n, int32_t* __restrict__ list, unsigned int* __restrict__ count) {
  __shared__ int32_t s_lbuf[kListBuf];
  __shared__ unsigned int s_lfill[4];
  BlockList blist{s_lbuf, s_lfill};
  blist.init();
  for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
    const int i = (int)base + threadIdx.x;
    blist.add(i < n && bit_get(active, i), i, list, count, 0u);
  }
  blist.finish(list, count, 0u);
}

// Work decomposition of the kernels that walk the out-edges of a listed active set: source i owns
// ceil(deg_i / kPieceEdges) pieces; off[i] = pieces before source i (off[nlist] = total).
// The edge kernels are launched with
// an upper bound of the total (out-edges / kPieceEdges + sources) and find their source by
// bisection -- a grid of sources x max-pieces would be almost entirely empty workgroups as soon as
// one hub is active (2.7 K sources x 837 pieces = 2.2 M launches for 18 K useful ones).
constexpr int kPieceEdges = kBlock * 4;
// exclusive scan of the piece counts in three small launches: per-workgroup totals, a one-workgroup
// scan of those (at most kSparseListCap / kBlock of them), per-workgroup scan plus its base
__device__ __forceinline__ unsigned int pieces_of(const gm_csr_t& S, int u) {
  return (unsigned int)((S.rowptr[u + 1] - S.rowptr[u] + kPieceEdges - 1) / kPieceEdges);
}
// inclusive scan of one value per thread over the workgroup; returns the inclusive value, *total = workgroup sum
__device__ __forceinline__ unsigned int block_scan_inclusive(unsigned int v, unsigned int* s_w /* kBlock/64 */, unsigned int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned int inc = v;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int o = (unsigned int)__shfl_up((int)inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();  // s_w may still be read from a previous call
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  unsigned int before = 0, all = 0;
  for (int w = 0; w < kBlock / 64; w++) { if (w < wv) before += s_w[w]; all += s_w[w]; }
  *total = all;
  return before + inc;
}
__global__ void __launch_bounds__(kBlock)
k_piece_count(gm_csr_t S, const int32_t* __restrict__ list, int nlist, unsigned int* __restrict__ block_sum) {
  __shared__ unsigned int s_w[kBlock / 64];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned int total;
  (void)block_scan_inclusive(i < nlist ? pieces_of(S, list[i]) : 0u, s_w, &total);
  if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kBlock)
k_piece_block_scan(unsigned int* __restrict__ block_sum, int nblocks) {  // in place: exclusive; [nblocks] = grand total
  __shared__ unsigned int s_w[kBlock / 64];
  unsigned int carry = 0;
  for (int base = 0; base < nblocks; base += kBlock) {
    const int i = base + threadIdx.x;
    const unsigned int v = i < nblocks ? block_sum[i] : 0u;
    unsigned int total;
    const unsigned int inc = block_scan_inclusive(v, s_w, &total);
    if (i < nblocks) block_sum[i] = carry + inc - v;
    carry += total;
  }
  if (threadIdx.x == 0) block_sum[nblocks] = carry;
}
__global__ void __launch_bounds__(kBlock)
k_piece_offsets(gm_csr_t S, const int32_t* __restrict__ list, int nlist, const unsigned int* __restrict__ block_base,
                unsigned int* __restrict__ off) {
  __shared__ unsigned int s_w[kBlock / 64];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const unsigned int v = i < nlist ? pieces_of(S, list[i]) : 0u;
  unsigned int total;
  const unsigned int inc = block_scan_inclusive(v, s_w, &total);
  if (i < nlist) off[i] = block_base[blockIdx.x] + inc - v;
  if (i == nlist - 1) off[nlist] = block_base[blockIdx.x] + inc;
}
// (source index, first edge, end edge) of workgroup b; false when b is past the last piece
__device__ __forceinline__ bool piece_of_block(const gm_csr_t& S, const int32_t* __restrict__ list, int nlist,
                                               const unsigned int* __restrict__ off, unsigned int b, int* u, int64_t* e0, int64_t* e1) {
  if (b >= off[nlist]) return false;
  int lo = 0, hi = nlist - 1;  // largest i with off[i] <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= b) lo = mid; else hi = mid - 1;
  }
  *u = list[lo];
  *e0 = S.rowptr[*u] + (int64_t)(b - off[lo]) * kPieceEdges;
  *e1 = S.rowptr[*u + 1];
  return *e0 < *e1;
}

// messages of the listed (active) vertices only
template <class P, class T, class V>
__global__ void __launch_bounds__(kBlock)
k_send_list(ProgArg<P> pa, const V* __restrict__ vp, const int32_t* __restrict__ list, int nlist, T* __restrict__ x, int row_base) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nlist) return;
  const int u = list[i];
  T m;
  p.P::send_message(vp[u], m);
  x[(size_t)row_base + u] = m;
}

// bids: one workgroup per 1024-edge piece of an active source's out-edges.  The first bid
// a destination receives also puts it on the `touched` list (its slot of `best` was 0 before).
__global__ void __launch_bounds__(kBlock)
k_push_bid(gm_csr_t S /* rows = sources */, const int32_t* __restrict__ list, int nlist,
           const unsigned int* __restrict__ off, const int32_t* __restrict__ native_of_dev,
           unsigned long long* __restrict__ best,
           const uint32_t* __restrict__ want /* row-filter bits of the destinations, or null */,
           int32_t* __restrict__ touched, unsigned int* __restrict__ tcount) {
  int u;
  int64_t e0, e1;
  if (!piece_of_block(S, list, nlist, off, blockIdx.x, &u, &e0, &e1)) return;
  const unsigned long long hi = (unsigned long long)((native_of_dev ? native_of_dev[u] : u) + 1) << 32;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int64_t e = e0 + threadIdx.x + j * kBlock;
    bool first = false;
    int c = 0;
    if (e < e1) {
      c = S.colidx[e];
      if (want == nullptr || ((want[c >> 5] >> (c & 31)) & 1u)) {  // else: destination ignores messages anyway
        // bids only grow: a plain read that already shows a larger bid makes the atomic pointless (a
        // stale, smaller value merely costs the atomic that decides anyway).  Hubs' neighbourhoods
        // overlap heavily, so most bids are dropped here instead of serialising in the L2.
        const unsigned long long key = hi | (unsigned long long)(uint32_t)e;
        if (best[c] < key) first = atomicMax(&best[c], key) == 0ull;
      }
    }
    const unsigned long long fm = touched != nullptr ? __ballot(first) : 0ull;
    if (fm) {
      unsigned int start = 0;
      if (lane == 0) start = atomicAdd(tcount, (unsigned int)__popcll(fm));
      start = (unsigned int)__shfl((int)start, 0, 64);
      if (first) touched[start + (unsigned int)__popcll(fm & ((1ull << lane) - 1ull))] = c;
    }
  }
}

// The same bids for an active set that is too large to list (more than kSparseListCap vertices) but owns few
// out-edges, each vertex only a handful (the late levels of a traversal: a million vertices of degree ~1): the
// active BITMAP is scanned, one lane per vertex, and every active lane walks its own short out-edge list -- no
// list, no piece offsets.  (A bottom-up level would scan every unvisited row for these few messages.)
__global__ void __launch_bounds__(kBlock)
k_push_bid_bits(gm_csr_t S /* rows = sources */, const uint32_t* __restrict__ active, int n,
                const int32_t* __restrict__ native_of_dev, unsigned long long* __restrict__ best,
                const uint32_t* __restrict__ want, int32_t* __restrict__ touched, unsigned int* __restrict__ tcount) {
  const int lane = threadIdx.x & 63;
  for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
    const int u = (int)base + threadIdx.x;
    int64_t e = 0, e1 = 0;
    unsigned long long hi = 0;
    if (u < n && ((active[u >> 5] >> (u & 31)) & 1u)) {
      e = S.rowptr[u];
      e1 = S.rowptr[u + 1];
      hi = (unsigned long long)((native_of_dev ? native_of_dev[u] : u) + 1) << 32;
    }
    while (__ballot(e < e1)) {
      bool first = false;
      int c = 0;
      if (e < e1) {
        c = S.colidx[e];
        if (want == nullptr || ((want[c >> 5] >> (c & 31)) & 1u)) {
          const unsigned long long key = hi | (unsigned long long)(uint32_t)e;
          if (best[c] < key) first = atomicMax(&best[c], key) == 0ull;
        }
        e++;
      }
      const unsigned long long fm = __ballot(first);
      if (fm) {
        unsigned int start = 0;
        if (lane == 0) start = atomicAdd(tcount, (unsigned int)__popcll(fm));
        start = (unsigned int)__shfl((int)start, 0, 64);
        if (first) touched[start + (unsigned int)__popcll(fm & ((1ull << lane) - 1ull))] = c;
      }
    }
  }
}

// Top-down step of a REDUCE_COMMUTATIVE program with a 4-byte reduction type (SSSP distances,
// labels, counts): every out-edge of the active set evaluates its message and folds it into the
// destination's slot of `acc` -- (1 << 32 | value bits), 0 = empty, because there is no additive
// identity: the first message assigns -- with a compare-and-swap loop around the user's
// reduce_function.  Any order is fine for such programs, so the result is the pull's.
template <class P, class T, class U, class V, class E, bool USE_VP>
__global__ void __launch_bounds__(kBlock)
k_push_combine(ProgArg<P> pa, gm_csr_t S /* rows = sources */, const int32_t* __restrict__ list, int nlist,
               const unsigned int* __restrict__ off, const T* __restrict__ x, const V* __restrict__ vp,
               unsigned long long* __restrict__ acc, const uint32_t* __restrict__ want, int32_t* __restrict__ touched,
               unsigned int* __restrict__ tcount) {
  static_assert(sizeof(U) == 4, "packed (flag, value) slots need a 4-byte reduction type");
  const P& p = *reinterpret_cast<const P*>(pa.b);
  int u;
  int64_t e0, e1;
  if (!piece_of_block(S, list, nlist, off, blockIdx.x, &u, &e0, &e1)) return;
  const T m = x[u];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int64_t e = e0 + threadIdx.x + j * kBlock;
    bool first = false;
    int c = 0;
    if (e < e1) {
      c = S.colidx[e];
      if (want == nullptr || ((want[c >> 5] >> (c & 31)) & 1u)) {
        V vprow;
        if constexpr (USE_VP) vprow = vp[c];
        U res;
        p.P::process_message(m, edge_at<E>(S.vals, e), vprow, res);
        unsigned long long old = acc[c];
        while (true) {
          U folded = res;
          if (old != 0ull) {
            const uint32_t ob = (uint32_t)old;
            memcpy(&folded, &ob, 4);
            p.P::reduce_function(folded, res);
          }
          uint32_t fb;
          memcpy(&fb, &folded, 4);
          const unsigned long long neu = (1ull << 32) | (unsigned long long)fb;
          if (neu == old) break;  // the message changes nothing
          const unsigned long long seen = atomicCAS(&acc[c], old, neu);
          if (seen == old) { first = old == 0ull; break; }
          old = seen;
        }
      }
    }
    const unsigned long long fm = __ballot(first);
    if (fm) {
      unsigned int start = 0;
      if (lane == 0) start = atomicAdd(tcount, (unsigned int)__popcll(fm));
      start = (unsigned int)__shfl((int)start, 0, 64);
      if (first) touched[start + (unsigned int)__popcll(fm & ((1ull << lane) - 1ull))] = c;
    }
  }
}

// the rest of a top-down step, one lane per touched destination: the winning bid's message is
// evaluated and applied on the spot (a=b: exactly one message per destination, no y round trip);
// changed vertices are activated, counted and listed for the next step.  The grid covers an
// upper bound of the touched count (the active set's out-edges), the real count is read here.
// COMBINED: the slot holds the already folded value (k_push_combine) instead of a bid (k_push_bid).
template <class P, class T, class U, class V, class E, bool USE_VP, bool COMBINED>
__global__ void __launch_bounds__(kBlock)
k_push_finish(ProgArg<P> pa, gm_csr_t S, const T* __restrict__ x, const int32_t* __restrict__ dev_of_native,
              V* __restrict__ vp, unsigned long long* __restrict__ best, const int32_t* __restrict__ touched,
              const unsigned int* __restrict__ tcount, uint32_t* __restrict__ active, int* __restrict__ changed_flag,
              unsigned long long* __restrict__ stats, uint32_t* __restrict__ want, int32_t* __restrict__ next_list,
              unsigned int* __restrict__ next_count) {
  const unsigned int i = blockIdx.x * kBlock + threadIdx.x;
  bool changed = false;
  int v = 0;
  if (i < *tcount) {
    v = touched[i];
    const unsigned long long key = best[v];
    best[v] = 0ull;  // leave the scratch clean for the next step
    ProgArg<P> local = pa;  // apply() is non-const in the API: give it a private copy
    P& p = *reinterpret_cast<P*>(local.b);
    V old_prop = vp[v];
    bool wanted = true;
    if constexpr (program_row_filter<P>::enabled) wanted = program_row_filter<P>::wants(p, old_prop);
    if (wanted && key != 0ull) {
      U res;
      if constexpr (COMBINED) {
        const uint32_t rb = (uint32_t)key;
        memcpy(&res, &rb, sizeof(U) < 4 ? sizeof(U) : 4);
      } else {
        const int un = (int)(key >> 32) - 1;
        const int64_t e = (int64_t)(uint32_t)key;
        const int ud = dev_of_native ? dev_of_native[un] : un;
        T m = message_of(p, x, (const V*)vp, ud);
        V vprow;
        if constexpr (USE_VP) vprow = old_prop;
        p.P::process_message(m, edge_at<E>(S.vals, e), vprow, res);
      }
      V cur = old_prop;
      p.P::apply(res, cur);
      vp[v] = cur;
      changed = old_prop != cur;
      if (changed) atomicOr(&active[v >> 5], 1u << (v & 31));
      if constexpr (program_row_filter<P>::enabled)
        if (want != nullptr && !program_row_filter<P>::wants(p, cur)) atomicAnd(&want[v >> 5], ~(1u << (v & 31)));
    }
  }
  // next active set: size (one set of atomics per workgroup) and, while it is small, its list
  __shared__ unsigned long long s_c[kBlock / 64], s_e[kBlock / 64], s_m[kBlock / 64];
  const unsigned long long m = __ballot(changed);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long deg = changed ? (unsigned long long)(S.rowptr[v + 1] - S.rowptr[v]) : 0ull, mx = deg;
  for (int off = 32; off > 0; off >>= 1) {
    deg += __shfl_down(deg, off, 64);
    const unsigned long long o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if (lane == 0) { s_c[wv] = (unsigned long long)__popcll(m); s_e[wv] = deg; s_m[wv] = mx; }
  if (m != 0ull) {
    unsigned int start = 0;
    if (lane == 0) start = atomicAdd(next_count, (unsigned int)__popcll(m));
    start = (unsigned int)__shfl((int)start, 0, 64);
    // entries beyond the cap are never read (the host takes a top-down step only for small sets)
    if (changed && start < (unsigned int)kSparseListCap + 64u) next_list[start + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long c = 0, e = 0, x2 = 0;
    for (int w = 0; w < kBlock / 64; w++) { c += s_c[w]; e += s_e[w]; x2 = s_m[w] > x2 ? s_m[w] : x2; }
    if (c) {
      *changed_flag = 1;
      unsigned long long* slot = stats + 4 * (blockIdx.x % kStatSlots);
      atomicAdd(&slot[0], c);
      atomicAdd(&slot[1], e);
      atomicMax(&slot[2], x2);
    }
  }
}

// The same for a top-down step whose active set has many out-edges: no touched list (building one
// costs an atomic with a result per 64 destinations), all vertices are looked at instead, and the
// ordinary k_apply follows.
template <class P, class T, class U, class V, class E, bool USE_VP>
__global__ void __launch_bounds__(kBlock)
k_push_resolve(ProgArg<P> pa, gm_csr_t S, const T* __restrict__ x, const int32_t* __restrict__ dev_of_native,
               const V* __restrict__ vp, unsigned long long* __restrict__ best, U* __restrict__ y,
               uint32_t* __restrict__ ybits, int n) {
  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int v = blockIdx.x * kBlock + threadIdx.x;
  bool got = false;
  if (v < n) {
    const unsigned long long key = best[v];
    if (key != 0ull) {
      best[v] = 0ull;  // leave the scratch clean for the next push step
      bool wanted = true;
      if constexpr (program_row_filter<P>::enabled) wanted = program_row_filter<P>::wants(p, vp[v]);
      if (wanted) {
        const int un = (int)(key >> 32) - 1;
        const int64_t e = (int64_t)(uint32_t)key;
        const int ud = dev_of_native ? dev_of_native[un] : un;
        T m = message_of(p, x, vp, ud);
        V vprow;
        if constexpr (USE_VP) vprow = vp[v];
        U res;
        p.P::process_message(m, edge_at<E>(S.vals, e), vprow, res);
        y[v] = res;
        got = true;
      }
    }
  }
  const unsigned long long bm = __ballot(got);  // ybits were cleared before the step: plain word stores
  if ((threadIdx.x & 63) == 0 && v < n && bm) {
    if ((uint32_t)bm) ybits[v >> 5] = (uint32_t)bm;
    if ((uint32_t)(bm >> 32)) ybits[(v >> 5) + 1] = (uint32_t)(bm >> 32);
  }
}

// 64:1 summary of a presence bit vector: bit j of the result = "some bit of entries [64j, 64j+64) is set".
// While the active set is small the bottom-up kernels test this 128 KB table first: most of their
// presence tests then never touch the 8 MB vector (level 1 of BFS RMAT-26: half of the short-row
// kernel's time was those gathers).
__global__ void __launch_bounds__(kBlock)
k_bits_summary(const uint32_t* __restrict__ bits, int nwords, uint32_t* __restrict__ sum, int nsum) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= nsum) return;
  uint32_t out = 0;
  for (int b = 0; b < 32; b++) {
    const int w = (t * 32 + b) * 2;
    uint32_t any = 0;
    if (w < nwords) any |= bits[w];
    if (w + 1 < nwords) any |= bits[w + 1];
    if (any) out |= 1u << b;
  }
  sum[t] = out;
}

// ------------------------------------------------------------------------------------
// Sparse exchange of the message vector between shards (ACTIVE_ONLY programs with a small active set; the
// reference compresses a segment the same way before sending it when few entries are set,
// include/GMDP/vectors/DenseSegment.h:532-538,665-700).  A shard packs the messages of its listed active
// vertices as (global device id, message) entries into its block of the gather buffer -- `cap` entries per
// shard, unused ones marked with id -1 --, the blocks are all-gathered, and every shard scatters all
// entries into x and sets their presence bits (cleared beforehand).
template <class T>
struct alignas(4) sparse_entry {
  int32_t idx;
  unsigned char msg[(sizeof(T) + 3) / 4 * 4];
};
template <class T>
__global__ void __launch_bounds__(kBlock)
k_pack_frontier(const int32_t* __restrict__ list, int nlist, const T* __restrict__ x, int row_base, sparse_entry<T>* __restrict__ block, int cap) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  sparse_entry<T> e;
  memset(&e, 0, sizeof(e));
  e.idx = -1;
  if (i < nlist) {
    const int u = list[i];
    e.idx = row_base + u;
    const T m = x[(size_t)row_base + u];
    memcpy(e.msg, &m, sizeof(T));
  }
  block[i] = e;
}
template <class T>
__global__ void __launch_bounds__(kBlock)
k_unpack_frontier(const sparse_entry<T>* __restrict__ all, int64_t n, T* __restrict__ x, uint32_t* __restrict__ xbits) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const sparse_entry<T> e = all[i];
  if (e.idx < 0) return;
  T m;
  memcpy(&m, e.msg, sizeof(T));
  x[e.idx] = m;
  atomicOr(&xbits[e.idx >> 5], 1u << (e.idx & 31));
}

// ---- rows with an active in-neighbour (engine.hpp: guided pull; round 6) -----------------------------------------------------------------
// mark[v] := 1 for every destination v of an out-edge of an active vertex (S: rows = sources).  One lane per vertex of the active bitmap; a
// vertex of more than 32 out-edges is walked by its whole wave.  The pull multiply of an ORDERED program then folds only the marked rows --
// every other row has no present message -- and folds them exactly as before.
__global__ void __launch_bounds__(kBlock)
k_mark_rows_of_active(gm_csr_t S, const uint32_t* __restrict__ active, int n, uint32_t* __restrict__ mark) {
  const int lane = threadIdx.x & 63;
  for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
    const int i = (int)base + (int)threadIdx.x;
    const bool act = i < n && bit_get(active, i);
    long long e0 = 0, e1 = 0;
    if (act) { e0 = S.rowptr[i]; e1 = S.rowptr[i + 1]; }
    const bool big = act && e1 - e0 > 32;
    if (act && !big)
      for (long long e = e0; e < e1; e++) { const int c = S.colidx[e]; if (!((mark[c >> 5] >> (c & 31)) & 1u)) atomicOr(&mark[c >> 5], 1u << (c & 31)); }
    unsigned long long todo = __ballot(big);
    while (todo) {
      const int l = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const long long b0 = __shfl(e0, l, 64), b1 = __shfl(e1, l, 64);
      for (long long e = b0 + lane; e < b1; e += 64) { const int c = S.colidx[e]; if (!((mark[c >> 5] >> (c & 31)) & 1u)) atomicOr(&mark[c >> 5], 1u << (c & 31)); }
    }
  }
}
// the same from the LIST of the active vertices (while they are few): one workgroup per 1024-edge piece of a listed vertex's out-edges
// (k_piece_offsets), so that a hub's 10^6 out-edges are spread over the chip instead of being walked by one wave (6 ms for RMAT-26's)
__global__ void __launch_bounds__(kBlock)
k_mark_rows_of_list(gm_csr_t S, const int32_t* __restrict__ list, int nlist, const unsigned int* __restrict__ off, uint32_t* __restrict__ mark) {
  int u;
  int64_t e0, e1;
  if (!piece_of_block(S, list, nlist, off, blockIdx.x, &u, &e0, &e1)) return;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int64_t e = e0 + threadIdx.x + j * kBlock;
    if (e < e1) { const int c = S.colidx[e]; if (!((mark[c >> 5] >> (c & 31)) & 1u)) atomicOr(&mark[c >> 5], 1u << (c & 31)); }
  }
}

// ---- the sharded swept schedule (engine.hpp: run_swept_sharded; round 6) ---------------------------------------------------------------
// nog = bits & ~(bits of the listed rows), only = bits of the listed rows: the rows every shard applies while its giant rows' fold
// passes are still running, and the giant rows themselves (both arrays initialised by the caller: nog = a copy of bits, only = 0)
__global__ void __launch_bounds__(kBlock)
k_split_row_bits(const int32_t* __restrict__ list, int nlist, uint32_t* __restrict__ nog, uint32_t* __restrict__ only) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nlist) return;
  const int r = list[i];
  const uint32_t b = 1u << (r & 31);
  if (atomicAnd(&nog[r >> 5], ~b) & b) atomicOr(&only[r >> 5], b);
}
// apply + send of the NEXT iteration for the listed rows (the shard's giant rows, once their folds have joined), the new message also
// packed as a (device id, message) entry of this shard's block of the gather buffer (entries past the list: id -1).  Same arithmetic as
// k_apply_send: apply on a private copy of the program, send with the program as captured.
template <class P, class T, class U, class V>
__global__ void __launch_bounds__(kBlock)
k_apply_send_list(ProgArg<P> pa, const U* __restrict__ y, const uint32_t* __restrict__ ybits, V* __restrict__ vp, uint32_t* __restrict__ active,
                  const int32_t* __restrict__ list, int nlist, int* __restrict__ changed_flag, T* __restrict__ x, int row_base, sparse_entry<T>* __restrict__ block, int cap) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= cap) return;
  sparse_entry<T> e;
  memset(&e, 0, sizeof(e));
  e.idx = -1;
  if (i < nlist) {
    const int r = list[i];
    ProgArg<P> local = pa;
    P& p = *reinterpret_cast<P*>(local.b);
    V cur = vp[r];
    if (bit_get(ybits, r)) {
      V old_prop = cur;
      p.P::apply(y[r], cur);
      vp[r] = cur;
      if (old_prop != cur) { atomicOr(&active[r >> 5], 1u << (r & 31)); *changed_flag = 1; }
    }
    const P& ps = *reinterpret_cast<const P*>(pa.b);
    T m;
    ps.P::send_message(cur, m);
    x[(size_t)row_base + r] = m;
    e.idx = row_base + r;
    memcpy(e.msg, &m, sizeof(T));
  }
  block[i] = e;
}

// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_fill_u32(uint32_t* __restrict__ p, int64_t n, uint32_t v) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace dev
}  // namespace GraphMat
