// graphmat/engine.hpp -- host driver of the device iteration loop.
//
// Restates the control flow of the reference's run_graph_program
// (include/GraphMatRuntime.h:93-279) around the kernels of kernels.hpp: the whole
// loop is device resident; per iteration only the 1-int "changed" flag (and only
// when running until convergence) and the host hook do_every_iteration cross back.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>

#include <limits>
#include <type_traits>
#include <vector>

#include "kernels.hpp"

namespace GraphMat {

enum edge_direction { OUT_EDGES, IN_EDGES, ALL_EDGES };  // GraphProgram.h:34
enum activity_type { ACTIVE_ONLY, ALL_VERTICES };        // GraphProgram.h:36

template <class T, class U, class V, class E>
class GraphProgram;  // include/GraphProgram.h

namespace detail {

// Does program P leave do_every_iteration to the base class (whose hook is empty)?  Only then may the engine compute
// messages of iteration i+1 BEFORE the hook of iteration i has run (the fused apply + send pass, the two-stage sharded
// schedule): the reference always runs do_every_iteration before the next send (GraphMatRuntime.h:236), and a hook
// may change state send_message reads that no comparison of the program object's bytes can see (a pointer member's
// target, a global).  A class-type test, evaluated at compile time: &P::do_every_iteration names the base's member
// exactly when no class between GraphProgram and P declares one.
template <class F>
struct member_class_of;
template <class C, class R, class... A>
struct member_class_of<R (C::*)(A...)> { typedef C type; };
template <class P>
constexpr bool inherits_iteration_hook() {
  typedef GraphProgram<typename P::message_type, typename P::message_reduction_type, typename P::vertex_property_type, typename P::edge_type> Base;
  return std::is_same<typename member_class_of<decltype(&P::do_every_iteration)>::type, Base>::value;
}

#define GM_HIP_OK(expr)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      printf("GraphMat(HIP): %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, \
             __LINE__);                                                                         \
      exit(1); /* the reference's error convention: message + exit(1) */                        \
    }                                                                                           \
  } while (0)

inline int grid_for(int64_t n) { return (int)((n + dev::kBlock - 1) / dev::kBlock); }

// The options of a run (which exact strategy, which kernel form, ablation switches) are NOT state of this header: they
// live in the library, as process defaults (gm_set_option) that a graph may override (gm_graph_set_option), and a run
// reads them once through gm_graph_engine_options (graphmat_hip.h: gm_engine_options_t documents every field).  An
// application binary that instantiates this header and the library therefore always agree on them.

// Persistent workgroups with a large LDS hot set (k_spmv_rowwave, k_spmv_wave16p) trade occupancy -- and LDS the
// auxiliary stream's kernels would use next to them -- for fewer L2 requests.  Measured on RMAT, PageRank, both forms on
// against both off: scale 22 +1.7 %, 24 -0.6 %, 25 -3.5 %, 26 +3.5 %, 27 +5.1 % (profiles/r03_persistent_kernels.md):
// they pay once the message vector is far larger than the caches, i.e. where the column tiles are many.
// Not on shards: their hot set is the first entries of EVERY slice of x (a slice lookup per gather) and they are not
// tiled -- a shard of 8 of RMAT-26 multiplies in 1.11 ms with the plain kernels, 1.30 ms with the persistent ones.
// (column tiles with row classes fixed per row -- the two-stream schedule -- from 24 M ids on: RMAT-25, 5 tiles, 3.39 -> 3.10 ms)
inline bool persistent_forms_pay(const gm_csr_t& A) {
  return A.hot_slices <= 1 && ((int64_t)A.ncols >= (48ll << 20) || ((int64_t)A.ncols >= (24ll << 20) && A.rows_keep_stream != 0));
}
inline int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}


// (per-iteration trace in the reference's own words -- include/GraphMatRuntime.h:150-248 under __TIMING: phase times and
// "Iteration %d :: %f msec :: updated %d vertices :: changed %d vertices" -- is the option iteration_trace: the host
// synchronises after every phase, so it is a diagnostic mode, set by run_graph_program in -D__TIMING builds or by
// GRAPHMAT_ITERATION_TRACE=1.)

// HIP-event phase timer: every mark closes an interval that is charged to `tag`.
enum { TAG_START = 0, TAG_SEND = 1, TAG_ROWBLOCK = 2, TAG_WAVE = 3, TAG_GIANT = 4, TAG_APPLY = 5 };
struct PhaseTimer {
  bool on;
  hipStream_t s;
  std::vector<hipEvent_t> ev;
  std::vector<int> tags;
  explicit PhaseTimer(bool on_, hipStream_t s_) : on(on_), s(s_) {}
  void mark(int tag) {
    if (!on) return;
    hipEvent_t e;
    GM_HIP_OK(hipEventCreate(&e));
    GM_HIP_OK(hipEventRecord(e, s));
    ev.push_back(e);
    tags.push_back(tag);
  }
  // explicit interval on another stream (the giant-row passes run on an auxiliary stream)
  std::vector<hipEvent_t> aux_ev;  // begin, end, begin, end ...
  void aux_mark(hipStream_t as) {
    if (!on) return;
    hipEvent_t e;
    GM_HIP_OK(hipEventCreate(&e));
    GM_HIP_OK(hipEventRecord(e, as));
    aux_ev.push_back(e);
  }
  void finish(gm_run_stats_t* st) {
    if (!on || ev.empty()) return;
    GM_HIP_OK(hipEventSynchronize(ev.back()));
    for (size_t i = 1; i < ev.size(); i++) {
      if (tags[i] == TAG_START) continue;
      float ms = 0;
      GM_HIP_OK(hipEventElapsedTime(&ms, ev[i - 1], ev[i]));
      if (tags[i] == TAG_SEND) st->send_ms += ms;
      else if (tags[i] == TAG_ROWBLOCK) { st->rowblock_ms += ms; st->rowblock_launches++; }
      else if (tags[i] == TAG_WAVE) { st->wave_ms += ms; st->wave_launches++; }
      else if (tags[i] == TAG_GIANT) { st->giant_ms += ms; st->giant_launches++; }
      else if (tags[i] == TAG_APPLY) st->apply_ms += ms;
    }
    for (size_t i = 0; i + 1 < aux_ev.size(); i += 2) {
      GM_HIP_OK(hipEventSynchronize(aux_ev[i + 1]));
      float ms = 0;
      GM_HIP_OK(hipEventElapsedTime(&ms, aux_ev[i], aux_ev[i + 1]));
      st->giant_ms += ms;
      st->giant_launches++;
    }
    for (hipEvent_t e : aux_ev) (void)hipEventDestroy(e);
    aux_ev.clear();
    st->spmv_ms = st->rowblock_ms + st->wave_ms + st->giant_ms;
    float t = 0;
    GM_HIP_OK(hipEventElapsedTime(&t, ev.front(), ev.back()));
    st->total_ms = t;
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear();
    tags.clear();
  }
};

// auxiliary stream for the giant-row passes: their long serial chains overlap the
// throughput-bound row-block and wave kernels instead of running after them
// (the stream and its events belong to the graph: gm_graph_run_resources)
struct AuxStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  // long_rows: the long wave rows of the passes launched next also go to this stream, behind the giant passes (set by
  // the column-tile loop: per tile the kernels are small and each ends with a tail of a few busy waves -- overlapping the
  // one-wave-per-row kernel with the throughput-bound ones is worth 3 % at RMAT-26; untiled it costs 3-15 %)
  bool long_rows = false;
  // keep: the passes launched next belong to a run of column-tile passes whose row classes are fixed per row
  // (gm_csr_t.rows_keep_stream): the giant and one-wave-per-row kernels ALWAYS go to this stream, it is forked once
  // (`forked`) and nobody waits for it until the caller joins after the last tile (`pending`)
  bool keep = false, forked = false, pending = false;
  // defer: the giant-row passes launched next are not waited for by their launch_spmv call; whoever needs their
  // rows calls wait_join (the two-stage schedule starts them before the tail stage and joins before the head apply)
  bool defer = false;
  void wait_join(hipStream_t main) {
    if (s) (void)hipStreamWaitEvent(main, join, 0);
  }
  void attach(void* stream, void* fork_ev, void* join_ev) {
    s = (hipStream_t)stream;
    fork = (hipEvent_t)fork_ev;
    join = (hipEvent_t)join_ev;
  }
  void finish() {
    if (s) (void)hipStreamSynchronize(s);
  }
};

// ---- which exact strategy may evaluate a program's reduce_function ------------------------------
// A program can say so (program_traits<P>::reduce).  Otherwise -- and only when the user opts in with
// GRAPHMAT_TRUST_PROBE=1 -- the runtime asks the function
// itself: reduce_function is an ordinary host-callable method, so it is run on the host on
// random operands and compared, bit for bit, with float a+b, with a=b, and (integral types) with
// wrapping +, min and max.  Only an exact match on every sample selects the corresponding
// strategy; anything else keeps the always-correct ordered fold.  GRAPHMAT_NO_PROBE=1 disables it.
// The probe is OPT-IN (GRAPHMAT_TRUST_PROBE=1): a finite set of operands plus a sampled device cross-check is
// evidence, not a proof, and this library's contract is the reference's bits (SPMV.h:54-59: `c = a; reduce(c, b)`
// in stored order).  Without the opt-in a program that declares no trait gets the ordered fold -- always exact.
// (probe_reduce_guess: the questions and what the answers look like, nothing decided.  Besides the opt-in above, the engine
// uses a "float addition" answer of a program that declares nothing to SPECULATE on its giant rows -- the replay of the sum is
// then proven chunk by chunk with the program's own function, kernels.hpp: k_giant_verify_chunks -- which needs no trust.)
template <class P, class U>
int probe_reduce_guess(const P* gp) {
  // arithmetic reduction types only: the function is called on the host with synthetic operands, which is
  // harmless for numbers and not for pointers or structures with invariants
  if constexpr (!std::is_arithmetic<U>::value || std::is_same<U, bool>::value || sizeof(U) > 8) {
    return REDUCE_ORDERED;
  } else {
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    bool is_last = true, is_fadd = std::is_same<U, float>::value, is_iadd = std::is_integral<U>::value,
         is_min = std::is_integral<U>::value, is_max = std::is_integral<U>::value;
    auto test = [&](U a, U b) {
      U c = a;
      gp->P::reduce_function(c, b);
      if (memcmp(&c, &b, sizeof(U)) != 0) is_last = false;
      if constexpr (std::is_same<U, float>::value) {
        volatile float sum = a + b;
        float sv = sum;
        if (memcmp(&c, &sv, 4) != 0) is_fadd = false;
      }
      if constexpr (std::is_integral<U>::value) {
        typedef typename std::make_unsigned<U>::type UU;
        U add = (U)((UU)a + (UU)b), mn = a < b ? a : b, mx = a < b ? b : a;
        if (c != add) is_iadd = false;
        if (c != mn) is_min = false;
        if (c != mx) is_max = false;
      }
    };
    if constexpr (std::is_floating_point<U>::value) {
      // every pairing of a list of edge values (zeros of both signs, the extremes of the format, powers of
      // two around 1, thresholds a clamping or saturating function might use) ...
      const U fmaxv = std::numeric_limits<U>::max(), fminv = std::numeric_limits<U>::min(), den = std::numeric_limits<U>::denorm_min();
      const U edge[] = {(U)0, -(U)0, (U)1, -(U)1, (U)2, (U)0.5, (U)3, (U)1e-30, (U)1e30, -(U)1e30, (U)3e38 < fmaxv ? (U)3e38 : fmaxv, fmaxv, -fmaxv, fminv,
                        den, (U)16777216, (U)16777217, (U)1e10, (U)65504, (U)1e-10};
      for (U a : edge)
        for (U b : edge) test(a, b);
      // ... and pseudo-random pairs with independent exponents over the whole range (sums that overflow,
      // underflow, cancel, or round at every distance), both signs, plus same-binade pairs rich in ties
      for (int k = 0; k < 4096; k++) {
        U a, b;
        if constexpr (std::is_same<U, float>::value) {
          uint32_t ba = (uint32_t)next(), bb = (uint32_t)next();
          if (((ba >> 23) & 0xff) == 0xff) ba ^= 0x00800000u;  // no inf / nan
          if (((bb >> 23) & 0xff) == 0xff) bb ^= 0x00800000u;
          if (k % 4 == 0) bb = (bb & 0x807fffffu) | (ba & 0x7f800000u);             // same binade
          if (k % 8 == 1) { ba &= 0xfffff000u; bb = (bb & 0x807ff800u) | (ba & 0x7f800000u) | 0x800u; }  // ties
          memcpy(&a, &ba, 4);
          memcpy(&b, &bb, 4);
        } else {
          unsigned long long ba = next(), bb = next();
          if (((ba >> 52) & 0x7ff) == 0x7ff) ba ^= 1ull << 52;
          if (((bb >> 52) & 0x7ff) == 0x7ff) bb ^= 1ull << 52;
          memcpy(&a, &ba, sizeof(U));
          memcpy(&b, &bb, sizeof(U));
        }
        test(a, b);
      }
    } else {
      const U lo = std::numeric_limits<U>::min(), hi = std::numeric_limits<U>::max();
      const U edge[] = {(U)0, (U)1, (U)2, (U)-1, lo, hi, (U)(hi - 1), (U)(lo + 1), (U)(hi / 2), (U)(hi / 2 + 1), (U)255, (U)256, (U)65535, (U)65536};
      for (U a : edge)
        for (U b : edge) test(a, b);
      for (int k = 0; k < 4096; k++) {
        unsigned long long ra = next(), rb = next();
        if (k % 5 == 0) { ra &= 0xff; rb &= 0xff; }
        if (k % 7 == 0) { ra >>= (next() % 64); rb >>= (next() % 64); }
        U a, b;
        memcpy(&a, &ra, sizeof(U));
        memcpy(&b, &rb, sizeof(U));
        test(a, b);
      }
    }
    if (is_last) return REDUCE_LAST;
    if (is_fadd) return REDUCE_F32_ADD;
    if (is_iadd || is_min || is_max) return REDUCE_COMMUTATIVE;
    return REDUCE_ORDERED;
  }
}
template <class P, class U>
int probe_reduce_kind(const P* gp) {
  const char* on = getenv("GRAPHMAT_TRUST_PROBE");
  if (!(on && on[0] == '1')) return REDUCE_ORDERED;
  const char* off = getenv("GRAPHMAT_NO_PROBE");
  if (off && off[0] == '1') return REDUCE_ORDERED;
  return probe_reduce_guess<P, U>(gp);
}
template <class P, class U>
int reduce_kind_of(const P* gp) {
  if constexpr ((int)program_traits<P>::reduce != (int)REDUCE_AUTO) return (int)program_traits<P>::reduce;
  else return probe_reduce_kind<P, U>(gp);
}

// which programs the 16-rows-per-wave ordered kernel takes
template <class U, bool USE_VP, int RK>
constexpr bool wave16_ok() {
  return !USE_VP && (RK == REDUCE_ORDERED || RK == REDUCE_F32_ADD) && std::is_trivially_copyable<U>::value &&
         (sizeof(U) == 4 || sizeof(U) == 8);
}

// what a multiply pass is launched with: the graph (scratch, tiles), the run's stream, the options of this run, and where to
// account launches and phase times; `aux` = the auxiliary stream's state or null
struct Launch {
  gm_graph_t* g;
  hipStream_t s;
  const gm_engine_options_t& opt;
  int* launches;
  PhaseTimer* timer;
  AuxStream* aux;
  bool tiled_untiled_pass = false;  // this launch is the untiled pass of a tiled multiply (set when aux is detached from it)
  bool terms_ready = false;  // the giant rows' products are in the products stream already (the sweep gathered them): fold passes only
  int32_t* spec_off = nullptr;  // device flag of this run: a speculation on the giant rows failed its proof, later passes fold in order right away
  bool guess_f32_add = false;  // the program declares no reduction, but its reduce_function answers like a float addition (probe_reduce_guess):
                               // giant rows are replayed as float sums and every chunk is PROVEN with the program's own function (k_giant_verify_chunks)
};

// one multiply+reduce pass over one direction of the adjacency, strategy RK
template <class P, class T, class U, class V, class E, bool USE_VP, int RK>
void launch_spmv_rk(const Launch& L, const dev::ProgArg<P>& pa, const gm_csr_t& A_in, const T* x, const uint32_t* xbits,
                    const V* vp, U* y, uint32_t* ybits, int accumulate, const uint32_t* want = nullptr, bool grouped = false,
                    const uint32_t* xsum = nullptr) {
  constexpr int WPB = dev::kBlock / 64;  // rows (waves) per workgroup of k_spmv_wave
  gm_graph_t* const g = L.g;
  const hipStream_t s = L.s;
  int* const launches = L.launches;
  PhaseTimer* const timer = L.timer;
  AuxStream* const aux = L.aux;
#ifdef GRAPHMAT_ABLATION
  gm_csr_t A = A_in;
  {
    int ntile = 1;
    gm_graph_tiles(g, GM_DIR_OUT, &ntile);
    const bool whole = A.hot_base == 0 && A.hot_len >= A.ncols;
    A.cold_from = (whole && ntile > 1) ? -L.opt.ablate_cold_short : L.opt.ablate_cold_from;
    static std::vector<const void*> seen;
    bool first = true;
    for (const void* q : seen) first = first && q != (const void*)A.colidx;
    if (first && A.cold_from != 0) {
      seen.push_back((const void*)A.colidx);
      if (ntile > 1) {
        int base[GM_MAX_TILES + 1] = {0};
        for (int t = 0; t < ntile; t++) {
          gm_csr_t At;
          const uint32_t* prev = nullptr;
          if (gm_graph_tile(g, GM_DIR_OUT, t, &At, &prev) == GM_OK) base[t] = At.hot_base;
        }
        GM_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(dev::g_abl_tile_base), base, sizeof(base)));
        GM_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(dev::g_abl_ntiles), &ntile, sizeof(int)));
      }
      unsigned long long* d_n = nullptr;
      unsigned long long h_n = 0;
      GM_HIP_OK(hipMalloc(&d_n, 8));
      GM_HIP_OK(hipMemset(d_n, 0, 8));
      hipLaunchKernelGGL(dev::k_abl_count, dim3(4096), dim3(256), 0, s, A.colidx, (int64_t)A.nnz, A.cold_from, A.hot_base, d_n);
      GM_HIP_OK(hipMemcpyAsync(&h_n, d_n, 8, hipMemcpyDeviceToHost, s));
      GM_HIP_OK(hipStreamSynchronize(s));
      GM_HIP_OK(hipFree(d_n));
      printf("GraphMat(HIP) ablation: view base %d len %d: %lld edges, %llu cold (cold_from %d)\n", A.hot_base, A.hot_len, (long long)A.nnz, h_n, A.cold_from);
    }
  }
#else
  const gm_csr_t& A = A_in;
#endif
  if (A.nnz == 0) return;
  const bool defer = aux != nullptr && aux->s != nullptr && aux->defer;
  const bool keep = aux != nullptr && aux->s != nullptr && aux->keep;
  const bool overlap = aux != nullptr && aux->s != nullptr && A.ngiant > 0 && (defer || keep || A.nblk > 0 || A.nmid > 0);
  // The long wave rows (one wave each, serial in the row's length: the launch ends with a tail in which a few waves
  // finish their rows on an otherwise idle chip) go to the auxiliary stream as well, behind the giant passes, while the
  // throughput-bound row-block and 16-rows-per-wave kernels run on the main stream.
  bool long_on_aux = false;
  if constexpr (wave16_ok<U, USE_VP, RK>())
    long_on_aux = aux != nullptr && aux->s != nullptr && aux->long_rows && !defer && !(grouped && want != nullptr) && A.nmid > 0 && A.nmid_long > 0 &&
                  (keep || A.nblk > 0 || A.nmid > A.nmid_long) && !(L.opt.debug_flags & dev::DBG_NO_WAVE16);
  bool forked = false;
  auto fork_aux = [&]() {
    if (forked) return;
    if (!(keep && aux->forked)) {
      GM_HIP_OK(hipEventRecord(aux->fork, s));
      GM_HIP_OK(hipStreamWaitEvent(aux->s, aux->fork, 0));
    }
    forked = true;
    if (keep) aux->forked = true;
  };
  if (A.ngiant > 0) {
    hipStream_t gs = s;
    if (overlap) {
      fork_aux();
      gs = aux->s;
      if (timer) timer->aux_mark(gs);
    }
    // giant rows: a workgroup each when the reduction kind has a block-wide strategy,
    // otherwise the ordered wave fold (always correct)
    if constexpr ((RK == REDUCE_F32_ADD && sizeof(U) == 4) ||
                  ((RK == REDUCE_COMMUTATIVE || RK == REDUCE_LAST) && sizeof(U) <= 8)) {
      U* terms = nullptr;
      unsigned long long* tpres = nullptr;
      dev::gchunk_state* maps = nullptr;
      if constexpr (RK == REDUCE_F32_ADD) {
        // pass 1: products of all giant-row edges, spread over the whole chip
        void *p6 = nullptr, *p7 = nullptr;
        gm_graph_workspace(g, 6, (size_t)A.giant_edges * sizeof(U) + 64, &p6);
        terms = (U*)p6;
        if (xbits != nullptr) {
          gm_graph_workspace(g, 14, (size_t)A.giant_edges / 8 + 64, &p7);  // (slot 7 holds the active-set list of sharded ACTIVE_ONLY runs)
          tpres = (unsigned long long*)p7;
        }
        // piece maps (kernels.hpp: gchunk_state): the exact replay of a giant row spread over the whole chip, its serial
        // part limited to the binade crossings.  float sums over a dense x only.
        if constexpr (std::is_same<U, float>::value) {
          if (xbits == nullptr && want == nullptr && L.opt.giant_maps != 0) maps = (dev::gchunk_state*)A.gchunk_state;
        }
        if (L.terms_ready) {
          // (the products are in the stream already -- the sweep gathered them: nothing to do here)
        } else {
          hipLaunchKernelGGL((dev::k_giant_terms<P, T, U, V, E, USE_VP>), dim3(A.ngchunk), dim3(dev::kBlock), 0, gs, pa, A, x, xbits, vp, terms, tpres
                             GM_DBG_ARG(L.opt.debug_flags), maps);
          (*launches)++;
        }
      }
      bool replayed = false;
      if constexpr (RK == REDUCE_F32_ADD && std::is_same<U, float>::value) {
        if (maps != nullptr) {  // the sub-pieces' maps for the binades predicted from this pass's products, then one wave per row walks them (kernels.hpp: gchunk_state)
          hipLaunchKernelGGL(dev::k_giant_sums, dim3(A.ngchunk), dim3(dev::kBlock), 0, gs, A, (const float*)terms, maps);
          hipLaunchKernelGGL(dev::k_giant_maps, dim3(A.ngchunk), dim3(dev::kBlock), 0, gs, A, (const float*)terms, maps, (const float*)y, (const uint32_t*)ybits, accumulate);
          hipLaunchKernelGGL((dev::k_giant_replay_maps<P, U>), dim3(A.ngiant), dim3(64), 0, gs, pa, A, y, ybits, accumulate, (const U*)terms, (const dev::gchunk_state*)maps,
                             (unsigned long long*)nullptr, (const int32_t*)nullptr);
          (*launches) += 2;
          replayed = true;
        }
      }
      if (!replayed)
        hipLaunchKernelGGL((dev::k_spmv_giant<P, T, U, V, E, USE_VP, RK>), dim3(A.ngiant), dim3(dev::kGiant), 0, gs, pa,
                           A, x, xbits, vp, y, ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), (const U*)terms,
                           (const unsigned long long*)tpres, want, (dev::gchunk_state*)nullptr);
    } else {
      // plain ordered fold (any reduce_function): products spread over the chip by k_giant_terms, then one wave per row
      // folds the dense products stream in stored order (kernels.hpp: k_giant_fold_ordered).  Larger reduction types keep
      // the one-wave-per-row kernel that gathers by itself.
      bool two_pass = false;
      // speculate-and-prove (kernels.hpp: k_giant_verify_chunks / k_giant_verify_chunks_any): the whole rows, nothing in y to start from
      const bool may_speculate = L.opt.ordered_giant_two_pass >= 2 && L.spec_off != nullptr && !(accumulate & dev::ACC_READ_PREV) && A.gchunk_row != nullptr;
      // workspace slot 15: [chunk boundaries, 8 bytes each][redo flag per giant row] (apps/speculated_float_sum.cpp reads these two), then the other arrays
      const size_t o_redo = ((size_t)A.ngchunk + 2) * 8, o_bhas = o_redo + (((size_t)A.ngiant * 4 + 7) & ~(size_t)7), o_tval = o_bhas + (((size_t)A.ngchunk * 4 + 7) & ~(size_t)7),
                   o_thas = o_tval + (size_t)A.ngchunk * 8, o_fin = o_thas + (((size_t)A.ngchunk * 4 + 7) & ~(size_t)7), o_finhas = o_fin + (size_t)A.ngiant * 8,
                   o_end = o_finhas + (size_t)A.ngiant * 4 + 64;
      if constexpr (std::is_same<U, float>::value) {
        // a float sum by its answers, a dense x, every row wanted: the exact replay, proven chunk by chunk
        void *p6 = nullptr, *p15 = nullptr;
        if (may_speculate && L.guess_f32_add && xbits == nullptr && want == nullptr &&
            gm_graph_workspace(g, 6, (size_t)A.giant_edges * sizeof(U) + 64, &p6) == GM_OK && gm_graph_workspace(g, 15, o_end, &p15) == GM_OK) {
          U* terms = (U*)p6;
          unsigned long long* bounds = (unsigned long long*)p15;
          int32_t* redo = (int32_t*)((char*)p15 + o_redo);
          dev::gchunk_state* maps = L.opt.giant_maps != 0 ? (dev::gchunk_state*)A.gchunk_state : nullptr;
          GM_HIP_OK(hipMemsetAsync(redo, 0, (size_t)A.ngiant * 4, gs));
          if (!L.terms_ready) {
            hipLaunchKernelGGL((dev::k_giant_terms<P, T, U, V, E, USE_VP>), dim3(A.ngchunk), dim3(dev::kBlock), 0, gs, pa, A, x, xbits, vp, terms,
                               (unsigned long long*)nullptr GM_DBG_ARG(L.opt.debug_flags), maps);
            (*launches)++;
          }
          if (maps != nullptr) {
            hipLaunchKernelGGL(dev::k_giant_sums, dim3(A.ngchunk), dim3(dev::kBlock), 0, gs, A, (const float*)terms, maps);
            hipLaunchKernelGGL(dev::k_giant_maps, dim3(A.ngchunk), dim3(dev::kBlock), 0, gs, A, (const float*)terms, maps, (const float*)y, (const uint32_t*)ybits, accumulate);
            hipLaunchKernelGGL((dev::k_giant_replay_maps<P, U>), dim3(A.ngiant), dim3(64), 0, gs, pa, A, y, ybits, accumulate, (const U*)terms, (const dev::gchunk_state*)maps, bounds,
                               (const int32_t*)L.spec_off);
            (*launches) += 2;
          } else
            hipLaunchKernelGGL((dev::k_spmv_giant<P, T, U, V, E, USE_VP, REDUCE_F32_ADD>), dim3(A.ngiant), dim3(dev::kGiant), 0, gs, pa, A, x, xbits, vp, y,
                               ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), (const U*)terms, (const unsigned long long*)nullptr, want, (dev::gchunk_state*)nullptr,
                               bounds, (const int32_t*)L.spec_off);
          hipLaunchKernelGGL((dev::k_giant_verify_chunks<P, U>), dim3((A.ngchunk + WPB - 1) / WPB), dim3(dev::kBlock), 0, gs, pa, A, (const U*)terms,
                             (const unsigned long long*)bounds, (const U*)y, redo, L.spec_off);
          hipLaunchKernelGGL((dev::k_giant_fold_ordered<P, U, V>), dim3((A.ngiant + WPB - 1) / WPB), dim3(dev::kBlock), 0, gs, pa, A, vp, y, ybits,
                             accumulate, (const U*)terms, (const unsigned long long*)nullptr, want, (const int32_t*)redo, (const int32_t*)L.spec_off);
          (*launches) += 2;
          two_pass = true;
        }
      }
      if constexpr (dev::stageable<U>::value && std::is_trivially_copyable<U>::value && (sizeof(U) == 4 || sizeof(U) == 8)) {
        // any function: associativity on the operands at hand, speculated and proven chunk by chunk
        void *p6 = nullptr, *p7 = nullptr, *p15 = nullptr;
        if (!two_pass && may_speculate && gm_graph_workspace(g, 6, (size_t)A.giant_edges * sizeof(U) + 64, &p6) == GM_OK &&
            (xbits == nullptr || gm_graph_workspace(g, 14, (size_t)A.giant_edges / 8 + 64, &p7) == GM_OK) && gm_graph_workspace(g, 15, o_end, &p15) == GM_OK) {
          const U* terms = (const U*)p6;
          const unsigned long long* tpres = (const unsigned long long*)p7;
          char* w = (char*)p15;
          U *bval = (U*)w, *tval = (U*)(w + o_tval), *fin = (U*)(w + o_fin);
          int32_t *redo = (int32_t*)(w + o_redo), *bhas = (int32_t*)(w + o_bhas), *thas = (int32_t*)(w + o_thas), *finhas = (int32_t*)(w + o_finhas);
          const unsigned cgrid = (unsigned)((A.ngchunk + WPB - 1) / WPB), rgrid = (unsigned)((A.ngiant + WPB - 1) / WPB);
          GM_HIP_OK(hipMemsetAsync(redo, 0, (size_t)A.ngiant * 4, gs));
          if (!L.terms_ready) {
            hipLaunchKernelGGL((dev::k_giant_terms<P, T, U, V, E, USE_VP>), dim3(A.ngchunk), dim3(dev::kBlock), 0, gs, pa, A, x, xbits, vp, (U*)p6,
                               (unsigned long long*)p7 GM_DBG_ARG(L.opt.debug_flags), (dev::gchunk_state*)nullptr);
            (*launches)++;
          }
          hipLaunchKernelGGL((dev::k_giant_chunk_totals<P, U, V>), dim3(cgrid), dim3(dev::kBlock), 0, gs, pa, A, vp, want, terms, tpres, tval, thas,
                             (const int32_t*)L.spec_off);
          hipLaunchKernelGGL((dev::k_giant_chunk_scan<P, U, V>), dim3(rgrid), dim3(dev::kBlock), 0, gs, pa, A, vp, want, y, ybits, accumulate, (const U*)tval,
                             (const int32_t*)thas, bval, bhas, fin, finhas, (const int32_t*)L.spec_off);
          hipLaunchKernelGGL((dev::k_giant_verify_chunks_any<P, U, V>), dim3(cgrid), dim3(dev::kBlock), 0, gs, pa, A, vp, want, terms, tpres, (const U*)bval,
                             (const int32_t*)bhas, (const U*)fin, (const int32_t*)finhas, redo, L.spec_off);
          hipLaunchKernelGGL((dev::k_giant_fold_ordered<P, U, V>), dim3(rgrid), dim3(dev::kBlock), 0, gs, pa, A, vp, y, ybits, accumulate, terms, tpres, want,
                             (const int32_t*)redo, (const int32_t*)L.spec_off);
          (*launches) += 3;
          two_pass = true;
        }
      }
      if constexpr (dev::stageable<U>::value && std::is_trivially_copyable<U>::value) {
        void *p6 = nullptr, *p7 = nullptr;
        if (two_pass) {
        } else
        if (L.opt.ordered_giant_two_pass != 0 && gm_graph_workspace(g, 6, (size_t)A.giant_edges * sizeof(U) + 64, &p6) == GM_OK &&
            (xbits == nullptr || gm_graph_workspace(g, 14, (size_t)A.giant_edges / 8 + 64, &p7) == GM_OK)) {
          if (!L.terms_ready) {
            hipLaunchKernelGGL((dev::k_giant_terms<P, T, U, V, E, USE_VP>), dim3(A.ngchunk), dim3(dev::kBlock), 0, gs, pa,
                               A, x, xbits, vp, (U*)p6, (unsigned long long*)p7 GM_DBG_ARG(L.opt.debug_flags), (dev::gchunk_state*)nullptr);
            (*launches)++;
          }
          hipLaunchKernelGGL((dev::k_giant_fold_ordered<P, U, V>), dim3((A.ngiant + WPB - 1) / WPB), dim3(dev::kBlock), 0, gs, pa, A, vp, y,
                             ybits, accumulate, (const U*)p6, (const unsigned long long*)p7, want);
          two_pass = true;
        }
      }
      if (!two_pass)
        hipLaunchKernelGGL((dev::k_spmv_wave<P, T, U, V, E, USE_VP, REDUCE_ORDERED>),
                           dim3((A.ngiant + WPB - 1) / WPB), dim3(dev::kBlock), 0, gs, pa, A, A.giant_row, A.ngiant, x,
                           xbits, vp, y, ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), want);
    }
    (*launches)++;
    if (overlap) {
      if (timer) timer->aux_mark(gs);
      GM_HIP_OK(hipEventRecord(aux->join, gs));
    } else if (timer) {
      timer->mark(TAG_GIANT);
    }
  }
  if (A.nblk > 0 && RK == REDUCE_LAST) {
    // a=b: one lane per row over the whole row range (short rows pick themselves by their length)
    if constexpr (RK == REDUCE_LAST) {
      hipLaunchKernelGGL((dev::k_spmv_short_last<P, T, U, V, E, USE_VP>), dim3(grid_for(A.nrows)), dim3(dev::kBlock), 0, s,
                         pa, A, x, xbits, vp, y, ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), want, xsum);
      (*launches)++;
      if (timer) timer->mark(TAG_ROWBLOCK);
    }
  } else if (A.nblk > 0) {
    bool done = false;
    if constexpr (!USE_VP && sizeof(T) == 4 && sizeof(U) == 4 && RK != REDUCE_LAST && std::is_trivially_copyable<T>::value) {
      // persistent workgroups sharing a large LDS hot set, row-blocks taken by waves (kernels.hpp: k_spmv_rowwave)
      const int form = L.opt.rowwave_form & 15;
      const bool any_size = (L.opt.rowwave_form & 16) != 0;  // (tests: also for small graphs)
      const bool whole_cols = A.hot_base == 0 && A.hot_len >= A.ncols;  // (not a column tile: the untiled pass of a tiled graph, or an untiled graph)
      if (form > 0 && xbits == nullptr && want == nullptr && !program_row_filter<P>::enabled && (persistent_forms_pay(A) || any_size) &&
          !(whole_cols && (keep || L.tiled_untiled_pass) && L.opt.untiled_pass_plain != 0)) {
        auto persistent = [&](auto block_c, auto hot_c, int fit) {
          constexpr int BLOCK = decltype(block_c)::value, HOT = decltype(hot_c)::value;
          const int per_cu = L.opt.persist_per_cu > 0 && L.opt.persist_per_cu < fit ? L.opt.persist_per_cu : fit;
          int grid = cu_count() * per_cu;
          const int need = (A.nblk + BLOCK / 64 - 1) / (BLOCK / 64);
          if (grid > need) grid = need;
          hipLaunchKernelGGL((dev::k_spmv_rowwave<P, T, U, V, E, BLOCK, HOT>), dim3(grid), dim3(BLOCK), 0, s, pa, A, x, y, ybits, accumulate);
          done = true;
        };
        // (1024 threads / 30720 entries: the row-block time falls by 15 % but the kernels of the auxiliary stream find no
        // LDS next to it and the iteration gains nothing; 20480 entries leave them room; 512 / 16384 and 256 / 8192: no gain)
        if (form == 4) persistent(std::integral_constant<int, 1024>(), std::integral_constant<int, 20480>(), 1);
      }
    }
    if (done) {
    } else if (xbits == nullptr)
      hipLaunchKernelGGL((dev::k_spmv_rowblock<P, T, U, V, E, USE_VP, true, RK>), dim3(A.nblk), dim3(dev::kBlock), 0, s, pa, A,
                         x, xbits, vp, y, ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), want);
    else
      hipLaunchKernelGGL((dev::k_spmv_rowblock<P, T, U, V, E, USE_VP, false, RK>), dim3(A.nblk), dim3(dev::kBlock), 0, s, pa, A,
                         x, xbits, vp, y, ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), want);
    (*launches)++;
    if (timer) timer->mark(TAG_ROWBLOCK);
  }
  if (A.nmid > 0) {
    if (grouped && want != nullptr) {  // a row bitmap governs: 64 list entries per wave, only the wanted rows are worked on
      const int groups = (A.nmid + 63) / 64;
      // (running this kernel on the auxiliary stream next to the short rows' kernel was measured: both want every wave
      // slot of the chip, side by side each takes twice as long -- RMAT-26, first bottom-up level: 1.55 against 1.52 ms)
      bool done = false;
      if constexpr (RK == REDUCE_LAST) {
        if (L.opt.last_rows_lanes == 16) {
          hipLaunchKernelGGL((dev::k_spmv_wave_grouped<P, T, U, V, E, USE_VP, RK, 16>), dim3((groups + WPB - 1) / WPB),
                             dim3(dev::kBlock), 0, s, pa, A, A.mid_row, A.nmid, x, xbits, vp, y, ybits, accumulate
                             GM_DBG_ARG(L.opt.debug_flags), want, xsum);
          done = true;
        }
      }
      if (!done)
        hipLaunchKernelGGL((dev::k_spmv_wave_grouped<P, T, U, V, E, USE_VP, RK>), dim3((groups + WPB - 1) / WPB),
                           dim3(dev::kBlock), 0, s, pa, A, A.mid_row, A.nmid, x, xbits, vp, y, ybits, accumulate
                           GM_DBG_ARG(L.opt.debug_flags), want, xsum);
    } else if (wave16_ok<U, USE_VP, RK>() && !(L.opt.debug_flags & dev::DBG_NO_WAVE16)) {
      if constexpr (wave16_ok<U, USE_VP, RK>()) {
        // ordered folds: the long rows at the head of the list get a wave each, the rest are folded 16 to a wave
        const int nlong = A.nmid_long < A.nmid ? A.nmid_long : A.nmid;
        const int rest = A.nmid - nlong;
        const int groups = (rest + dev::kWaveRows - 1) / dev::kWaveRows;
        // the long rows: a wave each (k_spmv_wave), on the auxiliary stream during column-tile passes.  (Handing them to
        // the waves of the persistent kernel below -- one launch, long rows first off the same work counter -- was
        // measured at RMAT-26: 6.6 -> 7.9 ms per iteration.  wave_row's ordered fold is a serial chain per wave and wants
        // the 32 waves per CU the plain kernel gets, not the 16 of a workgroup that holds 90 KB of LDS.)
        if (nlong > 0) {
          hipStream_t ls = s;
          if (long_on_aux) {
            fork_aux();
            ls = aux->s;
            if (timer) timer->aux_mark(ls);
          }
          hipLaunchKernelGGL((dev::k_spmv_wave<P, T, U, V, E, USE_VP, RK>), dim3((nlong + WPB - 1) / WPB), dim3(dev::kBlock), 0, ls,
                             pa, A, A.mid_row, nlong, x, xbits, vp, y, ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), want);
          if (long_on_aux) {
            if (timer) timer->aux_mark(ls);
            GM_HIP_OK(hipEventRecord(aux->join, ls));  // (re-recorded behind the giant passes' record: the wait below sees this one)
          }
        }
        // Large graphs: the 16-row groups go to persistent 1024-thread workgroups that load the slice's 22528 busiest x
        // entries into LDS once, their waves taking groups in an interleaved order (kernels.hpp: k_spmv_wave16p; measured
        // in profiles/r03_persistent_kernels.md)
        bool done = false;
        if constexpr (sizeof(T) == 4 && sizeof(U) == 4) {
          const int form = persistent_forms_pay(A) || (L.opt.wave16_form & 16) ? (L.opt.wave16_form & 15) : 0;
          if (rest > 0 && form == 2) {
            constexpr int BLOCK = 1024, HOT = 22528;
            int grid = cu_count();
            const int need = (groups + BLOCK / 64 - 1) / (BLOCK / 64);
            if (grid > need) grid = need;
            hipLaunchKernelGGL((dev::k_spmv_wave16p<P, T, U, V, E, BLOCK, HOT>), dim3(grid), dim3(BLOCK), 0, s, pa, A, A.mid_row + nlong, rest,
                               x, xbits, vp, y, ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), want);
            done = true;
          }
        }
        if (!done && rest > 0) {
          constexpr int W16 = dev::kWave16Block / 64;
          hipLaunchKernelGGL((dev::k_spmv_wave16<P, T, U, V, E>), dim3((groups + W16 - 1) / W16), dim3(dev::kWave16Block), 0, s, pa, A,
                             A.mid_row + nlong, rest, x, xbits, vp, y, ybits, accumulate GM_DBG_ARG(L.opt.debug_flags), want);
        }
      }
    } else
      hipLaunchKernelGGL((dev::k_spmv_wave<P, T, U, V, E, USE_VP, RK>), dim3((A.nmid + WPB - 1) / WPB),
                         dim3(dev::kBlock), 0, s, pa, A, A.mid_row, A.nmid, x, xbits, vp, y, ybits, accumulate
                         GM_DBG_ARG(L.opt.debug_flags), want);
    (*launches)++;
    if (timer && !long_on_aux) timer->mark(TAG_WAVE);
  }
  if (keep) {
    if (overlap || long_on_aux) aux->pending = true;  // (joined by the caller after the last tile)
  } else if ((overlap && !defer) || long_on_aux) GM_HIP_OK(hipStreamWaitEvent(s, aux->join, 0));
  if (long_on_aux && timer) timer->mark(TAG_WAVE);  // the wave rows are done when both streams are
}

// rk: REDUCE_* chosen for this run.  Programs with a declared kind only instantiate that one.
template <class P, class T, class U, class V, class E, bool USE_VP>
void launch_spmv(const Launch& L, const dev::ProgArg<P>& pa, const gm_csr_t& A, const T* x, const uint32_t* xbits, const V* vp, U* y,
                 uint32_t* ybits, int accumulate, int rk = REDUCE_ORDERED, const uint32_t* want = nullptr, bool grouped = false,
                 const uint32_t* xsum = nullptr) {
  if constexpr ((int)program_traits<P>::reduce != (int)REDUCE_AUTO) {
    launch_spmv_rk<P, T, U, V, E, USE_VP, (int)program_traits<P>::reduce>(L, pa, A, x, xbits, vp, y, ybits, accumulate, want, grouped, xsum);
  } else {
    if constexpr (std::is_same<U, float>::value) {
      if (rk == REDUCE_F32_ADD) {
        launch_spmv_rk<P, T, U, V, E, USE_VP, REDUCE_F32_ADD>(L, pa, A, x, xbits, vp, y, ybits, accumulate, want, grouped, xsum);
        return;
      }
    }
    if constexpr (std::is_trivially_copyable<U>::value && sizeof(U) <= 8) {
      if (rk == REDUCE_LAST) {
        launch_spmv_rk<P, T, U, V, E, USE_VP, REDUCE_LAST>(L, pa, A, x, xbits, vp, y, ybits, accumulate, want, grouped, xsum);
        return;
      }
      if (rk == REDUCE_COMMUTATIVE) {
        launch_spmv_rk<P, T, U, V, E, USE_VP, REDUCE_COMMUTATIVE>(L, pa, A, x, xbits, vp, y, ybits, accumulate, want, grouped, xsum);
        return;
      }
    }
    launch_spmv_rk<P, T, U, V, E, USE_VP, REDUCE_ORDERED>(L, pa, A, x, xbits, vp, y, ybits, accumulate, want, grouped, xsum);
  }
}
// ... with the 2- / 3-operand form chosen at run time (GraphProgram::process_message_requires_vertexprop, SPMV.h:67-71)
template <class P, class T, class U, class V, class E>
void launch_spmv_vp(bool use_vp, const Launch& L, const dev::ProgArg<P>& pa, const gm_csr_t& A, const T* x, const uint32_t* xbits, const V* vp,
                    U* y, uint32_t* ybits, int accumulate, int rk = REDUCE_ORDERED, const uint32_t* want = nullptr, bool grouped = false,
                    const uint32_t* xsum = nullptr) {
  if (use_vp) launch_spmv<P, T, U, V, E, true>(L, pa, A, x, xbits, vp, y, ybits, accumulate, rk, want, grouped, xsum);
  else launch_spmv<P, T, U, V, E, false>(L, pa, A, x, xbits, vp, y, ybits, accumulate, rk, want, grouped, xsum);
}

// ---- the iteration loop ---------------------------------------------------------------------------------------------
// One run of run_graph_program (GraphMatRuntime.h:93-279) on the device.  The state of a run lives in a Run object;
// what an iteration does is one of four SCHEDULES, each a member function that enqueues its kernels on the run's stream:
//   step_bits_push / step_list_push   top-down steps for small active sets (a=b and exact commutative programs)
//   step_pull                         send (or exchange) -> multiply (plain, column-tiled on two streams, or a
//                                     top-down bid pass over a larger active set) -> apply
//   run_two_stage                     the whole fixed-count loop of a sharded ALL_VERTICES run: tail rows, then head
//                                     rows, each multiplied / applied / sent while the other part's messages travel
// d_vp / d_active cover the shard's rows in device order; x / xbits are global-size scratch, y / ybits shard-size scratch.
template <class P, class T, class U, class V, class E>
class Run {
 public:
  Run(P* gp_, gm_graph_t* g_, edge_direction order_, activity_type act_, bool use_vp_, V* d_vp_, uint32_t* d_active_, T* x_, uint32_t* xbits_,
      U* y_, uint32_t* ybits_, int iterations_, hipStream_t s_)
      : gp(gp_), g(g_), order(order_), act(act_), use_vp(use_vp_), d_vp(d_vp_), d_active(d_active_), x(x_), xbits(xbits_), y(y_), ybits(ybits_),
        iterations(iterations_), s(s_), timer(false, s_) {}

  // Returns iterations done.
  int go() {
    setup();
    int it = -1;
    if (two_stage_applies()) it = run_two_stage();
    if (it < 0) {
      gm_sweep_t sw;
      if (swept_pipeline_applies(&sw)) it = run_swept_sharded(sw);
    }
    if (it < 0) it = run_loop();
    return it;
  }

 private:
  static constexpr bool kSparseT = std::is_trivially_copyable<T>::value && sizeof(T) <= 8;
  typedef dev::sparse_entry<typename std::conditional<kSparseT, T, int>::type> xentry_t;

  // ---- what the caller handed over -------------------------------------------------------------------------------
  P* gp;
  gm_graph_t* g;
  edge_direction order;
  activity_type act;
  bool use_vp;
  V* d_vp;
  uint32_t* d_active;
  T* x;
  uint32_t* xbits;
  U* y;
  uint32_t* ybits;
  int iterations;
  hipStream_t s;

  // ---- the run's fixed facts (setup) -------------------------------------------------------------------------------
  gm_engine_options_t opt;  // (gm_graph_engine_options: process defaults overlaid with the graph's own)
  bool verbose = false, trace = false;
  struct timeval tv0;
  gm_graph_desc_t desc;
  gm_csr_t Aout, Ain, Asrc;
  int n = 0, nwords = 0, n_live = 0;
  bool multi = false;
  int rk = REDUCE_ORDERED;
  bool rk_unverified = false;  // a probed strategy still to be cross-checked on the device (k_check_rows)
  bool guess_f32_add = false;  // rk is the ordered fold and reduce_function answers like a float addition: speculation for the giant rows only, proven chunk by chunk
  bool can_push = false, xsparse_ok = false, lazy_send = false;
  bool guided = false;         // an ORDERED fold over a sparse x on a large unsharded graph: levels with small active sets only fold the rows that have an active in-neighbour
  uint32_t* d_mark = nullptr;  // ... their bitmap (workspace slot 8)
  const int32_t *dev_of_native = nullptr, *native_of_dev = nullptr;
  xentry_t* d_gather = nullptr;
  unsigned long long* d_best = nullptr;
  int32_t* d_list = nullptr;
  int32_t* d_touched = nullptr;
  unsigned int* d_off = nullptr;
  uint32_t* d_want = nullptr;  // row-filter bits (program_row_filter), kept current by k_apply
  void* flag_v = nullptr;      // 4096 bytes: [2] list counter, [3] touched counter, [4..9] frontier stats, [127] changed flag, [128..] striped stats
  int* d_changed = nullptr;
  unsigned int *d_count = nullptr, *d_tcount = nullptr;
  unsigned long long *d_stats = nullptr, *d_striped = nullptr;
  static constexpr size_t striped_bytes = (size_t)dev::kStatSlots * 4 * sizeof(unsigned long long);
  int* h_changed = nullptr;  // pinned mirror of flag_v
  unsigned long long *h_stats = nullptr, *h_striped = nullptr;
  int stats_grid = 1;
  bool grouped_waves = true;

  // ---- what changes from iteration to iteration ------------------------------------------------------------------
  unsigned long long frontier_v = 0, frontier_e = 0, frontier_maxdeg = 0;
  int xs_max = 0, xs_total = 0;  // largest / total number of active vertices per shard (from the last GM_XCHG_STATE)
  bool list_ready = false;       // d_list holds exactly the current active set
  bool listed = false;           // the step just run wrote the list of its changed vertices
  bool x_presend = false;        // x already holds the next iteration's messages (fused apply + send)
  gm_run_stats_t st;
  PhaseTimer timer;
  AuxStream aux;
  struct timeval tr_iter, tr_last;
  long long tr_updated = -1;

  Launch launch_ctx() {
    Launch L{g, s, opt, &st.spmv_launches, &timer, &aux};
    L.guess_f32_add = guess_f32_add;
    L.spec_off = flag_v ? (int32_t*)flag_v + 720 : nullptr;
    return L;
  }
  static void die(const char* what) {
    printf("GraphMat(HIP): %s\n", what);
    exit(1);
  }
  void tick(const char* what, int k) {  // GRAPHMAT_VERBOSE=1: host-side timeline of one run
    if (!verbose) return;
    (void)hipStreamSynchronize(s);
    struct timeval tv1;
    gettimeofday(&tv1, 0);
    printf("GraphMat(HIP): +%9.3f ms  %s %d\n", (tv1.tv_sec - tv0.tv_sec) * 1e3 + (tv1.tv_usec - tv0.tv_usec) * 1e-3, what, k);
  }
  void lap(const char* label) {  // trace mode: host clock per phase (the reference's __TIMING lines)
    if (!trace) return;
    (void)hipStreamSynchronize(s);
    if (aux.s) (void)hipStreamSynchronize(aux.s);
    struct timeval now;
    gettimeofday(&now, 0);
    if (label) printf("%s = %.3f ms \n", label, (now.tv_sec - tr_last.tv_sec) * 1e3 + (now.tv_usec - tr_last.tv_usec) * 1e-3);
    tr_last = now;
  }
  long long count_bits(const uint32_t* bits, int nbits) {
    int64_t c = 0;
    if (bits == nullptr || nbits <= 0) return 0;
    if (gm_popcount_bits(bits, (int64_t)nbits, &c, (gm_stream_t)s) != GM_OK) return -1;
    return (long long)c;
  }
  void send_all(const dev::ProgArg<P>& pa, const uint32_t* active_bits, T* into) {  // x = send_message(vp) over the live rows
    hipLaunchKernelGGL((dev::k_send<P, T, V>), dim3(grid_for(n_live)), dim3(dev::kBlock), 0, s, pa, (const V*)d_vp, active_bits, into, xbits, n_live,
                       desc.row_lo);
  }
  void fill_active() {
    hipLaunchKernelGGL(dev::k_fill_u32, dim3(grid_for(nwords)), dim3(dev::kBlock), 0, s, d_active, (int64_t)nwords, 0xffffffffu);
  }
  void exchange_state(int* converged) {  // flag AND + active-set sizes over the shards
    int hf[4] = {*converged, frontier_v > 0x7fffffffull ? 0x7fffffff : (int)frontier_v, 0, 0};
    if (gm_graph_exchange(g, GM_XCHG_STATE, nullptr, 0, nullptr, hf) != 0) die("state exchange failed");
    *converged = hf[0];
    xs_max = hf[1];
    xs_total = hf[2];
  }
  // piece offsets of the listed sources (kernels.hpp: k_piece_*); the per-workgroup bases live behind them
  void piece_offsets(int nf) {
    const int nb = grid_for(nf);
    unsigned int* d_base = d_off + dev::kSparseListCap + 64;
    hipLaunchKernelGGL(dev::k_piece_count, dim3(nb), dim3(dev::kBlock), 0, s, Asrc, (const int32_t*)d_list, nf, d_base);
    hipLaunchKernelGGL(dev::k_piece_block_scan, dim3(1), dim3(dev::kBlock), 0, s, d_base, nb);
    hipLaunchKernelGGL(dev::k_piece_offsets, dim3(nb), dim3(dev::kBlock), 0, s, Asrc, (const int32_t*)d_list, nf, (const unsigned int*)d_base, d_off);
  }
  void list_active_set() {  // d_list = the current active set (when no step left it behind)
    if (list_ready) return;
    GM_HIP_OK(hipMemsetAsync(d_count, 0, 4, s));
    hipLaunchKernelGGL(dev::k_frontier_list, dim3(stats_grid), dim3(dev::kBlock), 0, s, (const uint32_t*)d_active, n, d_list, d_count);
  }

  // ---- set-up: adjacency views, strategy, scratch --------------------------------------------------------------------
  void setup() {
    verbose = getenv("GRAPHMAT_VERBOSE") != nullptr;
    gettimeofday(&tv0, 0);
    if (gm_graph_engine_options(g, &opt) != GM_OK) die(gm_last_error());
    trace = opt.iteration_trace != 0;
    gm_graph_desc(g, &desc);
    memset(&Aout, 0, sizeof(Aout));
    memset(&Ain, 0, sizeof(Ain));
    memset(&Asrc, 0, sizeof(Asrc));
    memset(&st, 0, sizeof(st));
    if (order != IN_EDGES && gm_graph_csr(g, GM_DIR_OUT, &Aout) != GM_OK)
      die("program needs OUT_EDGES adjacency (GM_DIR_OUT) which this graph was built without");
    if (order != OUT_EDGES && gm_graph_csr(g, GM_DIR_IN, &Ain) != GM_OK)
      die("program needs IN_EDGES adjacency (GM_DIR_IN) which this graph was built without");
    n = desc.row_hi - desc.row_lo;
    nwords = (n + 31) / 32;
    multi = gm_graph_has_exchange(g) != 0;
    gm_graph_set_run_stream(g, (gm_stream_t)s);  // a native (RCCL) exchange enqueues its collectives here
    n_live = (desc.xchg_rows > 0 && desc.xchg_rows < n && (desc.xchg_rows & 63) == 0) ? desc.xchg_rows : n;

    rk = reduce_kind_of<P, U>(gp);
    // a strategy the program did not declare but the probe inferred is cross-checked on the device against the
    // ordered fold the first time a pull multiply runs (k_check_rows); a disagreement falls back to the ordered fold
    rk_unverified = (int)program_traits<P>::reduce == (int)REDUCE_AUTO && rk != REDUCE_ORDERED && !getenv("GRAPHMAT_NO_PROBE_CHECK");
    if constexpr (std::is_same<U, float>::value && ((int)program_traits<P>::reduce == (int)REDUCE_AUTO || (int)program_traits<P>::reduce == (int)REDUCE_ORDERED)) {
      // (also for a program that DECLARES the ordered fold: the guess decides nothing about its bits, only which speculation its giant rows try)
      const char* off = getenv("GRAPHMAT_NO_PROBE");
      // (the host only calls the program's reduce_function with synthetic operands when the answer can be used at all: the adjacency has
      // giant rows and the speculation on them is enabled -- round-5 advice; GRAPHMAT_NO_PROBE=1 switches it off altogether)
      const bool has_giants = (order != IN_EDGES && Aout.ngiant > 0) || (order != OUT_EDGES && Ain.ngiant > 0);
      guess_f32_add = rk == REDUCE_ORDERED && has_giants && opt.ordered_giant_two_pass >= 2 && !(off && off[0] == '1') && probe_reduce_guess<P, U>(gp) == REDUCE_F32_ADD;
    }
    tick("reduce_function probed", rk);
    if (verbose) printf("GraphMat(HIP): reduce strategy %d (0 ordered, 1 commutative, 2 last, 3 float add)\n", rk);
    // Top-down steps for small active sets (kernels.hpp: k_push_*): REDUCE_LAST programs over OUT_EDGES, running until
    // convergence (the host already syncs once per iteration), unsharded, when the by-source adjacency is available;
    // REDUCE_COMMUTATIVE programs with a 4-byte reduction type take the list-based steps too, folding with
    // compare-and-swap (k_push_combine).  (A frontier-guided pull for the other reduction kinds -- mark the rows that
    // have an active in-neighbour, multiply only those -- was tried and dropped: on RMAT graphs an active set of any size
    // reaches most busy rows, and the extra passes plus the per-iteration statistics made SSSP 7.0 -> 13 ms on RMAT-22.)
    const bool comm_push = rk == REDUCE_COMMUTATIVE && sizeof(U) == 4 && std::is_trivially_copyable<U>::value;
    can_push = (rk == REDUCE_LAST || comm_push) && order == OUT_EDGES && act == ACTIVE_ONLY && iterations <= 0 && !multi &&
               !(opt.debug_flags & dev::DBG_NO_PUSH) && gm_graph_csr(g, GM_DIR_IN, &Asrc) == GM_OK && desc.row_lo == 0 && desc.row_hi == desc.ndevice;
    if (can_push) gm_graph_maps(g, &dev_of_native, &native_of_dev);
    // Sharded ACTIVE_ONLY programs running until convergence exchange a SMALL active set as (device id, message) lists
    // instead of all-gathering the whole dense x (graphmat_hip.h: GM_XCHG_STATE / GM_XCHG_GATHER; the reference
    // compresses sparse segments before sending them, DenseSegment.h:532-538,665-700)
    xsparse_ok = kSparseT && multi && act == ACTIVE_ONLY && iterations <= 0 && (gm_graph_exchange_caps(g) & GM_XCAP_SPARSE) &&
                 !(opt.debug_flags & dev::DBG_NO_SPARSE_XCHG);
    if (xsparse_ok && Asrc.rowptr == nullptr) (void)gm_graph_csr(g, GM_DIR_IN, &Asrc);  // out-degrees for the statistics, when available
    // a=b programs consume one message per row: unsharded, they evaluate it on demand from the sender's vertex property
    // (kernels.hpp: message_of) and the send pass disappears; the presence bits of x are the active bits themselves
    lazy_send = rk == REDUCE_LAST && sizeof(U) <= 8 && std::is_trivially_copyable<U>::value && !multi && act == ACTIVE_ONLY && order == OUT_EDGES &&
                desc.row_lo == 0 && desc.row_hi == desc.ndevice && !(opt.debug_flags & dev::DBG_NO_LAZY_SEND);
    if (verbose && lazy_send) printf("GraphMat(HIP): messages are evaluated on demand (no send pass)\n");
    // Guided pull (round 6).  A program whose reduce_function nothing is known about folds, in every iteration, every row's present
    // messages in stored order -- and finds them by testing the presence bit of every edge of the graph: 11 ms per BFS / SSSP level at
    // RMAT-26 however few vertices are active.  While the active set owns few out-edges the rows that can receive a message at all are
    // marked first (k_mark_rows_of_active: one atomic per out-edge of an active vertex) and only those are folded -- by the same kernels,
    // in the same order: the bits cannot change.  Large unsharded graphs only (below 2^27 edges the statistics' host round trip costs
    // more than the full scan: RMAT-22 SSSP 7.0 -> 13 ms when round 3 tried this on every graph).
    guided = !can_push && !multi && !lazy_send && act == ACTIVE_ONLY && iterations <= 0 && order == OUT_EDGES && rk != REDUCE_LAST &&
             !program_row_filter<P>::enabled && desc.row_lo == 0 && desc.row_hi == desc.ndevice && opt.guided_pull != 0 &&
             (Aout.nnz >= (1ll << 27) || opt.guided_pull >= 2) &&
             !(opt.debug_flags & dev::DBG_NO_PUSH) && gm_graph_csr(g, GM_DIR_IN, &Asrc) == GM_OK;
    if (guided) {
      void* pm = nullptr;
      if (gm_graph_workspace(g, 8, ((size_t)(n + 31) / 32 + 2) * 4, &pm) == GM_OK) d_mark = (uint32_t*)pm; else guided = false;
    }
    if (verbose && guided) printf("GraphMat(HIP): guided pull: iterations whose active set owns few out-edges only fold the rows it reaches\n");

    gm_graph_workspace(g, 0, 4096, &flag_v);
    // (the changed flag sits directly in front of the striped statistics, so that one memset clears and one copy fetches
    // "flag + statistics": every call the host makes between two short levels of a traversal shows up as idle time on the GPU)
    d_changed = (int*)flag_v + 127;
    GM_HIP_OK(hipMemsetAsync((int*)flag_v + 720, 0, 4, s));  // Launch::spec_off
    d_count = (unsigned int*)flag_v + 2;   // entries of d_list (the active set, when it is small)
    d_tcount = (unsigned int*)flag_v + 3;  // entries of d_touched (destinations bid for in a top-down step)
    d_stats = (unsigned long long*)flag_v + 2;    // byte offset 16
    d_striped = (unsigned long long*)flag_v + 64;  // byte offset 512: kStatSlots x 4 u64 (k_apply)
    void *res_stream = nullptr, *res_fork = nullptr, *res_join = nullptr, *res_pinned = nullptr;
    if (gm_graph_run_resources(g, &res_stream, &res_fork, &res_join, &res_pinned) != GM_OK) die(gm_last_error());
    h_changed = (int*)res_pinned + 127;  // (the pinned mirror has the layout of flag_v)
    h_stats = (unsigned long long*)res_pinned;
    h_striped = (unsigned long long*)res_pinned + 64;
    stats_grid = grid_for(n) < 2048 ? grid_for(n) : 2048;
    const bool xsparse_candidate = xsparse_ok;  // (the same on every shard: program, run mode and exchange capabilities)
    if (xsparse_ok) {
      void* pg = nullptr;
      if (gm_graph_workspace(g, GM_WS_GATHER, (size_t)desc.nshards * dev::kSparseListCap * sizeof(xentry_t) + 256, &pg) == GM_OK) d_gather = (xentry_t*)pg;
      else xsparse_ok = false;  // (an adopted buffer that is too small: dense exchanges only)
    }
    if (can_push || xsparse_ok || guided) {
      void *pb = nullptr, *pl = nullptr, *pt = nullptr;
      if (gm_graph_workspace(g, 7, (size_t)n * 4 + 1024 + ((size_t)dev::kSparseListCap + 64 + dev::kSparseListCap / dev::kBlock + 64) * 4, &pl) != GM_OK ||
          (can_push && (gm_graph_workspace(g, 6, (size_t)n * 8 + 64, &pb) != GM_OK || gm_graph_workspace(g, 10, (size_t)n * 4 + 1024, &pt) != GM_OK))) {
        can_push = false;
        xsparse_ok = false;
        guided = false;
      } else {
        d_best = (unsigned long long*)pb;
        d_list = (int32_t*)pl;
        d_off = (unsigned int*)((char*)pl + ((size_t)n * 4 + 1024) / 256 * 256);  // piece offsets of the listed sources
        d_touched = (int32_t*)pt;
        // (rows past n_live have no in-edge: nobody ever bids for them -- at RMAT-26 that halves a 537 MB memset)
        if (can_push) GM_HIP_OK(hipMemsetAsync(d_best, 0, (size_t)n_live * 8, s));
        GM_HIP_OK(hipMemsetAsync(d_stats, 0, 24, s));
        GM_HIP_OK(hipMemsetAsync(d_count, 0, 4, s));
        hipLaunchKernelGGL(dev::k_frontier_stats, dim3(stats_grid), dim3(dev::kBlock), 0, s, (const uint32_t*)d_active, Asrc.rowptr, n, d_stats, d_list,
                           d_count);
        GM_HIP_OK(hipMemcpyAsync(h_stats + 2, d_stats, 24, hipMemcpyDeviceToHost, s));
        GM_HIP_OK(hipStreamSynchronize(s));
        frontier_v = h_stats[2];
        frontier_e = h_stats[3];
        frontier_maxdeg = h_stats[4];
        list_ready = frontier_v <= (unsigned long long)dev::kSparseListCap;  // listed by the same pass
      }
    }
    if (xsparse_candidate) {
      // a shard that had to give up the sparse exchange (a workspace it could not get) takes the others with it: from here
      // on the shards must issue the same collectives (MIN over the shards of "still possible here")
      int still = xsparse_ok ? 1 : 0;
      gm_graph_exchange(g, GM_XCHG_CONVERGED, nullptr, 0, nullptr, &still);
      if (!still) xsparse_ok = false;
    }
    if (xsparse_ok) {
      int dummy = 0;
      exchange_state(&dummy);
    }
    if (act == ALL_VERTICES) fill_active();  // GraphMatRuntime.h:121-123 g.setAllActive()
    timer = PhaseTimer(gm_graph_timing_enabled(g) != 0, s);
    tick("first frontier counted", (int)frontier_v);
    if (!(opt.debug_flags & dev::DBG_NO_OVERLAP)) {
      aux.attach(res_stream, res_fork, res_join);
    }
    // row-filter bits (program_row_filter): one pass over the vertex properties now, kept current by k_apply
    if constexpr (program_row_filter<P>::enabled) {
      void* pw = nullptr;
      if (gm_graph_workspace(g, 8, ((size_t)(n + 31) / 32 + 2) * 4, &pw) == GM_OK) {
        d_want = (uint32_t*)pw;
        dev::ProgArg<P> pa0 = dev::make_prog_arg(gp);
        // (rows past n_live have no edges: no kernel ever looks at their bit)
        hipLaunchKernelGGL((dev::k_want_init<P, V>), dim3(grid_for(n_live)), dim3(dev::kBlock), 0, s, pa0, (const V*)d_vp, n_live, d_want);
      }
    }
    grouped_waves = !(opt.debug_flags & dev::DBG_NO_GROUPED);
  }

  void finish(int it) {
    GM_HIP_OK(hipStreamSynchronize(s));
    gm_graph_note_set(g, 3, (int64_t)sparse_sweeps);
    gm_graph_note_set(g, 4, (int64_t)short_folds);
    tick("loop done", it);
    aux.finish();
    st.iterations = it;
    timer.finish(&st);
    gm_graph_record_stats(g, &st);
    tick("teardown done", 0);
  }

  // ---- schedule: the overlapped two-stage loop of sharded fixed-count ALL_VERTICES / OUT_EDGES runs ------------------------
  // (graphmat_hip.h: GM_XCHG_PART).  Stage 1 multiplies, applies and sends the TAIL rows (most of the rows, a third of
  // the edges) and starts their exchange; stage 2 does the same for the HEAD rows (the busy ones) while stage 1's
  // messages travel.  Each row is still folded by one kernel in stored order, so results are those of the plain loop.
  // The stages send iteration i+1's messages before the hook of iteration i could run: only for programs that leave
  // do_every_iteration to the base class.
  bool two_stage_applies() const {
    if (!(multi && act == ALL_VERTICES && order == OUT_EDGES && iterations > 0 && !(opt.debug_flags & dev::DBG_NO_PIPELINE) && !trace && inherits_iteration_hook<P>()))
      return false;
    // a shard whose rows the sweep takes (gm_sweep_t.nsub): its device order is [slice][degree rank], the busy rows are no prefix of it
    // -- the plain loop with the swept multiply (round 6)
    if constexpr (sizeof(T) == 4 && sizeof(U) == 4 && std::is_trivially_copyable<T>::value && std::is_trivially_copyable<U>::value) {
      gm_sweep_t sw;
      // (decided by the LAYOUT -- sliced on every shard or on none -- not by whether this shard happens to hold swept rows: the shards
      // must choose the same schedule)
      if (!use_vp && gm_graph_sweep(g, &sw) == GM_OK && sw.nslices > 1 && desc.nshards > 1 && !(opt.debug_flags & (dev::DBG_NO_TILES | dev::DBG_NO_OVERLAP))) return false;
    }
    return true;
  }
  // returns the iterations done, or -1 when the shards could not agree on a split (the caller runs the plain loop)
  int run_two_stage() {
    void* x2v = nullptr;
    size_t x2_bytes = 0;
    int x2_ext = 0;
    int32_t rs = 0, bs = 0, ms = 0;
    bool staged = false;
    // the second message buffer: adopted from the caller (callback exchange: the collective library must know it), or
    // simply the library's own when the exchange is native
    bool have_x2 = gm_graph_workspace_info(g, 9, &x2v, &x2_bytes, &x2_ext) == GM_OK && x2_ext && x2v != nullptr && x2_bytes >= (size_t)desc.ndevice * sizeof(T);
    if (!have_x2 && gm_graph_exchange_is_native(g))
      have_x2 = gm_graph_workspace(g, 9, (size_t)desc.ndevice * sizeof(T) + 64, &x2v) == GM_OK && x2v != nullptr;
    if (have_x2) {
      // every shard must use the same split (the parts are the same rows of every slice): take the largest of the shards'
      // own choices, then check that it suits everybody
      int64_t agreed = 0;
      if (gm_graph_note_get(g, 0, &agreed) == GM_OK) {  // note 0: the split the shards agreed on in an earlier run (0 = none)
        rs = (int32_t)agreed;
        staged = rs > 0 && gm_graph_split(g, GM_DIR_OUT, opt.two_stage_head_permille, &rs, &bs, &ms) == GM_OK;
      } else {
        int mine = (gm_graph_split(g, GM_DIR_OUT, opt.two_stage_head_permille, &rs, &bs, &ms) == GM_OK) ? -rs : 0;
        gm_graph_exchange(g, GM_XCHG_CONVERGED, nullptr, 0, nullptr, &mine);  // MIN of -rs = -(largest rs)
        rs = -mine;
        int fine = (rs >= 64 && rs < n_live && gm_graph_split(g, GM_DIR_OUT, opt.two_stage_head_permille, &rs, &bs, &ms) == GM_OK) ? 1 : 0;
        gm_graph_exchange(g, GM_XCHG_CONVERGED, nullptr, 0, nullptr, &fine);
        staged = fine == 1;
        gm_graph_note_set(g, 0, staged ? (int64_t)rs : 0);
      }
    }
    if (!staged) return -1;
    if (verbose) printf("GraphMat(HIP): two-stage schedule, head rows [0,%d) tail rows [%d,%d)\n", rs, rs, n_live);
    T* xcur = x;
    T* xnext = (T*)x2v;
    gm_csr_t At = Aout, Ah = Aout;
    At.blk_seg += bs; At.nblk -= bs; At.mid_row += ms; At.nmid -= ms; At.ngiant = 0; At.ngchunk = 0;
    At.nmid_long = Aout.nmid_long > ms ? Aout.nmid_long - ms : 0;
    Ah.nblk = bs; Ah.nmid = ms;
    Ah.nmid_long = Aout.nmid_long < ms ? Aout.nmid_long : ms;
    // The giant rows belong to the head, but their serial chains are the longest thing in a shard's iteration (the hub row
    // does not shrink with the number of shards): they start on the auxiliary stream before the TAIL stage and are joined
    // only before the head rows are applied, so they overlap both stages' multiplies.
    gm_csr_t Ag = Aout;
    Ag.nblk = 0; Ag.nmid = 0; Ag.nmid_long = 0;
    const bool early_giants = Aout.ngiant > 0 && aux.s != nullptr && !(opt.debug_flags & dev::DBG_LATE_GIANTS);
    if (early_giants) { Ah.ngiant = 0; Ah.ngchunk = 0; }
    const Launch L = launch_ctx();
    auto multiply = [&](const dev::ProgArg<P>& pa, const gm_csr_t& A) {
      launch_spmv_vp<P, T, U, V, E>(use_vp, L, pa, A, (const T*)xcur, (const uint32_t*)nullptr, (const V*)d_vp, y, ybits, dev::ACC_STATIC_BITS, rk);
    };
    // per-stage times (send + exchange start, multiply, apply) come out of the phase timer; what the exchange adds on top
    // is visible as the difference between this schedule's wall clock and tools/shard_emulation.py's do-nothing exchange
    auto stage = [&](const dev::ProgArg<P>& pa, const gm_csr_t& A, int r0, int r1, bool more, bool join_giants) {
      multiply(pa, A);
      if (join_giants) aux.wait_join(s);
      const int cnt = r1 - r0;
      const int ag = grid_for(cnt) < dev::kApplyMaxBlocks ? grid_for(cnt) : dev::kApplyMaxBlocks;
      // apply, and -- when another iteration follows -- this part's messages of it out of the same pass (k_apply_send:
      // one launch and one read of the vertex properties less per stage; the program cannot change in between, see above)
      const bool fused = more && opt.fuse_apply_send != 0;
      if (fused)
        hipLaunchKernelGGL((dev::k_apply_send<P, T, U, V>), dim3(ag), dim3(dev::kBlock), 0, s, pa, (const U*)(y + r0), Aout.rowbits + r0 / 32, d_vp + r0,
                           d_active + r0 / 32, cnt, d_changed, (uint32_t*)nullptr, xnext, xbits, desc.row_lo + r0);
      else
        hipLaunchKernelGGL((dev::k_apply<P, U, V, false>), dim3(ag), dim3(dev::kBlock), 0, s, pa, (const U*)(y + r0), Aout.rowbits + r0 / 32, d_vp + r0,
                           d_active + r0 / 32, cnt, d_changed, (const int64_t*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr);
      timer.mark(TAG_APPLY);
      if (more) {
        if (!fused)
          hipLaunchKernelGGL((dev::k_send<P, T, V>), dim3(grid_for(cnt)), dim3(dev::kBlock), 0, s, pa, (const V*)(d_vp + r0), (const uint32_t*)nullptr, xnext,
                             xbits, cnt, desc.row_lo + r0);
        int part[2] = {r0, cnt};
        if (gm_graph_exchange(g, GM_XCHG_PART, xnext, (int64_t)sizeof(T), nullptr, part) != 0) die("partial message exchange failed");
        timer.mark(TAG_SEND);
      }
    };
    int it = 0;
    {
      dev::ProgArg<P> pa = dev::make_prog_arg(gp);
      timer.mark(TAG_START);
      send_all(pa, nullptr, xcur);
      if (gm_graph_exchange(g, GM_XCHG_MESSAGES, xcur, (int64_t)sizeof(T), xbits, nullptr) != 0) die("message exchange callback failed");
      timer.mark(TAG_SEND);
    }
    for (; it < iterations; it++) {
      dev::ProgArg<P> pa = dev::make_prog_arg(gp);
      const bool more = it + 1 < iterations;
      timer.mark(TAG_START);
      if (early_giants) {
        aux.defer = true;
        multiply(pa, Ag);
        aux.defer = false;
      }
      stage(pa, At, rs, n_live, more, false);
      stage(pa, Ah, 0, rs, more, early_giants);
      if (more && gm_graph_exchange(g, GM_XCHG_WAIT, xnext, (int64_t)sizeof(T), nullptr, nullptr) != 0) die("message exchange wait failed");
      if (n_live < n) GM_HIP_OK(hipMemsetAsync(d_active + n_live / 32, 0, (size_t)(nwords - n_live / 32) * 4, s));
      gp->do_every_iteration(it);  // (the base class's empty hook: this schedule is only taken by programs that do not override it)
      T* t = xcur; xcur = xnext; xnext = t;
    }
    fill_active();
    finish(it);
    return it;
  }

  // ---- schedule: top-down step straight from the active bitmap ---------------------------------------------------------
  // an active set too large to list whose vertices own only a few out-edges each (the late levels of a traversal:
  // 10^5..10^6 vertices of degree ~1): one lane per vertex bids for its destinations
  void step_bits_push(const dev::ProgArg<P>& pa) {
    send_all(pa, (const uint32_t*)d_active, x);
    timer.mark(TAG_SEND);
    lap("Send message time");
    GM_HIP_OK(hipMemsetAsync(d_tcount, 0, 4, s));
    const int bgrid = grid_for(n_live) < 4096 ? grid_for(n_live) : 4096;
    hipLaunchKernelGGL(dev::k_push_bid_bits, dim3(bgrid), dim3(dev::kBlock), 0, s, Asrc, (const uint32_t*)d_active, n_live, native_of_dev, d_best,
                       (const uint32_t*)d_want, d_touched, d_tcount);
    timer.mark(TAG_WAVE);
    lap("SPMV time");
    finish_push(pa, false);
  }
  // the bids of a top-down step are in d_best / d_touched: rewrite the active vector and the list, apply the winners
  void finish_push(const dev::ProgArg<P>& pa, bool combined) {
    if (trace) {  // vertices that received a message = destinations first touched by this step
      unsigned int tc = 0;
      GM_HIP_OK(hipMemcpy(&tc, d_tcount, 4, hipMemcpyDeviceToHost));
      tr_updated = (long long)tc;
    }
    GM_HIP_OK(hipMemsetAsync(d_active, 0, (size_t)nwords * 4, s));
    GM_HIP_OK(hipMemsetAsync(d_count, 0, 4, s));
    const unsigned long long bound = frontier_e < (unsigned long long)n ? frontier_e : (unsigned long long)n;
    if (bound > 0) {
      auto finish_k = [&](auto use_vp_c, auto combined_c) {
        hipLaunchKernelGGL((dev::k_push_finish<P, T, U, V, E, decltype(use_vp_c)::value, decltype(combined_c)::value>), dim3(grid_for((int64_t)bound)),
                           dim3(dev::kBlock), 0, s, pa, Asrc, (const T*)x, dev_of_native, d_vp, d_best, (const int32_t*)d_touched,
                           (const unsigned int*)d_tcount, d_active, d_changed, d_striped, d_want, d_list, d_count);
      };
      if (!combined) {
        if (use_vp) finish_k(std::true_type(), std::false_type()); else finish_k(std::false_type(), std::false_type());
      } else {
        if constexpr (sizeof(U) == 4 && std::is_trivially_copyable<U>::value) {
          if (use_vp) finish_k(std::true_type(), std::true_type()); else finish_k(std::false_type(), std::true_type());
        }
      }
    }
    st.spmv_launches += 2;
    listed = true;
    timer.mark(TAG_APPLY);
    lap("Apply time");
  }

  // ---- schedule: top-down step entirely on lists (few sources, few out-edges: nothing scans all vertices) ---------------------
  void step_list_push(const dev::ProgArg<P>& pa) {
    list_active_set();
    const int nf = (int)frontier_v;
    // The messages of the listed vertices are always materialised here, also for programs that otherwise evaluate them on
    // demand: k_push_finish applies while other lanes still fetch messages, so an on-demand send_message(vp[u]) could read
    // a vertex property that this very step is rewriting (any a=b program whose active vertices can change again).  The
    // list is small (<= kSparseListCap vertices), so this costs microseconds.
    hipLaunchKernelGGL((dev::k_send_list<P, T, V>), dim3(grid_for(nf)), dim3(dev::kBlock), 0, s, pa, (const V*)d_vp, (const int32_t*)d_list, nf, x,
                       desc.row_lo);
    timer.mark(TAG_SEND);
    lap("Send message time");
    GM_HIP_OK(hipMemsetAsync(d_tcount, 0, 4, s));
    const unsigned pieces = (unsigned)(frontier_e / dev::kPieceEdges + frontier_v);  // upper bound of the pieces
    piece_offsets(nf);
    if (rk == REDUCE_LAST) {
      hipLaunchKernelGGL(dev::k_push_bid, dim3(pieces > 0 ? pieces : 1), dim3(dev::kBlock), 0, s, Asrc, (const int32_t*)d_list, nf,
                         (const unsigned int*)d_off, native_of_dev, d_best, (const uint32_t*)d_want, d_touched, d_tcount);
    } else {
      if constexpr (sizeof(U) == 4 && std::is_trivially_copyable<U>::value) {
        if (use_vp)
          hipLaunchKernelGGL((dev::k_push_combine<P, T, U, V, E, true>), dim3(pieces > 0 ? pieces : 1), dim3(dev::kBlock), 0, s, pa, Asrc,
                             (const int32_t*)d_list, nf, (const unsigned int*)d_off, (const T*)x, (const V*)d_vp, d_best, (const uint32_t*)d_want, d_touched,
                             d_tcount);
        else
          hipLaunchKernelGGL((dev::k_push_combine<P, T, U, V, E, false>), dim3(pieces > 0 ? pieces : 1), dim3(dev::kBlock), 0, s, pa, Asrc,
                             (const int32_t*)d_list, nf, (const unsigned int*)d_off, (const T*)x, (const V*)d_vp, d_best, (const uint32_t*)d_want, d_touched,
                             d_tcount);
      }
    }
    timer.mark(TAG_WAVE);
    lap("SPMV time");
    finish_push(pa, rk != REDUCE_LAST);
  }

  // ---- schedule: the pull step (send -> multiply -> apply) ---------------------------------------------------------------
  // what this iteration's multiply reads messages from (a probed strategy that fails its cross-check changes them)
  const T* xq = nullptr;
  const uint32_t* xb = nullptr;

  // sparse exchange: messages of the listed active vertices only, as (device id, message) entries
  void send_sparse(const dev::ProgArg<P>& pa) {
    if constexpr (kSparseT) {
      list_active_set();
      const int nf = (int)frontier_v;
      const int cap = xs_max > 0 ? (xs_max + 63) / 64 * 64 : 64;
      GM_HIP_OK(hipMemsetAsync(xbits, 0, ((size_t)(desc.ndevice + 31) / 32) * 4, s));
      if (nf > 0)
        hipLaunchKernelGGL((dev::k_send_list<P, T, V>), dim3(grid_for(nf)), dim3(dev::kBlock), 0, s, pa, (const V*)d_vp, (const int32_t*)d_list, nf, x,
                           desc.row_lo);
      hipLaunchKernelGGL((dev::k_pack_frontier<T>), dim3(grid_for(cap)), dim3(dev::kBlock), 0, s, (const int32_t*)d_list, nf, (const T*)x, desc.row_lo,
                         d_gather + (size_t)desc.shard * cap, cap);
      int hf[2] = {cap, 0};
      if (gm_graph_exchange(g, GM_XCHG_GATHER, d_gather, (int64_t)sizeof(xentry_t), nullptr, hf) != 0) die("sparse message exchange failed");
      const int64_t nall = (int64_t)desc.nshards * cap;
      hipLaunchKernelGGL((dev::k_unpack_frontier<T>), dim3(grid_for(nall)), dim3(dev::kBlock), 0, s, (const xentry_t*)d_gather, nall, x, xbits);
      if (verbose)
        printf("GraphMat(HIP):   sparse exchange: %d active here, at most %d per shard, %zu bytes sent instead of %zu\n", nf, xs_max,
               (size_t)cap * sizeof(xentry_t), (size_t)n_live * sizeof(T) + (size_t)n_live / 8);
      st.sparse_exchanges++;
    }
  }

  // Cross-check of a probed strategy after the first pass of a pull multiply over adjacency A whose results (presence bits
  // yb) are in y: giant rows, the first wave rows and a strided sample of all rows are folded again in order.  On a
  // mismatch the pass is redone with the ordered fold, which also governs the rest of the run.
  void check_probed(const dev::ProgArg<P>& pa, const gm_csr_t& A, const uint32_t* yb, const uint32_t* rowfilter, int acc_flags, uint32_t* yb_write,
                    bool dense_x) {
    if (!rk_unverified || (acc_flags & dev::ACC_READ_PREV)) return;
    rk_unverified = false;
    unsigned int* d_mis = (unsigned int*)flag_v + 700;
    GM_HIP_OK(hipMemsetAsync(d_mis, 0, 4, s));
    const T* xc = xq;
    const uint32_t* xbc = xb;
    if (lazy_send) {  // the ordered fold reads materialised messages: write them once for the check
      send_all(pa, dense_x ? (const uint32_t*)nullptr : (const uint32_t*)d_active, x);
      xc = x;
      xbc = dense_x ? nullptr : (const uint32_t*)xbits;
    }
    auto sample = [&](const int32_t* rows, int cnt, int stride) {
      if (cnt <= 0) return;
      const int grid = (cnt + dev::kBlock / 64 - 1) / (dev::kBlock / 64);
      if constexpr ((int)program_traits<P>::reduce != (int)REDUCE_AUTO) return;  // (declared strategies are never checked)
      else if (use_vp)
        hipLaunchKernelGGL((dev::k_check_rows<P, T, U, V, E, true>), dim3(grid), dim3(dev::kBlock), 0, s, pa, A, rows, cnt, stride, xc, xbc, (const V*)d_vp,
                           (const U*)y, yb, rowfilter, d_mis);
      else
        hipLaunchKernelGGL((dev::k_check_rows<P, T, U, V, E, false>), dim3(grid), dim3(dev::kBlock), 0, s, pa, A, rows, cnt, stride, xc, xbc, (const V*)d_vp,
                           (const U*)y, yb, rowfilter, d_mis);
    };
    sample(A.giant_row, A.ngiant, 0);
    sample(A.mid_row, A.nmid < 2048 ? A.nmid : 2048, 0);
    const int live_rows = n_live < A.nrows ? n_live : A.nrows;
    const int stride = live_rows > 2048 ? live_rows / 2048 : 1;
    sample(nullptr, live_rows / stride, stride);
    unsigned int mis = 0;
    GM_HIP_OK(hipMemcpyAsync(&mis, d_mis, 4, hipMemcpyDeviceToHost, s));
    GM_HIP_OK(hipStreamSynchronize(s));
    if (verbose) printf("GraphMat(HIP):   probed reduce strategy %d cross-checked against the ordered fold: %u mismatching rows\n", rk, mis);
    if (mis == 0) return;
    printf("GraphMat(HIP): warning: reduce_function matched strategy %d on the probe's operands but not on this run's data (%u sampled "
           "rows differ from the ordered fold); using the ordered fold.  Declare GraphMat::program_traits<YourProgram>::reduce to choose explicitly.\n", rk, mis);
    rk = REDUCE_ORDERED;
    can_push = false;
    if (lazy_send) {  // the ordered kernels read materialised messages (written above for the check)
      lazy_send = false;
      xq = x;
      xb = dense_x ? nullptr : (const uint32_t*)xbits;
    }
    if (!(acc_flags & dev::ACC_STATIC_BITS)) GM_HIP_OK(hipMemsetAsync(yb_write, 0, (size_t)nwords * 4, s));
    launch_spmv_vp<P, T, U, V, E>(use_vp, launch_ctx(), pa, A, xq, xb, (const V*)d_vp, y, yb_write, acc_flags, rk, rowfilter);
  }

  // Can this run's pull multiply of the OUT adjacency take the row-stationary sweep (graphmat_hip.h: gm_sweep_t; kernels.hpp:
  // k_spmv_sell)?  Every x entry present, 2-operand program, 4-byte messages and reductions, edge values absent or 4 bytes and
  // carried by the structure, no row filter, nothing accumulated from an earlier pass; the structure must hold exactly the rows
  // the whole-graph CSR's short-row pass and giant passes leave out.
  bool sweep_usable(int acc, gm_sweep_t* sw) {
    if constexpr (sizeof(T) == 4 && sizeof(U) == 4 && std::is_trivially_copyable<T>::value && std::is_trivially_copyable<U>::value) {
      if (aux.s == nullptr || use_vp || xq == nullptr || d_want != nullptr || program_row_filter<P>::enabled || (acc & dev::ACC_READ_PREV)) return false;
      if (!(rk == REDUCE_ORDERED || rk == REDUCE_F32_ADD || rk == REDUCE_COMMUTATIVE)) return false;
      if (opt.debug_flags & (dev::DBG_NO_TILES | dev::DBG_NO_OVERLAP)) return false;
      if (gm_graph_sweep(g, sw) != GM_OK || sw->nrows <= 0 || sw->acc_rows != GM_SWEEP_ACC_ROWS || sw->long_slots != GM_SWEEP_LONG_SLOTS) return false;
      if (sw->short_row != Aout.short_row) return false;
      // a shard's rows (gm_sweep_t.nsub): the structure must describe THIS cut of the message vector; a single-shard structure is not
      // used by a run that exchanges messages (a world of one rank: nothing to gain)
      if (sw->nsub > 1 ? (sw->nsub != desc.nshards || sw->stride != n || sw->hot_words <= 0) : multi) return false;
      if (sw->waves != 0 && sw->waves != 16 && !(sw->waves == 12 && sw->nsub <= 1)) return false;  // (a block's groups dealt over a wave count no kernel here has)
      // a SPARSE message vector (ACTIVE_ONLY programs; round 6): k_spmv_sell_sparse -- single-shard structures, 16 waves, not under the static presence bits of a dense x
      // (large graphs only -- the sweep walks all its slices however few columns are present: unchanged SSSP.cpp RMAT-22 6.5 -> 12.3 ms and
      // RMAT-23 11.5 -> 14.4 ms when it was taken there, 21.6 -> 20.0 ms at RMAT-24, 84 -> 60 ms at RMAT-26; sweep_form bit 6 lifts the size
      // limit for tests, bit 5 refuses the form)
      if (xb != nullptr && (sw->nsub > 1 || (sw->waves != 0 && sw->waves != 16) || (acc & dev::ACC_STATIC_BITS) || (opt.sweep_form & 32) ||
                            (Aout.nnz < 200000000ll && !(opt.sweep_form & 64)))) return false;
      if (Aout.vals != nullptr && !(sw->val_bytes == 4 && sizeof(E) == 4 && std::is_trivially_copyable<E>::value)) return false;
      return true;
    } else {
      (void)acc; (void)sw;
      return false;
    }
  }

  // The OUT adjacency through the sweep: the giant rows' passes (products spread over the chip, then the exact replay or the
  // ordered fold: a chain of small latency-bound launches) on the auxiliary stream from the moment x is complete, the short
  // rows (row-blocks of the whole-graph CSR) and the sweep's launches on the main stream; joined before apply.  sweep_form:
  // 0 = short rows in front of the sweep, 1 = on the auxiliary stream behind the giant passes (next to the sweep), 2 = behind
  // the sweep.  The three row sets are disjoint, so no two kernels touch the same y entry.
  // defer_join: return without waiting for the auxiliary stream (the giant rows' fold passes): the caller joins (aux.wait_join) before it
  // touches the giant rows' y entries -- the sharded swept schedule applies, sends and starts exchanging all other rows meanwhile
  void multiply_out_swept(const dev::ProgArg<P>& pa, int acc, const gm_sweep_t& sw, bool defer_join = false) {
    if constexpr (sizeof(T) == 4 && sizeof(U) == 4 && std::is_trivially_copyable<T>::value && std::is_trivially_copyable<U>::value) {
      const Launch L = launch_ctx();
      aux.keep = true;
      aux.pending = false;
      // Giant rows.  Their fold passes read a products stream (gm_csr_t.gterm_off); when the reduction has such a two-pass form
      // (float sums: the exact replay; plain ordered folds) the SWEEP gathers for them -- slice by slice, with its LDS hot sets,
      // a slice's giant edges dealt evenly over the workgroups -- and only the fold passes remain, on the auxiliary stream next to
      // the short rows' pass BEHIND the sweep.  (Round 5, first form: k_giant_terms gathered the 36 M giant-row messages of
      // RMAT-26 untiled -- 64 % L2 misses -- on the auxiliary stream next to the short rows' kernel, which it stretched from 1.27
      // to 1.55 ms.)  sweep_form bit 3 keeps that first form; other reductions (commutative) gather in their own kernels as before.
      U* gterms = nullptr;
      if constexpr (dev::stageable<U>::value) {
        // (float sums only.  A plain ordered fold of a giant row is a chain of dependent reduce_function calls -- 4-5 ms for RMAT-26's
        // hub row -- that has to start as early as possible: such programs keep the first form, k_giant_terms and the fold on the
        // auxiliary stream from the moment x is complete, next to the short rows AND the sweep; unchanged PageRank.cpp, RMAT-26:
        // 459 ms with its chain behind the sweep, 444 with it beside everything)
        const bool two_pass = rk == REDUCE_F32_ADD && std::is_same<U, float>::value;
        void* p6 = nullptr;
        if (two_pass && xb == nullptr && Aout.ngiant > 0 && sw.ngiant_edges > 0 && sw.gcol != nullptr && !(opt.sweep_form & 8) &&
            gm_graph_workspace(g, 6, (size_t)Aout.giant_edges * sizeof(U) + 64, &p6) == GM_OK)
          gterms = (U*)p6;
      }
      GM_HIP_OK(hipEventRecord(aux.fork, s));
      GM_HIP_OK(hipStreamWaitEvent(aux.s, aux.fork, 0));
      aux.forked = true;
      gm_csr_t Ag = Aout;
      Ag.nblk = 0; Ag.nmid = 0; Ag.nmid_long = 0;
      if (Aout.ngiant > 0 && gterms == nullptr) launch_spmv_vp<P, T, U, V, E>(use_vp, L, pa, Ag, xq, xb, (const V*)d_vp, y, ybits, acc, rk);
      gm_csr_t As = Aout;  // the short rows
      As.nmid = 0; As.nmid_long = 0; As.ngiant = 0; As.ngchunk = 0;
      Launch La = L;
      La.aux = nullptr;
      La.tiled_untiled_pass = true;
      // (a graph with few busy rows and a column-blocked stream of its short rows, graphmat_hip.h: gm_blocked_t: the stream instead of the
      // row-blocks, on the main stream -- it wants the whole chip, like the sweep)
      gm_blocked_t bl;
      const bool shorts_blocked = blocked_usable(acc, &bl);
      // The 768-thread form (gm_sweep_t.waves = 12; experiment of round 6): the sweep leaves every CU a quarter of its registers and 36 KB of LDS,
      // and EVERYTHING else runs beside it on the auxiliary stream from the moment x is complete: the giant rows' gathers in slice order
      // (k_giant_gather_sliced), their fold passes, then the short rows' kernel.
      const bool w12 = sw.waves == 12 && sw.nsub <= 1 && !shorts_blocked && !defer_join && xb == nullptr;
      // (the giant rows' gathers in a kernel of their own instead of inside the sweep: sweep_form bit 4, or the 768-thread form)
      const bool gather_apart = gterms != nullptr && sw.nsub <= 1 && ((opt.sweep_form & 16) != 0 || w12);
      auto giant_gather = [&]() {
        bool done = false;
        if constexpr (sizeof(E) == 4 && std::is_trivially_copyable<E>::value) {
          if (Aout.vals != nullptr && sw.gval != nullptr) {
            hipLaunchKernelGGL((dev::k_giant_gather_sliced<P, T, U, V, E, true>), dim3(2048), dim3(dev::kBlock), 0, aux.s, pa, sw.gcol, sw.gval, sw.gdst, (int64_t)sw.ngiant_edges, xq, gterms);
            done = true;
          }
        }
        if (!done)
          hipLaunchKernelGGL((dev::k_giant_gather_sliced<P, T, U, V, E, false>), dim3(2048), dim3(dev::kBlock), 0, aux.s, pa, sw.gcol, (const uint32_t*)nullptr, sw.gdst, (int64_t)sw.ngiant_edges, xq, gterms);
        st.spmv_launches++;
      };
      bool chain_done = false;
      if (w12 && gterms != nullptr) {  // gathers and fold passes of the giant rows right away, beside the sweep
        giant_gather();
        Launch Lg = L;
        Lg.terms_ready = true;
        launch_spmv_vp<P, T, U, V, E>(use_vp, Lg, pa, Ag, xq, xb, (const V*)d_vp, y, ybits, acc, rk);
        chain_done = true;
      }
      // The short rows through the sweep (gm_sweep_t.nstream; round 6, last session): their edges sit in STREAM groups of the first launch's
      // blocks, the sweep gathers for them -- from ~1.3 MB of x at a time and its LDS hot sets, where the row-block kernel gathers from the
      // whole message vector -- and leaves the products in a stream that k_short_fold folds bin by bin behind the sweep.  Dense x, 16 waves
      // (a shard's structure too); sweep_form bit 7 keeps the row-block kernel (every other form walks wrow, which leaves the groups out).
      // From 2^25 short-row edges on (bit 8: on any structure -- tests): below, the blocks' few stream rows cost the sweep more than the row-block
      // kernel saves -- RMAT-23 (27 M) 0.595 -> 0.610 ms, a shard of 8 of RMAT-26 (20 M) 0.82 -> 0.89; RMAT-24 (53 M) 1.049 -> 0.946, a shard of 4 (40 M) 1.28 -> 1.25,
      // of 2 (81 M) 2.25 -> 1.92, RMAT-25 (107 M) 2.08 -> 1.76, RMAT-26 3.78 -> 3.28, RMAT-27 9.35 -> 7.34 (with the stream rows weighted 3 : 1 in the waves' shares).
      U* sterms = nullptr;
      if (sw.nstream > 0 && sw.sinv != nullptr && sw.wrow_stream != nullptr && xb == nullptr && !w12 && !shorts_blocked && !(opt.sweep_form & 128) &&
          sw.bin_cap == GM_STREAM_BIN && (sw.nstream >= (1ll << 25) || (opt.sweep_form & 256))) {
        void* p16 = nullptr;
        if (gm_graph_workspace(g, 16, (size_t)sw.nstream_slots * sizeof(U) + 256, &p16) == GM_OK) sterms = (U*)p16;
      }
      const int where = sterms != nullptr ? 4 : shorts_blocked ? 3 : (w12 ? 1 : (gterms != nullptr ? 2 : (opt.sweep_form & 3)));  // (3: behind everything, below; 4: folded from the sweep's products)
      if (where == 1) { La.s = aux.s; La.timer = nullptr; }
      auto short_rows = [&]() {
        if (As.nblk <= 0) return;
        launch_spmv_vp<P, T, U, V, E>(use_vp, La, pa, As, xq, xb, (const V*)d_vp, y, ybits, acc, rk);
        if (where == 1) { GM_HIP_OK(hipEventRecord(aux.join, aux.s)); aux.pending = true; }
      };
      if (where < 2) short_rows();
      // the pool of LDS words is split between the slice's hot entries and the long rows' stage: a stage that takes the
      // largest block in one round where that leaves most of the pool to the hot set
      int stage = 64;
      if (sw.nrows_long > 0) {
        stage = (sw.max_long_block + 63) / 64 * 64;
        if (stage > GM_SWEEP_MAX_STAGE) stage = GM_SWEEP_MAX_STAGE;
        if (stage < 1024) stage = 1024;
        if (opt.sweep_form & 4) stage = 1024;  // (tests: blocks staged in several rounds)
      }
      if (w12 && stage > GM_SWEEP_MAX_STAGE_W12) stage = GM_SWEEP_MAX_STAGE_W12;
      for (int set = 0; set < sw.nsets; set++) {
        U* gt = (set == 0 && !gather_apart) ? gterms : (U*)nullptr;  // (the first launch gathers for the giant rows)
        bool with_vals = false;
        // (the launch that holds the short rows' stream groups walks the wave ranges that include them and stores their products)
        const uint32_t* st_rows = (sterms != nullptr && set == 0) ? sw.wrow_stream : sw.wrow;
        U* st_terms = (sterms != nullptr && set == 0) ? sterms : (U*)nullptr;
        if (sw.nsub > 1) {  // a shard's rows: the message vector is made of nsub owners' ranges (kernels.hpp: k_spmv_sell_sharded)
          if constexpr (sizeof(E) == 4 && std::is_trivially_copyable<E>::value) {
            if (Aout.vals != nullptr) {
              hipLaunchKernelGGL((dev::k_spmv_sell_sharded<P, T, U, V, E, true>), dim3(256), dim3(1024), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base,
                                 sw.scol, sw.sval, st_rows, sw.row_of_slot, sw.lcol, sw.lval, sw.lps, sw.lrow_of_slot, sw.gcol, sw.gval, sw.gdst, sw.gslice, gt, xq, y,
                                 sw.nsub, sw.stride, sw.hot_words, st_terms);
              with_vals = true;
            }
          }
          if (!with_vals)
            hipLaunchKernelGGL((dev::k_spmv_sell_sharded<P, T, U, V, E, false>), dim3(256), dim3(1024), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base,
                               sw.scol, (const uint32_t*)nullptr, st_rows, sw.row_of_slot, sw.lcol, (const uint32_t*)nullptr, sw.lps, sw.lrow_of_slot, sw.gcol,
                               (const uint32_t*)nullptr, sw.gdst, sw.gslice, gt, xq, y, sw.nsub, sw.stride, sw.hot_words, st_terms);
          continue;
        }
        if (xb != nullptr) {  // a sparse message vector: presence tests per entry, y's presence bits OR-ed in (kernels.hpp: k_spmv_sell_sparse)
          if (verbose && !said_sparse_sweep) { printf("GraphMat(HIP):   sparse message vector through the sweep (k_spmv_sell_sparse)\n"); said_sparse_sweep = true; }
          if (set == 0) sparse_sweeps++;
          if constexpr (sizeof(E) == 4 && std::is_trivially_copyable<E>::value) {
            if (Aout.vals != nullptr) {
              hipLaunchKernelGGL((dev::k_spmv_sell_sparse<P, T, U, V, E, true>), dim3(256), dim3(1024), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base, sw.scol,
                                 sw.sval, sw.wrow, sw.row_of_slot, sw.lcol, sw.lval, sw.lps, sw.lrow_of_slot, xq, xb, y, ybits);
              with_vals = true;
            }
          }
          if (!with_vals)
            hipLaunchKernelGGL((dev::k_spmv_sell_sparse<P, T, U, V, E, false>), dim3(256), dim3(1024), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base, sw.scol,
                               (const uint32_t*)nullptr, sw.wrow, sw.row_of_slot, sw.lcol, (const uint32_t*)nullptr, sw.lps, sw.lrow_of_slot, xq, xb, y, ybits);
          continue;
        }
        if (w12) {  // 768-thread workgroups (kernels.hpp: k_spmv_sell_w12)
          if constexpr (sizeof(E) == 4 && std::is_trivially_copyable<E>::value) {
            if (Aout.vals != nullptr) {
              hipLaunchKernelGGL((dev::k_spmv_sell_w12<P, T, U, V, E, true>), dim3(256), dim3(768), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base, sw.scol,
                                 sw.sval, sw.wrow, sw.row_of_slot, sw.lcol, sw.lval, sw.lps, sw.lrow_of_slot, sw.gcol, sw.gval, sw.gdst, sw.gslice, gt, xq, y);
              with_vals = true;
            }
          }
          if (!with_vals)
            hipLaunchKernelGGL((dev::k_spmv_sell_w12<P, T, U, V, E, false>), dim3(256), dim3(768), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base, sw.scol,
                               (const uint32_t*)nullptr, sw.wrow, sw.row_of_slot, sw.lcol, (const uint32_t*)nullptr, sw.lps, sw.lrow_of_slot, sw.gcol,
                               (const uint32_t*)nullptr, sw.gdst, sw.gslice, gt, xq, y);
          continue;
        }
        if (sterms != nullptr && set == 0) {  // the launch that holds the short rows' stream groups stores their products
          if constexpr (sizeof(E) == 4 && std::is_trivially_copyable<E>::value) {
            if (Aout.vals != nullptr) {
              hipLaunchKernelGGL((dev::k_spmv_sell_stream<P, T, U, V, E, true>), dim3(256), dim3(1024), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base, sw.scol,
                                 sw.sval, sw.wrow_stream, sw.row_of_slot, sw.lcol, sw.lval, sw.lps, sw.lrow_of_slot, sw.gcol, sw.gval, sw.gdst, sw.gslice, gt, xq, y, sterms);
              with_vals = true;
            }
          }
          if (!with_vals)
            hipLaunchKernelGGL((dev::k_spmv_sell_stream<P, T, U, V, E, false>), dim3(256), dim3(1024), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base, sw.scol,
                               (const uint32_t*)nullptr, sw.wrow_stream, sw.row_of_slot, sw.lcol, (const uint32_t*)nullptr, sw.lps, sw.lrow_of_slot, sw.gcol,
                               (const uint32_t*)nullptr, sw.gdst, sw.gslice, gt, xq, y, sterms);
          continue;
        }
        if constexpr (sizeof(E) == 4 && std::is_trivially_copyable<E>::value) {
          if (Aout.vals != nullptr) {
            hipLaunchKernelGGL((dev::k_spmv_sell<P, T, U, V, E, true>), dim3(256), dim3(1024), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base, sw.scol,
                               sw.sval, sw.wrow, sw.row_of_slot, sw.lcol, sw.lval, sw.lps, sw.lrow_of_slot, sw.gcol, sw.gval, sw.gdst, sw.gslice, gt, xq, y);
            with_vals = true;
          }
        }
        if (!with_vals)
          hipLaunchKernelGGL((dev::k_spmv_sell<P, T, U, V, E, false>), dim3(256), dim3(1024), 0, s, pa, set, stage, sw.nslices, sw.nrows_long, sw.slice_base, sw.scol,
                             (const uint32_t*)nullptr, sw.wrow, sw.row_of_slot, sw.lcol, (const uint32_t*)nullptr, sw.lps, sw.lrow_of_slot, sw.gcol,
                             (const uint32_t*)nullptr, sw.gdst, sw.gslice, gt, xq, y);
      }
      st.spmv_launches += sw.nsets;
      timer.mark(TAG_WAVE);
      if (gterms != nullptr && !chain_done) {  // the giant rows' fold passes behind the sweep, on the auxiliary stream next to the short rows
        GM_HIP_OK(hipEventRecord(aux.fork, s));
        GM_HIP_OK(hipStreamWaitEvent(aux.s, aux.fork, 0));
        if (gather_apart) giant_gather();
        Launch Lg = L;
        Lg.terms_ready = true;
        launch_spmv_vp<P, T, U, V, E>(use_vp, Lg, pa, Ag, xq, xb, (const V*)d_vp, y, ybits, acc, rk);
      }
      if (where == 2) short_rows();
      if (where == 4) {  // the short rows' fold from the products the sweep left (next to the giant rows' fold passes on the auxiliary stream)
        hipLaunchKernelGGL((dev::k_short_fold<P, U>), dim3((unsigned)sw.nbins), dim3(512), 0, s, pa, (const U*)sterms, sw.sinv, sw.schunk, sw.nslices, sw.bin_cap, sw.sbin_row,
                           sw.soff, sw.srow, y, (acc & dev::ACC_STATIC_BITS) ? (uint32_t*)nullptr : ybits);
        st.spmv_launches++;
        short_folds++;
        if (verbose && !said_short_fold) { printf("GraphMat(HIP):   the rows of up to %d edges ride the sweep (k_short_fold)\n", (int)sw.short_row); said_short_fold = true; }
        timer.mark(TAG_ROWBLOCK);  // (the short rows' share of the multiply that is not inside the sweep)
      }
      if (aux.pending && !defer_join) GM_HIP_OK(hipStreamWaitEvent(s, aux.join, 0));
      aux.keep = aux.forked = aux.pending = false;
      timer.mark(defer_join ? TAG_ROWBLOCK : TAG_GIANT);  // (the multiply ends when the auxiliary stream has joined: the wait is charged to the giant rows' passes)
      if (shorts_blocked) {  // (with the chip to itself: its workgroups walk the slices in step only while all of them are resident)
        launch_blocked(pa, bl);
        st.spmv_launches += 1;
        timer.mark(TAG_ROWBLOCK);
      }
      if (!defer_join) check_probed(pa, Aout, xb != nullptr ? (const uint32_t*)ybits : Aout.rowbits, nullptr, acc, ybits, xb == nullptr);
    } else {
      (void)pa; (void)acc; (void)sw; (void)defer_join;
    }
  }

  // ---- schedule: the fixed-count loop of a SHARDED ALL_VERTICES run whose rows the sweep takes (round 6) ------------------------------
  // A shard's iteration is sweep -> [short rows | the giant rows' fold passes on the auxiliary stream] -> apply -> send -> all-gather of x.
  // The giant rows' passes are a serial chain that does not shrink with the number of shards (the hub row: ~0.5 ms on a shard of 8 of
  // RMAT-26) during which most of the chip idles -- and they only hold up the apply of the giant rows themselves, a few hundred per shard.
  // So: every OTHER row is applied and sends its next message as soon as the short rows are done, the all-gather of the whole live range
  // starts on the exchange's side stream (GM_XCHG_PART into the second message buffer) and travels WHILE the giant rows fold; when they
  // have joined, a list kernel applies them, sends their messages and packs them as (device id, message) entries, which one small
  // all-gather of equal blocks (GM_XCHG_GATHER: the sparse exchange's mechanism) carries to every shard; GM_XCHG_WAIT puts the parts in
  // place and the entries are scattered over them.  Every row is folded by the same kernels in the same order as in the plain loop: same
  // bits.  Needs what the two-stage schedule needs (a second message buffer; a program that leaves do_every_iteration to the base class)
  // plus the sparse exchange; DBG_NO_PIPELINE (128) keeps the plain loop.
  bool swept_pipeline_applies(gm_sweep_t* sw) {
    if constexpr (kSparseT && sizeof(T) == 4 && sizeof(U) == 4 && std::is_trivially_copyable<U>::value) {
      if (!(multi && act == ALL_VERTICES && order == OUT_EDGES && iterations > 1 && !(opt.debug_flags & dev::DBG_NO_PIPELINE) && !trace && inherits_iteration_hook<P>())) return false;
      if (!(gm_graph_exchange_caps(g) & GM_XCAP_SPARSE) || rk_unverified || opt.fuse_apply_send == 0) return false;
      xq = x;
      xb = nullptr;
      // (everything above is the same on every shard; what follows is not -- a shard may hold no swept row -- and the shards must take the
      // same schedule, or their collectives do not match: MIN over the shards of "fine here")
      int fine = (sweep_usable(dev::ACC_STATIC_BITS, sw) && sw->nsub > 1 && Aout.ngiant <= dev::kSparseListCap && n_live >= 64) ? 1 : 0;
      gm_graph_exchange(g, GM_XCHG_CONVERGED, nullptr, 0, nullptr, &fine);
      return fine == 1;
    } else {
      (void)sw;
      return false;
    }
  }
  // returns the iterations done, or -1 when the schedule cannot run here (the caller runs the plain loop)
  int run_swept_sharded(const gm_sweep_t& sw) {
    if constexpr (kSparseT && sizeof(T) == 4 && sizeof(U) == 4 && std::is_trivially_copyable<U>::value) {
      void* x2v = nullptr;
      size_t x2_bytes = 0;
      int x2_ext = 0;
      bool have_x2 = gm_graph_workspace_info(g, 9, &x2v, &x2_bytes, &x2_ext) == GM_OK && x2_ext && x2v != nullptr && x2_bytes >= (size_t)desc.ndevice * sizeof(T);
      if (!have_x2 && gm_graph_exchange_is_native(g)) have_x2 = gm_graph_workspace(g, 9, (size_t)desc.ndevice * sizeof(T) + 64, &x2v) == GM_OK && x2v != nullptr;
      // the shards' blocks of the gather buffer are equally long: the largest giant-row count, agreed once per graph (note 2; 0 = the
      // schedule cannot run: no shard has a giant row, or one has no second message buffer)
      int cap = 0;
      int64_t agreed = 0;
      if (gm_graph_note_get(g, 2, &agreed) == GM_OK) {
        cap = (int)agreed;
      } else {
        int mine = have_x2 ? -Aout.ngiant : -(1 << 30);  // (MIN over the shards of -count = -(largest count); a shard without the buffer makes it too large)
        gm_graph_exchange(g, GM_XCHG_CONVERGED, nullptr, 0, nullptr, &mine);
        cap = (-mine >= (1 << 30)) ? 0 : (-mine + 63) / 64 * 64;
        gm_graph_note_set(g, 2, (int64_t)cap);
      }
      if (cap <= 0 || cap > dev::kSparseListCap) return -1;
      void *pg = nullptr, *pm = nullptr;
      if (gm_graph_workspace(g, GM_WS_GATHER, (size_t)desc.nshards * (size_t)cap * sizeof(xentry_t) + 256, &pg) != GM_OK) pg = nullptr;
      const size_t mwords = (size_t)nwords + 2;
      if (gm_graph_workspace(g, 8, 2 * mwords * 4, &pm) != GM_OK) pm = nullptr;
      {  // (a shard that could not get its buffers takes the others with it: they must issue the same collectives)
        int fine = (pg != nullptr && pm != nullptr) ? 1 : 0;
        gm_graph_exchange(g, GM_XCHG_CONVERGED, nullptr, 0, nullptr, &fine);
        if (!fine) return -1;
      }
      xentry_t* gather = (xentry_t*)pg;
      uint32_t *bits_rest = (uint32_t*)pm, *bits_giant = (uint32_t*)pm + mwords;
      GM_HIP_OK(hipMemcpyAsync(bits_rest, Aout.rowbits, (size_t)nwords * 4, hipMemcpyDeviceToDevice, s));
      GM_HIP_OK(hipMemsetAsync(bits_giant, 0, mwords * 4, s));
      if (Aout.ngiant > 0)  // (a shard without a giant row takes part with an empty list)
        hipLaunchKernelGGL(dev::k_split_row_bits, dim3(grid_for(Aout.ngiant)), dim3(dev::kBlock), 0, s, Aout.giant_row, Aout.ngiant, bits_rest, bits_giant);
      if (verbose) printf("GraphMat(HIP): sharded swept schedule: %d giant rows here, blocks of %d entries; the all-gather travels while they fold\n", Aout.ngiant, cap);
      T* xcur = x;
      T* xnext = (T*)x2v;
      const int ag = grid_for(n_live) < dev::kApplyMaxBlocks ? grid_for(n_live) : dev::kApplyMaxBlocks;
      int it = 0;
      {
        dev::ProgArg<P> pa = dev::make_prog_arg(gp);
        timer.mark(TAG_START);
        send_all(pa, nullptr, xcur);
        if (gm_graph_exchange(g, GM_XCHG_MESSAGES, xcur, (int64_t)sizeof(T), xbits, nullptr) != 0) die("message exchange callback failed");
        timer.mark(TAG_SEND);
      }
      for (; it < iterations; it++) {
        dev::ProgArg<P> pa = dev::make_prog_arg(gp);
        const bool more = it + 1 < iterations;
        timer.mark(TAG_START);
        xq = xcur;
        xb = nullptr;
        multiply_out_swept(pa, dev::ACC_STATIC_BITS, sw, more);
        if (more) {
          // every row but the giant ones: apply, the next message, and off they go
          hipLaunchKernelGGL((dev::k_apply_send<P, T, U, V>), dim3(ag), dim3(dev::kBlock), 0, s, pa, (const U*)y, (const uint32_t*)bits_rest, d_vp, d_active, n_live, d_changed,
                             (uint32_t*)nullptr, xnext, xbits, desc.row_lo);
          timer.mark(TAG_APPLY);
          int part[2] = {0, n_live};
          if (gm_graph_exchange(g, GM_XCHG_PART, xnext, (int64_t)sizeof(T), nullptr, part) != 0) die("partial message exchange failed");
          timer.mark(TAG_SEND);
          // the giant rows, once their folds have joined
          aux.wait_join(s);
          timer.mark(TAG_GIANT);
          hipLaunchKernelGGL((dev::k_apply_send_list<P, T, U, V>), dim3(grid_for(cap)), dim3(dev::kBlock), 0, s, pa, (const U*)y, (const uint32_t*)bits_giant, d_vp, d_active,
                             Aout.giant_row, Aout.ngiant, d_changed, xnext, desc.row_lo, gather + (size_t)desc.shard * cap, cap);
          timer.mark(TAG_APPLY);
          int hf[2] = {cap, 0};
          if (gm_graph_exchange(g, GM_XCHG_GATHER, gather, (int64_t)sizeof(xentry_t), nullptr, hf) != 0) die("giant rows' message exchange failed");
          if (gm_graph_exchange(g, GM_XCHG_WAIT, xnext, (int64_t)sizeof(T), nullptr, nullptr) != 0) die("message exchange wait failed");
          const int64_t nall = (int64_t)desc.nshards * cap;
          hipLaunchKernelGGL((dev::k_unpack_frontier<T>), dim3(grid_for(nall)), dim3(dev::kBlock), 0, s, (const xentry_t*)gather, nall, xnext, xbits);
          timer.mark(TAG_SEND);
        } else {
          hipLaunchKernelGGL((dev::k_apply<P, U, V, false>), dim3(ag), dim3(dev::kBlock), 0, s, pa, (const U*)y, Aout.rowbits, d_vp, d_active, n_live, d_changed,
                             (const int64_t*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr, (unsigned int*)nullptr);
          timer.mark(TAG_APPLY);
        }
        if (n_live < n && it == 0) GM_HIP_OK(hipMemsetAsync(d_active + n_live / 32, 0, (size_t)(nwords - n_live / 32) * 4, s));
        gp->do_every_iteration(it);  // (the base class's empty hook: this schedule is only taken by programs that do not override it)
        T* t = xcur; xcur = xnext; xnext = t;
      }
      fill_active();
      finish(it);
      return it;
    } else {
      (void)sw;
      return -1;
    }
  }

  // Can this run's pull multiply of the OUT adjacency take the column-blocked stream of the short rows (graphmat_hip.h: gm_blocked_t;
  // kernels.hpp: k_spmv_blocked)?  The conditions of the sweep; the structure exists only for graphs without skew (edge values: none, or 4 bytes in its entries).
  bool said_blocked = false, said_sparse_sweep = false, said_short_fold = false;
  int sparse_sweeps = 0;  // multiplies of this run that took a sparse x through the sweep (note 3 of the graph: tests read it)
  int short_folds = 0;    // multiplies of this run whose short rows were folded from the sweep's products stream (note 4)
  bool blocked_usable(int acc, gm_blocked_t* bl) {
    if constexpr (sizeof(T) == 4 && sizeof(U) == 4 && std::is_trivially_copyable<T>::value && std::is_trivially_copyable<U>::value) {
      if (use_vp || xq == nullptr || xb != nullptr || d_want != nullptr || program_row_filter<P>::enabled || (acc & dev::ACC_READ_PREV)) return false;
      if (!(rk == REDUCE_ORDERED || rk == REDUCE_F32_ADD || rk == REDUCE_COMMUTATIVE)) return false;
      if (opt.debug_flags & dev::DBG_NO_TILES) return false;
      if (gm_graph_blocked(g, bl) != GM_OK || bl->nrows <= 0 || bl->short_row != Aout.short_row) return false;
      if (Aout.vals != nullptr && !(bl->val_bytes == 4 && bl->eval != nullptr && sizeof(E) == 4 && std::is_trivially_copyable<E>::value)) return false;
      return true;
    } else {
      (void)acc; (void)bl;
      return false;
    }
  }
  // the launch of k_spmv_blocked on the run's stream (the kernel wants the whole chip: 256 workgroups x 128 KB of LDS)
  void launch_blocked(const dev::ProgArg<P>& pa, const gm_blocked_t& bl) {
    if constexpr (sizeof(T) == 4 && sizeof(U) == 4 && std::is_trivially_copyable<T>::value && std::is_trivially_copyable<U>::value) {
      // (per instantiation and device: the kernels' 128 KB of dynamic LDS have to be allowed once, and the device's CU count is asked once)
      static int cus_of[64] = {0};
      constexpr bool kValsOk = sizeof(E) == 4 && std::is_trivially_copyable<E>::value;
      int dev_id = 0;
      GM_HIP_OK(hipGetDevice(&dev_id));
      int& cus = cus_of[dev_id & 63];
      if (cus == 0) {
        GM_HIP_OK(hipFuncSetAttribute((const void*)&dev::k_spmv_blocked<P, T, U, V, E, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_BLOCKED_ROWS * 4));
        GM_HIP_OK(hipFuncSetAttribute((const void*)&dev::k_spmv_blocked<P, T, U, V, E, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_BLOCKED_ROWS * 4));
        if constexpr (kValsOk) {
          GM_HIP_OK(hipFuncSetAttribute((const void*)&dev::k_spmv_blocked<P, T, U, V, E, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_BLOCKED_ROWS * 4));
          GM_HIP_OK(hipFuncSetAttribute((const void*)&dev::k_spmv_blocked<P, T, U, V, E, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_BLOCKED_ROWS * 4));
        }
        int n = 0;
        GM_HIP_OK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_id));
        cus = n > 0 ? n : 1;
      }
      if (verbose && !said_blocked) { printf("GraphMat(HIP):   the short rows take the column-blocked stream (%d rows, %lld edges%s)\n", bl.nrows, (long long)bl.nentries, Aout.vals ? ", with their edge values" : ""); said_blocked = true; }
      // workgroups that are not all resident at once (a partitioned or masked device) cannot walk the slices in step: they are not asked to
      // -- the device must have a CU per workgroup AND the runtime must say that a workgroup of this kernel (1024 threads, 128 KB of LDS)
      // fits a CU (round-5 advice; the kernel's own wait is bounded and sticky on top of that)
      static int fits_of[64] = {0};  // 0: not asked yet, 1: fits, -1: does not
      int& fits = fits_of[dev_id & 63];
      if (fits == 0) {
        int per_cu = 0;
        const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)&dev::k_spmv_blocked<P, T, U, V, E, false, 2>, 1024, (size_t)GM_BLOCKED_ROWS * 4);
        fits = (oe == hipSuccess && per_cu >= 1) ? 1 : -1;
        (void)hipGetLastError();
      }
      const int window = (cus >= 256 && fits > 0) ? (opt.blocked_form & 15) : 0;
      if (window > 0) GM_HIP_OK(hipMemsetAsync(bl.step_count, 0, (size_t)8 * bl.nsteps * 4, s));
      auto launch = [&](auto vals_c, auto ub_c) {
        constexpr bool HV = decltype(vals_c)::value;
        constexpr int UBV = decltype(ub_c)::value;
        hipLaunchKernelGGL((dev::k_spmv_blocked<P, T, U, V, E, HV, UBV>), dim3(256), dim3(1024), GM_BLOCKED_ROWS * 4, s, pa, bl.ecol, bl.erow, bl.eval, bl.woff, bl.nslices, bl.nblocks,
                           bl.row_of, xq, y, bl.step_count, bl.nsteps, window);
      };
      const bool ub4 = (opt.blocked_form & 16) != 0;
      bool launched = false;
      if constexpr (kValsOk) {
        if (Aout.vals != nullptr) {
          if (ub4) launch(std::true_type(), std::integral_constant<int, 4>()); else launch(std::true_type(), std::integral_constant<int, 2>());
          launched = true;
        }
      }
      if (!launched) { if (ub4) launch(std::false_type(), std::integral_constant<int, 4>()); else launch(std::false_type(), std::integral_constant<int, 2>()); }
    } else {
      (void)pa; (void)bl;
    }
  }
  // The OUT adjacency of a graph without skew: the short rows -- (nearly) all of it -- through the column-blocked stream, whatever rows
  // are longer through the whole-graph CSR's wave / giant kernels in front of it.  The row sets are disjoint.
  void multiply_out_blocked(const dev::ProgArg<P>& pa, int acc, const gm_blocked_t& bl) {
    if constexpr (sizeof(T) == 4 && sizeof(U) == 4 && std::is_trivially_copyable<T>::value && std::is_trivially_copyable<U>::value) {
      gm_csr_t Ar = Aout;  // the rows above the short-row limit
      Ar.nblk = 0;
      if (Ar.nmid > 0 || Ar.ngiant > 0) {
        Launch La = launch_ctx();
        La.aux = nullptr;  // (few rows, if any: everything on the run's stream)
        launch_spmv_vp<P, T, U, V, E>(use_vp, La, pa, Ar, xq, xb, (const V*)d_vp, y, ybits, acc, rk);
      }
      launch_blocked(pa, bl);
      st.spmv_launches += 1;
      timer.mark(TAG_ROWBLOCK);
      check_probed(pa, Aout, Aout.rowbits, nullptr, acc, ybits, true);
    } else {
      (void)pa; (void)acc; (void)bl;
    }
  }

  // the OUT adjacency tile by tile on two streams (graphmat_hip.h: gm_graph_tile).  With every x entry present and an
  // ordered or commutative fold, the rows of more than tile_min_row edges are multiplied tile by tile -- each pass gathers
  // from one slice of x and continues the row's fold from the value y holds -- and only the short rows take the
  // untiled row-blocks.  Same fold order, same bits.
  void multiply_out_tiled(const dev::ProgArg<P>& pa, int ntile, int acc) {
    const Launch L = launch_ctx();
    // Tiles whose row classes are fixed per row (gm_csr_t.rows_keep_stream): the giant / one-wave-per-row kernels and the
    // row-block / 16-row kernels never touch the same y entry, so the auxiliary stream is forked once -- before the untiled
    // pass -- and joined once after the last tile instead of after every tile, where the main stream used to wait 70-120 us
    // per tile for the tail of the one-wave-per-row kernel.
    bool keep_streams = aux.s != nullptr && !use_vp && (rk == REDUCE_ORDERED || rk == REDUCE_F32_ADD) && std::is_trivially_copyable<U>::value &&
                        (sizeof(U) == 4 || sizeof(U) == 8) && !rk_unverified &&
                        !(opt.debug_flags & (dev::DBG_NO_WAVE16 | dev::DBG_LONG_ON_MAIN | dev::DBG_NO_OVERLAP));
    for (int t = 0; t < ntile && keep_streams; t++) {
      gm_csr_t At;
      const uint32_t* prev = nullptr;
      if (gm_graph_tile(g, GM_DIR_OUT, t, &At, &prev) != GM_OK || !At.rows_keep_stream) keep_streams = false;
    }
    if (keep_streams) {
      aux.keep = true;
      aux.pending = false;
      GM_HIP_OK(hipEventRecord(aux.fork, s));  // x is complete here: the auxiliary stream may start on tile 0 during the untiled pass
      GM_HIP_OK(hipStreamWaitEvent(aux.s, aux.fork, 0));
      aux.forked = true;
    }
    gm_csr_t As = Aout;  // the rows that are not tiled: row-blocks and the shorter wave rows
    As.mid_row = Aout.umid_row; As.nmid = Aout.numid; As.nmid_long = Aout.numid_long; As.ngiant = 0; As.ngchunk = 0;
    if (Aout.tile_min_row == 0) { As.nblk = 0; As.nmid = 0; }  // every row is tiled
    // (running the untiled pass on a stream of its own next to the tile passes -- its rows are no tile's rows -- was measured:
    // 7.06 -> 7.03 ms at RMAT-26, not worth a third stream)
    launch_spmv_vp<P, T, U, V, E>(use_vp, L, pa, As, xq, xb, (const V*)d_vp, y, ybits, acc, rk);
    aux.long_rows = !(opt.debug_flags & dev::DBG_LONG_ON_MAIN);
    for (int t = 0; t < ntile; t++) {
      gm_csr_t At;
      const uint32_t* prev = nullptr;
      if (gm_graph_tile(g, GM_DIR_OUT, t, &At, &prev) != GM_OK) die(gm_last_error());
      // y's presence bits are static (dense x): `prev` says which rows already carry a value
      launch_spmv_vp<P, T, U, V, E>(use_vp, L, pa, At, xq, xb, (const V*)d_vp, y, const_cast<uint32_t*>(prev), dev::ACC_STATIC_BITS | dev::ACC_READ_PREV, rk);
    }
    aux.long_rows = false;
    if (aux.keep) {
      if (aux.pending) GM_HIP_OK(hipStreamWaitEvent(s, aux.join, 0));
      aux.keep = aux.forked = aux.pending = false;
    }
    // (a probed strategy is cross-checked against the ordered fold of the WHOLE rows; a mismatch redoes this iteration
    // untiled with the ordered fold, which then also governs the tiled iterations that follow)
    check_probed(pa, Aout, Aout.rowbits, nullptr, acc, ybits, true);
  }

  // top-down step over a larger active set inside the pull step: bids, then one pass over all vertices picks the winners
  void multiply_dense_push(const dev::ProgArg<P>& pa) {
    list_active_set();
    const unsigned pieces = (unsigned)(frontier_e / dev::kPieceEdges + frontier_v);
    piece_offsets((int)frontier_v);
    hipLaunchKernelGGL(dev::k_push_bid, dim3(pieces > 0 ? pieces : 1), dim3(dev::kBlock), 0, s, Asrc, (const int32_t*)d_list, (int)frontier_v,
                       (const unsigned int*)d_off, native_of_dev, d_best, (const uint32_t*)d_want, (int32_t*)nullptr, (unsigned int*)nullptr);
    if (use_vp)
      hipLaunchKernelGGL((dev::k_push_resolve<P, T, U, V, E, true>), dim3(grid_for(n_live)), dim3(dev::kBlock), 0, s, pa, Asrc, xq, dev_of_native,
                         (const V*)d_vp, d_best, y, ybits, n_live);  // (only live rows can have been bid for)
    else
      hipLaunchKernelGGL((dev::k_push_resolve<P, T, U, V, E, false>), dim3(grid_for(n_live)), dim3(dev::kBlock), 0, s, pa, Asrc, xq, dev_of_native,
                         (const V*)d_vp, d_best, y, ybits, n_live);
    st.spmv_launches += 2;
    timer.mark(TAG_WAVE);
  }

  void step_pull(const dev::ProgArg<P>& pa, int it, bool xsp, bool dense_push, bool want_stats) {
    // Clear(&x) / Clear(&y) (:139-140): x presence words are fully rewritten by send.  With every x entry present
    // (ALL_VERTICES) y's presence is the static set of non-empty rows.
    const bool static_bits = (act == ALL_VERTICES);
    const bool dense_x = (act == ALL_VERTICES);
    if (!static_bits) GM_HIP_OK(hipMemsetAsync(ybits, 0, (size_t)nwords * 4, s));
    // send (:145).  Rows past n_live have no edge in either direction (degree-ranked order puts them at the tail): nobody
    // reads their messages and they never receive one
    if (xsp) {
      send_sparse(pa);
    } else {
      const bool presend_valid = x_presend;  // (only ever set for programs that cannot change between the iterations)
      x_presend = false;
      if (!lazy_send && !presend_valid) send_all(pa, dense_x ? (const uint32_t*)nullptr : (const uint32_t*)d_active, x);
      if (multi && gm_graph_exchange(g, GM_XCHG_MESSAGES, x, (int64_t)sizeof(T), xbits, nullptr) != 0) die("message exchange callback failed");
    }
    timer.mark(TAG_SEND);
    lap("Send message time");
    // multiply + reduce (:160-176)
    xq = lazy_send ? (const T*)nullptr : (const T*)x;
    xb = dense_x ? nullptr : (lazy_send ? (const uint32_t*)d_active : (const uint32_t*)xbits);
    const uint32_t* apply_bits = ybits;
    const uint32_t* row_bits = d_want;  // which rows the multiply works on
    bool guided_few = true;
    if (guided && !dense_push && !xsp && frontier_v > 0 && frontier_e * 50ull < (unsigned long long)Aout.nnz) {
      // few out-edges leave the active set: mark the rows they reach, fold only those (same kernels, same order)
      GM_HIP_OK(hipMemsetAsync(d_mark, 0, ((size_t)(n + 31) / 32 + 2) * 4, s));
      if (frontier_v <= (unsigned long long)dev::kSparseListCap) {  // few vertices: from their list, a workgroup per 1024 out-edges (a hub is spread over the chip)
        list_active_set();
        const int nf = (int)frontier_v;
        const unsigned pieces = (unsigned)(frontier_e / dev::kPieceEdges + frontier_v);
        piece_offsets(nf);
        hipLaunchKernelGGL(dev::k_mark_rows_of_list, dim3(pieces > 0 ? pieces : 1), dim3(dev::kBlock), 0, s, Asrc, (const int32_t*)d_list, nf, (const unsigned int*)d_off, d_mark);
      } else {
        const int mgrid = grid_for(n_live) < 4096 ? grid_for(n_live) : 4096;
        hipLaunchKernelGGL(dev::k_mark_rows_of_active, dim3(mgrid), dim3(dev::kBlock), 0, s, Asrc, (const uint32_t*)d_active, n_live, d_mark);
      }
      row_bits = d_mark;
      // (the kernel that takes 64 list entries per wave and works on the wanted ones one after the other pays for a handful of wanted rows;
      // with thousands of them the 16-rows-per-wave kernels, which skip unwanted rows as well, are several times faster: RMAT-26 BFS level
      // with 857 K out-edges 14.3 ms against ~4)
      guided_few = frontier_e <= 4096ull;
      if (verbose) printf("GraphMat(HIP):   guided pull: %llu active vertices, %llu out-edges\n", frontier_v, frontier_e);
    }
    if (dense_push) {
      multiply_dense_push(pa);
    } else if (order == OUT_EDGES || order == ALL_EDGES) {
      const int acc = static_bits ? dev::ACC_STATIC_BITS : 0;
      // sparse active set of an a=b program: a 64:1 summary of the presence bits for the short-row kernel
      const uint32_t* xsum = nullptr;
      const unsigned long long present_x = xsparse_ok ? (unsigned long long)xs_total : frontier_v;  // entries of x that are present
      if (want_stats && rk == REDUCE_LAST && xb != nullptr && present_x * 512ull < (unsigned long long)n_live * (unsigned long long)desc.nshards) {
        void* ps = nullptr;
        const int xwords = (desc.ndevice + 31) / 32;  // x (and its presence bits) cover every shard's rows
        const int nsum = (xwords / 2 + 31) / 32 + 1;
        if (gm_graph_workspace(g, 11, (size_t)nsum * 4 + 64, &ps) == GM_OK) {
          hipLaunchKernelGGL(dev::k_bits_summary, dim3(grid_for(nsum)), dim3(dev::kBlock), 0, s, xb, xwords, (uint32_t*)ps, nsum);
          xsum = (const uint32_t*)ps;
        }
      }
      int ntile = 1;
      if (dense_x && !multi && row_bits == nullptr && rk != REDUCE_LAST && dev::stageable<T>::value && !(opt.debug_flags & dev::DBG_NO_TILES))
        gm_graph_tiles(g, GM_DIR_OUT, &ntile);
      gm_sweep_t sw;
      gm_blocked_t bl;
      if (row_bits == nullptr && !xsp && sweep_usable(acc, &sw)) {  // (sharded graphs too: gm_sweep_t.nsub; a sparse x too: k_spmv_sell_sparse)
        multiply_out_swept(pa, acc, sw);
      } else if (dense_x && !multi && row_bits == nullptr && blocked_usable(acc, &bl)) {
        multiply_out_blocked(pa, acc, bl);
      } else if (ntile > 1) {
        multiply_out_tiled(pa, ntile, acc);
      } else {
        launch_spmv_vp<P, T, U, V, E>(use_vp, launch_ctx(), pa, Aout, xq, xb, (const V*)d_vp, y, ybits, acc, rk, row_bits, grouped_waves && guided_few, xsum);
        check_probed(pa, Aout, static_bits ? Aout.rowbits : (const uint32_t*)ybits, row_bits, acc, ybits, dense_x);
      }
      if (static_bits) apply_bits = Aout.rowbits;
    }
    if (!dense_push && (order == IN_EDGES || order == ALL_EDGES)) {
      int acc = (order == ALL_EDGES) ? dev::ACC_READ_PREV : 0;
      uint32_t* yb = ybits;
      if (static_bits) {
        acc |= dev::ACC_STATIC_BITS;
        yb = const_cast<uint32_t*>(Aout.rowbits);  // only read (presence of the OUT pass's results)
        apply_bits = Ain.rowbits;
        if (order == ALL_EDGES && gm_graph_rowbits_all(g, &apply_bits) != GM_OK) die(gm_last_error());
      }
      launch_spmv_vp<P, T, U, V, E>(use_vp, launch_ctx(), pa, Ain, (const T*)x, xb, (const V*)d_vp, y, yb, acc, rk, (const uint32_t*)d_want, grouped_waves);
      if (order == IN_EDGES) check_probed(pa, Ain, static_bits ? Ain.rowbits : (const uint32_t*)ybits, d_want, acc, ybits, dense_x);
    }
    lap("SPMV time");
    if (trace) tr_updated = count_bits(apply_bits, n_live);  // y.getNNZ(): rows that received a message
    // setAllInactive (:184) + apply (:195-225): the active vector is rewritten by k_apply (when top-down steps are
    // possible the kernel also sizes and lists the next active set).  The changed vertices are also listed when the next
    // active set is bound to be small: it cannot have more vertices than the current one has out-edges.  (If it turns
    // out small without having been listed, a k_frontier_list pass builds the list when it is needed.)
    const bool build_list = want_stats && frontier_e <= (4ull << 20);
    if (want_stats) GM_HIP_OK(hipMemsetAsync(d_count, 0, 4, s));
    listed = build_list;
    // (a workgroup that lists changed vertices ends with one global atomic: fewer, longer-running workgroups then)
    const int apply_cap = build_list ? dev::kApplyMaxBlocks / 4 : dev::kApplyMaxBlocks;
    const int apply_grid = grid_for(n_live) < apply_cap ? grid_for(n_live) : apply_cap;
    if (want_stats)
      hipLaunchKernelGGL((dev::k_apply<P, U, V, true>), dim3(apply_grid), dim3(dev::kBlock), 0, s, pa, (const U*)y, apply_bits, d_vp, d_active, n_live,
                         d_changed, Asrc.rowptr, d_striped, d_want, build_list ? d_list : (int32_t*)nullptr, build_list ? d_count : (unsigned int*)nullptr);
    else if (inherits_iteration_hook<P>() && dense_x && !lazy_send && !trace && opt.fuse_apply_send != 0 && iterations > 0 && it + 1 < iterations) {
      // another iteration follows: its messages come out of the same pass.  Only for programs without a do_every_iteration
      // of their own (nothing can change what send_message reads between the two iterations), and only in fixed-count
      // runs: until convergence the last iteration is not known in advance, and a fused pass there would leave x holding
      // messages of an iteration that never runs (the reference's px would not)
      hipLaunchKernelGGL((dev::k_apply_send<P, T, U, V>), dim3(apply_grid), dim3(dev::kBlock), 0, s, pa, (const U*)y, apply_bits, d_vp, d_active, n_live,
                         d_changed, d_want, x, xbits, desc.row_lo);
      x_presend = true;
    } else
      hipLaunchKernelGGL((dev::k_apply<P, U, V, false>), dim3(apply_grid), dim3(dev::kBlock), 0, s, pa, (const U*)y, apply_bits, d_vp, d_active, n_live,
                         d_changed, (const int64_t*)nullptr, (unsigned long long*)nullptr, d_want, (int32_t*)nullptr, (unsigned int*)nullptr);
    if (n_live < n && it == 0)  // setAllInactive for the rows k_apply does not visit (they have no edges: once clear, nothing sets them again)
      GM_HIP_OK(hipMemsetAsync(d_active + n_live / 32, 0, (size_t)(nwords - n_live / 32) * 4, s));
    timer.mark(TAG_APPLY);
    lap("Apply time");
  }

  // ---- the loop (GraphMatRuntime.h:136-261) ----------------------------------------------------------------------------
  int run_loop() {
    int it = 0;
    tick("setup done", 0);
    while (true) {
      tick("iteration", it);
      if (trace) { (void)hipStreamSynchronize(s); gettimeofday(&tr_iter, 0); tr_last = tr_iter; }
      tr_updated = -1;
      if (verbose && can_push) printf("GraphMat(HIP):   active set: %llu vertices, %llu out-edges (max %llu)\n", frontier_v, frontier_e, frontier_maxdeg);
      dev::ProgArg<P> pa = dev::make_prog_arg(gp);  // re-captured every iteration (do_every_iteration may change it)
      const bool want_stats = (can_push || xsparse_ok || guided) && iterations <= 0;
      // the changed flag and, for steered runs, the striped statistics behind it (k_apply / k_push_finish add to them);
      // a fixed-count run never reads the flag (:254-256), so it is not cleared either: one tiny fill kernel less per iteration
      if (iterations <= 0) GM_HIP_OK(hipMemsetAsync(d_changed, 0, want_stats ? sizeof(int) + striped_bytes : sizeof(int), s));
      timer.mark(TAG_START);
      // this iteration's x travels as lists when every shard's active set is small: fewer bytes than the dense slices
      // (entry = id + message against one message per live row) and within the list capacity
      const bool xsp = xsparse_ok && xs_max <= dev::kSparseListCap &&
                       (unsigned long long)xs_max * sizeof(xentry_t) * 2ull < (unsigned long long)n_live * sizeof(T);
      // top-down step for small active sets only: few sources and few out-edges
      const bool push = can_push && frontier_v > 0 && frontier_v <= (unsigned long long)dev::kSparseListCap &&
                        frontier_e * 1000ull < (unsigned long long)Aout.nnz * (unsigned long long)opt.push_edge_permille;
      // ... and among those, active sets with few out-edges run entirely on lists (nothing scans all vertices)
      const bool sparse = push && frontier_e <= (unsigned long long)opt.sparse_step_edges;
      const bool dense_push = push && !sparse && rk == REDUCE_LAST;
      // ... and an active set too large to list whose vertices own only a few out-edges each bids straight from the bitmap
      const bool bits_push = can_push && !push && rk == REDUCE_LAST && frontier_v > (unsigned long long)dev::kSparseListCap &&
                             frontier_e <= (unsigned long long)opt.bits_step_edges && frontier_maxdeg <= 64ull;
      if (bits_push) step_bits_push(pa);
      else if (sparse) step_list_push(pa);
      else step_pull(pa, it, xsp, dense_push, want_stats);

      int converged = 0;
      if (iterations <= 0) {  // the flag only matters when running until convergence (:257-259)
        // the flag and, behind it, the size of the next active set (written by k_apply): one copy
        GM_HIP_OK(hipMemcpyAsync(h_changed, d_changed, want_stats ? sizeof(int) + striped_bytes : sizeof(int), hipMemcpyDeviceToHost, s));
        GM_HIP_OK(hipStreamSynchronize(s));
        if (want_stats) {
          frontier_v = frontier_e = frontier_maxdeg = 0;
          for (int k = 0; k < dev::kStatSlots; k++) {
            frontier_v += h_striped[4 * k];
            frontier_e += h_striped[4 * k + 1];
            frontier_maxdeg = h_striped[4 * k + 2] > frontier_maxdeg ? h_striped[4 * k + 2] : frontier_maxdeg;
          }
          list_ready = listed && frontier_v <= (unsigned long long)dev::kSparseListCap;  // k_apply / k_push_finish listed it
        }
        converged = (*h_changed == 0) ? 1 : 0;
        if (xsparse_ok) exchange_state(&converged);  // the flag and the shards' active-set sizes in one step
        else if (multi) gm_graph_exchange(g, GM_XCHG_CONVERGED, nullptr, 0, nullptr, &converged);  // :226 Allreduce(LAND)
      }
      gp->do_every_iteration(it);  // :236
      if (trace) {
        lap("Do every iteration time");
        const long long changed = count_bits(d_active, n);  // g.active->getNNZ() before ALL_VERTICES re-activates (:248-252)
        (void)hipStreamSynchronize(s);
        struct timeval now;
        gettimeofday(&now, 0);
        printf("Iteration %d :: %f msec :: updated %lld vertices :: changed %lld vertices \n", it,
               (now.tv_sec - tr_iter.tv_sec) * 1e3 + (now.tv_usec - tr_iter.tv_usec) * 1e-3, tr_updated, changed);
      }
      const bool last_fixed = iterations > 0 && it + 1 == iterations;
      const bool done = last_fixed || (iterations <= 0 && converged == 1);
      it++;
      if (done) {
        if (act == ALL_VERTICES) fill_active();  // leave the graph all-active (:250-252)
        break;
      }
    }
    finish(it);
    return it;
  }
};

template <class P, class T, class U, class V, class E>
int run_on_device(P* gp, gm_graph_t* g, edge_direction order, activity_type act, bool use_vp, V* d_vp, uint32_t* d_active, T* x, uint32_t* xbits, U* y,
                  uint32_t* ybits, int iterations, hipStream_t s) {
  Run<P, T, U, V, E> run(gp, g, order, act, use_vp, d_vp, d_active, x, xbits, y, ybits, iterations, s);
  return run.go();
}

}  // namespace detail
}  // namespace GraphMat
