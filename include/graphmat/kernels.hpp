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
//  * rows longer than GM_LONG_ROW get a workgroup each (k_spmv_longrow) with a
//    reduction strategy chosen by program_traits<P>::reduce.
//  * presence bit vectors keep the reference layout (bit i&31 of word i>>5).
//  * the vertex program is passed by value as raw bytes and its methods are
//    called qualified (p.P::f(...)), i.e. never through the host vtable.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../graphmat_hip.h"

namespace GraphMat {

// How a program's reduce_function may be evaluated.  The default is always safe.
enum reduce_kind {
  REDUCE_ORDERED = 0,      // fold strictly in ascending column order (any functor)
  REDUCE_COMMUTATIVE = 1,  // associative+commutative and exact (integer +, min, max): any order
  REDUCE_LAST = 2,         // reduce(a,b) is a=b: the result is the last present message
  REDUCE_F32_ADD = 3       // float a+=b: ordered result, reproduced bit-exactly in parallel
};

// Optional, per-program knowledge the runtime may exploit; specialise for your
// program type.  Nothing here changes results, only how they are computed.
template <class P>
struct program_traits {
  static constexpr reduce_kind reduce = REDUCE_ORDERED;
};

namespace dev {

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
// apply on rows whose y bit is set; a changed vertex (V::operator!=) becomes active and
// raises the changed flag (zeroed by the host before the launch).  The active vector is fully rewritten (the reference clears
// it right before, GraphMatRuntime.h:184).
template <class P, class U, class V>
__global__ void __launch_bounds__(kBlock)
k_apply(ProgArg<P> pa, const U* __restrict__ y, const uint32_t* __restrict__ ybits, V* __restrict__ vp,
        uint32_t* __restrict__ active, int n, int* __restrict__ changed_flag) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  bool changed = false;
  if (i < n && bit_get(ybits, i)) {
    ProgArg<P> local = pa;  // apply() is non-const in the API: give it a private copy
    P& p = *reinterpret_cast<P*>(local.b);
    V old_prop = vp[i];
    V cur = old_prop;
    p.P::apply(y[i], cur);
    vp[i] = cur;
    if (old_prop != cur) changed = true;
  }
  unsigned long long m = __ballot(changed);
  if ((threadIdx.x & 63) == 0 && i < n) {
    active[i >> 5] = (uint32_t)m;
    if (i + 32 < n) active[(i >> 5) + 1] = (uint32_t)(m >> 32);
    if (m) *changed_flag = 1;
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
// multiply+reduce over row-blocks (rows of at most GM_LONG_ROW edges).
//   USE_VP : 3-operand form, process_message sees vp[row] (spmspv3.h:70)
//   xbits == nullptr : every x entry present (ALL_VERTICES programs)
//   accumulate : y may already hold partial results (second pass of ALL_EDGES)
template <class P, class T, class U, class V, class E, bool USE_VP>
__global__ void __launch_bounds__(kBlock)
k_spmv_rowblock(ProgArg<P> pa, gm_csr_t A, const T* __restrict__ x, const uint32_t* __restrict__ xbits,
                const V* __restrict__ vp, U* __restrict__ y, uint32_t* __restrict__ ybits, int accumulate) {
  constexpr bool STAGE = stageable<T>::value;
  typedef typename raw_of<STAGE ? (int)sizeof(T) : 1>::type raw_t;
  __shared__ int s_col[kStage];
  __shared__ raw_t s_msg[STAGE ? kStage : 1];

  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int b = blockIdx.x;
  const int r0 = A.blk_row[b], r1 = A.blk_row[b + 1];
  const int64_t e0 = A.rowptr[r0], e1 = A.rowptr[r1];
  const int n = (int)(e1 - e0);
  if (n == 0) return;
  if (r1 - r0 == 1 && n > GM_LONG_ROW) return;  // handled by k_spmv_longrow

  // phase 1: coalesced column ids, parallel gathers, into LDS in edge order
  for (int k = threadIdx.x; k < n; k += kBlock) {
    int c = A.colidx[e0 + k];
    bool present = (xbits == nullptr) || bit_get(xbits, c);
    s_col[k] = present ? c : -1;
    if (STAGE && present) s_msg[k] = reinterpret_cast<const raw_t*>(x)[c];
  }
  __syncthreads();

  // phase 2: one lane per row folds its segment in ascending column order
  const int row = r0 + threadIdx.x;
  if (row < r1) {
    const int kb = (int)(A.rowptr[row] - e0), ke = (int)(A.rowptr[row + 1] - e0);
    if (ke > kb) {
      bool has = accumulate && bit_get(ybits, row);
      U acc;
      if (has) acc = y[row];
      V vprow;
      if (USE_VP) vprow = vp[row];
      for (int k = kb; k < ke; k++) {
        int c = s_col[k];
        if (c < 0) continue;
        T m;
        if (STAGE) {
          raw_t r = s_msg[k];
          memcpy(&m, &r, sizeof(T));
        } else {
          m = x[c];
        }
        fold_one<P, T, U, V, E>(p, m, edge_at<E>(A.vals, e0 + k), vprow, acc, has);
      }
      if (has) {
        y[row] = acc;
        atomicOr(&ybits[row >> 5], 1u << (row & 31));
      }
    }
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
// (delta if S even, delta if S odd).  A chunk is folded with an ordered tree of these
// pairs; it is accepted iff all terms are finite, non-negative and smaller than 2^e and
// the chunk total keeps S below 2^24 (no binade crossing, S is monotone).  Otherwise the
// chunk is replayed serially.  Results are bit-identical to the serial loop.
struct ulp_map {
  uint32_t de, dod;  // ulps added when the incoming S is even / odd
};
// Deltas are non-negative and a chunk is only accepted when its total stays below 2^23,
// so partial results saturate at 2^24 (no 32-bit wrap; a saturated value forces a reject).
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
  if (abits >> 31) return false;  // negative (or -0: replay serially)
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

// ------------------------------------------------------------------------------------
// multiply+reduce for one long row per workgroup.
template <class P, class T, class U, class V, class E, bool USE_VP, int RK>
__global__ void __launch_bounds__(kBlock)
k_spmv_longrow(ProgArg<P> pa, gm_csr_t A, const T* __restrict__ x, const uint32_t* __restrict__ xbits,
               const V* __restrict__ vp, U* __restrict__ y, uint32_t* __restrict__ ybits, int accumulate) {
  constexpr int CH = kStage;  // edges per chunk
  __shared__ int s_col[CH];
  __shared__ __attribute__((aligned(16))) unsigned char s_res_raw[(RK == REDUCE_COMMUTATIVE || RK == REDUCE_F32_ADD) ? kBlock * sizeof(U) : 16];
  __shared__ int s_has[kBlock];
  __shared__ ulp_map s_map[kBlock / 64];
  __shared__ int s_flag;

  const P& p = *reinterpret_cast<const P*>(pa.b);
  const int row = A.long_row[blockIdx.x];
  const int64_t e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
  const int tid = threadIdx.x;
  V vprow;
  if constexpr (USE_VP) vprow = vp[row];

  if constexpr (RK == REDUCE_COMMUTATIVE) {
    // any order: strided private folds, then an LDS tree with the user's reduce_function
    U* s_res = reinterpret_cast<U*>(s_res_raw);
    bool has = false;
    U acc;
    for (int64_t k = e0 + tid; k < e1; k += kBlock) {
      int c = A.colidx[k];
      if (xbits != nullptr && !bit_get(xbits, c)) continue;
      T m = x[c];
      fold_one<P, T, U, V, E>(p, m, edge_at<E>(A.vals, k), vprow, acc, has);
    }
    if (tid == 0 && accumulate && bit_get(ybits, row)) {
      U prev = y[row];
      if (has) { U t = acc; acc = prev; p.P::reduce_function(acc, t); } else { acc = prev; has = true; }
    }
    s_has[tid] = has;
    if (has) s_res[tid] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
      if (tid < s && s_has[tid + s]) {
        if (s_has[tid]) { U a = s_res[tid]; p.P::reduce_function(a, s_res[tid + s]); s_res[tid] = a; }
        else { s_res[tid] = s_res[tid + s]; s_has[tid] = 1; }
      }
      __syncthreads();
    }
    if (tid == 0 && s_has[0]) {
      y[row] = s_res[0];
      atomicOr(&ybits[row >> 5], 1u << (row & 31));
    }
    return;
  }

  if constexpr (RK == REDUCE_LAST) {
    // reduce is a=b: the answer is the last present edge of the row; scan backwards
    if (tid == 0) s_flag = -1;
    __syncthreads();
    for (int64_t hi = e1; hi > e0; hi -= CH) {
      int64_t lo = hi - CH < e0 ? e0 : hi - CH;
      int best = -1;
      for (int64_t k = lo + tid; k < hi; k += kBlock) {
        int c = A.colidx[k];
        if (xbits == nullptr || bit_get(xbits, c)) best = (int)(k - lo);
      }
      if (best >= 0) atomicMax(&s_flag, best);
      __syncthreads();
      int f = s_flag;
      if (f >= 0) {
        if (tid == 0) {
          int64_t k = lo + f;
          T m = x[A.colidx[k]];
          U res;
          p.P::process_message(m, edge_at<E>(A.vals, k), vprow, res);
          y[row] = res;
          atomicOr(&ybits[row >> 5], 1u << (row & 31));
        }
        return;
      }
    }
    return;  // no present message: with accumulate an earlier pass's value simply stays
  }

  // ordered kinds: chunked; thread 0 carries the running value
  bool has = false;
  U acc;
  if (tid == 0 && accumulate && bit_get(ybits, row)) { acc = y[row]; has = true; }
  for (int64_t base = e0; base < e1; base += CH) {
    const int n = (int)((e1 - base) < CH ? (e1 - base) : CH);
    __syncthreads();  // previous chunk fully consumed
    for (int k = tid; k < n; k += kBlock) {
      int c = A.colidx[base + k];
      bool present = (xbits == nullptr) || bit_get(xbits, c);
      s_col[k] = present ? c : -1;
    }
    __syncthreads();
    bool done = false;
    if constexpr (RK == REDUCE_F32_ADD) {
      static_assert(sizeof(U) == 4 && sizeof(T) >= 1, "REDUCE_F32_ADD needs a float reduction type");
      // requires U = float and process_message(m, e, vp) independent of order (it is per edge)
      float* s_res = reinterpret_cast<float*>(s_res_raw);
      // each thread owns CH/kBlock consecutive edges of the chunk: products first
      constexpr int PER = CH / kBlock;
      float term[PER];
      bool pres[PER];
#pragma unroll
      for (int j = 0; j < PER; j++) {
        int k = tid * PER + j;
        pres[j] = false;
        term[j] = 0.f;
        if (k < n) {
          int c = s_col[k];
          if (c >= 0) {
            T m = x[c];
            U res;
            p.P::process_message(m, edge_at<E>(A.vals, base + k), vprow, res);
            memcpy(&term[j], &res, sizeof(float));
            pres[j] = true;
          }
        }
      }
      // running sum known to everyone
      if (tid == 0) { s_has[0] = has; if (has) s_res[0] = *reinterpret_cast<float*>(&acc); }
      __syncthreads();
      const bool has0 = s_has[0];
      const float S0 = has0 ? s_res[0] : 0.f;
      const uint32_t sb = __float_as_uint(S0);
      const int eS = (int)((sb >> 23) & 0xff);
      bool ok = has0 && !(sb >> 31) && eS > 0 && eS < 255;
      ulp_map mine = {0u, 0u};
      if (ok) {
#pragma unroll
        for (int j = 0; j < PER; j++) {
          if (pres[j]) {
            ulp_map t;
            if (!ulp_term(__float_as_uint(term[j]), eS, t)) ok = false;
            else mine = ulp_compose(mine, t);
          }
        }
      }
      __syncthreads();  // s_has[0]/s_res[0] read by all before reuse
      if (tid == 0) s_flag = 1;
      __syncthreads();
      if (!ok) s_flag = 0;
      // ordered combine across the wave (lane order), then across waves
      ulp_map v = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        ulp_map o;
        o.de = __shfl_down(v.de, off, 64);
        o.dod = __shfl_down(v.dod, off, 64);
        if (((tid & 63) & (2 * off - 1)) == 0) v = ulp_compose(v, o);
      }
      if ((tid & 63) == 0) s_map[tid >> 6] = v;
      __syncthreads();
      if (tid == 0) {
        if (s_flag) {
          ulp_map t = s_map[0];
          for (int w = 1; w < kBlock / 64; w++) t = ulp_compose(t, s_map[w]);
          uint32_t Sint = (sb & 0x7fffffu) | 0x800000u;  // S in ulps, in [2^23, 2^24)
          uint32_t add = (Sint & 1u) ? t.dod : t.de;     // saturated at 2^24 (ulp_compose)
          uint32_t Snew = Sint + add;
          if (add < 0x800000u && Snew < 0x1000000u) {
            float r = __uint_as_float((sb & 0xff800000u) | (Snew & 0x7fffffu));
            memcpy(&acc, &r, sizeof(float));
            s_flag = 2;  // accepted
          } else {
            s_flag = 0;
          }
        }
      }
      __syncthreads();
      done = (s_flag == 2);
    }
    if (!done && tid == 0) {
      for (int k = 0; k < n; k++) {
        int c = s_col[k];
        if (c < 0) continue;
        T m = x[c];
        fold_one<P, T, U, V, E>(p, m, edge_at<E>(A.vals, base + k), vprow, acc, has);
      }
    }
  }
  if (tid == 0 && has) {
    y[row] = acc;
    atomicOr(&ybits[row >> 5], 1u << (row & 31));
  }
}

// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_fill_u32(uint32_t* __restrict__ p, int64_t n, uint32_t v) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace dev
}  // namespace GraphMat
