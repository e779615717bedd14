// One slice's line of a long neighbourhood held across the registers of a wavefront, and its sort.  gfx950.
//
// Blocked layout: lane l holds elements l*M .. l*M+M-1 (keys, and one 32-bit payload per key when PAYLOAD: the weight
// in the forward kernels, the element index in the backward kernels -- payloads only ever move, bit-exactly).  sort():
// every lane sorts its M keys with the register network of sortnet.h, then six merge levels double the sorted run
// (2, 4, .. 64 lanes): one "flip" exchange with lane ^ (lanes-1) and mirrored registers, log2(lanes)-1 exchanges
// with lane ^ stride, log2(M) in-register half-cleaners.  merge_chunk(): the tail of a merge level for a chunk whose
// halves were separated elsewhere (embed_wsort.hip, rows longer than one chunk).
#pragma once
#include <type_traits>
#include "fsw_common.h"
#include "sortnet.h"

#ifndef FSW_PERMLANE_SWAP
#define FSW_PERMLANE_SWAP 0   // 1: v_permlane16/32_swap for the half-cleaner distances 16 / 32 -- measured SLOWER than ds_swizzle /
#endif                        // ds_bpermute on the 64M-edge RMAT graph (hub<4,24> 10.2 -> 11.2 ms, hub<8,32> 7.2 -> 7.3, giant 6.2 -> 6.5)

namespace fsw {

// value of lane ^ MASK.  Distances inside a row of 16 lanes go through DPP (VALU rate, no LDS crossbar): xor 1, 2, 3 are
// quad permutes, xor 7 / 15 the half-row / row mirrors, xor 4 = mirror7 o quad3, xor 8 = mirror15 o mirror7; xor 16 and 31
// are ds_swizzle bit-mode patterns (inside 32 lanes); only 32 and 63 need ds_bpermute.
template <int MASK>
__device__ __forceinline__ float xor_lane(float v) {
  const int x = __float_as_int(v);
  auto dpp = [](int y, auto ctrl) { return __builtin_amdgcn_update_dpp(0, y, decltype(ctrl)::value, 0xF, 0xF, true); };
  using std::integral_constant;
  if constexpr (MASK == 1) return __int_as_float(dpp(x, integral_constant<int, 0xB1>{}));
  else if constexpr (MASK == 2) return __int_as_float(dpp(x, integral_constant<int, 0x4E>{}));
  else if constexpr (MASK == 3) return __int_as_float(dpp(x, integral_constant<int, 0x1B>{}));
  else if constexpr (MASK == 7) return __int_as_float(dpp(x, integral_constant<int, 0x141>{}));
  else if constexpr (MASK == 15) return __int_as_float(dpp(x, integral_constant<int, 0x140>{}));
  else if constexpr (MASK == 4) return __int_as_float(dpp(dpp(x, integral_constant<int, 0x141>{}), integral_constant<int, 0x1B>{}));
  else if constexpr (MASK == 8) return __int_as_float(dpp(x, integral_constant<int, 0x128>{}));   // row_ror:8 = xor 8 inside a row of 16
  else if constexpr (MASK == 16) return __int_as_float(__builtin_amdgcn_ds_swizzle(x, 0x401F));
  else if constexpr (MASK == 31) return __int_as_float(__builtin_amdgcn_ds_swizzle(x, 0x7C1F));
  else return __shfl_xor(v, MASK);
}

// min(a, b) where lim == -inf, max(a, b) where lim == +inf: ONE v_med3_f32 instead of v_min + v_max + v_cndmask when the
// side of the pair is a per-lane property (keys are never NaN: the inputs are validated)
__device__ __forceinline__ float minmax_by_limit(float a, float b, float lim) { return __builtin_amdgcn_fmed3f(a, b, lim); }

// keys (and weights) of one line across the wave, blocked layout; ascending over element index l*M + j afterwards
// TIEBREAK: the payload is the element index and equal keys are ordered by it (= a stable sort by key, the order the
// reference's sort and the oracle produce; the backward kernels need it to hand tied neighbours the same coefficients)
// LL: lanes per line (64: one line across the wavefront; 16: four independent lines, one per DPP row -- every exchange of its
// merge levels is then a single DPP move)
template <int M, bool WEIGHTED, bool TIEBREAK = false, int LL = kWave>
struct WaveLine {
  float k[M];
  float w[WEIGHTED ? M : 1];

  // exchange with another lane: this lane keeps the smaller (lower == true) or the larger key of each pair
  template <int JREV, int MASK>
  __device__ __forceinline__ void exchange(bool lower) {
#if FSW_PERMLANE_SWAP
    // distances 16 and 32 of the half-cleaners (lower == ((lane & MASK) == 0)): gfx950's v_permlane16/32_swap puts both members
    // of a pair into ONE lane for two registers at a time -- swap, v_min + v_max, swap back: two instructions per key like the DPP
    // path, and no trip through the LDS crossbar (ds_swizzle / ds_bpermute + wait) that these two distances needed
    if constexpr (!WEIGHTED && JREV == 0 && (MASK == 16 || MASK == 32) && M % 2 == 0) {
#pragma unroll
      for (int j = 0; j < M; j += 2) {
        const unsigned a = __float_as_uint(k[j]), b = __float_as_uint(k[j + 1]);
        unsigned x, y;
        if constexpr (MASK == 32) {
          auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
          x = r[0]; y = r[1];
        } else {
          auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
          x = r[0]; y = r[1];
        }
        const float mn = fminf(__uint_as_float(x), __uint_as_float(y)), mx = fmaxf(__uint_as_float(x), __uint_as_float(y));
        if constexpr (MASK == 32) {
          auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(mn), __float_as_uint(mx), false, false);
          k[j] = __uint_as_float(q[0]); k[j + 1] = __uint_as_float(q[1]);
        } else {
          auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(mn), __float_as_uint(mx), false, false);
          k[j] = __uint_as_float(q[0]); k[j + 1] = __uint_as_float(q[1]);
        }
      }
      return;
    }
#endif
    float ok[M], ow[WEIGHTED ? M : 1];
    const float lim = lower ? -__builtin_inff() : __builtin_inff();
#pragma unroll
    for (int j = 0; j < M; ++j) {
      ok[j] = xor_lane<MASK>(k[JREV ? M - 1 - j : j]);
      if constexpr (WEIGHTED) ow[j] = xor_lane<MASK>(w[JREV ? M - 1 - j : j]);
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      if constexpr (WEIGHTED && TIEBREAK) {
        bool take = lower ? (ok[j] < k[j]) : (ok[j] > k[j]);
        const int oi = __float_as_int(ow[j]), mi = __float_as_int(w[j]);
        take = take || (ok[j] == k[j] && (lower ? oi < mi : oi > mi));
        k[j] = take ? ok[j] : k[j];
        w[j] = take ? ow[j] : w[j];
      } else if constexpr (WEIGHTED) {
        // the key as in the unit line (one v_med3); the payload follows the key: it moves exactly when the kept key is not this
        // lane's own (ties: both lanes keep their own element).  Three instructions per element next to the two lane moves --
        // `lower ? ok < k : ok > k` compiled to two compares plus selects of the compare results per element
        const float kn = minmax_by_limit(k[j], ok[j], lim);
        w[j] = kn != k[j] ? ow[j] : w[j];
        k[j] = kn;
      } else {
        k[j] = minmax_by_limit(k[j], ok[j], lim);
      }
    }
  }
  __device__ __forceinline__ void cx(int i, int j) {
    if constexpr (WEIGHTED) {
      bool sw = k[j] < k[i];
      if constexpr (TIEBREAK) sw = sw || (k[j] == k[i] && __float_as_int(w[j]) < __float_as_int(w[i]));
      const float ki = k[i], kj = k[j], wi = w[i], wj = w[j];
      k[i] = sw ? kj : ki;
      k[j] = sw ? ki : kj;
      w[i] = sw ? wj : wi;
      w[j] = sw ? wi : wj;
    } else {
      const float lo = fminf(k[i], k[j]), hi = fmaxf(k[i], k[j]);
      k[i] = lo;
      k[j] = hi;
    }
  }
  __device__ __forceinline__ void sort() {
    const int lane = lane_id();
    // every lane: its own M elements
    if constexpr (WEIGHTED && TIEBREAK) {
      IndexedNet<M> net;
#pragma unroll
      for (int j = 0; j < M; ++j) { net.k[j] = k[j]; net.w[j] = w[j]; }
      sort_network<M>(net);
#pragma unroll
      for (int j = 0; j < M; ++j) { k[j] = net.k[j]; w[j] = net.w[j]; }
    } else if constexpr (WEIGHTED) {
      PairNet<M> net;
#pragma unroll
      for (int j = 0; j < M; ++j) { net.k[j] = k[j]; net.w[j] = w[j]; }
      sort_network<M>(net);
#pragma unroll
      for (int j = 0; j < M; ++j) { k[j] = net.k[j]; w[j] = net.w[j]; }
    } else {
      KeyNet<M> net;
#pragma unroll
      for (int j = 0; j < M; ++j) net.k[j] = k[j];
      sort_network<M>(net);
#pragma unroll
      for (int j = 0; j < M; ++j) k[j] = net.k[j];
    }
    merge_levels<2>(lane);
  }
  // merge levels: sorted runs of LANES/2 lanes -> runs of LANES lanes (template recursion: every exchange mask is a
  // compile-time constant, so the compiler can use DPP / swizzles for the short ones)
  template <int LANES>
  __device__ __forceinline__ void merge_levels(int lane) {
    if constexpr (LANES <= LL) {
      exchange<1, LANES - 1>((lane & (LANES >> 1)) == 0);          // element i against i ^ (LANES*M - 1)
      half_cleaners<(LANES >> 2)>(lane);
      in_register_merge<M, 0>();
      merge_levels<LANES * 2>(lane);
    }
  }
  // the 64 M elements form a bitonic sequence whose halves were already separated: finish the merge
  __device__ __forceinline__ void merge_chunk() {
    const int lane = lane_id();
    half_cleaners<(kWave >> 1)>(lane);
    in_register_merge<M, 0>();
  }
  // bitonic merge of the lane's registers BASE .. BASE + LEN - 1 (a bitonic sequence): half-cleaners while the length is even,
  // a 3-sorter for an odd remainder -- M = 32 / 16 / 8 are the classic log2(M) stages, M = 24 / 12 / 6 (lines of 1.5 x a
  // power of two) halve down to groups of three
  template <int LEN, int BASE>
  __device__ __forceinline__ void in_register_merge() {
    if constexpr (LEN >= 2 && LEN % 2 == 0) {
#pragma unroll
      for (int j = 0; j < LEN / 2; ++j) cx(BASE + j, BASE + j + LEN / 2);
      in_register_merge<LEN / 2, BASE>();
      in_register_merge<LEN / 2, BASE + LEN / 2>();
    } else if constexpr (LEN == 3) {
      cx(BASE, BASE + 1);
      cx(BASE + 1, BASE + 2);
      cx(BASE, BASE + 1);
    } else {
      static_assert(LEN == 1, "keys per lane: a power of two, or three times one");
    }
  }
  template <int ST>
  __device__ __forceinline__ void half_cleaners(int lane) {
    if constexpr (ST >= 1) {
      exchange<0, ST>((lane & ST) == 0);
      half_cleaners<(ST >> 1)>(lane);
    }
  }
};

// The same line layout and merge structure for 64-bit words (pack_key_index: key, then element index): the backward
// kernels sort these -- one unsigned 64-bit compare per pair orders by key with ties by index (= the reference's stable
// order), and the element index comes back out of the low word.
template <int M>
struct WaveLine64 {
  unsigned long long e[M];

  template <int MASK>
  static __device__ __forceinline__ unsigned long long xor_lane64(unsigned long long v) {
    const float lo = xor_lane<MASK>(__uint_as_float((unsigned int)v));
    const float hi = xor_lane<MASK>(__uint_as_float((unsigned int)(v >> 32)));
    return ((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo);
  }
  template <int JREV, int MASK>
  __device__ __forceinline__ void exchange(bool lower) {
    unsigned long long o[M];
#pragma unroll
    for (int j = 0; j < M; ++j) o[j] = xor_lane64<MASK>(e[JREV ? M - 1 - j : j]);
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const bool take = (o[j] < e[j]) == lower;   // the words of a line are distinct (element index in the low half): one compare
      e[j] = take ? o[j] : e[j];
    }
  }
  __device__ __forceinline__ void cx(int i, int j) {
    const unsigned long long a = e[i], b = e[j];
    const bool sw = b < a;
    e[i] = sw ? b : a;
    e[j] = sw ? a : b;
  }
  template <int ST>
  __device__ __forceinline__ void half_cleaners(int lane) {
    if constexpr (ST >= 1) {
      exchange<0, ST>((lane & ST) == 0);
      half_cleaners<(ST >> 1)>(lane);
    }
  }
  __device__ __forceinline__ void in_register_cleaners() {
#pragma unroll
    for (int st = M >> 1; st >= 1; st >>= 1)
#pragma unroll
      for (int j = 0; j < M; ++j)
        if ((j & st) == 0) cx(j, j + st);
  }
  template <int LANES>
  __device__ __forceinline__ void merge_levels(int lane) {
    if constexpr (LANES <= kWave) {
      exchange<1, LANES - 1>((lane & (LANES >> 1)) == 0);
      half_cleaners<(LANES >> 2)>(lane);
      in_register_cleaners();
      merge_levels<LANES * 2>(lane);
    }
  }
  __device__ __forceinline__ void sort() {
    U64Net<M> net;
#pragma unroll
    for (int j = 0; j < M; ++j) net.e[j] = e[j];
    sort_network<M>(net);
#pragma unroll
    for (int j = 0; j < M; ++j) e[j] = net.e[j];
    merge_levels<2>(lane_id());
  }
  __device__ __forceinline__ void merge_chunk() {
    half_cleaners<(kWave >> 1)>(lane_id());
    in_register_cleaners();
  }
  __device__ __forceinline__ float key(int j) const { return from_orderable_bits((unsigned int)(e[j] >> 32)); }
  __device__ __forceinline__ int index(int j) const { return (int)(unsigned int)e[j]; }
};

}  // namespace fsw
