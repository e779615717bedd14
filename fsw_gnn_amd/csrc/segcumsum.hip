// Stand-alone segmented cumulative sum + the reference's legacy entry points.  gfx950.
//
// fsw_segcumsum replaces segcumsum / segcumsum_cuda (reference fsw_embedding.py:2795-3012): the
// reference scans each 256-element block with a Hillis-Steele loop in shared memory that re-reads the
// int64 ids from global memory every round, then recurses over block sums level by level with a
// device-wide synchronise around every launch (fsw_embedding.cu:34-117, 194-228): >= 28 bytes of HBM
// traffic per element and 9 launches at 2.56e9 elements.  Here the scan is ONE streaming pass (chained
// scan with decoupled look-back) over (sum, segment-head flag) pairs, which form a monoid
//        (s1, f1) . (s2, f2) = (f2 ? s2 : s1 + s2,  f1 | f2):
// every element is read once and written once (4 + 8 + 4 bytes for float32 values / int64 ids).
//   * a persistent grid (a fixed number of workgroups per CU, all co-resident) walks the tiles in order:
//     workgroup g takes tiles g, g + G, g + 2G, ...  Every predecessor of a tile is therefore owned by a
//     resident workgroup that reaches it first -- no ticket counter, no dependence on dispatch order;
//   * per tile: 16-byte loads of a thread's 16 (float32) / 8 (float64) consecutive elements, in-register
//     scan, wavefront shuffles across the threads, then the tile publishes its AGGREGATE in a 64-bit
//     descriptor word {status | flag, value bits} with one relaxed agent-scope store (an 8-byte granule: no
//     fence; float64 sums take two tagged words), looks back over its predecessors 64 at a time -- the walk
//     stops at the first tile that holds a segment head or has published its inclusive PREFIX -- and
//     publishes its own prefix.  With the short segments of the FSW path (one neighbourhood, ~10 elements)
//     the walk ends at the immediate predecessor's aggregate, which depends on nothing but that tile's data.
// A head is an element whose id differs from its predecessor's (successor's when reverse != 0), exactly
// the restart rule of segcumsum_slow (fsw_embedding.py:3016-3027).
#include <algorithm>
#include <atomic>
#include "fsw_common.h"

namespace fsw {

constexpr int kMaxDevices = 64;
constexpr int kSegThreads = 256;
#ifndef FSW_SEG_WG_PER_CU
#define FSW_SEG_WG_PER_CU 6
#endif
constexpr int kSegMaxWgPerCu = FSW_SEG_WG_PER_CU;

template <class V>
struct SegPair {
  V s;
  int f;
};

template <class V>
__device__ __forceinline__ SegPair<V> seg_combine(SegPair<V> left, SegPair<V> right) {
  SegPair<V> r;
  r.s = right.f ? right.s : left.s + right.s;
  r.f = left.f | right.f;
  return r;
}

// inclusive segmented scan across the lanes of a wave
template <class V>
__device__ __forceinline__ SegPair<V> wave_segscan(SegPair<V> v) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    SegPair<V> l;
    l.s = __shfl_up(v.s, off);
    l.f = __shfl_up(v.f, off);
    if (lane_id() >= off) v = seg_combine(l, v);
  }
  return v;
}

// inclusive segmented scan across all threads of a workgroup (any blockDim <= 1024, thread order)
template <class V>
__device__ __forceinline__ SegPair<V> block_segscan(SegPair<V> v, SegPair<V>* wave_tot /* LDS [16] */) {
  const int wv = threadIdx.x >> 6;
  const int nw = (blockDim.x + kWave - 1) >> 6;
  v = wave_segscan(v);
  if (lane_id() == kWave - 1 || threadIdx.x == blockDim.x - 1) wave_tot[wv] = v;
  __syncthreads();
  SegPair<V> carry;
  carry.s = V(0);
  carry.f = 0;
  for (int i = 0; i < wv && i < nw; ++i) carry = seg_combine(carry, wave_tot[i]);
  __syncthreads();
  return wv ? seg_combine(carry, v) : v;
}

// ---- tile descriptors ------------------------------------------------------------------------------------
// tag = status (1 aggregate, 2 inclusive prefix) | flag << 2 in the high half of every 64-bit word, 32 value bits in the
// low half: float32 sums one word, float64 sums two (low / high half of the double), both carrying the same tag.
constexpr unsigned kDescAggregate = 1u, kDescPrefix = 2u;
template <class V>
struct DescWords {
  static constexpr int kWords = sizeof(V) == 4 ? 1 : 2;
};

__device__ __forceinline__ void desc_store(unsigned long long* d, float v, unsigned tag) {
  __hip_atomic_store(d, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(unsigned long long* d, double v, unsigned tag) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(d, ((unsigned long long)tag << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(d + 1, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns the tag (0 = not published yet, or the two words of a float64 descriptor disagree: poll again)
__device__ __forceinline__ unsigned desc_load(const unsigned long long* d, float& v) {
  const unsigned long long w = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __uint_as_float((unsigned)w);
  return (unsigned)(w >> 32);
}
__device__ __forceinline__ unsigned desc_load(const unsigned long long* d, double& v) {
  const unsigned long long w0 = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long w1 = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __longlong_as_double((long long)((w1 << 32) | (w0 & 0xffffffffull)));
  return (w0 >> 32) == (w1 >> 32) ? (unsigned)(w0 >> 32) : 0u;
}

// ---- the single-pass kernel ----------------------------------------------------------------------------------
// Data layout of a tile (TILE = 4 wavefronts x WCH elements).  Everything below is written in LOGICAL element order: logical
// element L of the array sits at memory L when !REV and at memory n - 1 - L when REV, so a reverse scan is the same code with
// mirrored addresses (no direction-dependent shuffle).  Wavefront w owns the logical range [w * WCH, (w + 1) * WCH) of its tile
// and walks it in Q groups of 64 * EPL elements; in group q lane l holds the EPL consecutive elements q * 64 * EPL + l * EPL + u.
// EPL = 16 bytes / the wider of (value, id): every load and store instruction of a wavefront covers ONE contiguous run of
// memory (64 x 16 B for the wider type, 64 x 8 B for float32 values beside int64 ids) -- no instruction touches a fraction of
// the lines it spans (the first layout of this kernel read int64 ids as two 16-byte pieces per lane at a 32-byte stride).
// Cross-lane traffic is DPP only: row_shr 1/2/4/8 inside the rows of 16 lanes, row_bcast:15 and row_bcast:31 across them,
// wave_shr:1 for the neighbour's last element; lanes without a source receive the monoid's identity (0, no head), so no
// lane-index tests are needed.
template <unsigned CTRL, unsigned ROW_MASK, class T>
__device__ __forceinline__ T dpp_or_zero(T v) {
  static_assert(sizeof(T) % 4 == 0, "32-bit pieces");
  unsigned w[sizeof(T) / 4];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w[i], CTRL, ROW_MASK, 0xf, false);
  T r;
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}
// lane `src` of a 4- or 8-byte value as a wave-uniform value (32-bit readlanes)
template <class T>
__device__ __forceinline__ T readlane_any(T v, int src) {
  unsigned w[sizeof(T) / 4];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = (unsigned)__builtin_amdgcn_readlane((int)w[i], src);
  T r;
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}
constexpr unsigned kDppRowShr = 0x110, kDppWaveShr1 = 0x138, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

template <unsigned CTRL, unsigned ROW_MASK, class V>
__device__ __forceinline__ SegPair<V> seg_dpp_step(SegPair<V> v) {
  SegPair<V> l;
  l.s = dpp_or_zero<CTRL, ROW_MASK>(v.s);
  l.f = dpp_or_zero<CTRL, ROW_MASK>(v.f);
  return seg_combine(l, v);
}

// inclusive segmented scan over the 64 lanes, lane order
template <class V>
__device__ __forceinline__ SegPair<V> wave_segscan_dpp(SegPair<V> v) {
  v = seg_dpp_step<kDppRowShr + 1, 0xf>(v);
  v = seg_dpp_step<kDppRowShr + 2, 0xf>(v);
  v = seg_dpp_step<kDppRowShr + 4, 0xf>(v);
  v = seg_dpp_step<kDppRowShr + 8, 0xf>(v);
  v = seg_dpp_step<kDppRowBcast15, 0xa>(v);   // lane 15 -> row 1, lane 47 -> row 3
  v = seg_dpp_step<kDppRowBcast31, 0xc>(v);   // lane 31 -> rows 2 and 3
  return v;
}

#ifndef FSW_SEG_ABL
#define FSW_SEG_ABL 0   // timing experiments only: 1 no look-back (carry 0), 2 no scan (values stored as read), 4 ids not read
#endif
#ifndef FSW_SEG_HALO
#define FSW_SEG_HALO 1    // carry from the previous tile's last group when it holds a segment head (no descriptor round trip)
#endif
#ifndef FSW_SEG_PREFETCH
#define FSW_SEG_PREFETCH 0   // 1: issue the loads of the workgroup's next tile before the current tile is scanned.  Measured slower
#endif                       // (150 registers -> 3 workgroups per CU): 4.68 against 5.27 TB/s at 2.56e9 elements; 6 per CU cover the loads

template <class V, class I>
struct SegTile {
  static constexpr int EPL = 16 / (int)(sizeof(V) > sizeof(I) ? sizeof(V) : sizeof(I));   // elements per lane and group
  static constexpr int EPT = sizeof(V) == 4 ? 16 : 8;                                     // elements per thread and tile
  static constexpr int Q = EPT / EPL;
  static constexpr int GRP = kWave * EPL;                                                 // elements per group
  static constexpr int WCH = kWave * EPT;                                                 // elements per wavefront and tile
  static constexpr int NWV = kSegThreads / kWave;
  static constexpr int TILE = NWV * WCH;
  V val[Q][EPL];
  I id[Q][EPL];
  I idprev;          // id of the element logically just before this wavefront's chunk (wave-uniform)
  bool has_prev;
  V hval[EPL];       // wavefront 0 only: the HALO, the last group of 64 * EPL elements of the previous tile (lane l: its elements
  I hid[EPL];        // l * EPL .. of that group) -- see the look-back
};

#ifndef FSW_SEG_MINWAVES
#define FSW_SEG_MINWAVES 1   // waves per SIMD the kernel is compiled for (register cap); measured: tools/exp_r3_segscan_registers.sh
#endif
template <class V, class I, bool REV, bool VEC>
__global__ void __launch_bounds__(kSegThreads, FSW_SEG_MINWAVES) k_segscan_chained(const V* __restrict__ values, V* __restrict__ out,
                                                                 const I* __restrict__ ids, int64_t n,
                                                                 unsigned long long* __restrict__ desc, int64_t ntiles, int use_halo) {
  // use_halo = 0 for an IN-PLACE scan (out == values): the previous tile belongs to another workgroup, which may already have
  // overwritten its values with its results -- the halo must be raw input, so such calls take the descriptor path for every tile
  using T = SegTile<V, I>;
  constexpr int EPL = T::EPL, Q = T::Q, GRP = T::GRP, WCH = T::WCH, NWV = T::NWV, TILE = T::TILE;
  constexpr int DW = DescWords<V>::kWords;
  typedef V vecv __attribute__((ext_vector_type(EPL)));
  typedef I veci __attribute__((ext_vector_type(EPL)));
  __shared__ SegPair<V> wave_tot[NWV];
  __shared__ V carry_s;
  const int lane = lane_id(), wv = wave_id();

  // memory index of logical element L (L may lie past the end: the caller tests)
  auto mem = [&](int64_t L) { return REV ? n - 1 - L : L; };
  // logical index of the lane's first element of group q in `tile`
  auto first_of = [&](int64_t tile, int q) { return tile * TILE + (int64_t)wv * WCH + q * GRP + lane * EPL; };

  auto load_tile = [&](int64_t tile, T& t) {
    const bool whole = (tile + 1) * TILE <= n;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int64_t L = first_of(tile, q);
      if (VEC && whole) {
        // the lane's EPL elements are contiguous in memory either way; reversed, the vector starts at the LAST logical one
        const int64_t m0 = REV ? mem(L + EPL - 1) : L;
        const vecv tv = *reinterpret_cast<const vecv*>(values + m0);
        veci ti;
        if constexpr (FSW_SEG_ABL & 4) {
#pragma unroll
          for (int u = 0; u < EPL; ++u) ti[u] = (I)((m0 + u) / 10);
        } else {
          ti = *reinterpret_cast<const veci*>(ids + m0);
        }
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
          t.val[q][u] = tv[REV ? EPL - 1 - u : u];
          t.id[q][u] = ti[REV ? EPL - 1 - u : u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
          const bool ok = L + u < n;
          t.val[q][u] = ok ? values[mem(L + u)] : V(0);
          t.id[q][u] = ok ? ids[mem(L + u)] : I(0);
        }
      }
    }
    const int64_t L0 = tile * TILE + (int64_t)wv * WCH;
    t.has_prev = L0 > 0 && L0 < n;
    t.idprev = t.has_prev ? ids[mem(L0 - 1)] : I(0);
    if (FSW_SEG_HALO && use_halo && wv == 0 && tile > 0) {      // wave-uniform; the halo lies wholly inside the array (tile * TILE <= n - 1)
      const int64_t L = tile * TILE - GRP + lane * EPL;
      if (VEC) {
        const int64_t m0 = REV ? mem(L + EPL - 1) : L;
        const vecv tv = *reinterpret_cast<const vecv*>(values + m0);
        const veci ti = *reinterpret_cast<const veci*>(ids + m0);
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
          t.hval[u] = tv[REV ? EPL - 1 - u : u];
          t.hid[u] = ti[REV ? EPL - 1 - u : u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < EPL; ++u) {
          t.hval[u] = values[mem(L + u)];
          t.hid[u] = ids[mem(L + u)];
        }
      }
    }
  };

  T cur;
  int64_t tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile, cur);
  for (; tile < ntiles; tile += gridDim.x) {
    // ---- heads (bit q * EPL + u) -- the ids are dead afterwards ----
    unsigned heads = 0;
    I carry_id = cur.idprev;                      // id logically before lane 0 of group q (wave-uniform)
    bool carry_has = cur.has_prev;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      I idp = dpp_or_zero<kDppWaveShr1, 0xf>(cur.id[q][EPL - 1]);
      bool hp = true;
      if (lane == 0) {
        idp = carry_id;
        hp = carry_has;
      }
      const int64_t L = first_of(tile, q);
#pragma unroll
      for (int u = 0; u < EPL; ++u) {
        const bool valid = L + u < n;
        const bool head = valid && (u == 0 ? (!hp || cur.id[q][0] != idp) : cur.id[q][u] != cur.id[q][u - 1]);
        heads |= (unsigned)head << (q * EPL + u);
      }
      carry_id = readlane_any(cur.id[q][EPL - 1], kWave - 1);   // the last lane's last id, wave-uniform
      carry_has = true;
    }
    V val[Q][EPL];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int u = 0; u < EPL; ++u) val[q][u] = (first_of(tile, q) + u < n) ? cur.val[q][u] : V(0);
    // The halo (wavefront 0): when the last group of the previous tile holds a segment head, the sum from that head to the end of
    // the previous tile IS the carry into this tile -- the value the predecessor publishes as its aggregate, formed by the same
    // operations in the same order -- and it depends on nothing but 64 * EPL elements this wavefront has just read itself.  With
    // neighbourhood-sized segments that is every tile: the look-back below (a descriptor round trip through the fabric, ~2 us that
    // the whole workgroup waits for: 26 % of the kernel at 2.56e8 elements) then never runs.  Tiles keep publishing their
    // descriptors: a tile whose halo holds no head (segments longer than 64 * EPL) falls back to them.  The first halo element counts
    // as "no head" (its predecessor is not read): conservative.
    SegPair<V> halo;
    halo.s = V(0);
    halo.f = 0;
    if (FSW_SEG_HALO && use_halo && wv == 0 && tile > 0) {
      const I idp = dpp_or_zero<kDppWaveShr1, 0xf>(cur.hid[EPL - 1]);
#pragma unroll
      for (int u = 0; u < EPL; ++u) {
        SegPair<V> e;
        e.s = cur.hval[u];
        e.f = u == 0 ? (lane > 0 && cur.hid[0] != idp) : (cur.hid[u] != cur.hid[u - 1]);
        halo = seg_combine(halo, e);
      }
      halo = wave_segscan_dpp(halo);
      halo.s = readlane_any(halo.s, kWave - 1);
      halo.f = __builtin_amdgcn_readlane(halo.f, kWave - 1);
    }
#if FSW_SEG_PREFETCH
    // the next tile of this workgroup: its loads are in flight while this tile is scanned, looked back and stored
    if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x, cur);
#endif
    // ---- lane-local scans and the scan across the lanes, group by group ----
    SegPair<V> pre[Q];      // everything of this wavefront's chunk that logically precedes the lane's elements of group q
    SegPair<V> wcarry;      // running aggregate of the chunk
    wcarry.s = V(0);
    wcarry.f = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      SegPair<V> agg;
      agg.s = V(0);
      agg.f = 0;
#pragma unroll
      for (int u = 0; u < EPL; ++u) {
        SegPair<V> e;
        e.s = val[q][u];
        e.f = (heads >> (q * EPL + u)) & 1u;
        agg = seg_combine(agg, e);
      }
      const SegPair<V> inc = wave_segscan_dpp(agg);
      SegPair<V> ex;
      ex.s = dpp_or_zero<kDppWaveShr1, 0xf>(inc.s);
      ex.f = dpp_or_zero<kDppWaveShr1, 0xf>(inc.f);
      pre[q] = seg_combine(wcarry, ex);
      SegPair<V> tot;     // the last lane's inclusive value, wave-uniform
      tot.s = readlane_any(inc.s, kWave - 1);
      tot.f = __builtin_amdgcn_readlane(inc.f, kWave - 1);
      wcarry = seg_combine(wcarry, tot);
    }
    // ---- across the wavefronts of the tile ----
    if (lane == 0) wave_tot[wv] = wcarry;
    __syncthreads();
    SegPair<V> wpre, tagg;
    wpre.s = V(0);
    wpre.f = 0;
    tagg = wpre;
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      if (i == wv) wpre = tagg;
      tagg = seg_combine(tagg, wave_tot[i]);
    }
    if (threadIdx.x == 0) {   // the tile's aggregate
      if (tile == 0) desc_store(desc, tagg.s, kDescPrefix | ((unsigned)tagg.f << 2));
      else desc_store(desc + tile * DW, tagg.s, kDescAggregate | ((unsigned)tagg.f << 2));
    }
    // decoupled look-back by wavefront 0: 64 predecessors per step, newest in lane 0
    if (wv == 0) {
      V carry = halo.f ? halo.s : V(0);
      int64_t t0 = tile - 1;
      bool done = tile == 0 || halo.f || (FSW_SEG_ABL & 1);
      while (!done) {
        const int64_t t = t0 - lane;
        V dv = V(0);
        unsigned tag = kDescPrefix;                 // lanes before tile 0: a stopper that contributes nothing
        if (t >= 0) {
          do {
            tag = desc_load(desc + t * DW, dv);
            if ((tag & 3u) == 0) __builtin_amdgcn_s_sleep(1);
          } while ((tag & 3u) == 0);
        }
        const bool stop = (tag & 3u) == kDescPrefix || (tag & 4u);
        const unsigned long long sm = __ballot(stop);
        const int lstop = sm ? __ffsll((long long)sm) - 1 : kWave;     // first (newest) stopper
        V part = (lane <= lstop && t >= 0) ? dv : V(0);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        carry = part + carry;
        done = sm != 0;
        t0 -= kWave;
      }
      if (lane == 0) {
        carry_s = carry;
        if (tile > 0) desc_store(desc + tile * DW, tagg.f ? tagg.s : carry + tagg.s, kDescPrefix | ((unsigned)tagg.f << 2));
      }
    }
    __syncthreads();
    SegPair<V> tc;
    tc.s = carry_s;
    tc.f = 0;
    const SegPair<V> wenter = seg_combine(tc, wpre);               // everything before this wavefront's chunk
    const bool whole = (tile + 1) * TILE <= n;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      V r = seg_combine(wenter, pre[q]).s;
      V res[EPL];
#pragma unroll
      for (int u = 0; u < EPL; ++u) {
        r = ((heads >> (q * EPL + u)) & 1u) ? val[q][u] : r + val[q][u];
        res[u] = (FSW_SEG_ABL & 2) ? val[q][u] : r;
      }
      const int64_t L = first_of(tile, q);
      if (VEC && whole) {
        vecv tv;
#pragma unroll
        for (int u = 0; u < EPL; ++u) tv[REV ? EPL - 1 - u : u] = res[u];
        *reinterpret_cast<vecv*>(out + (REV ? mem(L + EPL - 1) : L)) = tv;
      } else {
#pragma unroll
        for (int u = 0; u < EPL; ++u)
          if (L + u < n) out[mem(L + u)] = res[u];
      }
    }
    __syncthreads();   // carry_s / wave_tot are reused by the next tile
#if !FSW_SEG_PREFETCH
    if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x, cur);
#endif
  }
}

// ---- legacy kernels: semantics of reference fsw_embedding.cu:34-117 ------------------------------------
// Block-local segmented inclusive scan of `size` values (one element per thread) restarted at id changes
// INSIDE the block; the last thread of a full block exports its running sum and id for the next level.
template <class V>
__global__ void k_legacy_block_scan(V* __restrict__ values, const int64_t* __restrict__ segment_ids, int64_t size,
                                    V* __restrict__ block_sums_out, int64_t* __restrict__ block_last_ids_out,
                                    bool return_next_level) {
  __shared__ SegPair<V> wave_tot[16];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = i < size;
  SegPair<V> e;
  e.s = ok ? values[i] : V(0);
  e.f = ok && threadIdx.x > 0 && segment_ids[i] != segment_ids[i - 1];
  e = block_segscan(e, wave_tot);
  if (ok) values[i] = e.s;
  if (return_next_level && ok && threadIdx.x == blockDim.x - 1) {
    block_sums_out[blockIdx.x] = e.s;
    block_last_ids_out[blockIdx.x] = segment_ids[i];
  }
}

template <class V>
__global__ void k_legacy_add_block_sums(V* __restrict__ output, const V* __restrict__ block_sums,
                                        const int64_t* __restrict__ segment_ids, const int64_t* __restrict__ block_last_id,
                                        int64_t size) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < size && blockIdx.x >= 1 && block_last_id[blockIdx.x - 1] == segment_ids[i]) output[i] += block_sums[blockIdx.x - 1];
}

template <class V>
static int64_t seg_tile_elems() { return (int64_t)kSegThreads * (sizeof(V) == 4 ? 16 : 8); }   // = TILE of k_segscan_chained

template <class V, class I, bool REV>
static int launch_segscan(const void* values, void* out, const void* ids, int64_t n, void* ws, hipStream_t stream) {
  const int64_t ntiles = ceil_div(n, seg_tile_elems<V>());
  unsigned long long* desc = reinterpret_cast<unsigned long long*>(ws);
  FSW_CHECK_HIP(hipMemsetAsync(desc, 0, sizeof(unsigned long long) * DescWords<V>::kWords * (size_t)ntiles, stream));
  // persistent grid: every workgroup must be resident (the look-back spins on predecessors).  4 workgroups of 256
  // threads per CU need <= 128 registers and no LDS to speak of; the occupancy query guards against surprises.
  // The grid is a property of (kernel instantiation, DEVICE): one atomic slot per device ordinal, so a process that drives
  // several GPUs -- or several host threads -- never sizes the grid of one device by another's CU count (a benign race:
  // every writer stores the same value).
  static std::atomic<int> grid_cache[kMaxDevices];
  int dev = 0;
  FSW_CHECK_HIP(hipGetDevice(&dev));
  int resident = (dev >= 0 && dev < kMaxDevices) ? grid_cache[dev].load(std::memory_order_relaxed) : 0;
  if (!resident) {
    int cus = 0, per_cu = 0;
    FSW_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    FSW_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_segscan_chained<V, I, REV, true>, kSegThreads, 0));
    resident = std::max(1, cus) * std::max(1, std::min(per_cu - 1, kSegMaxWgPerCu));   // one below the query: it can be one high (MI355X guide)
    if (dev >= 0 && dev < kMaxDevices) grid_cache[dev].store(resident, std::memory_order_relaxed);
  }
  const unsigned grid = (unsigned)std::min<int64_t>(ntiles, resident);
  // vector accesses (EPL elements per lane): aligned pointers, and scanning from the end the groups must start on a multiple of EPL
  constexpr int EPL = SegTile<V, I>::EPL;
  const bool vec = ((uintptr_t)values % (EPL * sizeof(V)) == 0) && ((uintptr_t)out % (EPL * sizeof(V)) == 0) &&
                   ((uintptr_t)ids % (EPL * sizeof(I)) == 0) && (!REV || n % EPL == 0);
  // the halo shortcut reads other tiles' INPUT values: only when the output cannot overwrite them (any overlap of the two arrays
  // counts as in place)
  const char* vb = (const char*)values;
  const char* ob = (const char*)out;
  const int use_halo = (ob + n * sizeof(V) <= vb || vb + n * sizeof(V) <= ob) ? 1 : 0;
  if (vec)
    k_segscan_chained<V, I, REV, true><<<grid, kSegThreads, 0, stream>>>((const V*)values, (V*)out, (const I*)ids, n, desc, ntiles, use_halo);
  else
    k_segscan_chained<V, I, REV, false><<<grid, kSegThreads, 0, stream>>>((const V*)values, (V*)out, (const I*)ids, n, desc, ntiles, use_halo);
  FSW_LAUNCH_CHECK();
  return 0;
}

template <class V, class I>
static int run_segcumsum(const void* values, void* out, const void* ids, int64_t n, int reverse, void* ws, hipStream_t stream) {
  return reverse ? launch_segscan<V, I, true>(values, out, ids, n, ws, stream) : launch_segscan<V, I, false>(values, out, ids, n, ws, stream);
}

}  // namespace fsw

using namespace fsw;

extern "C" size_t fsw_segcumsum_workspace_bytes(int64_t n) {
  const size_t nt = (size_t)ceil_div(n > 0 ? n : 1, (int64_t)kSegThreads * 8);   // the smaller (float64) tile
  return ((2 * sizeof(unsigned long long) * nt + 255) / 256) * 256;
}

extern "C" int fsw_segcumsum(int value_dtype, const void* values, void* out, const void* segment_ids, int id_bytes, int64_t n,
                             int reverse, void* workspace, size_t workspace_bytes, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n == 0) return 0;
  FSW_REQUIRE(n > 0 && n < (1ll << 40), "fsw_segcumsum: bad size %lld", (long long)n);
  FSW_REQUIRE(values && out && segment_ids && workspace, "fsw_segcumsum: null pointer");
  FSW_REQUIRE(workspace_bytes >= fsw_segcumsum_workspace_bytes(n), "fsw_segcumsum: workspace too small");
  FSW_REQUIRE(((uintptr_t)workspace & 7) == 0, "fsw_segcumsum: workspace must be 8-byte aligned");
  FSW_REQUIRE((value_dtype == 0 || value_dtype == 1) && (id_bytes == 4 || id_bytes == 8),
              "fsw_segcumsum: value_dtype must be 0 (float32) or 1 (float64), id_bytes 4 or 8");
  reverse = reverse ? 1 : 0;
  if (value_dtype == 0 && id_bytes == 4) return run_segcumsum<float, int32_t>(values, out, segment_ids, n, reverse, workspace, stream);
  if (value_dtype == 0) return run_segcumsum<float, int64_t>(values, out, segment_ids, n, reverse, workspace, stream);
  if (id_bytes == 4) return run_segcumsum<double, int32_t>(values, out, segment_ids, n, reverse, workspace, stream);
  return run_segcumsum<double, int64_t>(values, out, segment_ids, n, reverse, workspace, stream);
}

// ---- legacy ABI (reference fsw_embedding.cu:194, 212, 231): default stream, synchronous, void -----------
// The reference synchronises the whole DEVICE before the launch and again after it (fsw_embedding.cu:197, 208, 215, 226).  The
// first one is what makes its default-stream launch safe when the caller's producer work is queued on a NON-BLOCKING side
// stream (torch's current stream need not be the null stream, and such a stream does not order itself against it).
static bool legacy_sync_before(const char* what) {
  const hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) fsw::set_error("%s: %s", what, hipGetErrorString(e));
  return e == hipSuccess;
}
static void legacy_report(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) fsw::set_error("%s: %s", what, hipGetErrorString(e));
}

extern "C" void segcumsum_wrapper(int dtype, void* values, const int64_t* segment_ids, int64_t size, int64_t max_seg_size,
                                  void* block_sums_out, int64_t* block_last_ids_out, bool return_next_level, int64_t num_blocks,
                                  int64_t threads_per_block, size_t shared_memory_size) {
  (void)max_seg_size;         // the shuffle scan always runs to the block width
  (void)shared_memory_size;   // no dynamic LDS needed
  if (size <= 0 || num_blocks <= 0 || threads_per_block <= 0 || threads_per_block > 1024) {
    fsw::set_error("segcumsum_wrapper: bad launch geometry");
    return;
  }
  if (!legacy_sync_before("segcumsum_wrapper")) return;
  if (dtype == 0)
    k_legacy_block_scan<float><<<(unsigned)num_blocks, (unsigned)threads_per_block, 0, nullptr>>>(
        (float*)values, segment_ids, size, (float*)block_sums_out, block_last_ids_out, return_next_level);
  else
    k_legacy_block_scan<double><<<(unsigned)num_blocks, (unsigned)threads_per_block, 0, nullptr>>>(
        (double*)values, segment_ids, size, (double*)block_sums_out, block_last_ids_out, return_next_level);
  legacy_report("segcumsum_wrapper");
}

extern "C" void add_block_sums_wrapper(int dtype, void* output, const void* block_sums, const int64_t* segment_ids,
                                       const int64_t* block_last_id, int64_t size, int64_t num_blocks, int64_t threads_per_block) {
  if (size <= 0 || num_blocks <= 0 || threads_per_block <= 0 || threads_per_block > 1024) {
    fsw::set_error("add_block_sums_wrapper: bad launch geometry");
    return;
  }
  if (!legacy_sync_before("add_block_sums_wrapper")) return;
  if (dtype == 0)
    k_legacy_add_block_sums<float><<<(unsigned)num_blocks, (unsigned)threads_per_block, 0, nullptr>>>(
        (float*)output, (const float*)block_sums, segment_ids, block_last_id, size);
  else
    k_legacy_add_block_sums<double><<<(unsigned)num_blocks, (unsigned)threads_per_block, 0, nullptr>>>(
        (double*)output, (const double*)block_sums, segment_ids, block_last_id, size);
  legacy_report("add_block_sums_wrapper");
}

// The four launch helpers the reference library also exports (fsw_embedding.cu:125-183; nothing in the reference's Python
// binds them): same kernels, default stream, no synchronisation -- exactly the reference's behaviour.
extern "C" void launch_segcumsum_kernel_float(float* values, const int64_t* segment_ids, int64_t size, int64_t max_seg_size,
                                              float* block_sums_out, int64_t* block_last_ids_out, bool return_next_level,
                                              int64_t num_blocks, int64_t threads_per_block, int64_t shared_memory_size) {
  (void)max_seg_size;
  (void)shared_memory_size;
  if (size <= 0 || num_blocks <= 0 || threads_per_block <= 0 || threads_per_block > 1024) return;
  k_legacy_block_scan<float><<<(unsigned)num_blocks, (unsigned)threads_per_block, 0, nullptr>>>(values, segment_ids, size, block_sums_out,
                                                                                             block_last_ids_out, return_next_level);
}
extern "C" void launch_segcumsum_kernel_double(double* values, const int64_t* segment_ids, int64_t size, int64_t max_seg_size,
                                               double* block_sums_out, int64_t* block_last_ids_out, bool return_next_level,
                                               int64_t num_blocks, int64_t threads_per_block, int64_t shared_memory_size) {
  (void)max_seg_size;
  (void)shared_memory_size;
  if (size <= 0 || num_blocks <= 0 || threads_per_block <= 0 || threads_per_block > 1024) return;
  k_legacy_block_scan<double><<<(unsigned)num_blocks, (unsigned)threads_per_block, 0, nullptr>>>(values, segment_ids, size, block_sums_out,
                                                                                              block_last_ids_out, return_next_level);
}
extern "C" void launch_add_block_sums_kernel_float(float* output, const float* block_sums, const int64_t* segment_ids,
                                                   const int64_t* block_last_id, int64_t size, int64_t num_blocks,
                                                   int64_t threads_per_block) {
  if (size <= 0 || num_blocks <= 0 || threads_per_block <= 0 || threads_per_block > 1024) return;
  k_legacy_add_block_sums<float><<<(unsigned)num_blocks, (unsigned)threads_per_block, 0, nullptr>>>(output, block_sums, segment_ids,
                                                                                                 block_last_id, size);
}
extern "C" void launch_add_block_sums_kernel_double(double* output, const double* block_sums, const int64_t* segment_ids,
                                                    const int64_t* block_last_id, int64_t size, int64_t num_blocks,
                                                    int64_t threads_per_block) {
  if (size <= 0 || num_blocks <= 0 || threads_per_block <= 0 || threads_per_block > 1024) return;
  k_legacy_add_block_sums<double><<<(unsigned)num_blocks, (unsigned)threads_per_block, 0, nullptr>>>(output, block_sums, segment_ids,
                                                                                                  block_last_id, size);
}

extern "C" int get_max_threads_per_block(int device_index) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_index) != hipSuccess) return 0;
  return prop.maxThreadsPerBlock;
}
