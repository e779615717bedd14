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
#include "fsw_common.h"

namespace fsw {

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
// Data layout of a tile (TILE = 4 wavefronts x WCH elements): wavefront wv owns memory offsets [wv * WCH, (wv + 1) * WCH) of
// the tile's memory range, and inside it lane l holds, for q < Q, the four elements q * 256 + 4 l .. 4 l + 3: every
// wave-instruction reads or writes 64 x 16 B of CONSECUTIVE memory (32 B per lane for 8-byte items).  The scan runs in logical order,
// which is memory order when !REV and the exact mirror image when REV (waves, q, lanes and the four elements of a lane
// all descending), so one code path with mirrored indices serves both directions.
// VEC: 16-byte accesses (pointers 16-byte aligned and the tile's memory range starting on a multiple of 4 elements).
template <bool REV>
__device__ __forceinline__ int logical_lane() { return REV ? kWave - 1 - lane_id() : lane_id(); }

// value of the logically previous lane (undefined for logical lane 0)
template <bool REV, class T>
__device__ __forceinline__ T from_prev_lane(T v, int off = 1) { return REV ? __shfl_down(v, off) : __shfl_up(v, off); }

template <bool REV, class V>
__device__ __forceinline__ SegPair<V> wave_segscan_dir(SegPair<V> v) {
  const int ll = logical_lane<REV>();
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    SegPair<V> l;
    l.s = from_prev_lane<REV>(v.s, off);
    l.f = from_prev_lane<REV>(v.f, off);
    if (ll >= off) v = seg_combine(l, v);
  }
  return v;
}

template <class V, class I, bool REV, bool VEC>
__global__ void __launch_bounds__(kSegThreads) k_segscan_chained(const V* __restrict__ values, V* __restrict__ out,
                                                                 const I* __restrict__ ids, int64_t n,
                                                                 unsigned long long* __restrict__ desc, int64_t ntiles) {
  constexpr int Q = sizeof(V) == 4 ? 4 : 2;           // 16 (float32) / 8 (float64) elements per thread
  constexpr int WCH = kWave * 4 * Q;                  // elements per wavefront and tile
  constexpr int NWV = kSegThreads / kWave;
  constexpr int TILE = NWV * WCH;
  constexpr int DW = DescWords<V>::kWords;
  __shared__ SegPair<V> wave_tot[NWV];
  __shared__ V carry_s;
  const int lane = lane_id(), wv = wave_id();
  const int ll = logical_lane<REV>();
  const int lwv = REV ? NWV - 1 - wv : wv;            // logical wave index
  const int lastlane = REV ? 0 : kWave - 1;           // physical lane that is logically last
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // memory range of the tile: [M0, M0 + TILE); logical element L of the whole array sits at memory n - 1 - L when REV
    const int64_t M0 = REV ? n - (tile + 1) * TILE : tile * TILE;
    const int64_t mw = M0 + (int64_t)wv * WCH;        // this wavefront's memory range starts here
    V val[Q][4];
    I id[Q][4];
    const bool whole = M0 >= 0 && M0 + TILE <= n;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int64_t m = mw + q * 256 + lane * 4;
      if (VEC && whole) {
        typedef V vecv __attribute__((ext_vector_type(4)));
        typedef I veci __attribute__((ext_vector_type(4)));
        const vecv tv = *reinterpret_cast<const vecv*>(values + m);
        const veci ti = *reinterpret_cast<const veci*>(ids + m);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          val[q][u] = tv[u];
          id[q][u] = ti[u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = m + u >= 0 && m + u < n;
          val[q][u] = ok ? values[m + u] : V(0);
          id[q][u] = ok ? ids[m + u] : I(0);
        }
      }
    }
    // id of the element logically just before this wavefront's chunk (wave-uniform)
    const int64_t L0 = tile * TILE + (int64_t)lwv * WCH;          // first logical element of the chunk
    const bool chunk_has_prev = L0 > 0 && L0 < n;
    I idchunk = I(0);
    if (chunk_has_prev) idchunk = ids[REV ? n - L0 : L0 - 1];
    // heads and the lane-local scans, q in logical order
    SegPair<V> pre[Q];      // everything of this wavefront's chunk that logically precedes the lane's four elements of q
    SegPair<V> wcarry;      // running aggregate of the chunk
    wcarry.s = V(0);
    wcarry.f = 0;
    unsigned heads = 0;     // bit q * 4 + u
#pragma unroll
    for (int lq = 0; lq < Q; ++lq) {
      const int q = REV ? Q - 1 - lq : lq;
      const int ul = REV ? 0 : 3;                                  // logically last of the lane's four
      // predecessor id of the lane's logically first element
      I idp = from_prev_lane<REV>(id[q][ul]);
      bool has_prev = true;
      if (lq == 0) {
        if (ll == 0) {
          idp = idchunk;
          has_prev = chunk_has_prev;
        }
      } else {
        const int qp = REV ? q + 1 : q - 1;
        const I wrap = __shfl(id[qp][ul], lastlane);
        if (ll == 0) idp = wrap;
      }
      SegPair<V> agg;
      agg.s = V(0);
      agg.f = 0;
#pragma unroll
      for (int lu = 0; lu < 4; ++lu) {
        const int u = REV ? 3 - lu : lu;
        const int64_t m = mw + q * 256 + lane * 4 + u;
        const bool valid = m >= 0 && m < n;
        bool head;
        if (lu == 0) head = !has_prev || id[q][u] != idp;
        else head = id[q][u] != id[q][REV ? u + 1 : u - 1];
        head = head && valid;
        heads |= (unsigned)head << (q * 4 + u);
        SegPair<V> e;
        e.s = valid ? val[q][u] : V(0);
        e.f = head;
        agg = seg_combine(agg, e);
      }
      const SegPair<V> inc = wave_segscan_dir<REV>(agg);            // inclusive over the lanes, logical order
      SegPair<V> ex;
      ex.s = from_prev_lane<REV>(inc.s);
      ex.f = from_prev_lane<REV>(inc.f);
      if (ll == 0) {
        ex.s = V(0);
        ex.f = 0;
      }
      pre[q] = seg_combine(wcarry, ex);
      SegPair<V> tot;
      tot.s = __shfl(inc.s, lastlane);
      tot.f = __shfl(inc.f, lastlane);
      wcarry = seg_combine(wcarry, tot);
    }
    // across the wavefronts of the tile
    if (lane == 0) wave_tot[lwv] = wcarry;
    __syncthreads();
    SegPair<V> wpre, tagg;
    wpre.s = V(0);
    wpre.f = 0;
    tagg = wpre;
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      if (i == lwv) wpre = tagg;
      tagg = seg_combine(tagg, wave_tot[i]);
    }
    if (threadIdx.x == 0) {   // the tile's aggregate
      if (tile == 0) desc_store(desc, tagg.s, kDescPrefix | ((unsigned)tagg.f << 2));
      else desc_store(desc + tile * DW, tagg.s, kDescAggregate | ((unsigned)tagg.f << 2));
    }
    // decoupled look-back by wavefront 0: 64 predecessors per step, newest in lane 0
    if (wv == 0) {
      V carry = V(0);
      int64_t t0 = tile - 1;
      bool done = tile == 0;
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
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      V r = seg_combine(wenter, pre[q]).s;
      V res[4];
#pragma unroll
      for (int lu = 0; lu < 4; ++lu) {
        const int u = REV ? 3 - lu : lu;
        r = ((heads >> (q * 4 + u)) & 1u) ? val[q][u] : r + val[q][u];
        res[u] = r;
      }
      const int64_t m = mw + q * 256 + lane * 4;
      if (VEC && whole) {
        typedef V vecv __attribute__((ext_vector_type(4)));
        vecv tv;
#pragma unroll
        for (int u = 0; u < 4; ++u) tv[u] = res[u];
        *reinterpret_cast<vecv*>(out + m) = tv;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (m + u >= 0 && m + u < n) out[m + u] = res[u];
      }
    }
    __syncthreads();   // carry_s / wave_tot are reused by the next tile
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
  static int grid_cache = 0;
  if (!grid_cache) {
    int dev = 0, cus = 0, per_cu = 0;
    FSW_CHECK_HIP(hipGetDevice(&dev));
    FSW_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    FSW_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_segscan_chained<V, I, REV, true>, kSegThreads, 0));
    grid_cache = std::max(1, cus) * std::max(1, std::min(per_cu - 1, kSegMaxWgPerCu));   // one below the query: it can be one high (MI355X guide)
  }
  const unsigned grid = (unsigned)std::min<int64_t>(ntiles, grid_cache);
  // 16-byte accesses: aligned pointers, and when scanning from the end the tiles must start on a multiple of 4 elements
  const bool vec = ((uintptr_t)values % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)ids % 16 == 0) && (!REV || n % 4 == 0);
  if (vec)
    k_segscan_chained<V, I, REV, true><<<grid, kSegThreads, 0, stream>>>((const V*)values, (V*)out, (const I*)ids, n, desc, ntiles);
  else
    k_segscan_chained<V, I, REV, false><<<grid, kSegThreads, 0, stream>>>((const V*)values, (V*)out, (const I*)ids, n, desc, ntiles);
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
static void legacy_report(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
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
