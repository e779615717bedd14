// Stand-alone segmented cumulative sum + the reference's legacy entry points.  gfx950.
//
// fsw_segcumsum replaces segcumsum / segcumsum_cuda (reference fsw_embedding.py:2795-3012): the
// reference scans each 256-element block with a Hillis-Steele loop in shared memory that re-reads the
// int64 ids from global memory every round, then recurses over block sums level by level with a
// device-wide synchronise around every launch (fsw_embedding.cu:34-117, 194-228).  Here the scan is a
// reduce-then-scan over (sum, segment-head flag) pairs, which form a monoid
//        (s1, f1) . (s2, f2) = (f2 ? s2 : s1 + s2,  f1 | f2)
// so wavefront shuffles (64 lanes) do the in-block work, no id is read more than twice, and the whole
// thing is three stream-ordered launches whatever the input size:
//   k_tile_reduce  per 2048-element tile: aggregate of the tile
//   k_tile_scan    one workgroup: exclusive scan of the tile aggregates (carry into every tile)
//   k_tile_apply   per tile: in-register scan seeded with the tile's carry, written in place or out of place
// A head is an element whose id differs from its predecessor's (successor's when reverse != 0), exactly
// the restart rule of segcumsum_slow (fsw_embedding.py:3016-3027).
#include "fsw_common.h"

namespace fsw {

constexpr int kSegThreads = 256;
constexpr int kSegItems = 8;
constexpr int kSegTile = kSegThreads * kSegItems;

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

template <class I>
__device__ __forceinline__ bool is_head(const I* __restrict__ ids, int64_t i, int64_t n, bool reverse) {
  // i is the LOGICAL position (scan order); memory position is n-1-i when scanning from the end
  if (i == 0) return true;
  const int64_t a = reverse ? n - 1 - i : i;
  const int64_t b = reverse ? a + 1 : a - 1;
  return ids[a] != ids[b];
}

template <class V, class I>
__global__ void __launch_bounds__(kSegThreads) k_tile_reduce(const V* __restrict__ values, const I* __restrict__ ids,
                                                             int64_t n, int reverse, V* __restrict__ tile_sum,
                                                             int* __restrict__ tile_flag) {
  __shared__ SegPair<V> wave_tot[16];
  const int64_t base = (int64_t)blockIdx.x * kSegTile + (int64_t)threadIdx.x * kSegItems;
  SegPair<V> agg;
  agg.s = V(0);
  agg.f = 0;
#pragma unroll
  for (int j = 0; j < kSegItems; ++j) {
    const int64_t i = base + j;
    if (i < n) {
      SegPair<V> e;
      e.s = values[reverse ? n - 1 - i : i];
      e.f = is_head(ids, i, n, reverse);
      agg = seg_combine(agg, e);
    }
  }
  agg = block_segscan(agg, wave_tot);
  if (threadIdx.x == kSegThreads - 1) {
    tile_sum[blockIdx.x] = agg.s;
    tile_flag[blockIdx.x] = agg.f;
  }
}

template <class V>
__global__ void __launch_bounds__(kSegThreads) k_tile_scan(V* __restrict__ tile_sum, int* __restrict__ tile_flag, int64_t nt) {
  // in-place: tile_sum[b] becomes the carry entering tile b (exclusive segmented scan of the aggregates)
  __shared__ SegPair<V> wave_tot[16];
  __shared__ SegPair<V> incl[kSegThreads];
  SegPair<V> run;
  run.s = V(0);
  run.f = 0;
  for (int64_t a = 0; a < nt; a += kSegThreads) {
    const int64_t i = a + threadIdx.x;
    SegPair<V> v;
    v.s = i < nt ? tile_sum[i] : V(0);
    v.f = i < nt ? tile_flag[i] : 0;
    SegPair<V> sc = block_segscan(v, wave_tot);
    incl[threadIdx.x] = sc;
    __syncthreads();
    SegPair<V> ex = run;
    if (threadIdx.x > 0) ex = seg_combine(run, incl[threadIdx.x - 1]);
    if (i < nt) tile_sum[i] = ex.s;
    run = seg_combine(run, incl[kSegThreads - 1]);
    __syncthreads();
  }
}

template <class V, class I>
__global__ void __launch_bounds__(kSegThreads) k_tile_apply(const V* __restrict__ values, V* __restrict__ out,
                                                            const I* __restrict__ ids, int64_t n, int reverse,
                                                            const V* __restrict__ tile_carry) {
  __shared__ SegPair<V> wave_tot[16];
  __shared__ V incl[kSegThreads];
  const int64_t base = (int64_t)blockIdx.x * kSegTile + (int64_t)threadIdx.x * kSegItems;
  V val[kSegItems];
  int head[kSegItems];
  SegPair<V> agg;
  agg.s = V(0);
  agg.f = 0;
#pragma unroll
  for (int j = 0; j < kSegItems; ++j) {
    const int64_t i = base + j;
    val[j] = V(0);
    head[j] = 0;
    if (i < n) {
      val[j] = values[reverse ? n - 1 - i : i];
      head[j] = is_head(ids, i, n, reverse);
    }
    SegPair<V> e;
    e.s = val[j];
    e.f = head[j];
    agg = seg_combine(agg, e);
  }
  SegPair<V> sc = block_segscan(agg, wave_tot);
  incl[threadIdx.x] = sc.s;
  __syncthreads();
  // running sum entering this thread: the previous thread's inclusive value (which already restarts at
  // heads) plus the tile carry when no head precedes this thread inside the tile
  __shared__ int inclf[kSegThreads];
  inclf[threadIdx.x] = sc.f;
  __syncthreads();
  V run = tile_carry[blockIdx.x];
  if (threadIdx.x > 0) run = inclf[threadIdx.x - 1] ? incl[threadIdx.x - 1] : incl[threadIdx.x - 1] + run;
#pragma unroll
  for (int j = 0; j < kSegItems; ++j) {
    const int64_t i = base + j;
    run = head[j] ? val[j] : run + val[j];
    if (i < n) out[reverse ? n - 1 - i : i] = run;
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

template <class V, class I>
static int run_segcumsum(const void* values, void* out, const void* ids, int64_t n, int reverse, void* ws, hipStream_t stream) {
  const int64_t nt = ceil_div(n, kSegTile);
  V* tile_sum = reinterpret_cast<V*>(ws);
  int* tile_flag = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + ((sizeof(double) * (size_t)nt + 255) / 256) * 256);
  k_tile_reduce<V, I><<<(unsigned)nt, kSegThreads, 0, stream>>>((const V*)values, (const I*)ids, n, reverse, tile_sum, tile_flag);
  FSW_LAUNCH_CHECK();
  k_tile_scan<V><<<1, kSegThreads, 0, stream>>>(tile_sum, tile_flag, nt);
  FSW_LAUNCH_CHECK();
  k_tile_apply<V, I><<<(unsigned)nt, kSegThreads, 0, stream>>>((const V*)values, (V*)out, (const I*)ids, n, reverse, tile_sum);
  FSW_LAUNCH_CHECK();
  return 0;
}

}  // namespace fsw

using namespace fsw;

extern "C" size_t fsw_segcumsum_workspace_bytes(int64_t n) {
  const size_t nt = (size_t)ceil_div(n > 0 ? n : 1, kSegTile);
  return ((sizeof(double) * nt + 255) / 256) * 256 + ((sizeof(int) * nt + 255) / 256) * 256;
}

extern "C" int fsw_segcumsum(int value_dtype, const void* values, void* out, const void* segment_ids, int id_bytes, int64_t n,
                             int reverse, void* workspace, size_t workspace_bytes, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n == 0) return 0;
  FSW_REQUIRE(n > 0 && ceil_div(n, kSegTile) < (1ll << 31), "fsw_segcumsum: bad size %lld", (long long)n);
  FSW_REQUIRE(values && out && segment_ids && workspace, "fsw_segcumsum: null pointer");
  FSW_REQUIRE(workspace_bytes >= fsw_segcumsum_workspace_bytes(n), "fsw_segcumsum: workspace too small");
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

extern "C" int get_max_threads_per_block(int device_index) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_index) != hipSuccess) return 0;
  return prop.maxThreadsPerBlock;
}
