// Edge list -> CSR by recipient + degree bins.  gfx950.
//
// Replaces FSW_conv.edge_index_to_adj (reference fsw_conv.py:384-447) and the sp.get_slice_info sorts of
// FSW_embedding.forward (reference fsw_embedding.py:778-821).  The reference sorts E int64 COO keys
// (coalesce + stable sort); here the adjacency is grouped by recipient with a counting sort:
//   histogram of recipients (int atomics on an L2-resident counter array) -> exclusive scan -> scatter
// and the rows are then bucketed by in-degree so the fused neighbourhood kernels can run exact-size
// sorting networks on runs of equal-degree rows.  All passes are coalesced streams over the edge list
// (16 B/edge read twice, 4-8 B/edge written once).
#include <algorithm>
#include "fsw_common.h"

namespace fsw {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanThreads * kScanItems;  // rows per scan block

struct GraphWs {
  int32_t* cursor;      // [num_rows + 1] degree counters, then running insertion cursors
  int32_t* block_sums;  // [num_scan_blocks]
  int32_t* bin_count;   // [FSW_NUM_BINS]
  int32_t* bin_cursor;  // [FSW_NUM_BINS]
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static GraphWs carve(void* ws, int64_t num_rows) {
  char* p = reinterpret_cast<char*>(ws);
  GraphWs g;
  g.cursor = reinterpret_cast<int32_t*>(p);
  p += align_up(sizeof(int32_t) * (size_t)(num_rows + 1), 256);
  g.block_sums = reinterpret_cast<int32_t*>(p);
  p += align_up(sizeof(int32_t) * (size_t)(ceil_div(num_rows, kScanTile) + 1), 256);
  g.bin_count = reinterpret_cast<int32_t*>(p);
  p += 256;
  g.bin_cursor = reinterpret_cast<int32_t*>(p);
  return g;
}

// ---- pass 1: in-degree histogram + input validation --------------------------------------------
__global__ void __launch_bounds__(256) k_degree_hist(const int64_t* __restrict__ recipients,
                                                     const int64_t* __restrict__ senders,
                                                     const float* __restrict__ edge_w, int64_t num_edges,
                                                     int64_t num_rows, int64_t num_cols, int32_t* __restrict__ cnt,
                                                     int32_t* __restrict__ stats) {
  int flags = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < num_edges; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = recipients[e];
    int64_t c = senders[e];
    if (r < 0 || r >= num_rows || c < 0 || c >= num_cols) {
      flags |= FSW_FLAG_INDEX_RANGE;
      continue;
    }
    if (edge_w) {
      float w = edge_w[e];
      if (!(fabsf(w) <= 3.402823466e38f)) flags |= FSW_FLAG_W_NONFINITE;  // NaN or Inf
      if (w < 0.f) flags |= FSW_FLAG_W_NEGATIVE;
    }
    atomicAdd(&cnt[r], 1);
  }
  if (flags) atomicOr(&stats[FSW_STAT_FLAGS], flags);
}

// ---- pass 2: exclusive scan of the degrees (3 small kernels) -------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    int t = __shfl_up(v, off);
    if (lane_id() >= off) v += t;
  }
  return v;
}

// exclusive scan of one int per thread across a 256-thread block; returns exclusive prefix, total in *total
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[kScanThreads / kWave];
  int inc = wave_inclusive_scan(v);
  int w = threadIdx.x >> 6;
  if (lane_id() == kWave - 1) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / kWave; ++i) {
    int s = wsum[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(kScanThreads) k_scan_partial(const int32_t* __restrict__ cnt, int64_t n,
                                                               int32_t* __restrict__ block_sums) {
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += cnt[base + i];
  int tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kScanThreads) k_scan_block_sums(int32_t* __restrict__ block_sums, int64_t nb) {
  int carry = 0;
  for (int64_t a = 0; a < nb; a += kScanThreads) {
    int64_t i = a + threadIdx.x;
    int v = i < nb ? block_sums[i] : 0;
    int tot;
    int ex = block_exclusive_scan(v, &tot);
    if (i < nb) block_sums[i] = carry + ex;
    carry += tot;
  }
}

// final pass: rowptr, insertion cursors, degree-bin histogram, max degree
__global__ void __launch_bounds__(kScanThreads) k_scan_final(int32_t* __restrict__ cursor /* in: degrees, out: row starts */,
                                                             int64_t n, const int32_t* __restrict__ block_sums,
                                                             int32_t* __restrict__ rowptr, int32_t* __restrict__ bin_count,
                                                             int32_t* __restrict__ stats) {
  __shared__ int lbin[FSW_NUM_BINS];
  __shared__ int lmax;
  if (threadIdx.x < FSW_NUM_BINS) lbin[threadIdx.x] = 0;
  if (threadIdx.x == 0) lmax = 0;
  __syncthreads();
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int deg[kScanItems];
  int s = 0, mx = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    deg[i] = base + i < n ? cursor[base + i] : 0;
    s += deg[i];
    mx = max(mx, deg[i]);
  }
  int tot;
  int ex = block_exclusive_scan(s, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) {
      rowptr[base + i] = ex;
      cursor[base + i] = ex;
      atomicAdd(&lbin[degree_bin(deg[i])], 1);
      ex += deg[i];
    }
  }
  if (base <= n - 1 && base + kScanItems > n - 1) rowptr[n] = ex;  // the thread owning the last row
  atomicMax(&lmax, mx);
  __syncthreads();
  if (threadIdx.x < FSW_NUM_BINS && lbin[threadIdx.x]) atomicAdd(&bin_count[threadIdx.x], lbin[threadIdx.x]);
  if (threadIdx.x == 0 && lmax) atomicMax(&stats[FSW_STAT_MAX_DEGREE], lmax);
}

__global__ void k_bin_offsets(const int32_t* __restrict__ bin_count, int32_t* __restrict__ bin_start,
                              int32_t* __restrict__ bin_cursor, int32_t* __restrict__ stats) {
  if (threadIdx.x == 0) {
    int acc = 0, reg = 0;
    for (int b = 0; b < FSW_NUM_BINS; ++b) {
      bin_start[b] = acc;
      bin_cursor[b] = acc;
      if (b >= 1 && b <= FSW_REG_MAX_DEG) reg += bin_count[b];
      acc += bin_count[b];
    }
    bin_start[FSW_NUM_BINS] = acc;
    stats[FSW_STAT_NUM_ZERO_DEG] = bin_count[0];
    stats[FSW_STAT_NUM_REG] = reg;
    stats[FSW_STAT_NUM_LDS] = bin_count[FSW_BIN_LDS];
    stats[FSW_STAT_NUM_GLOBAL] = bin_count[FSW_BIN_GLOBAL];
  }
}

// ---- pass 3: scatter senders (and weights) into their rows ----------------------------------------
__global__ void __launch_bounds__(256) k_scatter(const int64_t* __restrict__ recipients, const int64_t* __restrict__ senders,
                                                 const float* __restrict__ edge_w, int64_t num_edges, int64_t num_rows,
                                                 int64_t num_cols, int32_t* __restrict__ cursor, int32_t* __restrict__ col,
                                                 float* __restrict__ w) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < num_edges; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = recipients[e];
    int64_t c = senders[e];
    if (r < 0 || r >= num_rows || c < 0 || c >= num_cols) continue;
    int pos = atomicAdd(&cursor[r], 1);
    col[pos] = (int32_t)c;
    if (edge_w) w[pos] = edge_w[e];
  }
}

// ---- pass 4: rows ordered by degree bin ------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bin_rows(const int32_t* __restrict__ rowptr, int64_t n,
                                                  int32_t* __restrict__ bin_cursor, int32_t* __restrict__ perm) {
  __shared__ int lcount[FSW_NUM_BINS];
  __shared__ int lbase[FSW_NUM_BINS];
  if (threadIdx.x < FSW_NUM_BINS) lcount[threadIdx.x] = 0;
  __syncthreads();
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int bin = -1, rank = 0;
  if (r < n) {
    bin = degree_bin(rowptr[r + 1] - rowptr[r]);
    rank = atomicAdd(&lcount[bin], 1);
  }
  __syncthreads();
  if (threadIdx.x < FSW_NUM_BINS && lcount[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&bin_cursor[threadIdx.x], lcount[threadIdx.x]);
  __syncthreads();
  if (r < n) perm[lbase[bin] + rank] = (int32_t)r;
}

}  // namespace fsw

using namespace fsw;

extern "C" size_t fsw_graph_workspace_bytes(int64_t num_rows, int64_t num_edges) {
  (void)num_edges;
  return align_up(sizeof(int32_t) * (size_t)(num_rows + 1), 256) +
         align_up(sizeof(int32_t) * (size_t)(ceil_div(num_rows, kScanTile) + 1), 256) + 512;
}

extern "C" int fsw_graph_build(const int64_t* recipients, const int64_t* senders, const float* edge_w, int64_t num_edges,
                               int64_t num_rows, int64_t num_cols, int32_t* rowptr, int32_t* col, float* w, int32_t* perm,
                               int32_t* bin_start, int32_t* stats, void* workspace, size_t workspace_bytes,
                               fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(num_rows >= 1 && num_rows < (1ll << 31) && num_cols >= 1 && num_cols < (1ll << 31) && num_edges >= 0 &&
                  num_edges < (1ll << 31),
              "fsw_graph_build: sizes must satisfy 1 <= rows, cols < 2^31 and 0 <= edges < 2^31 (got %lld, %lld, %lld)",
              (long long)num_rows, (long long)num_cols, (long long)num_edges);
  FSW_REQUIRE(workspace && workspace_bytes >= fsw_graph_workspace_bytes(num_rows, num_edges),
              "fsw_graph_build: workspace too small");
  FSW_REQUIRE(rowptr && perm && bin_start && stats && (num_edges == 0 || (col && recipients && senders)),
              "fsw_graph_build: null pointer");
  FSW_REQUIRE(!edge_w || w, "fsw_graph_build: edge_w given but w is null");
  GraphWs g = carve(workspace, num_rows);
  const int64_t nb = ceil_div(num_rows, kScanTile);

  FSW_CHECK_HIP(hipMemsetAsync(g.cursor, 0, sizeof(int32_t) * (size_t)(num_rows + 1), stream));
  FSW_CHECK_HIP(hipMemsetAsync(g.bin_count, 0, 512, stream));
  FSW_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(int32_t) * FSW_NUM_STATS, stream));

  const int edge_blocks = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(num_edges, 256 * 4), 1), 256 * 16);
  if (num_edges > 0) {
    k_degree_hist<<<edge_blocks, 256, 0, stream>>>(recipients, senders, edge_w, num_edges, num_rows, num_cols, g.cursor, stats);
    FSW_LAUNCH_CHECK();
  }
  k_scan_partial<<<(int)nb, kScanThreads, 0, stream>>>(g.cursor, num_rows, g.block_sums);
  FSW_LAUNCH_CHECK();
  k_scan_block_sums<<<1, kScanThreads, 0, stream>>>(g.block_sums, nb);
  FSW_LAUNCH_CHECK();
  k_scan_final<<<(int)nb, kScanThreads, 0, stream>>>(g.cursor, num_rows, g.block_sums, rowptr, g.bin_count, stats);
  FSW_LAUNCH_CHECK();
  k_bin_offsets<<<1, 64, 0, stream>>>(g.bin_count, bin_start, g.bin_cursor, stats);
  FSW_LAUNCH_CHECK();
  if (num_edges > 0) {
    k_scatter<<<edge_blocks, 256, 0, stream>>>(recipients, senders, edge_w, num_edges, num_rows, num_cols, g.cursor, col, w);
    FSW_LAUNCH_CHECK();
  }
  k_bin_rows<<<(int)ceil_div(num_rows, 256), 256, 0, stream>>>(rowptr, num_rows, g.bin_cursor, perm);
  FSW_LAUNCH_CHECK();
  return 0;
}
