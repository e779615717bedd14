// Edge list -> CSR by recipient + degree bins.  gfx950.
//
// Replaces FSW_conv.edge_index_to_adj (reference fsw_conv.py:384-447) and the sp.get_slice_info sorts of
// FSW_embedding.forward (reference fsw_embedding.py:778-821).  The reference coalesces and stable-sorts E
// int64 COO keys with torch.sort; here the edges are grouped by recipient with a hand-written stable LSD
// radix sort on the recipient index only (ceil(log2(rows+1)) bits, 7-8 bits per pass):
//   upsweep    per 4096-edge tile: digit histogram (wave-aggregated LDS atomics)
//   scan       exclusive scan of the digit-major [digit][tile] count table = global base of every (tile, digit)
//   downsweep  wave64 match ranking (ballot per digit bit), per-wave digit counters in LDS, tile-local
//              reorder through LDS so the global writes are runs of consecutive addresses per digit
// The first pass reads the int64 edge_index directly (and validates it), later passes move int32 pairs.
// The sort is stable, so every CSR row keeps its senders in edge-list order: the build is deterministic.
// rowptr comes from the boundaries of the sorted keys; rows are then bucketed by in-degree so the fused
// neighbourhood kernels run exact-size sorting networks on runs of equal-degree rows.
// (A first version used one global int atomic per edge for the histogram and one for the scatter:
//  1.13 ms at 10M edges against ~0.3 ms for the sort -- profiles/r01_v1_bench_kernel_stats.csv.)
#include <algorithm>
#include <atomic>
#include "fsw_common.h"

namespace fsw {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanThreads * kScanItems;

#ifndef FSW_RS_THREADS
#define FSW_RS_THREADS 512   // 8 wavefronts x 8 rounds per 4096-edge tile (256 x 16: 0.273 -> 0.253 ms for the build at config 3)
#endif
constexpr int kRsThreads = FSW_RS_THREADS;
constexpr int kRsItems = 4096 / kRsThreads;                      // rounds of 64 consecutive edges per wave
constexpr int kRsTile = kRsThreads * kRsItems;    // 4096 edges per tile
constexpr int kRsWaves = kRsThreads / kWave;
constexpr int kRsMaxDigits = 512;                 // 9 bits: the bucket pass of the two-level build (below)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- generic device-wide exclusive scan of int32 (three launches) ----------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    int t = __shfl_up(v, off);
    if (lane_id() >= off) v += t;
  }
  return v;
}

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

__global__ void __launch_bounds__(kScanThreads) k_scan_partial(const int32_t* __restrict__ in, int64_t n,
                                                               int32_t* __restrict__ block_sums) {
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += in[base + i];
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

__global__ void __launch_bounds__(kScanThreads) k_scan_apply(int32_t* __restrict__ data, int64_t n,
                                                             const int32_t* __restrict__ block_sums) {
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = base + i < n ? data[base + i] : 0;
    s += v[i];
  }
  int tot;
  int ex = block_exclusive_scan(s, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) data[base + i] = ex;
    ex += v[i];
  }
}

static int exclusive_scan_i32(int32_t* data, int64_t n, int32_t* block_sums, hipStream_t stream) {
  const int64_t nb = ceil_div(n, kScanTile);
  k_scan_partial<<<(unsigned)nb, kScanThreads, 0, stream>>>(data, n, block_sums);
  FSW_LAUNCH_CHECK();
  k_scan_block_sums<<<1, kScanThreads, 0, stream>>>(block_sums, nb);
  FSW_LAUNCH_CHECK();
  k_scan_apply<<<(unsigned)nb, kScanThreads, 0, stream>>>(data, n, block_sums);
  FSW_LAUNCH_CHECK();
  return 0;
}

// Lanes of the wavefront whose digit equals this lane's (wave64 "match any" from one ballot per digit bit).  The lanes that
// differ from me in some bit are accumulated in two 32-bit halves -- per bit one v_bfe, one v_cmp and two or/xor pairs, no
// 64-bit selects and no branches (bits above the digit width are zero in every lane and change nothing).  Inactive lanes
// (ok == false) form a group of their own.
template <int MAXB>
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool ok) {
  uint32_t dlo = 0u, dhi = 0u;
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    const uint32_t bit = (d >> b) & 1u;
    const unsigned long long bb = __ballot(bit != 0u);
    const uint32_t m = 0u - bit;
    dlo |= (uint32_t)bb ^ m;
    dhi |= (uint32_t)(bb >> 32) ^ m;
  }
  const unsigned long long okm = __ballot(ok);
  const unsigned long long same = ~(((unsigned long long)dhi << 32) | (unsigned long long)dlo);
  return ok ? (same & okm) : ~okm;
}

// ---- radix sort by recipient ----------------------------------------------------------------------------
// Key of edge e: its recipient, or `num_rows` (a sentinel that sorts last) when an endpoint is out of range.
template <bool FIRST>
__device__ __forceinline__ uint32_t load_key(const int64_t* __restrict__ recipients, const int64_t* __restrict__ senders,
                                             const uint32_t* __restrict__ keys_in, int64_t e, int64_t num_rows,
                                             int64_t num_cols, int& flags) {
  if constexpr (FIRST) {
    const int64_t r = recipients[e], c = senders[e];
    if (r < 0 || r >= num_rows || c < 0 || c >= num_cols) {
      flags |= FSW_FLAG_INDEX_RANGE;
      return (uint32_t)num_rows;
    }
    return (uint32_t)r;
  } else {
    return keys_in[e];
  }
}

template <bool FIRST>
__global__ void __launch_bounds__(kRsThreads) k_rs_upsweep(const int64_t* __restrict__ recipients,
                                                           const int64_t* __restrict__ senders,
                                                           const float* __restrict__ edge_w,
                                                           const uint32_t* __restrict__ keys_in, int64_t num_edges,
                                                           int64_t num_rows, int64_t num_cols, int shift, int ndigits,
                                                           int32_t* __restrict__ counts /* [ndigits][ntiles] */, int64_t ntiles,
                                                           int32_t* __restrict__ stats) {
  __shared__ int hist[kRsMaxDigits];
  for (int d = threadIdx.x; d < ndigits; d += kRsThreads) hist[d] = 0;
  __syncthreads();
  const int64_t tile0 = (int64_t)blockIdx.x * kRsTile;
  const uint32_t mask = (uint32_t)ndigits - 1;
  int flags = 0;
  // half a tile's loads in flight, then its LDS atomics
  constexpr int kHalf = kRsItems / 2;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t key[kHalf];
#pragma unroll
    for (int i = 0; i < kHalf; ++i) {
      const int64_t e = tile0 + (int64_t)(h * kHalf + i) * kRsThreads + threadIdx.x;
      key[i] = 0xffffffffu;
      if (e < num_edges) {
        key[i] = load_key<FIRST>(recipients, senders, keys_in, e, num_rows, num_cols, flags);
        if (FIRST && edge_w) {
          const float w = edge_w[e];
          if (!(fabsf(w) <= 3.402823466e38f)) flags |= FSW_FLAG_W_NONFINITE;
          if (w < 0.f) flags |= FSW_FLAG_W_NEGATIVE;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kHalf; ++i)
      if (tile0 + (int64_t)(h * kHalf + i) * kRsThreads + threadIdx.x < num_edges) atomicAdd(&hist[(key[i] >> shift) & mask], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < ndigits; d += kRsThreads) counts[(int64_t)d * ntiles + blockIdx.x] = hist[d];
  if (FIRST && flags) atomicOr(&stats[FSW_STAT_FLAGS], flags);
}

// VAL = uint32_t (sender) for unit weights, uint64_t (sender | weight bits << 32) when weights are carried
template <bool FIRST, class VAL, bool EID = false>   // EID: carry the edge id instead of the sender
__global__ void __launch_bounds__(kRsThreads) k_rs_downsweep(const int64_t* __restrict__ recipients,
                                                             const int64_t* __restrict__ senders,
                                                             const float* __restrict__ edge_w,
                                                             const uint32_t* __restrict__ keys_in,
                                                             const VAL* __restrict__ vals_in, int64_t num_edges,
                                                             int64_t num_rows, int64_t num_cols, int shift, int nbits,
                                                             const int32_t* __restrict__ bases /* scanned [ndigits][ntiles] */,
                                                             int64_t ntiles, uint32_t* __restrict__ keys_out,
                                                             VAL* __restrict__ vals_out) {
  __shared__ int wcnt[kRsWaves][kRsMaxDigits];   // per-wave digit counters, then exclusive wave bases
  __shared__ int dstart[kRsMaxDigits];           // tile-local exclusive start of each digit
  __shared__ int gbase[kRsMaxDigits];            // global base of (this tile, digit)
  __shared__ uint32_t skey[kRsTile];
  __shared__ VAL sval[kRsTile];
  const int ndigits = 1 << nbits;
  const uint32_t mask = (uint32_t)ndigits - 1;
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  for (int d = threadIdx.x; d < ndigits; d += kRsThreads) {
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) wcnt[w][d] = 0;
    gbase[d] = bases[(int64_t)d * ntiles + blockIdx.x];
  }
  __syncthreads();

  // wave wv owns tile elements [wv*1024, wv*1024 + 1024) as 16 rounds of 64 consecutive edges (stable order)
  const int64_t wave0 = (int64_t)blockIdx.x * kRsTile + (int64_t)wv * (kRsItems * kWave);
  uint32_t key[kRsItems];
  VAL val[kRsItems];
  int rank[kRsItems];
  int dummy = 0;
  // two half-tiles of 8 rounds: all loads of a half are issued before its first match (the matches are ~60 vector
  // instructions a round; with the loads inside the rounds every round waited for its own HBM access)
  constexpr int kHalf = kRsItems / 2;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = h * kHalf; i < (h + 1) * kHalf; ++i) {
      const int64_t e = wave0 + (int64_t)i * kWave + lane;
      const bool ok = e < num_edges;
      key[i] = 0xffffffffu;
      val[i] = VAL(0);
      if (ok) {
        key[i] = load_key<FIRST>(recipients, senders, keys_in, e, num_rows, num_cols, dummy);
        if constexpr (FIRST && EID) {
          val[i] = (VAL)(uint32_t)e;
        } else if constexpr (FIRST) {
          const uint32_t s = (uint32_t)senders[e];
          if constexpr (sizeof(VAL) == 8)
            val[i] = (VAL)s | ((VAL)__float_as_uint(edge_w[e]) << 32);
          else
            val[i] = (VAL)s;
        } else {
          val[i] = vals_in[e];
        }
      }
    }
#pragma unroll
    for (int i = h * kHalf; i < (h + 1) * kHalf; ++i) {
      const bool ok = wave0 + (int64_t)i * kWave + lane < num_edges;
      const uint32_t d = (key[i] >> shift) & mask;
      const unsigned long long peers = match_digit<9>(d, ok);
      const int leader = __ffsll((long long)peers) - 1;
      const int below = __popcll(peers & ((1ull << lane) - 1ull));
      int prev = 0;
      if (ok && lane == leader) {
        prev = wcnt[wv][d];
        wcnt[wv][d] = prev + __popcll(peers);
      }
      prev = __shfl(prev, leader);
      rank[i] = prev + below;
    }
  }
  __syncthreads();
  // per digit: exclusive bases across waves and the tile-local digit start
  for (int d = threadIdx.x; d < ndigits; d += kRsThreads) {
    int acc = 0;
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) {
      const int c = wcnt[w][d];
      wcnt[w][d] = acc;
      acc += c;
    }
    dstart[d] = acc;  // tile total for now
  }
  __syncthreads();
  if (threadIdx.x < kWave) {  // exclusive scan of the tile totals over the digits (<= kRsMaxDigits = 8 per lane)
    constexpr int kPer = kRsMaxDigits / kWave;
    int v[kPer], s = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int d = lane * kPer + j;
      v[j] = d < ndigits ? dstart[d] : 0;
      s += v[j];
    }
    int ex = wave_inclusive_scan(s) - s;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int d = lane * kPer + j;
      if (d < ndigits) dstart[d] = ex;
      ex += v[j];
    }
  }
  __syncthreads();
  // tile-local reorder through LDS
#pragma unroll
  for (int i = 0; i < kRsItems; ++i) {
    const int64_t e = wave0 + (int64_t)i * kWave + lane;
    if (e < num_edges) {
      const uint32_t d = (key[i] >> shift) & mask;
      const int lpos = dstart[d] + wcnt[wv][d] + rank[i];
      skey[lpos] = key[i];
      sval[lpos] = val[i];
    }
  }
  __syncthreads();
  const int64_t tile0 = (int64_t)blockIdx.x * kRsTile;
  const int count = (int)min((int64_t)kRsTile, num_edges - tile0);
  for (int i = threadIdx.x; i < count; i += kRsThreads) {
    const uint32_t k = skey[i];
    const uint32_t d = (k >> shift) & mask;
    const int64_t pos = (int64_t)gbase[d] + (i - dstart[d]);
    keys_out[pos] = k;
    vals_out[pos] = sval[i];
  }
}

// ---- rowptr from the sorted keys, col / w from the sorted values --------------------------------------------
template <class VAL>
__global__ void __launch_bounds__(256) k_finish_csr(const uint32_t* __restrict__ keys, const VAL* __restrict__ vals,
                                                    int64_t num_edges, int64_t num_rows, int32_t* __restrict__ rowptr,
                                                    int32_t* __restrict__ col, float* __restrict__ w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= num_edges; i += (int64_t)gridDim.x * blockDim.x) {
    // key sequence extended with -1 in front and num_rows behind; invalid edges carry key == num_rows
    const int64_t prev = i == 0 ? -1 : (int64_t)keys[i - 1];
    const int64_t cur = i == num_edges ? num_rows : (int64_t)keys[i];
    for (int64_t r = prev + 1; r <= cur && r <= num_rows; ++r) rowptr[r] = (int32_t)i;
    if (i < num_edges && cur < num_rows) {
      const VAL v = vals[i];
      col[i] = (int32_t)(uint32_t)v;
      if constexpr (sizeof(VAL) == 8) w[i] = __uint_as_float((uint32_t)(v >> 32));
    }
  }
}

// ---- degree bins ----------------------------------------------------------------------------------------------
// Most rows of a graph share a handful of degrees, so per-thread LDS atomics on the bin counters serialise (48 us for
// 1M rows).  Rows are matched inside the wavefront instead (one ballot per bit of the bin id): one LDS atomic per
// distinct bin and wave, and the lane's rank among its peers for free.
__device__ __forceinline__ unsigned long long match_bin(int bin, bool valid) {
  unsigned long long peers = __ballot(valid);
  if (!valid) peers = ~peers;
  static_assert(FSW_NUM_BINS <= 64, "bin ids must fit 6 bits");
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    const unsigned long long bb = __ballot((bin >> b) & 1);
    peers &= ((bin >> b) & 1) ? bb : ~bb;
  }
  return peers;
}

// Row chunks: with chunk_rows > 0 the rows are binned separately inside every chunk of chunk_rows consecutive rows
// (a multiple of kBinRowsPerBlock, so a workgroup never straddles two chunks): perm lists chunk 0's rows by degree bin, then
// chunk 1's, ..., and bin_start holds one row of FSW_NUM_BINS + 1 offsets per chunk.  A kernel handed chunk c's row of
// bin_start visits exactly the recipients [c * chunk_rows, (c + 1) * chunk_rows) -- that is how the multi-GPU path
// overlaps the collective of one node range with the kernels of the next (dist.py).
constexpr int kBinItems = 8;
constexpr int kBinRowsPerBlock = 256 * kBinItems;
constexpr int kMaxRowChunks = 256;
static_assert(kBinRowsPerBlock == FSW_BIN_BLOCK_ROWS, "include/fsw_hip.h documents the chunk granularity");

__global__ void __launch_bounds__(256) k_bin_count(const int32_t* __restrict__ rowptr, int64_t n, int64_t chunk_rows,
                                                   int32_t* __restrict__ bin_count, int32_t* __restrict__ stats) {
  __shared__ int lbin[FSW_NUM_BINS];
  __shared__ int lmax;
  if (threadIdx.x < FSW_NUM_BINS) lbin[threadIdx.x] = 0;
  if (threadIdx.x == 0) lmax = 0;
  __syncthreads();
  int mx = 0;
  const int64_t r0 = (int64_t)blockIdx.x * kBinRowsPerBlock + threadIdx.x;
#pragma unroll
  for (int i = 0; i < kBinItems; ++i) {
    const int64_t r = r0 + (int64_t)i * 256;
    const bool valid = r < n;
    const int deg = valid ? rowptr[r + 1] - rowptr[r] : 0;
    const int bin = degree_bin(deg);
    mx = max(mx, deg);
    const unsigned long long peers = match_bin(bin, valid);
    if (valid && lane_id() == __ffsll((long long)peers) - 1) atomicAdd(&lbin[bin], __popcll(peers));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
  if (lane_id() == 0) atomicMax(&lmax, mx);
  __syncthreads();
  const int64_t chunk = ((int64_t)blockIdx.x * kBinRowsPerBlock) / chunk_rows;
  if (threadIdx.x < FSW_NUM_BINS && lbin[threadIdx.x]) atomicAdd(&bin_count[chunk * FSW_NUM_BINS + threadIdx.x], lbin[threadIdx.x]);
  if (threadIdx.x == 0 && lmax) atomicMax(&stats[FSW_STAT_MAX_DEGREE], lmax);
}

__global__ void __launch_bounds__(kMaxRowChunks) k_bin_offsets(const int32_t* __restrict__ bin_count, int32_t* __restrict__ bin_start,
                                                               int32_t* __restrict__ bin_cursor, int32_t* __restrict__ stats,
                                                               int num_chunks, const int32_t* __restrict__ rowptr, int64_t num_rows) {
  __shared__ int ctot[kMaxRowChunks];
  const int c = threadIdx.x;
  if (c == 0) stats[FSW_STAT_NNZ] = rowptr[num_rows];   // CSR entries in use (invalid edges dropped, parallel edges merged or not)
  int tot = 0;
  if (c < num_chunks)
    for (int b = 0; b < FSW_NUM_BINS; ++b) tot += bin_count[c * FSW_NUM_BINS + b];
  ctot[c] = tot;
  __syncthreads();
  if (c == 0) {
    int acc = 0;
    for (int i = 0; i < num_chunks; ++i) {
      const int t = ctot[i];
      ctot[i] = acc;
      acc += t;
    }
  }
  __syncthreads();
  if (c < num_chunks) {
    int acc = ctot[c], reg = 0, mid = 0;
    const int32_t* cnt = bin_count + c * FSW_NUM_BINS;
    int32_t* bs = bin_start + c * (FSW_NUM_BINS + 1);
    for (int b = 0; b < FSW_NUM_BINS; ++b) {
      bs[b] = acc;
      bin_cursor[c * FSW_NUM_BINS + b] = acc;
      if (b >= 1 && b <= FSW_REG_MAX_DEG) reg += cnt[b];
      if (b >= FSW_BIN_MID0 && b < FSW_BIN_HUB0) mid += cnt[b];
      acc += cnt[b];
    }
    bs[FSW_NUM_BINS] = acc;
    if (cnt[0]) atomicAdd(&stats[FSW_STAT_NUM_ZERO_DEG], cnt[0]);
    if (reg) atomicAdd(&stats[FSW_STAT_NUM_REG], reg);
    if (mid) atomicAdd(&stats[FSW_STAT_NUM_LDS], mid);
    int glob = 0;
    for (int b = FSW_BIN_HUB0; b <= FSW_BIN_GLOBAL; ++b) glob += cnt[b];
    if (glob) atomicAdd(&stats[FSW_STAT_NUM_GLOBAL], glob);
  }
}

// A workgroup places kBinRowsPerBlock consecutive rows: per-bin counts accumulate in LDS over the rounds (a lane's rank
// is the running count before its round + its rank among its wave peers), then ONE global atomic per bin claims the
// workgroup's range -- the cursors of a chunk are shared by all its workgroups, so same-address atomics are the cost to
// keep low.
__global__ void __launch_bounds__(256) k_bin_rows(const int32_t* __restrict__ rowptr, int64_t n, int64_t chunk_rows,
                                                  int32_t* __restrict__ bin_cursor, int32_t* __restrict__ perm,
                                                  int32_t* __restrict__ invperm) {
  __shared__ int lcount[FSW_NUM_BINS];
  __shared__ int lbase[FSW_NUM_BINS];
  if (threadIdx.x < FSW_NUM_BINS) lcount[threadIdx.x] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * kBinRowsPerBlock + threadIdx.x;
  int bin[kBinItems], rank[kBinItems];
#pragma unroll
  for (int i = 0; i < kBinItems; ++i) {
    const int64_t r = r0 + (int64_t)i * 256;
    const bool valid = r < n;
    bin[i] = valid ? degree_bin(rowptr[r + 1] - rowptr[r]) : 0;
    const unsigned long long peers = match_bin(bin[i], valid);
    const int leader = __ffsll((long long)peers) - 1;
    int prev = 0;
    if (valid && lane_id() == leader) prev = atomicAdd(&lcount[bin[i]], __popcll(peers));   // one LDS atomic per bin and wave
    rank[i] = __shfl(prev, leader) + __popcll(peers & ((1ull << lane_id()) - 1ull));
  }
  __syncthreads();
  const int64_t chunk = ((int64_t)blockIdx.x * kBinRowsPerBlock) / chunk_rows;
  if (threadIdx.x < FSW_NUM_BINS && lcount[threadIdx.x])
    lbase[threadIdx.x] = atomicAdd(&bin_cursor[chunk * FSW_NUM_BINS + threadIdx.x], lcount[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kBinItems; ++i) {
    const int64_t r = r0 + (int64_t)i * 256;
    if (r < n) {
      const int pos = lbase[bin[i]] + rank[i];
      perm[pos] = (int32_t)r;
      if (invperm) invperm[r] = pos;
    }
  }
}

// ---- workspace ---------------------------------------------------------------------------------------------------
constexpr size_t kBinTableBytes = sizeof(int32_t) * kMaxRowChunks * FSW_NUM_BINS;

struct GraphWs {
  uint32_t* keys[2];
  void* vals[2];
  int32_t* counts;      // [ndigits][ntiles]
  int32_t* block_sums;  // scan scratch
  int32_t* block_sums_big;  // bucket starts of the two-level build (multi-pass case): rows / 2^11 + 2 entries
  int32_t* bin_count;
  int32_t* bin_cursor;
  size_t total;
};

static GraphWs carve(void* ws, int64_t num_edges, int64_t num_rows = 0) {
  const size_t E = (size_t)std::max<int64_t>(num_edges, 1);
  const size_t ntiles = (size_t)ceil_div((int64_t)E, kRsTile);
  char* p = reinterpret_cast<char*>(ws);
  GraphWs g;
  auto take = [&](size_t bytes) {
    char* q = p;
    p += align_up(bytes, 256);
    return q;
  };
  g.keys[0] = reinterpret_cast<uint32_t*>(take(4 * E));
  g.keys[1] = reinterpret_cast<uint32_t*>(take(4 * E));
  g.vals[0] = take(8 * E);
  g.vals[1] = take(8 * E);
  g.counts = reinterpret_cast<int32_t*>(take(4 * (size_t)kRsMaxDigits * ntiles));
  g.block_sums = reinterpret_cast<int32_t*>(take(4 * (size_t)(ceil_div((int64_t)(kRsMaxDigits * ntiles), kScanTile) + 1)));
  g.block_sums_big = reinterpret_cast<int32_t*>(take(4 * (size_t)((num_rows >> 10) + 3)));
  g.bin_count = reinterpret_cast<int32_t*>(take(kBinTableBytes));
  g.bin_cursor = reinterpret_cast<int32_t*>(take(kBinTableBytes));
  g.total = (size_t)(p - reinterpret_cast<char*>(ws));
  return g;
}

// LSD radix sort of E (key, value) pairs.  first_a / first_b: the int64 arrays the FIRST pass reads directly
// (key = first_a[e], valid when 0 <= first_a[e] < range_a and 0 <= first_b[e] < range_b; invalid edges get the
// sentinel key range_a and sort last).  keys_in != NULL: pre-built uint32 keys instead (values from g.vals[*cur]).
// On return *cur selects the ping-pong buffer that holds the result.
static int key_bits(int64_t range_a) {
  int keybits = 1;
  while ((1ll << keybits) <= range_a) ++keybits;  // values 0 .. range_a (sentinel) must fit
  return keybits;
}

// lo_bit > 0: only the key bits from lo_bit upwards are sorted (the two-level build finishes the low bits bucket by bucket),
// in passes of up to max_bits bits.
template <class VAL, bool EID>
static int radix_sort_pairs(const int64_t* first_a, const int64_t* first_b, const float* edge_w, bool from_keys,
                            int64_t num_edges, int64_t range_a, int64_t range_b, int32_t* stats, GraphWs& g, int* cur_io,
                            hipStream_t stream, int lo_bit = 0, int max_bits = 8, int* last_bits = nullptr) {
  const int keybits = key_bits(range_a) - lo_bit;
  const int npass = (keybits + max_bits - 1) / max_bits;
  const int bits = (keybits + npass - 1) / npass;
  if (last_bits) *last_bits = bits;
  const int64_t ntiles = ceil_div(num_edges, kRsTile);
  int cur = *cur_io;
  for (int pass = 0; pass < npass; ++pass) {
    const int shift = lo_bit + pass * bits;
    const int ndigits = 1 << bits;
    const bool first = (pass == 0) && !from_keys;
    const uint32_t* kin = g.keys[cur];
    const VAL* vin = reinterpret_cast<const VAL*>(g.vals[cur]);
    uint32_t* kout = g.keys[cur ^ 1];
    VAL* vout = reinterpret_cast<VAL*>(g.vals[cur ^ 1]);
    if (first)
      k_rs_upsweep<true><<<(unsigned)ntiles, kRsThreads, 0, stream>>>(first_a, first_b, edge_w, nullptr, num_edges, range_a,
                                                                      range_b, shift, ndigits, g.counts, ntiles, stats);
    else
      k_rs_upsweep<false><<<(unsigned)ntiles, kRsThreads, 0, stream>>>(nullptr, nullptr, nullptr, kin, num_edges, range_a,
                                                                       range_b, shift, ndigits, g.counts, ntiles, stats);
    FSW_LAUNCH_CHECK();
    int rc = exclusive_scan_i32(g.counts, (int64_t)ndigits * ntiles, g.block_sums, stream);
    if (rc) return rc;
    if (first)
      k_rs_downsweep<true, VAL, EID><<<(unsigned)ntiles, kRsThreads, 0, stream>>>(first_a, first_b, edge_w, nullptr, nullptr, num_edges,
                                                                                  range_a, range_b, shift, bits, g.counts, ntiles, kout, vout);
    else
      k_rs_downsweep<false, VAL, EID><<<(unsigned)ntiles, kRsThreads, 0, stream>>>(nullptr, nullptr, nullptr, kin, vin, num_edges, range_a,
                                                                                   range_b, shift, bits, g.counts, ntiles, kout, vout);
    FSW_LAUNCH_CHECK();
    cur ^= 1;
  }
  *cur_io = cur;
  return 0;
}

template <class VAL>
static int sort_and_finish(const int64_t* recipients, const int64_t* senders, const float* edge_w, int64_t num_edges,
                           int64_t num_rows, int64_t num_cols, int32_t* rowptr, int32_t* col, float* w, int32_t* stats,
                           GraphWs& g, hipStream_t stream) {
  int cur = 0;
  int rc = radix_sort_pairs<VAL, false>(recipients, senders, edge_w, false, num_edges, num_rows, num_cols, stats, g, &cur, stream);
  if (rc) return rc;
  const int blocks = (int)std::min<int64_t>(ceil_div(num_edges + 1, 256), 256 * 32);
  k_finish_csr<VAL><<<blocks, 256, 0, stream>>>(g.keys[cur], reinterpret_cast<const VAL*>(g.vals[cur]), num_edges, num_rows, rowptr, col, w);
  FSW_LAUNCH_CHECK();
  return 0;
}

// ---- coalescing build (edge features; exact reference adjacency) ----------------------------------------------------
// Sort by (recipient, sender) = stable sort by sender, then stable sort by recipient, carrying the edge id; a run of equal
// (recipient, sender) pairs becomes ONE CSR entry whose weight and feature vector are the sums over the run -- what
// torch.sparse_coo_tensor(...).coalesce() does to adj and X_edge in the reference (fsw_conv.py:397-398, 436-437).
__global__ void __launch_bounds__(256) k_gather_keys(const int64_t* __restrict__ recipients, const uint32_t* __restrict__ eid,
                                                     const uint32_t* __restrict__ sender_keys, int64_t num_edges, int64_t num_rows,
                                                     int64_t num_cols, uint32_t* __restrict__ keys_out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_edges; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = recipients[eid[i]];
    // an edge whose sender was out of range carries the sentinel sender key: keep it invalid
    keys_out[i] = (r < 0 || r >= num_rows || sender_keys[i] >= (uint32_t)num_cols) ? (uint32_t)num_rows : (uint32_t)r;
  }
}

__global__ void __launch_bounds__(256) k_mark_heads(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ eid,
                                                    const int64_t* __restrict__ senders, int64_t num_edges, int64_t num_rows,
                                                    int32_t* __restrict__ head) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_edges; i += (int64_t)gridDim.x * blockDim.x) {
    const bool valid = keys[i] < (uint32_t)num_rows;
    const bool differs = i == 0 || keys[i] != keys[i - 1] || senders[eid[i]] != senders[eid[i - 1]];
    head[i] = (valid && differs) ? 1 : 0;
  }
}

// pos = exclusive scan of head.  One thread per run head sums the run (duplicates are rare and short) in edge-list order.
__global__ void __launch_bounds__(256) k_emit_coalesced(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ eid,
                                                        const int32_t* __restrict__ head, const int32_t* __restrict__ pos,
                                                        const int64_t* __restrict__ senders, const float* __restrict__ edge_w,
                                                        const float* __restrict__ edge_feat, int d_edge, int64_t num_edges,
                                                        int64_t num_rows, uint32_t* __restrict__ keyc, int32_t* __restrict__ col,
                                                        float* __restrict__ w, float* __restrict__ ef,
                                                        int32_t* __restrict__ slot_of_edge, int32_t* __restrict__ nnz_out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_edges; i += (int64_t)gridDim.x * blockDim.x) {
    if (i == num_edges - 1) *nnz_out = pos[i] + head[i];
    if (!head[i]) continue;
    const int p = pos[i];
    double wsum = 0.0;
    for (int q = 0; q < d_edge; ++q) ef[(int64_t)p * d_edge + q] = 0.f;
    for (int64_t j = i; j < num_edges && (j == i || (!head[j] && keys[j] < (uint32_t)num_rows)); ++j) {
      const uint32_t e = eid[j];
      wsum += edge_w ? (double)edge_w[e] : 1.0;
      for (int q = 0; q < d_edge; ++q) ef[(int64_t)p * d_edge + q] += edge_feat[(int64_t)e * d_edge + q];
      if (slot_of_edge) slot_of_edge[e] = p;
    }
    keyc[p] = keys[i];
    col[p] = (int32_t)senders[eid[i]];
    w[p] = (float)wsum;
  }
}

__global__ void __launch_bounds__(256) k_rowptr_from_keys(const uint32_t* __restrict__ keyc, const int32_t* __restrict__ nnz_ptr,
                                                          int64_t num_rows, int32_t* __restrict__ rowptr) {
  const int64_t nnz = *nnz_ptr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nnz; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prev = i == 0 ? -1 : (int64_t)keyc[i - 1];
    const int64_t cur = i == nnz ? num_rows : (int64_t)keyc[i];
    for (int64_t r = prev + 1; r <= cur && r <= num_rows; ++r) rowptr[r] = (int32_t)i;
  }
}

static int finish_bins(int32_t* rowptr, int64_t num_rows, int64_t chunk_rows, int32_t* perm, int32_t* invperm, int32_t* bin_start,
                       int32_t* stats, GraphWs& g, hipStream_t stream) {
  if (chunk_rows <= 0) chunk_rows = ceil_div(num_rows, kBinRowsPerBlock) * kBinRowsPerBlock;   // one chunk
  const int num_chunks = (int)ceil_div(num_rows, chunk_rows);
  const int row_blocks = (int)ceil_div(num_rows, kBinRowsPerBlock);
  k_bin_count<<<row_blocks, 256, 0, stream>>>(rowptr, num_rows, chunk_rows, g.bin_count, stats);
  FSW_LAUNCH_CHECK();
  k_bin_offsets<<<1, kMaxRowChunks, 0, stream>>>(g.bin_count, bin_start, g.bin_cursor, stats, num_chunks, rowptr, num_rows);
  FSW_LAUNCH_CHECK();
  k_bin_rows<<<row_blocks, 256, 0, stream>>>(rowptr, num_rows, chunk_rows, g.bin_cursor, perm, invperm);
  FSW_LAUNCH_CHECK();
  return 0;
}

// ---- two-level build: one partition pass over the high key bits, then every bucket of 2^rb rows finished by ONE workgroup ----
// The LSD build above moves every edge three times at 1M rows (3 x 7 bits) and pays 3 x (upsweep, three scan launches,
// downsweep).  Here one pass (same kernels, up to 9 bits) groups the edges by bucket = row >> rb, stably; a bucket then holds
// ~E / buckets edges of 2^rb <= 2048 consecutive rows, and one workgroup finishes it without further global passes:
//   1. every wave histograms ITS contiguous part of the bucket over the 2^rb rows (wave-private LDS table),
//   2. exclusive scan over the waves per row (= where each wave's entries of a row start inside the row) and over the rows
//      (= rowptr of the bucket's rows: the bucket's start is already the number of edges of all smaller rows),
//   3. every wave walks its part again in order: wave64 match on the row bits gives the rank among the round's 64 edges, the
//      wave-private cursor the rest -> col[] / w[] written in place, in edge-list order inside every row (stable, deterministic).
// The scattered 4-byte stores of step 3 land in the bucket's own window of col[] (80 KB at config 3) from one workgroup, i.e. one
// L2: they merge there and reach HBM as whole lines.  A bucket of any size is handled (the workgroup just loops longer), so the
// result never depends on the edge distribution -- but a hub-heavy bucket serialises on its workgroup, which is why the host
// side keeps the LSD build for skewed graphs (graph.py).
constexpr int kBucketWaves = 8;
#ifndef FSW_BUCKET_ROW_BITS
#define FSW_BUCKET_ROW_BITS 11
#endif
constexpr int kBucketMaxRowBits = FSW_BUCKET_ROW_BITS;   // LDS: (kBucketWaves + 1) * 2^bits ints (72 KB at 11, 144 KB at 12)
constexpr int kBucketAhead = 8;                          // rounds of 64 edges whose loads are in flight together

template <class VAL>
__global__ void __launch_bounds__(kBucketWaves * kWave) k_bucket_rows(const uint32_t* __restrict__ keys, const VAL* __restrict__ vals,
                                                                      const int32_t* __restrict__ bstart, int64_t bstride, int ntable,
                                                                      int64_t num_edges, int64_t num_rows, int rb,
                                                                      int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                                      float* __restrict__ w, int stage_cap) {
  extern __shared__ int bsm[];   // wcnt[kBucketWaves][R] | tot[R] | stage[stage_cap] values (VAL)
  const int R = 1 << rb;
  const uint32_t rmask = (uint32_t)R - 1u;
  int* wcnt = bsm;
  int* tot = bsm + kBucketWaves * R;
  VAL* stage = reinterpret_cast<VAL*>(bsm + (kBucketWaves + 1) * R);
  __shared__ int wsum[kBucketWaves];
  const int b = blockIdx.x, lane = lane_id(), wv = threadIdx.x >> 6;
  const int64_t s = bstart[(int64_t)b * bstride];
  const int64_t e = b + 1 < ntable ? (int64_t)bstart[(int64_t)(b + 1) * bstride] : num_edges;
  for (int i = threadIdx.x; i < kBucketWaves * R; i += blockDim.x) wcnt[i] = 0;
  __syncthreads();
  // the wave's contiguous part of the bucket, a whole number of 64-edge rounds
  const int64_t L = ceil_div(e - s, (int64_t)kBucketWaves * kWave) * kWave;
  const int64_t w0 = min(s + wv * L, e), w1 = min(w0 + L, e);
  int* mine = wcnt + wv * R;
  // a bucket that fits the staging tile is put in order in LDS and leaves as whole lines; a larger one scatters its 4-byte
  // stores over its window of col[] (8x the write traffic when 500 buckets' windows compete for the L2s: PMC WRITE_SIZE)
  const bool staged = e - s <= stage_cap;
  {  // kBucketAhead loads in flight (the next group's issued before this group's LDS atomics)
    uint32_t kn[kBucketAhead];
#pragma unroll
    for (int u = 0; u < kBucketAhead; ++u) kn[u] = w0 + lane + u * kWave < w1 ? keys[w0 + lane + u * kWave] : 0u;
    for (int64_t i = w0 + lane; i < w1; i += kBucketAhead * kWave) {
      uint32_t kk[kBucketAhead];
#pragma unroll
      for (int u = 0; u < kBucketAhead; ++u) kk[u] = kn[u];
      const int64_t nx = i + kBucketAhead * kWave;
      if (nx - lane < w1) {
#pragma unroll
        for (int u = 0; u < kBucketAhead; ++u) kn[u] = nx + u * kWave < w1 ? keys[nx + u * kWave] : 0u;
      }
#pragma unroll
      for (int u = 0; u < kBucketAhead; ++u)
        if (i + u * kWave < w1) atomicAdd(&mine[kk[u] & rmask], 1);
    }
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    int acc = 0;
#pragma unroll
    for (int q = 0; q < kBucketWaves; ++q) {
      const int c = wcnt[q * R + r];
      wcnt[q * R + r] = acc;
      acc += c;
    }
    tot[r] = acc;
  }
  __syncthreads();
  {  // exclusive scan of tot[0..R) in place: thread t owns R / blockDim.x consecutive rows
    constexpr int kPerMax = (1 << kBucketMaxRowBits) / (kBucketWaves * kWave);
    const int per = max(R / (int)blockDim.x, 1);
    const int r0 = threadIdx.x * per;
    int v[kPerMax], sum = 0;
#pragma unroll
    for (int j = 0; j < kPerMax; ++j) {
      v[j] = (j < per && r0 + j < R) ? tot[r0 + j] : 0;
      sum += v[j];
    }
    const int inc = wave_inclusive_scan(sum);
    if (lane == kWave - 1) wsum[wv] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int q = 0; q < kBucketWaves; ++q)
      if (q < wv) base += wsum[q];
    int ex = base + inc - sum;
#pragma unroll
    for (int j = 0; j < kPerMax; ++j)
      if (j < per && r0 + j < R) {
        tot[r0 + j] = ex;
        const int64_t row = (int64_t)b * R + r0 + j;
        if (row <= num_rows) rowptr[row] = (int32_t)(s + ex);   // row == num_rows: the run of invalid edges (sentinel key) = nnz
        ex += v[j];
      }
  }
  __syncthreads();
  // the next group's loads are issued before the current group is ranked (two wavefronts per SIMD do not hide an HBM round
  // trip per group on their own)
  uint32_t keyn[kBucketAhead];
  VAL valn[kBucketAhead];
  auto load_group = [&](int64_t g) {
#pragma unroll
    for (int u = 0; u < kBucketAhead; ++u) {
      const int64_t i = g + u * kWave + lane;
      keyn[u] = i < w1 ? keys[i] : 0xffffffffu;
      valn[u] = i < w1 ? vals[i] : VAL(0);
    }
  };
  load_group(w0);
  for (int64_t g0 = w0; g0 < w1; g0 += kBucketAhead * kWave) {
    uint32_t keyv[kBucketAhead];
    VAL valv[kBucketAhead];
#pragma unroll
    for (int u = 0; u < kBucketAhead; ++u) {
      keyv[u] = keyn[u];
      valv[u] = valn[u];
    }
    if (g0 + kBucketAhead * kWave < w1) load_group(g0 + kBucketAhead * kWave);
#pragma unroll
    for (int u = 0; u < kBucketAhead; ++u) {
      if (g0 + u * kWave >= w1) break;                       // wave-uniform
      const bool ok = g0 + u * kWave + lane < w1;
      const uint32_t key = keyv[u];
      const uint32_t k = key & rmask;
      const unsigned long long peers = match_digit<kBucketMaxRowBits>(k, ok);
      const int leader = __ffsll((long long)peers) - 1;
      const int below = __popcll(peers & ((1ull << lane) - 1ull));
      int prev = 0;
      if (ok && lane == leader) {
        prev = mine[k];
        mine[k] = prev + __popcll(peers);
      }
      prev = __shfl(prev, leader);
      if (ok && (int64_t)key < num_rows) {
        const int lpos = tot[k] + prev + below;
        if (staged) {
          stage[lpos] = valv[u];
        } else {
          col[s + lpos] = (int32_t)(uint32_t)valv[u];
          if constexpr (sizeof(VAL) == 8) w[s + lpos] = __uint_as_float((uint32_t)(valv[u] >> 32));
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    // valid entries of the bucket: everything below the run of the sentinel row (invalid edges), if that row lives here
    const int64_t sent = num_rows - (int64_t)b * R;
    const int nvalid = (sent >= 0 && sent < R) ? tot[sent] : (int)(e - s);
    for (int i = threadIdx.x; i < nvalid; i += blockDim.x) {
      const VAL v = stage[i];
      col[s + i] = (int32_t)(uint32_t)v;
      if constexpr (sizeof(VAL) == 8) w[s + i] = __uint_as_float((uint32_t)(v >> 32));
    }
  }
}

// first index of every bucket in the sorted keys (buckets = key >> rb): the multi-pass case of the two-level build
__global__ void __launch_bounds__(256) k_bucket_starts(const uint32_t* __restrict__ keys, int64_t num_edges, int rb, int64_t nbuckets,
                                                       int32_t* __restrict__ bstart) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= num_edges; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prev = i == 0 ? -1 : (int64_t)(keys[i - 1] >> rb);
    const int64_t cur = i == num_edges ? nbuckets : (int64_t)(keys[i] >> rb);
    for (int64_t q = prev + 1; q <= cur && q <= nbuckets; ++q) bstart[q] = (int32_t)i;
  }
}

// shapes the two-level build is meant for: enough rows for a partition pass, buckets that one workgroup finishes quickly
static bool two_level_ok(int64_t num_rows, int64_t num_edges) {
  if (num_rows < (1 << 15) || num_edges < (1 << 18)) return false;
  const int64_t nbuckets = ceil_div(num_rows + 1, (int64_t)1 << kBucketMaxRowBits);
  return num_edges / nbuckets <= (1 << 16);
}

template <class VAL>
static int sort_and_finish_two_level(const int64_t* recipients, const int64_t* senders, const float* edge_w, int64_t num_edges,
                                     int64_t num_rows, int64_t num_cols, int32_t* rowptr, int32_t* col, float* w, int32_t* stats,
                                     GraphWs& g, hipStream_t stream) {
  const int rb = std::min(kBucketMaxRowBits, key_bits(num_rows));
  const int64_t nbuckets = ceil_div(num_rows + 1, (int64_t)1 << rb);
  const int64_t ntiles = ceil_div(num_edges, kRsTile);
  int cur = 0, bits = 0;
  const int upper = key_bits(num_rows) - rb;
  int rc = radix_sort_pairs<VAL, false>(recipients, senders, edge_w, false, num_edges, num_rows, num_cols, stats, g, &cur, stream, rb, 9, &bits);
  if (rc) return rc;
  const int32_t* bstart = g.counts;     // single pass: the scanned [digit][tile] table, digit = bucket
  int64_t bstride = ntiles;
  int ntable = 1 << bits;
  if (bits != upper) {                  // several passes: bucket boundaries from the sorted keys
    const int blocks = (int)std::min<int64_t>(ceil_div(num_edges + 1, 256), 256 * 32);
    k_bucket_starts<<<blocks, 256, 0, stream>>>(g.keys[cur], num_edges, rb, nbuckets, g.block_sums_big);
    FSW_LAUNCH_CHECK();
    bstart = g.block_sums_big;
    bstride = 1;
    ntable = (int)nbuckets + 1;
  }
  // LDS: the tables, and -- when the average bucket (+ 3 %) fits what is left of 160 KB -- a staging tile for the bucket's values
  // (one workgroup per CU then); without it two workgroups share a CU
  const size_t tables = sizeof(int) * (size_t)(kBucketWaves + 1) * ((size_t)1 << rb);
  const size_t room = (size_t)160 * 1024 - 1024 - tables;                      // 1 KB for the static arrays
  const int64_t avg = ceil_div(num_edges, nbuckets);
  const int64_t want = avg + avg / 32 + 256;   // random graphs: max bucket ~ avg + 4 sqrt(avg); a larger bucket scatters
  const int stage_cap = (want * (int64_t)sizeof(VAL) <= (int64_t)room) ? (int)(room / sizeof(VAL)) : 0;
  const size_t lds = tables + (size_t)stage_cap * sizeof(VAL);
  FSW_SET_MAX_LDS_ONCE((&k_bucket_rows<VAL>), 160 * 1024 - 1024);
  k_bucket_rows<VAL><<<(unsigned)nbuckets, kBucketWaves * kWave, lds, stream>>>(g.keys[cur], reinterpret_cast<const VAL*>(g.vals[cur]), bstart,
                                                                              bstride, ntable, num_edges, num_rows, rb, rowptr, col, w,
                                                                              stage_cap);
  FSW_LAUNCH_CHECK();
  return 0;
}

// keys = the sender of every CSR entry, values = the entry's position: the input of the sender-major sort (fsw_graph_transpose)
__global__ void __launch_bounds__(256) k_entry_keys(const int32_t* __restrict__ col, int64_t nnz, int64_t num_cols,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t c = col[i];
    keys[i] = (c < 0 || c >= num_cols) ? (uint32_t)num_cols : (uint32_t)c;
    vals[i] = (uint32_t)i;
  }
}

static int check_chunks(int64_t num_rows, int64_t chunk_rows) {
  FSW_REQUIRE(chunk_rows >= 0 && chunk_rows % kBinRowsPerBlock == 0,
              "fsw_graph_build: chunk_rows must be 0 or a positive multiple of %d", kBinRowsPerBlock);
  FSW_REQUIRE(chunk_rows == 0 || ceil_div(num_rows, chunk_rows) <= kMaxRowChunks, "fsw_graph_build: more than %d row chunks",
              kMaxRowChunks);
  return 0;
}

}  // namespace fsw

using namespace fsw;

extern "C" size_t fsw_graph_workspace_bytes(int64_t num_rows, int64_t num_edges) {
  GraphWs g = carve(nullptr, num_edges, num_rows);
  return g.total;
}

static int graph_build_impl(const int64_t* recipients, const int64_t* senders, const float* edge_w, int64_t num_edges,
                            int64_t num_rows, int64_t num_cols, int64_t chunk_rows, int32_t* rowptr, int32_t* col, float* w,
                            int32_t* perm, int32_t* invperm, int32_t* bin_start, int32_t* stats, void* workspace,
                            size_t workspace_bytes, fsw_stream_t stream_, bool two_level) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(num_rows >= 1 && num_rows < (1ll << 31) - 1 && num_cols >= 1 && num_cols < (1ll << 31) && num_edges >= 0 &&
                  num_edges < (1ll << 31) - kRsTile,
              "fsw_graph_build: sizes must satisfy 1 <= rows, cols < 2^31 and 0 <= edges < 2^31 (got %lld, %lld, %lld)",
              (long long)num_rows, (long long)num_cols, (long long)num_edges);
  FSW_REQUIRE(workspace && workspace_bytes >= fsw_graph_workspace_bytes(num_rows, num_edges),
              "fsw_graph_build: workspace too small");
  FSW_REQUIRE(rowptr && perm && bin_start && stats && (num_edges == 0 || (col && recipients && senders)),
              "fsw_graph_build: null pointer");
  FSW_REQUIRE(!edge_w || w, "fsw_graph_build: edge_w given but w is null");
  if (int rc = check_chunks(num_rows, chunk_rows)) return rc;
  GraphWs g = carve(workspace, num_edges, num_rows);

  FSW_CHECK_HIP(hipMemsetAsync(g.bin_count, 0, kBinTableBytes, stream));
  FSW_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(int32_t) * FSW_NUM_STATS, stream));
  if (num_edges == 0) {
    FSW_CHECK_HIP(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (size_t)(num_rows + 1), stream));
  } else if (two_level && two_level_ok(num_rows, num_edges)) {
    int rc = edge_w ? sort_and_finish_two_level<unsigned long long>(recipients, senders, edge_w, num_edges, num_rows, num_cols, rowptr, col, w, stats, g, stream)
                    : sort_and_finish_two_level<uint32_t>(recipients, senders, edge_w, num_edges, num_rows, num_cols, rowptr, col, w, stats, g, stream);
    if (rc) return rc;
  } else {
    int rc = edge_w ? sort_and_finish<unsigned long long>(recipients, senders, edge_w, num_edges, num_rows, num_cols, rowptr, col, w, stats, g, stream)
                    : sort_and_finish<uint32_t>(recipients, senders, edge_w, num_edges, num_rows, num_cols, rowptr, col, w, stats, g, stream);
    if (rc) return rc;
  }
  return finish_bins(rowptr, num_rows, chunk_rows, perm, invperm, bin_start, stats, g, stream);
}

extern "C" int fsw_graph_build(const int64_t* recipients, const int64_t* senders, const float* edge_w, int64_t num_edges,
                               int64_t num_rows, int64_t num_cols, int64_t chunk_rows, int32_t* rowptr, int32_t* col, float* w,
                               int32_t* perm, int32_t* invperm, int32_t* bin_start, int32_t* stats, void* workspace,
                               size_t workspace_bytes, fsw_stream_t stream) {
  return graph_build_impl(recipients, senders, edge_w, num_edges, num_rows, num_cols, chunk_rows, rowptr, col, w, perm, invperm, bin_start,
                          stats, workspace, workspace_bytes, stream, false);
}

// Same contract and same result, bit for bit; the edges are grouped by a single partition pass + one workgroup per bucket of 2048
// rows (above) when the shape suits that (>= 32768 rows, <= 65536 edges per bucket on average), by the LSD passes otherwise.
extern "C" int fsw_graph_build_two_level(const int64_t* recipients, const int64_t* senders, const float* edge_w, int64_t num_edges,
                                         int64_t num_rows, int64_t num_cols, int64_t chunk_rows, int32_t* rowptr, int32_t* col, float* w,
                                         int32_t* perm, int32_t* invperm, int32_t* bin_start, int32_t* stats, void* workspace,
                                         size_t workspace_bytes, fsw_stream_t stream) {
  return graph_build_impl(recipients, senders, edge_w, num_edges, num_rows, num_cols, chunk_rows, rowptr, col, w, perm, invperm, bin_start,
                          stats, workspace, workspace_bytes, stream, true);
}

extern "C" int fsw_graph_build_coalesced(const int64_t* recipients, const int64_t* senders, const float* edge_w,
                                         const float* edge_feat, int d_edge, int64_t num_edges, int64_t num_rows, int64_t num_cols,
                                         int32_t* rowptr, int32_t* col, float* w, float* ef, int32_t* slot_of_edge, int32_t* perm,
                                         int32_t* invperm, int32_t* bin_start, int32_t* stats, void* workspace,
                                         size_t workspace_bytes, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(num_rows >= 1 && num_rows < (1ll << 31) - 1 && num_cols >= 1 && num_cols < (1ll << 31) - 1 && num_edges >= 0 &&
                  num_edges < (1ll << 31) - kRsTile,
              "fsw_graph_build_coalesced: sizes must satisfy 1 <= rows, cols < 2^31 and 0 <= edges < 2^31");
  FSW_REQUIRE(workspace && workspace_bytes >= fsw_graph_workspace_bytes(num_rows, num_edges),
              "fsw_graph_build_coalesced: workspace too small");
  FSW_REQUIRE(rowptr && perm && bin_start && stats && w && (num_edges == 0 || (col && recipients && senders)),
              "fsw_graph_build_coalesced: null pointer");
  FSW_REQUIRE(d_edge >= 0 && (d_edge == 0 || (edge_feat && ef)), "fsw_graph_build_coalesced: edge features need edge_feat and ef");
  GraphWs g = carve(workspace, num_edges, num_rows);
  FSW_CHECK_HIP(hipMemsetAsync(g.bin_count, 0, kBinTableBytes, stream));
  FSW_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(int32_t) * FSW_NUM_STATS, stream));
  if (num_edges == 0) {
    FSW_CHECK_HIP(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (size_t)(num_rows + 1), stream));
    return finish_bins(rowptr, num_rows, 0, perm, invperm, bin_start, stats, g, stream);
  }
  if (slot_of_edge) FSW_CHECK_HIP(hipMemsetAsync(slot_of_edge, 0xff, sizeof(int32_t) * (size_t)num_edges, stream));
  const int edge_blocks = (int)std::min<int64_t>(ceil_div(num_edges, 256), 256 * 16);
  int cur = 0, rc;
  // 1. stable sort by sender, carrying the edge id (validates both endpoints and the weights)
  if ((rc = radix_sort_pairs<uint32_t, true>(senders, recipients, edge_w, false, num_edges, num_cols, num_rows, stats, g, &cur, stream))) return rc;
  // 2. stable sort by recipient: (recipient, sender) order
  k_gather_keys<<<edge_blocks, 256, 0, stream>>>(recipients, reinterpret_cast<const uint32_t*>(g.vals[cur]), g.keys[cur], num_edges,
                                                 num_rows, num_cols, g.keys[cur ^ 1]);
  FSW_LAUNCH_CHECK();
  {  // the gathered keys live in the other key buffer: make it the current one, values stay where they are
    uint32_t* t = g.keys[cur];
    g.keys[cur] = g.keys[cur ^ 1];
    g.keys[cur ^ 1] = t;
  }
  if ((rc = radix_sort_pairs<uint32_t, true>(nullptr, nullptr, nullptr, true, num_edges, num_rows, num_cols, stats, g, &cur, stream))) return rc;
  // 3. runs of equal (recipient, sender) -> one entry each
  const uint32_t* keys = g.keys[cur];
  const uint32_t* eid = reinterpret_cast<const uint32_t*>(g.vals[cur]);
  int32_t* head = reinterpret_cast<int32_t*>(g.vals[cur ^ 1]);
  int32_t* pos = head + num_edges;
  uint32_t* keyc = g.keys[cur ^ 1];
  k_mark_heads<<<edge_blocks, 256, 0, stream>>>(keys, eid, senders, num_edges, num_rows, head);
  FSW_LAUNCH_CHECK();
  FSW_CHECK_HIP(hipMemcpyAsync(pos, head, sizeof(int32_t) * (size_t)num_edges, hipMemcpyDeviceToDevice, stream));
  if ((rc = exclusive_scan_i32(pos, num_edges, g.block_sums, stream))) return rc;
  int32_t* nnz_dev = stats + FSW_STAT_NNZ;
  k_emit_coalesced<<<edge_blocks, 256, 0, stream>>>(keys, eid, head, pos, senders, edge_w, edge_feat, d_edge, num_edges, num_rows,
                                                    keyc, col, w, ef, slot_of_edge, nnz_dev);
  FSW_LAUNCH_CHECK();
  k_rowptr_from_keys<<<edge_blocks, 256, 0, stream>>>(keyc, nnz_dev, num_rows, rowptr);
  FSW_LAUNCH_CHECK();
  return finish_bins(rowptr, num_rows, 0, perm, invperm, bin_start, stats, g, stream);
}

// Sender-major view of a built graph: cptr[j] .. cptr[j + 1] index the entries of `order` that list, in CSR order, the CSR
// positions e with col[e] == j.  The backward pass sums the stored key gradients over it (fsw_segment_sum_rows_f32) instead of
// scattering them with float atomics.  A stable LSD sort of (col[e], e): the CSR build's own passes.
extern "C" int fsw_graph_transpose(const int32_t* col, int64_t nnz, int64_t num_cols, int32_t* cptr, int32_t* order, void* workspace,
                                   size_t workspace_bytes, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(num_cols >= 1 && num_cols < (1ll << 31) - 1 && nnz >= 0 && nnz < (1ll << 31) - kRsTile,
              "fsw_graph_transpose: sizes must satisfy 1 <= cols < 2^31 and 0 <= nnz < 2^31");
  FSW_REQUIRE(cptr && workspace && workspace_bytes >= fsw_graph_workspace_bytes(num_cols, nnz) && (nnz == 0 || (col && order)),
              "fsw_graph_transpose: null pointer or workspace too small");
  if (nnz == 0) {
    FSW_CHECK_HIP(hipMemsetAsync(cptr, 0, sizeof(int32_t) * (size_t)(num_cols + 1), stream));
    return 0;
  }
  GraphWs g = carve(workspace, nnz, num_cols);
  const int blocks = (int)std::min<int64_t>(ceil_div(nnz + 1, 256), 256 * 32);
  k_entry_keys<<<blocks, 256, 0, stream>>>(col, nnz, num_cols, g.keys[0], reinterpret_cast<uint32_t*>(g.vals[0]));
  FSW_LAUNCH_CHECK();
  int cur = 0;
  if (int rc = radix_sort_pairs<uint32_t, true>(nullptr, nullptr, nullptr, true, nnz, num_cols, num_cols, nullptr, g, &cur, stream)) return rc;
  k_finish_csr<uint32_t><<<blocks, 256, 0, stream>>>(g.keys[cur], reinterpret_cast<const uint32_t*>(g.vals[cur]), nnz, num_cols, cptr, order,
                                                     nullptr);
  FSW_LAUNCH_CHECK();
  return 0;
}
