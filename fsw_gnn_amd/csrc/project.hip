// Per-slice projection  Xp[n, S] = X[n, d] . V[S, d]^T  on the fp32 matrix cores.  gfx950.
//
// Replaces torch.tensordot(X, projVecs, dims=((-1,),(1,))) (reference fsw_embedding.py:909-913), the one
// dense contraction of the path.  v_mfma_f32_32x32x2_f32 keeps exact fp32 products and accumulation
// (bit-for-bit a k-ordered fmaf chain), which the 1e-5 parity target needs; gfx950 has no xf32.
// Xp is written row-major with the slice index minor so that the neighbourhood kernels gather one
// contiguous 256-byte run per (neighbour, 64-slice chunk).
//
// Tiling: 256 threads = 4 waves as 2x2; workgroup tile 128 rows x 128 slices, K step 32 through LDS;
// each wave owns a 64x64 sub-tile = 2x2 MFMA blocks of 32x32 (64 accumulator registers).
#include <stdlib.h>
#include <algorithm>
#include "fsw_common.h"

namespace fsw {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PBM = 128, PBN = 128, PBK = 32, PLD = PBK + 1;

// Stage 128 rows x 32 k of a row-major matrix into LDS; rowp(r) returns the global row pointer of tile row r
// (or nullptr for rows past the end, which are zero filled).  thread t covers k-quad (t & 7) of rows (t >> 3) + 32 i.
template <bool VEC, class RowPtr>
__device__ __forceinline__ void stage_tile(RowPtr rowp, int k0, int kmax, float (*dst)[PLD], int& nonfinite,
                                           float* __restrict__ copy_dst, int64_t ld_copy, int64_t row0) {
  const int kq = (threadIdx.x & 7) * 4;
  const int r0 = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 32 * i;
    const float* src = rowp(r);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (src) {
      const float* p = src + k0 + kq;
      if (VEC && k0 + kq + 3 < kmax) {
        float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + kq + j < kmax) v[j] = p[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      nonfinite |= !(fabsf(v[j]) <= 3.402823466e38f);
      dst[r][kq + j] = v[j];
      if (copy_dst && src && k0 + kq + j < kmax) copy_dst[(row0 + r) * ld_copy + k0 + kq + j] = v[j];
    }
  }
}

// Columns 0 .. S-1 of the product go to Xp (rows of V); columns S .. S+H2-1 are an optional second block
// Y2 = X . W2^T + b2 (rows of W2): the x half of FSW_conv's first Linear layer (conv_fused.hip).
template <bool VEC>
__global__ void __launch_bounds__(256) k_project(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                 const float* __restrict__ V, int S, int64_t ldv, float* __restrict__ Xp,
                                                 int64_t ldp, int32_t* __restrict__ stats, int nct,
                                                 float* __restrict__ x_copy, int64_t ld_copy,
                                                 const float* __restrict__ W2, int H2, int64_t ldw2,
                                                 const float* __restrict__ b2, float* __restrict__ Y2, int64_t ldy2,
                                                 const int32_t* __restrict__ row_map) {
  __shared__ float As[PBM][PLD];
  __shared__ float Bs[PBN][PLD];
  const int ct = blockIdx.x % nct;
  const int64_t rt = blockIdx.x / nct;
  const int64_t row0 = rt * PBM;
  const int col0 = ct * PBN;
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const int N = S + H2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto a_row = [&](int r) -> const float* { return row0 + r < n ? X + (row0 + r) * ldx : nullptr; };
  auto b_row = [&](int r) -> const float* {
    const int c = col0 + r;
    return c < S ? V + (int64_t)c * ldv : (c < N ? W2 + (int64_t)(c - S) * ldw2 : nullptr);
  };
  int nonfinite = 0, dummy = 0;
  for (int k0 = 0; k0 < d; k0 += PBK) {
    stage_tile<VEC>(a_row, k0, d, As, nonfinite, ct == 0 ? x_copy : nullptr, ld_copy, row0);
    stage_tile<VEC>(b_row, k0, d, Bs, dummy, nullptr, 0, 0);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < PBK; kk += 2) {
      const float a0 = As[wr * 64 + fr][kk + fh];
      const float a1 = As[wr * 64 + 32 + fr][kk + fh];
      const float b0 = Bs[wc * 64 + fr][kk + fh];
      const float b1 = Bs[wc * 64 + 32 + fr][kk + fh];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = col0 + wc * 64 + j * 32 + fr;
      if (c >= N) continue;
      const bool second = c >= S;
      float* dst = second ? Y2 + (c - S) : Xp + c;
      const int64_t ld = second ? ldy2 : ldp;
      const float add = (second && b2) ? b2[c - S] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gr = row0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (gr < n) dst[((second && row_map) ? (int64_t)row_map[gr] : gr) * ld] = acc[i][j][r] + add;
      }
    }
  if (stats && nonfinite) atomicOr(&stats[FSW_STAT_FLAGS], FSW_FLAG_X_NONFINITE);
}

// ---- B-stationary variant (d <= 128, 16-byte aligned rows) --------------------------------------------------------
// The weight block [V; W2] is small (N x d <= 384 x 128 floats), so every wave keeps its 32-column slab of it in
// registers for the whole kernel (one VGPR per k-pair: exactly the MFMA B operand) and the workgroup -- one wave per
// slab, up to 12 -- streams 32-row tiles of X through a double-buffered LDS tile.  X is read from HBM exactly
// once, nothing but X tiles moves through LDS, and the MFMA pipe sees back-to-back accumulate chains:
//   per tile:  prefetch tile t+1 (global -> registers) | d/2 MFMAs from LDS tile t | store the 32x32 block,
//              registers -> LDS tile t+1, one barrier.
constexpr int BS_ROWS = 32, BS_LD = 129;

template <int KQ>
__global__ void __launch_bounds__(768) k_project_bs(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                    const float* __restrict__ V, int S, int64_t ldv,
                                                    float* __restrict__ Xp, int64_t ldp, int32_t* __restrict__ stats,
                                                    float* __restrict__ x_copy, int64_t ld_copy,
                                                    const float* __restrict__ W2, int H2, int64_t ldw2,
                                                    const float* __restrict__ b2, float* __restrict__ Y2, int64_t ldy2,
                                                    const int32_t* __restrict__ row_map, int64_t ntiles, int nslab_waves) {
  __shared__ float As[2][BS_ROWS][BS_LD];
  __shared__ int rmap[2][BS_ROWS];
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int N = S + H2;
  const int slab = blockIdx.y * nslab_waves + wv;
  const bool slab_active = wv < nslab_waves && slab * 32 < N;
  const int c = slab_active ? slab * 32 + fr : N;       // this lane's output column (N = none)

  // the slab of [V; W2] this wave multiplies by, as MFMA B operands: b[q] = W[c][2q + fh]
  float b[KQ];
  {
    const float* wrow = c < S ? V + (int64_t)c * ldv : (c < N ? W2 + (int64_t)(c - S) * ldw2 : nullptr);
#pragma unroll
    for (int q = 0; q < KQ; ++q) b[q] = (wrow && 2 * q + fh < d) ? wrow[2 * q + fh] : 0.f;
  }
  const bool second = c >= S;
  float* dst = c < N ? (second ? Y2 + (c - S) : Xp + c) : nullptr;
  const int64_t ldd = second ? ldy2 : ldp;
  const float add = (second && b2 && c < N) ? b2[c - S] : 0.f;

  // zero the k-padding columns of both LDS tiles once (columns d .. 2 KQ)
  for (int i = threadIdx.x; i < 2 * BS_ROWS * BS_LD; i += blockDim.x) (&As[0][0][0])[i] = 0.f;
  __syncthreads();

  const int d4 = d >> 2;                      // float4 per row
  const int per_tile = BS_ROWS * d4;          // float4 per tile (<= 1024)
  int nonfinite = 0;
  auto load_tile = [&](int64_t tile, float4& q0, float4& q1) {
    const int64_t row0 = tile * BS_ROWS;
    q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    q1 = q0;
    const int i0 = threadIdx.x, i1 = threadIdx.x + blockDim.x;
    if (i0 < per_tile) {
      const int r = i0 / d4, c4 = i0 - r * d4;
      if (row0 + r < n) q0 = *reinterpret_cast<const float4*>(X + (row0 + r) * ldx + 4 * c4);
    }
    if (i1 < per_tile) {
      const int r = i1 / d4, c4 = i1 - r * d4;
      if (row0 + r < n) q1 = *reinterpret_cast<const float4*>(X + (row0 + r) * ldx + 4 * c4);
    }
  };
  auto store_tile = [&](int buf, int64_t tile, const float4& q0, const float4& q1) {
    const int64_t row0 = tile * BS_ROWS;
    if (row_map && threadIdx.x < BS_ROWS) rmap[buf][threadIdx.x] = row0 + threadIdx.x < n ? row_map[row0 + threadIdx.x] : 0;
    const int idx[2] = {(int)threadIdx.x, (int)(threadIdx.x + blockDim.x)};
    const float4 q[2] = {q0, q1};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (idx[u] < per_tile) {
        const int r = idx[u] / d4, c4 = idx[u] - r * d4;
        const float v[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nonfinite |= !(fabsf(v[j]) <= 3.402823466e38f);
          As[buf][r][4 * c4 + j] = v[j];
          if (blockIdx.y == 0 && x_copy && row0 + r < n) x_copy[(row0 + r) * ld_copy + 4 * c4 + j] = v[j];
        }
      }
    }
  };

  // software pipeline, two tiles deep: registers hold tile t+1 (loaded during iteration t-1) and go to LDS at the top
  // of iteration t; the HBM loads of tile t+2 are issued right after and have the whole iteration to land
  int64_t tile = blockIdx.x;
  float4 q0, q1;
  if (tile < ntiles) {
    load_tile(tile, q0, q1);
    store_tile(0, tile, q0, q1);
  }
  if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x, q0, q1);
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t next = tile + gridDim.x, next2 = next + gridDim.x;
    if (next < ntiles) store_tile(buf ^ 1, next, q0, q1);
    if (next2 < ntiles) load_tile(next2, q0, q1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ap = &As[buf][fr][fh];
    if (slab_active) {   // wave-uniform: helper waves (no slab) only move X tiles
#pragma unroll
      for (int q = 0; q < KQ; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * q], b[q], acc, 0, 0, 0);
    }
    if (dst) {
      const int64_t row0 = tile * BS_ROWS;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gr = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (gr < n) dst[((second && row_map) ? (int64_t)rmap[buf][(r & 3) + 8 * (r >> 2) + 4 * fh] : gr) * ldd] = acc[r] + add;
      }
    }
    __syncthreads();
    buf ^= 1;
  }
  if (stats && nonfinite) atomicOr(&stats[FSW_STAT_FLAGS], FSW_FLAG_X_NONFINITE);
}

// ---- bf16x3 variant: fp32-accurate product on the bf16 matrix cores ------------------------------------------------
// x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) carries 24 significand bits, the same
// for the weights; the product keeps the six terms down to 2^-16 relative (x1 v1, x1 v2, x2 v1, x1 v3, x2 v2, x3 v1),
// each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, and drops terms below 2^-24.  Six bf16 MFMAs at 16x
// the fp32 MFMA rate = 2.7x the throughput of the exact-fp32 kernel above at the same ~1e-7 accuracy (bf16 keeps
// fp32's exponent range, so no scaling is needed).  Same B-stationary structure as k_project_bs: every wave keeps
// the three bf16 planes of its 32-column weight slab in registers, X tiles are split once when they enter LDS.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// bf16 per LDS row of the A planes: 16 KS + 8 (16-byte pad: row stride = 4 banks mod 64, conflict-free 16-byte fragment reads)
#ifndef FSW_PROJECT_DIRECT_C
#define FSW_PROJECT_DIRECT_C 0   // 1: C tile stored straight from the accumulators instead of through the LDS staging tile
#endif
#ifndef FSW_PROJECT_STAMPS
#define FSW_PROJECT_STAMPS 0   // 1: s_memtime stamps at the phase boundaries of k_project_bf3's tile loop, summed per wave role
#endif                         //    (tools/exp_project_stamps.py reads them through fsw_debug_project_stamps)
#if FSW_PROJECT_STAMPS
__device__ unsigned long long g_proj_stamps[1024][12][8];   // [workgroup][wave][phase]: cycles; [..][7]: iterations
#define FSW_STAMP(i)                                         \
  do {                                                       \
    const unsigned long long now_ = clock64();               \
    stamp_sum[i] += now_ - stamp_last;                       \
    stamp_last = now_;                                       \
  } while (0)
#else
#define FSW_STAMP(i) do { } while (0)
#endif
#ifndef FSW_PROJECT_STAGGER
#define FSW_PROJECT_STAGGER 0  // 1: the matrix block first / in the middle / last for the three wavefronts of a SIMD (see the tile loop)
#endif
#ifndef FSW_PROJECT_ABL
#define FSW_PROJECT_ABL 0   // timing experiments (tools/exp_variants.sh): 1 = no MFMA, 2 = no output stores
#endif

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r = v - (float)a;
  b = (__bf16)r;
  c = (__bf16)(r - (float)b);
}

template <int KS>
__global__ void __launch_bounds__(KS <= 8 ? 768 : 512) k_project_bf3(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                     const float* __restrict__ V, int S, int64_t ldv,
                                                     float* __restrict__ Xp, int64_t ldp, int32_t* __restrict__ stats,
                                                     float* __restrict__ x_copy, int64_t ld_copy,
                                                     const float* __restrict__ W2, int H2, int64_t ldw2,
                                                     const float* __restrict__ b2, float* __restrict__ Y2, int64_t ldy2,
                                                     const int32_t* __restrict__ row_map, int64_t ntiles, int nslab_waves,
                                                     int nsl1, int y2_vec, int cbufs) {
  // cbufs: buffers of the C staging tile.  2: one barrier per tile orders everything; 1: a second barrier per tile, half the
  // staging LDS -- with the slabs split over two column groups (6 slab waves each) two workgroups then share a CU and cover
  // each other's load / matrix / store phases.
  // KS <= 8 (d <= 128): up to 12 slab waves keep their whole slab of [V; W2] in registers (24 KS registers each).
  // KS == 16 (d <= 256): the slab takes 192 registers, so a workgroup is 4 slab waves + 4 waves that only move X, two waves
  // per SIMD at up to 256 registers, and the column groups beyond the first re-read X (from L2 / the Infinity Cache mostly).
  constexpr int B3_LD = 16 * KS + 8;
  constexpr int NQ = KS <= 8 ? 3 : 4;     // float4 of an X tile per thread (32 rows x 4 KS float4 over >= 384 (d <= 128) / 512 threads)
  // LDS: A planes [2][3][32][B3_LD] bf16 | C staging [2][32][ldc] float | row map [3][32] int
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __bf16 (*APlanes)[3][BS_ROWS][B3_LD];
  APlanes As = reinterpret_cast<APlanes>(smem);
  const int ldc = nslab_waves * 32 + 4;
  float* Cs = reinterpret_cast<float*>(smem + sizeof(__bf16) * 2 * 3 * BS_ROWS * B3_LD);
  int* rmap = reinterpret_cast<int*>(Cs + cbufs * BS_ROWS * ldc);

  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  // slabs 0 .. nsl1-1 cover the S projection columns, the following slabs the H2 columns of the second block
  // (each block starts on a slab boundary so that both are 16-byte aligned in the C staging tile)
  const int slab = blockIdx.y * nslab_waves + wv;
  const int nsl2 = (H2 + 31) >> 5;
  const bool slab_active = wv < nslab_waves && slab < nsl1 + nsl2;
  const bool second = slab >= nsl1;
  const int c = second ? (slab - nsl1) * 32 + fr : slab * 32 + fr;       // column inside its block
  const bool col_ok = slab_active && (second ? c < H2 : c < S);

  // weight slab as MFMA B operands: lane (fr, fh), k-step s holds W[c][16 s + 8 fh + j], j = 0..7, in three planes
  bf16x8 bw[3][KS];
  {
    const float* wrow = col_ok ? (second ? W2 + (int64_t)c * ldw2 : V + (int64_t)c * ldv) : nullptr;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * fh + j;
        const float v = (wrow && k < d) ? wrow[k] : 0.f;
        __bf16 h1, h2, h3;
        split3(v, h1, h2, h3);
        bw[0][s][j] = h1;
        bw[1][s][j] = h2;
        bw[2][s][j] = h3;
      }
  }
  const float add = (col_ok && second && b2) ? b2[c] : 0.f;

  for (int i = threadIdx.x; i < 2 * 3 * BS_ROWS * B3_LD / 2; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;   // k padding
  __syncthreads();

  const int d4 = d >> 2;
  const int per_tile = BS_ROWS * d4;
  // i / d4, i / w1, i / w2 for the element indices i < 32 * 96 of a tile: multiply-shift instead of a division by a run-time value
  // (~35 instructions each, eleven of them per thread and tile: the s_memtime stamps of tools/exp_project_stamps.py put the address
  // arithmetic of the load / split / write-out phases at 46 % of a tile).  Exact for i * divisor < 2^20.
  auto magic = [](int dv) { return dv > 0 ? (unsigned)(((1u << 20) + dv - 1) / dv) : 0u; };
  auto fdiv = [](int i, unsigned m) { return (int)(((unsigned)i * m) >> 20); };
  const unsigned m_d4 = magic(d4);
  int nonfinite = 0;
  struct TileRegs {
    float4 q[NQ];
  };
  auto load_tile = [&](int64_t tile, TileRegs& t) {
    const int64_t row0 = tile * BS_ROWS;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      t.q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int i = threadIdx.x + u * blockDim.x;
      if (i < per_tile) {
        const int r = fdiv(i, m_d4), c4 = i - r * d4;
        if (row0 + r < n) t.q[u] = *reinterpret_cast<const float4*>(X + (row0 + r) * ldx + 4 * c4);
      }
    }
  };
  // rslot: slot of the tile's row map (three slots: the write-out of tile t may still read its slot while the A planes of tile
  // t + 2 -- same A buffer, same parity -- are being written; with two slots a fast wavefront 0 could overwrite the map under a
  // slow wavefront's write-out)
  auto store_tile = [&](int buf, int rslot, int64_t tile, const TileRegs& t) {
    const int64_t row0 = tile * BS_ROWS;
    if (threadIdx.x < BS_ROWS)
      rmap[rslot * BS_ROWS + threadIdx.x] = (row_map && row0 + threadIdx.x < n) ? row_map[row0 + threadIdx.x] : (int)(row0 + threadIdx.x);
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int i = threadIdx.x + u * blockDim.x;
      if (i < per_tile) {
        const int r = fdiv(i, m_d4), c4 = i - r * d4;
        const float v[4] = {t.q[u].x, t.q[u].y, t.q[u].z, t.q[u].w};
        bf16x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nonfinite |= !(fabsf(v[j]) <= 3.402823466e38f);
          __bf16 h1, h2, h3;
          split3(v[j], h1, h2, h3);
          p1[j] = h1;
          p2[j] = h2;
          p3[j] = h3;
          if (blockIdx.y == 0 && x_copy && row0 + r < n) x_copy[(row0 + r) * ld_copy + 4 * c4 + j] = v[j];
        }
        *reinterpret_cast<bf16x4*>(&As[buf][0][r][4 * c4]) = p1;
        *reinterpret_cast<bf16x4*>(&As[buf][1][r][4 * c4]) = p2;
        *reinterpret_cast<bf16x4*>(&As[buf][2][r][4 * c4]) = p3;
      }
    }
  };
  // C staging tile -> global memory as whole rows: 16-byte stores, every row of Xp / Y2 is one contiguous run
  const int g1 = min(nsl1 - (int)blockIdx.y * nslab_waves, nslab_waves);   // slabs of this column group in block 1
  const int w1 = max(g1, 0) * 8;                                           // float4 per staged row, first block
  const int w2 = (nslab_waves - max(g1, 0)) * 8;                           // float4 per staged row, second block
  const int col1 = blockIdx.y * nslab_waves * 32;                          // first Xp column of this group
  const int col2 = max((int)blockIdx.y * nslab_waves - nsl1, 0) * 32;      // first Y2 column of this group
  const unsigned m_w1 = magic(w1), m_w2 = magic(w2);
  auto write_out = [&](int cb, int64_t tile, int rb) {
    const int64_t row0 = tile * BS_ROWS;
    const float* cs = Cs + cb * BS_ROWS * ldc;
    for (int i = threadIdx.x; i < BS_ROWS * w1; i += blockDim.x) {
      const int r = fdiv(i, m_w1), c4 = i - r * w1;
      if (row0 + r < n && col1 + 4 * c4 < ldp)   // Xp rows are padded to ldp >= 32 ceil(S/32): whole float4 always fit
        *reinterpret_cast<float4*>(Xp + (row0 + r) * ldp + col1 + 4 * c4) = *reinterpret_cast<const float4*>(cs + r * ldc + 4 * c4);
    }
    for (int i = threadIdx.x; i < BS_ROWS * w2; i += blockDim.x) {
      const int r = fdiv(i, m_w2), c4 = i - r * w2;
      if (row0 + r >= n) continue;
      const int cc = col2 + 4 * c4;
      const float4 v = *reinterpret_cast<const float4*>(cs + r * ldc + max(g1, 0) * 32 + 4 * c4);
      float* yrow = Y2 + (int64_t)rmap[rb * BS_ROWS + r] * ldy2;
      if (y2_vec && cc + 3 < H2) {
        *reinterpret_cast<float4*>(yrow + cc) = v;
      } else {
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (cc + j < H2) yrow[cc + j] = e[j];
      }
    }
  };

  // software pipeline, two tiles deep: registers hold tile t+1 (loaded during iteration t-1) and go to LDS at the top
  // of iteration t; the HBM loads of tile t+2 are issued right after and have the whole iteration to land.  The C
  // tile is double buffered too, so one barrier per tile orders everything.
  int64_t tile = blockIdx.x;
  TileRegs tr;
  if (tile < ntiles) {
    load_tile(tile, tr);
    store_tile(0, 0, tile, tr);
  }
  if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x, tr);
  __syncthreads();
  int buf = 0, rs = 0;                                   // A / C buffer and row-map slot of the current tile
#if FSW_PROJECT_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_last = clock64();
#endif
#if FSW_PROJECT_STAGGER && !FSW_PROJECT_DIRECT_C
  // the three blocks of an iteration
  auto move_in = [&](int64_t t) {                        // tile t + 1 from registers into the A planes, loads of tile t + 2 issued
    const int64_t next = t + gridDim.x, next2 = next + gridDim.x;
    if (next < ntiles) store_tile(buf ^ 1, rs == 2 ? 0 : rs + 1, next, tr);
    FSW_STAMP(0);                                        // waited for the tile's loads, split, wrote the A planes
    if (next2 < ntiles) load_tile(next2, tr);
    FSW_STAMP(1);                                        // loads of tile t + 2 issued
  };
  auto matrix = [&](int64_t t) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (slab_active && !(FSW_PROJECT_ABL & 1)) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[buf][0][fr][16 * s + 8 * fh]);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&As[buf][1][fr][16 * s + 8 * fh]);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(&As[buf][2][fr][16 * s + 8 * fh]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bw[0][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bw[1][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[2][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bw[0][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[1][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[0][s], acc, 0, 0, 0);
      }
      FSW_STAMP(2);                                      // matrix instructions issued
#if FSW_PROJECT_DIRECT_C
      // straight from the accumulators: every store instruction covers two rows x 32 columns = two whole 128-byte lines
      if (col_ok && !(FSW_PROJECT_ABL & 2)) {
        const int64_t row0 = t * BS_ROWS;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int R = (r & 3) + 8 * (r >> 2) + 4 * fh;   // C/D map of the 32x32 MFMA
          if (row0 + R < n) {
            if (second) Y2[(int64_t)rmap[rs * BS_ROWS + R] * ldy2 + c] = acc[r] + add;
            else Xp[(row0 + R) * ldp + c] = acc[r] + add;
          }
        }
      }
#else
      if (cbufs == 1) __syncthreads();   // uniform: the previous tile has left the single staging buffer
      float* cs = Cs + (cbufs == 2 ? buf : 0) * BS_ROWS * ldc + wv * 32 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * fh) * ldc] = acc[r] + add;   // C/D map of the 32x32 MFMA
      FSW_STAMP(3);                                      // accumulators drained into the staging tile
#endif
    } else if (cbufs == 1) {
      __syncthreads();                   // helper waves (no slab) join the same barrier
    }
  };
  // The matrix block at a different place of the iteration for each of the three wavefronts that share a SIMD (wavefront w runs on
  // SIMD w % 4): one group's matrix instructions run under the other groups' loads / splits / stores instead of all twelve
  // wavefronts moving data together and then queueing for the matrix pipe together.  The write-out of tile t moves into iteration
  // t + 1 (its staging buffer and row-map slot are not reused before the barrier that ends it).
  const int grp = cbufs == 2 ? (wv >> 2) % 3 : 2;
  bool have_prev = false;
  int64_t prev_tile = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    auto out_prev = [&]() {
      if (have_prev && !(FSW_PROJECT_ABL & 2)) write_out(cbufs == 2 ? buf ^ 1 : 0, prev_tile, rs == 0 ? 2 : rs - 1);
      FSW_STAMP(5);
    };
    FSW_STAMP(6);
    if (grp == 0) {
      matrix(tile);
      move_in(tile);
      out_prev();
    } else if (grp == 1) {
      move_in(tile);
      matrix(tile);
      out_prev();
    } else {
      move_in(tile);
      out_prev();
      matrix(tile);
    }
    __syncthreads();
    FSW_STAMP(4);                                        // barrier
#if FSW_PROJECT_STAMPS
    stamp_sum[7] += 1;
#endif
    have_prev = true;
    prev_tile = tile;
    buf ^= 1;
    rs = rs == 2 ? 0 : rs + 1;
  }
  if (have_prev && !(FSW_PROJECT_ABL & 2)) write_out(cbufs == 2 ? buf ^ 1 : 0, prev_tile, rs == 0 ? 2 : rs - 1);
#else
  // (written out inline, not through the lambdas above: the same statements through move_in() / matrix() compiled to a loop that
  // measured 0.70-0.72 ms against 0.65-0.68 ms for this one)
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t next = tile + gridDim.x, next2 = next + gridDim.x;
    FSW_STAMP(6);                                        // tail of the previous iteration: output stores issued
    if (next < ntiles) store_tile(buf ^ 1, rs == 2 ? 0 : rs + 1, next, tr);
    FSW_STAMP(0);                                        // waited for the tile's loads, split, wrote the A planes
    if (next2 < ntiles) load_tile(next2, tr);
    FSW_STAMP(1);                                        // loads of tile t + 2 issued
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (slab_active && !(FSW_PROJECT_ABL & 1)) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[buf][0][fr][16 * s + 8 * fh]);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&As[buf][1][fr][16 * s + 8 * fh]);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(&As[buf][2][fr][16 * s + 8 * fh]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bw[0][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bw[1][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[2][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bw[0][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[1][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[0][s], acc, 0, 0, 0);
      }
      FSW_STAMP(2);                                      // matrix instructions issued
#if FSW_PROJECT_DIRECT_C
      // straight from the accumulators: every store instruction covers two rows x 32 columns = two whole 128-byte lines
      if (col_ok && !(FSW_PROJECT_ABL & 2)) {
        const int64_t row0 = tile * BS_ROWS;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int R = (r & 3) + 8 * (r >> 2) + 4 * fh;   // C/D map of the 32x32 MFMA
          if (row0 + R < n) {
            if (second) Y2[(int64_t)rmap[rs * BS_ROWS + R] * ldy2 + c] = acc[r] + add;
            else Xp[(row0 + R) * ldp + c] = acc[r] + add;
          }
        }
      }
#else
      if (cbufs == 1) __syncthreads();   // uniform: the previous tile has left the single staging buffer
      float* cs = Cs + (cbufs == 2 ? buf : 0) * BS_ROWS * ldc + wv * 32 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * fh) * ldc] = acc[r] + add;   // C/D map of the 32x32 MFMA
      FSW_STAMP(3);                                      // accumulators drained into the staging tile
#endif
    } else if (cbufs == 1) {
      __syncthreads();                   // helper waves (no slab) join the same barrier
    }
    __syncthreads();
    FSW_STAMP(4);                                        // barrier
#if !FSW_PROJECT_DIRECT_C
    if (!(FSW_PROJECT_ABL & 2)) write_out(cbufs == 2 ? buf : 0, tile, rs);
#endif
    FSW_STAMP(5);                                        // staging tile read, output stores issued
#if FSW_PROJECT_STAMPS
    stamp_sum[7] += 1;
#endif
    buf ^= 1;
    rs = rs == 2 ? 0 : rs + 1;
  }
#endif
  if (stats && nonfinite) atomicOr(&stats[FSW_STAT_FLAGS], FSW_FLAG_X_NONFINITE);
#if FSW_PROJECT_STAMPS
  if (lane == 0 && blockIdx.x < 1024 && blockIdx.y == 0 && wv < 12)
    for (int i = 0; i < 8; ++i) g_proj_stamps[blockIdx.x][wv][i] += stamp_sum[i];
#endif
}

// ---- narrow outputs (S <= 64, no second block: one rank's slice block of a slice-sharded layer, dist.py) ----------------------------
// With one or two 32-column slabs the workgroup kernel above has one or two wavefronts on the matrix pipe and six helpers, one
// workgroup per CU, and a barrier per 32 rows: 0.24 ms for 32 slices of the 1M x 128 input, which it only has to read once (0.1 ms).
// Here every WAVEFRONT is on its own: it keeps the slab's three bf16 planes in registers, reads its 32-row tile of X straight from
// global memory in the matrix instruction's A layout (lane (r, h), k-step s: X[row0 + r][16 s + 8 h .. + 7] = two 16-byte loads; a
// 64-byte segment per row and k-step over the two h lanes), splits it in registers and stores C from the accumulators (two whole
// 128-byte lines per instruction).  No LDS, no barrier; blockIdx.y = slab.
template <int KS>
__global__ void __launch_bounds__(256, 2) k_project_narrow(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                           const float* __restrict__ V, int S, int64_t ldv, float* __restrict__ Xp,
                                                           int64_t ldp, int32_t* __restrict__ stats, int64_t ntiles) {
  const int lane = lane_id();
  const int fr = lane & 31, fh = lane >> 5;
  const int slab = blockIdx.y;
  const int c = slab * 32 + fr;
  bf16x8 bw[3][KS];
  {
    const float* wrow = c < S ? V + (int64_t)c * ldv : nullptr;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * fh + j;
        const float v = (wrow && k < d) ? wrow[k] : 0.f;
        __bf16 h1, h2, h3;
        split3(v, h1, h2, h3);
        bw[0][s][j] = h1;
        bw[1][s][j] = h2;
        bw[2][s][j] = h3;
      }
  }
  int nonfinite = 0;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t row0 = tile * BS_ROWS;
    const int64_t row = min(row0 + fr, n - 1);              // rows past the end re-read the last row; their outputs are not stored
    const float* xr = X + row * ldx + 8 * fh;
    float4 q[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = min(16 * s, d - 16);                   // d is a multiple of 16 (launch condition): k-steps past d re-read the
      q[s][0] = *reinterpret_cast<const float4*>(xr + k0);  // last one against zero weights
      q[s][1] = *reinterpret_cast<const float4*>(xr + k0 + 4);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float v[8] = {q[s][0].x, q[s][0].y, q[s][0].z, q[s][0].w, q[s][1].x, q[s][1].y, q[s][1].z, q[s][1].w};
      bf16x8 a1, a2, a3;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        nonfinite |= !(fabsf(v[j]) <= 3.402823466e38f);
        __bf16 h1, h2, h3;
        split3(v[j], h1, h2, h3);
        a1[j] = h1;
        a2[j] = h2;
        a3[j] = h3;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bw[0][s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bw[1][s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[2][s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bw[0][s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[1][s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[0][s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t R = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;   // C/D map of the 32x32 MFMA
      if (R < n) Xp[R * ldp + c] = acc[r];
    }
  }
  if (stats && nonfinite) atomicOr(&stats[FSW_STAT_FLAGS], FSW_FLAG_X_NONFINITE);
}

static size_t bf3_lds_bytes(int nwaves, int ks, int cbufs) {
  return sizeof(__bf16) * 2 * 3 * BS_ROWS * (16 * ks + 8) + sizeof(float) * cbufs * BS_ROWS * (nwaves * 32 + 4) + sizeof(int) * 3 * BS_ROWS;
}

}  // namespace fsw

using namespace fsw;

static int project_launch(const float* X, int64_t n, int d, int64_t ldx, const float* V, int S, int64_t ldv, float* Xp,
                          int64_t ldp, float* x_copy, int64_t ld_copy, const float* W2, int H2, int64_t ldw2, const float* b2,
                          float* Y2, int64_t ldy2, const int32_t* row_map, int32_t* stats, hipStream_t stream) {
  FSW_REQUIRE(X && V && Xp, "fsw_project: null pointer");
  FSW_REQUIRE(!x_copy || ld_copy >= d, "fsw_project: ld_copy must be >= d");
  FSW_REQUIRE(n >= 1 && d >= 1 && S >= 1 && ldx >= d && ldv >= d && ldp >= S, "fsw_project: bad sizes n=%lld d=%d S=%d",
              (long long)n, d, S);
  FSW_REQUIRE(H2 == 0 || (W2 && Y2 && ldw2 >= d && ldy2 >= H2), "fsw_project: bad second output block");
  const bool vec = (ldx % 4 == 0) && (ldv % 4 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)V % 16 == 0) &&
                   (H2 == 0 || ((ldw2 % 4 == 0) && ((uintptr_t)W2 % 16 == 0)));
  if (vec && d % 4 == 0 && d <= 256) {
    // B-stationary kernels: one wave per 32-column slab, persistent over 32-row tiles (d <= 128: <= 12 slab waves so that
    // the B-operand registers fit without spilling, d <= 256: 4; >= 8 waves so that every X tile is two / four 16-byte
    // loads per thread -- waves without a slab only help moving X)
    const bool exact = d <= 128 && getenv("FSW_PROJECT_EXACT_FP32") && atoi(getenv("FSW_PROJECT_EXACT_FP32")) != 0;
    const int nsl1 = (int)ceil_div(S, 32), nsl2 = (int)ceil_div(H2, 32);
#ifndef FSW_PROJECT_NARROW_MAX
#define FSW_PROJECT_NARROW_MAX 64   // widest output (columns) of the wavefront-per-tile kernel; 0: never
#endif
    if (!exact && S <= FSW_PROJECT_NARROW_MAX && H2 == 0 && !x_copy && d % 16 == 0 && d <= 128 && ldp >= (int64_t)nsl1 * 32) {
      const int64_t ntiles = ceil_div(n, BS_ROWS);
      dim3 grid((unsigned)std::min<int64_t>(ceil_div(ntiles, 4), 512), (unsigned)nsl1);   // two workgroups of four wavefronts per CU
      if (d <= 32) k_project_narrow<2><<<grid, 256, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, ntiles);
      else if (d <= 64) k_project_narrow<4><<<grid, 256, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, ntiles);
      else k_project_narrow<8><<<grid, 256, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, ntiles);
      FSW_LAUNCH_CHECK();
      return 0;
    }
    const int nslabs = exact ? (int)ceil_div(S + H2, 32) : nsl1 + nsl2;
    // FSW_PROJECT_GROUPS=2 (experiment): d <= 128 with the slabs in two column groups of <= 6 waves, single C staging buffer, two
    // workgroups per CU.  Measured SLOWER at config 3 (0.99 against 0.67 ms: X is read twice and every tile pays a second barrier)
    static const int split_groups = [] { const char* e = getenv("FSW_PROJECT_GROUPS"); return e ? atoi(e) : 1; }();
    const bool split = split_groups == 2 && d <= 128 && !exact;
    const int max_slab_waves = d <= 128 ? (split ? 6 : 12) : 4;
    const int ngroups = (int)ceil_div(nslabs, max_slab_waves);
    const int nwaves = (int)ceil_div(nslabs, ngroups);
    const int64_t ntiles = ceil_div(n, BS_ROWS);
    dim3 grid((unsigned)std::min<int64_t>(ntiles, 256), (unsigned)ngroups);
    const int threads = std::max(nwaves, split ? 6 : 8) * 64;
    const int cbufs = split ? 1 : 2;
    if (exact) {
      // FSW_PROJECT_EXACT_FP32=1: exact-fp32 MFMA kernel instead of bf16x3 (same accuracy class, 2.7x the matrix cycles)
#define FSW_LAUNCH_BS(KQ)                                                                                                   \
  k_project_bs<KQ><<<grid, threads, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, x_copy, ld_copy, W2, H2, ldw2, b2, Y2, \
                                                 ldy2, row_map, ntiles, nwaves)
      if (d <= 32) FSW_LAUNCH_BS(16);
      else if (d <= 64) FSW_LAUNCH_BS(32);
      else FSW_LAUNCH_BS(64);
#undef FSW_LAUNCH_BS
    } else {
      FSW_REQUIRE(ldp >= (int64_t)nsl1 * 32 && ldp % 4 == 0 && (uintptr_t)Xp % 16 == 0,
                  "fsw_project: Xp must be 16-byte aligned with ldp >= 32*ceil(S/32), ldp %% 4 == 0");
      const int y2_vec = (H2 > 0 && ldy2 % 4 == 0 && (uintptr_t)Y2 % 16 == 0) ? 1 : 0;
      const int ks = d <= 32 ? 2 : d <= 64 ? 4 : d <= 128 ? 8 : 16;
      const size_t lds = bf3_lds_bytes(nwaves, ks, cbufs);
#define FSW_LAUNCH_BF3(KS)                                                                                                  \
  do {                                                                                                                      \
    FSW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_project_bf3<KS>),                                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
    k_project_bf3<KS><<<grid, threads, lds, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, x_copy, ld_copy, W2, H2, ldw2, \
                                                      b2, Y2, ldy2, row_map, ntiles, nwaves, nsl1, y2_vec, cbufs);          \
  } while (0)
      if (ks == 2) FSW_LAUNCH_BF3(2);
      else if (ks == 4) FSW_LAUNCH_BF3(4);
      else if (ks == 8) FSW_LAUNCH_BF3(8);
      else FSW_LAUNCH_BF3(16);
#undef FSW_LAUNCH_BF3
    }
    FSW_LAUNCH_CHECK();
    return 0;
  }
  const int nct = (int)ceil_div(S + H2, PBN);
  const int64_t nblocks = ceil_div(n, PBM) * nct;
  FSW_REQUIRE(nblocks < (1ll << 31), "fsw_project: grid too large");
  if (vec)
    k_project<true><<<(unsigned)nblocks, 256, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, nct, x_copy, ld_copy, W2, H2,
                                                           ldw2, b2, Y2, ldy2, row_map);
  else
    k_project<false><<<(unsigned)nblocks, 256, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, nct, x_copy, ld_copy, W2, H2,
                                                            ldw2, b2, Y2, ldy2, row_map);
  FSW_LAUNCH_CHECK();
  return 0;
}

extern "C" int fsw_project_f32(const float* X, int64_t n, int d, int64_t ldx, const float* V, int S, int64_t ldv,
                               float* Xp, int64_t ldp, float* x_copy, int64_t ld_copy, int32_t* stats, fsw_stream_t stream) {
  return project_launch(X, n, d, ldx, V, S, ldv, Xp, ldp, x_copy, ld_copy, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, stats,
                        reinterpret_cast<hipStream_t>(stream));
}

extern "C" int fsw_project_linear_f32(const float* X, int64_t n, int d, int64_t ldx, const float* V, int S, int64_t ldv,
                                      float* Xp, int64_t ldp, const float* W2, int H2, int64_t ldw2, const float* b2, float* Y2,
                                      int64_t ldy2, const int32_t* row_map, int32_t* stats, fsw_stream_t stream) {
  return project_launch(X, n, d, ldx, V, S, ldv, Xp, ldp, nullptr, 0, W2, H2, ldw2, b2, Y2, ldy2, row_map, stats,
                        reinterpret_cast<hipStream_t>(stream));
}

#if FSW_PROJECT_STAMPS
// timing experiment only (not in include/fsw_hip.h): copies and clears the per-wave stamp sums of k_project_bf3
// (out: [1024][12][8] unsigned 64-bit words)
extern "C" int fsw_debug_project_stamps(unsigned long long* out) {
  FSW_CHECK_HIP(hipDeviceSynchronize());
  FSW_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fsw::g_proj_stamps), sizeof(unsigned long long) * 1024 * 12 * 8));
  void* p = nullptr;
  FSW_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(fsw::g_proj_stamps)));
  FSW_CHECK_HIP(hipMemset(p, 0, sizeof(unsigned long long) * 1024 * 12 * 8));
  return 0;
}
#endif
