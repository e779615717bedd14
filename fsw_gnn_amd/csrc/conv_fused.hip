// FSW_conv fast path: neighbourhood embedding fused with the first Linear layer of the MLP.  gfx950.
//
//   Y[i, :] = act( [ mw * E(N(i)) , x_i ] . W^T + b )            (reference fsw_conv.py:355-362)
//           = act( mw * E(N(i)) . W1^T  +  ( x_i . W2^T + b ) )   with W = [W1 | W2]
//
// for unit edge weights, tau <= 1 and in-degrees 0 .. FSW_REG_MAX_DEG.  The reference (and the unfused path of
// this library) writes the embedding E to HBM (n x embed_dim floats), copies x next to it (torch.cat), reads
// both back in a GEMM and makes one more pass for the activation.  Here
//   * the x . W2^T + b half is folded into the projection GEMM (project.hip: extra output block), which already
//     has every X tile in LDS, and lands in Y;
//   * a workgroup keeps the 32 embedding rows it has just produced in LDS and multiplies the 32 x embed_dim tile
//     by W1^T on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32) while other workgroups of the CU
//     are in their memory-bound gather phase -- the MFMA pipe is otherwise idle in this kernel; the epilogue adds
//     the row of Y, applies the activation and stores it back.
// Per 32 rows:
//   phase 1  (= embed_reg.hip: k_embed_reg_unit) lane = slice, wave = 64-slice chunk: D coalesced 256-B gathers
//            of Xp[col, k0..k0+63], exact-size min/max sorting network, D FMAs with the float64-evaluated
//            coefficient table  ->  H[row][mass | slices] in LDS (odd row stride: conflict-free MFMA operand reads)
//   phase 2  wave w owns output columns 32w..32w+31: A from LDS; B = W1^T packed on the host so that ONE 16-byte
//            load per lane feeds four MFMAs (group of 8 k: [k, k+2, k+4, k+6 | k+1, k+3, k+5, k+7] per column),
//            eight groups in flight per wave (the loads queue behind the CU's HBM gathers)
//   phase 3  + Y row (x . W2^T + b), activation, rows scattered back to Y by node id
// HBM traffic per forward drops by ~3.6 GB at BASELINE config 3 (no E write, no E/x re-read, no concat, no
// activation pass).
#include <stdlib.h>
#include <algorithm>
#include "fsw_common.h"
#include "sortnet.h"
#include "row_pipeline.h"
#ifndef FSW_FUSED_PIPE_BARRIER
#define FSW_FUSED_PIPE_BARRIER 0   // measured: tools/exp_variants.sh
#endif

#ifndef FSW_FUSED_STAMPS
#define FSW_FUSED_STAMPS 0   // 1: s_memtime stamps at the phase boundaries of k_conv_fused_unit, one record per workgroup and wavefront
#endif                       //    (tools/exp_fused_stamps.py reads them through fsw_debug_fused_stamps)

namespace fsw {

#if FSW_FUSED_STAMPS
constexpr int kFusedStampWgs = 40000;
__device__ unsigned long long g_fused_stamps[kFusedStampWgs][4][8];
#define FSW_FSTAMP(i)                                                                      \
  do {                                                                                     \
    const unsigned long long now_ = clock64();                                             \
    if (lane_id() == 0 && blockIdx.x < kFusedStampWgs) g_fused_stamps[blockIdx.x][wave_id()][i] = now_ - fst_last; \
    fst_last = now_;                                                                       \
  } while (0)
#else
#define FSW_FSTAMP(i) do { } while (0)
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kFusedRows = 32;
constexpr int kLdT = 132;   // LDS row stride of the staged output tile (Hout <= 128)
#ifndef FSW_FUSED_PREFETCH
#define FSW_FUSED_PREFETCH 8
#endif
#ifndef FSW_FUSED_NARROW_WAVES
#define FSW_FUSED_NARROW_WAVES 4   // waves per SIMD the narrow-block variant (RPW = 2) is compiled for; measured: tools/exp_r3_narrow_occupancy.sh
#endif

struct FusedArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* perm;
  const int32_t* bin_start;
  const float* Xp;
  int64_t ldp;
  int S;
  const float* table;
  int64_t ldt;
  const float* bias;  // embedding bias [has_mass + S] or null
  float out_scale;    // message_weight_vs_self
  int has_mass, mass_fn;
  float mass_scale;
  const float* Wq;  // packed W1^T: [Kp/8][ldw][8], zero padded (Kp = K rounded up to 8, ldw = Hout rounded up to 32)
  int64_t ldw;
  const float* lin_bias;  // [Hout] or null; used only when y_accumulate == 0
  const float* Yin;       // [n][ldyin] x . W2^T + b or null: in perm order (written by the projection kernel), or by node id when
  int64_t ldyin;          // yin_by_node (the block comes from a BLAS GEMM: no row permutation needed, whole 4 Hout-byte rows are read)
  int yin_by_node;
  int Hout;
  int act;  // 0 none, 1 relu, 2 leaky relu
  float slope;
  float* Y;
  int64_t ldy;
  int ldh;  // LDS row stride of H (odd)
  int tile_floats;  // LDS floats reserved for H / the staged output tile
  int Kp;
};

__device__ __forceinline__ float mass_encode_f(float m, int fn) {
  if (fn == 1) return 2.f * (m / (sqrtf(m + 1.f) + 1.f));
  if (fn == 2) return log1pf(m);
  return m;
}

// phase 1 for one (degree, 64-slice chunk): embedding values of the block's rows into LDS (row_pipeline.h)
// RPW = rows per wavefront.  1: lane = slice of one row (col indices wave-uniform, scalar loads).  2: a slice block of at most
// 32 slices (one rank's share of a slice-sharded layer at 8 ranks x 256 slices, dist.py) -- lanes 0..31 and 32..63 work on two
// DIFFERENT rows of the same degree, so every lane gathers: the col indices become per-lane vector loads (two distinct
// addresses per instruction), a gather instruction reads two 128-byte runs, the network and the FMAs are unchanged.
template <int D, int RPW>
__device__ __forceinline__ void fused_embed_rows(const FusedArgs& a, int p, int nrows, float* __restrict__ H, int chunk) {
  constexpr int LPR = kWave / RPW;   // lanes per row
  const int lane = lane_id();
  const int sub = RPW == 1 ? 0 : lane / LPR;
  const int k = chunk * kWave + (RPW == 1 ? lane : lane % LPR);
  const bool kvalid = k < a.S;
  const int kc = kvalid ? k : a.S - 1;
  const float b = a.bias ? a.out_scale * a.bias[a.has_mass + kc] : 0.f;
  if constexpr (D == 0) {
    for (int r = sub; r < nrows; r += RPW)
      if (kvalid) H[r * a.ldh + a.has_mass + k] = b;
  } else {
    float coef[D];
    const float* tab = a.table + (int64_t)(D * (D - 1) / 2) * a.ldt + kc;
#pragma unroll
    for (int t = 0; t < D; ++t) coef[t] = a.out_scale * tab[(int64_t)t * a.ldt];
    const int startv = a.rowptr[a.perm[p + min(lane, nrows - 1)]];   // lane r: CSR offset of the block's row r
    float* hk = H + a.has_mass + kc;   // lanes past the last slice recompute slice S-1 (row_pipeline.h)
    const int ldh = a.ldh;
    const int nsteps = (nrows + RPW - 1) / RPW;
    // step r works on row r (RPW = 1) or on rows 2r and 2r + 1 (clamped to the last row: recomputed, same value stored again)
    pipelined_rows<D, pipeline_depth<D>(), FSW_FUSED_PIPE_BARRIER>(
        nsteps, a.col, a.Xp + kc, a.ldp,
        [&](int r) {
          if constexpr (RPW == 1) return __builtin_amdgcn_readlane(startv, r);
          else return __shfl(startv, min(r * RPW + sub, nrows - 1));
        },
        [&](KeyNet<D>& net, int r) {
          sort_network<D>(net);
          float acc = b;
#pragma unroll
          for (int t = 0; t < D; ++t) acc = fmaf(coef[t], net.k[t], acc);
          const int row = RPW == 1 ? r : min(r * RPW + sub, nrows - 1);
          hk[row * ldh] = acc;
        });
  }
}

#define FSW_CASES_0_32(X)                                                                                                   \
  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) \
  X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

// phase 2 for one 32-column slab: acc = H[32 x Kp] . W1^T[Kp x 32] on the fp32 matrix cores
template <int ABL>
__device__ __forceinline__ void slab_mma(const FusedArgs& a, const float* __restrict__ H, int slab, int fr, int fh, f32x16& acc) {
  constexpr int kPrefetch = FSW_FUSED_PREFETCH;   // 16-byte W loads in flight per wave (they queue behind the CU's HBM gathers)
  const int ngroups = a.Kp >> 3;
  const int j = min(slab * 32 + fr, (int)a.ldw - 1);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* hp = H + fr * a.ldh + fh;
  // packed W1^T: group g of this lane at wq[g * gstride]; the host pads 16 zero groups past the end
  const float4* wq = reinterpret_cast<const float4*>(a.Wq) + ((int64_t)j * 2 + fh);
  const int64_t gstride = a.ldw * 2;
  auto mma4 = [&](const float* h, const float4& bq) {
    if constexpr (ABL & 1) { asm volatile("" ::"v"(bq.x), "v"(bq.y), "v"(bq.z), "v"(bq.w)); return; }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h[0], bq.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h[2], bq.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h[4], bq.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h[6], bq.w, acc, 0, 0, 0);
  };
  float4 bq[kPrefetch];
#pragma unroll
  for (int u = 0; u < kPrefetch; ++u) bq[u] = wq[u * gstride];
  wq += kPrefetch * gstride;
  int g0 = 0;
  for (; g0 + kPrefetch <= ngroups; g0 += kPrefetch) {
#pragma unroll
    for (int u = 0; u < kPrefetch; ++u) {
      mma4(hp + 8 * (g0 + u), bq[u]);
      bq[u] = wq[u * gstride];
    }
    wq += kPrefetch * gstride;
  }
#pragma unroll
  for (int u = 0; u < kPrefetch; ++u)
    if (g0 + u < ngroups) mma4(hp + 8 * (g0 + u), bq[u]);
}

// TR = rows of a workgroup's tile.  32: the output staging tile reuses H's LDS.  128 (narrow slice blocks, Hout <= 128): four
// 32-row sub-tiles go through the matrix phase one after the other with a staging tile of their own -- a workgroup's start-up chain
// (tile search, perm -> rowptr -> col -> first gather: three dependent round trips, ~5 us) is paid once per 128 rows instead of once
// per 32, which at 32 slices per row is a quarter of a tile's whole life.
template <int DLO, int DHI, int WAVES_PER_SIMD, int ABL = 0, int RPW = 1, int TR = 32>  // ABL: timing experiments only (tools/exp_fused.py)
__global__ void __launch_bounds__(256, WAVES_PER_SIMD) k_conv_fused_unit(const FusedArgs a) {
  static_assert(TR == kFusedRows || (TR == 128 && RPW == 2), "tile rows");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* H = smem;                                                        // [TR][ldh]
  int* nodeS = reinterpret_cast<int*>(smem + a.tile_floats);              // [TR]

#if FSW_FUSED_STAMPS
  unsigned long long fst_last = clock64();
#endif
  // workgroup -> (degree bin, perm range): one degree per workgroup, highest degrees first, bin 0 last
  int D, p = 0, pe = 0;
  if (!find_degree_tile<TR>(a.bin_start, DLO, DHI, (int)blockIdx.x, D, p, pe)) return;
  const int nrows = pe - p;
  const int wv = wave_id();
  const int lane = lane_id();
  const int K = a.has_mass + a.S;  // embedding width = K of the fused product
  const int fr = lane & 31, fh = lane >> 5;

  // zero the K padding column(s) and the unused rows of H, record node ids, mass column
  if (threadIdx.x < TR) {
    const int r = threadIdx.x;
    for (int c = (r < nrows ? K : 0); c < a.ldh; ++c) H[r * a.ldh + c] = 0.f;
    nodeS[r] = r < nrows ? a.perm[p + r] : -1;
    if (a.has_mass && r < nrows)
      H[r * a.ldh] = a.out_scale * (mass_encode_f((float)D, a.mass_fn) * a.mass_scale + (a.bias ? a.bias[0] : 0.f));
  }
  FSW_FSTAMP(0);                                           // tile found, H padding / node ids / mass column
  // phase 1: embedding rows
  // a wave = one 64-slice chunk of a group of rows.  Four or more chunks: every wave walks all 32 rows of its chunks.
  // Narrow slice blocks (one rank's share of a slice-sharded layer, dist.py): the rows are split into 4 / nchunks
  // groups so that all four waves still gather.
  const int nchunks = (a.S + kWave - 1) / kWave;
  const int ngroups = nchunks >= 3 ? 1 : 4 / nchunks;
  const int grows = TR / ngroups;
  for (int item = wv; item < nchunks * ngroups; item += 4) {
    const int chunk = item / ngroups, r0 = (item - chunk * ngroups) * grows;
    const int gn = min(nrows - r0, grows);
    if (gn <= 0) continue;
    switch (D) {
#define X(d)                                                  \
  case d:                                                     \
    if constexpr (d >= DLO && d <= DHI) fused_embed_rows<d, RPW>(a, p + r0, gn, H + r0 * a.ldh, chunk); \
    break;
      FSW_CASES_0_32(X)
#undef X
      default:
        break;
    }
  }
  FSW_FSTAMP(1);                                           // phase 1: this wavefront's rows gathered, sorted, read out into H
  const int nslabs = (a.Hout + 31) / 32;
  if (nslabs <= 4) {
    // ---- Hout <= 128: one slab per wave, output tile staged through LDS so that Y is written as whole rows ----
    // Rows of Yin (= x . W2^T + b, stored by the projection kernel in perm order: this workgroup's 32 rows are one
    // contiguous run).  Wave w finishes rows 8w..8w+7; lane owns columns lane and lane+64.  Issued after phase 1
    // (registers are free again) and before the barrier: in flight while the other waves finish their rows.
    // Branch-free (rows past the tile's end re-read its last row, columns past Hout the last column; neither is used): with a
    // branch per load every load sat in its own basic block and waited for the one before it (s_waitcnt vmcnt(0) per block) --
    // 10 us of a workgroup's 91 (tools/exp_fused_stamps.py)
    float lb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) lb[h] = (!a.Yin && a.lin_bias && lane + 64 * h < a.Hout) ? a.lin_bias[lane + 64 * h] : 0.f;
    // TR == 32: the staging tile reuses H; TR == 128: its own LDS behind the node ids (four sub-tiles read H one after the other)
    float* T = TR == kFusedRows ? smem : reinterpret_cast<float*>(nodeS + TR);   // [32][kLdT]
#pragma unroll 1
    for (int r0 = 0; r0 < TR; r0 += kFusedRows) {
      if (TR > kFusedRows && r0 >= nrows) break;           // uniform
      const int nsub = min(nrows - r0, kFusedRows);
      float yin[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) yin[q] = 0.f;
      if ((ABL & 2) == 0 && a.Yin) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = r0 + min(wv * 8 + (q >> 1), nsub - 1), c = min(lane + 64 * (q & 1), a.Hout - 1);
          yin[q] = a.Yin[(int64_t)(a.yin_by_node ? a.perm[p + row] : p + row) * a.ldyin + c];
        }
      }
      FSW_FSTAMP(2);                                       // Yin loads issued
      __syncthreads();                       // first sub-tile: phase 1 complete; later ones: the previous epilogue has read T
      FSW_FSTAMP(3);                                       // barrier: waited for the slowest wavefront's phase 1
      f32x16 acc;
      if (wv < nslabs) slab_mma<ABL>(a, H + r0 * a.ldh, wv, fr, fh, acc);
      FSW_FSTAMP(4);                                       // matrix phase
      if (TR == kFusedRows) __syncthreads(); // every wave has finished reading H: reuse it for the output tile
      if (wv < nslabs) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * fh) * kLdT + wv * 32 + fr] = acc[r];
      }
      __syncthreads();
      FSW_FSTAMP(5);                                       // barrier, accumulators into the staging tile, barrier
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int row = wv * 8 + rr;
        const int node = row < nsub ? nodeS[r0 + row] : -1;
        if (node < 0) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = lane + 64 * h;
          if (c < a.Hout) {
            float y = T[row * kLdT + c] + lb[h] + yin[rr * 2 + h];
            if (a.act == 1) y = fmaxf(y, 0.f);
            else if (a.act == 2) y = y >= 0.f ? y : a.slope * y;
            if ((ABL & 4) == 0 || y == 12345.f) a.Y[(int64_t)node * a.ldy + c] = y;
          }
        }
      }
    }
    FSW_FSTAMP(6);                                         // epilogue: + Yin, activation, Y rows stored
    return;
  }

  // ---- wide layers (Hout > 128): slabs round-robin over the waves, every wave stores its own 32 x 32 tiles ----
  __syncthreads();
  for (int slab = wv; slab < nslabs; slab += 4) {
    const int j = slab * 32 + fr;
    float yin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;     // C/D map of the 32x32 MFMA
      yin[r] = (a.Yin && row < nrows && j < a.Hout) ? a.Yin[(int64_t)(a.yin_by_node ? a.perm[p + row] : p + row) * a.ldyin + j] : 0.f;
    }
    f32x16 acc;
    slab_mma<ABL>(a, H, slab, fr, fh, acc);
    if (j < a.Hout) {
      const float lb = (!a.Yin && a.lin_bias) ? a.lin_bias[j] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int node = nodeS[(r & 3) + 8 * (r >> 2) + 4 * fh];
        if (node >= 0) {
          float y = acc[r] + lb + yin[r];
          if (a.act == 1) y = fmaxf(y, 0.f);
          else if (a.act == 2) y = y >= 0.f ? y : a.slope * y;
          a.Y[(int64_t)node * a.ldy + j] = y;
        }
      }
    }
  }
}

// W = [W1 | W2] of the first Linear layer -> the packed operand of slab_mma (layout: include/fsw_hip.h) and a
// contiguous copy of W2.  One launch per forward instead of a host-side cache that an in-place weight edit could outdate.
__global__ void __launch_bounds__(256) k_pack_linear(const float* __restrict__ W, int64_t ldw_in, int Hout, int col0, int K,
                                                     float* __restrict__ Wq, int64_t ldwq, int64_t ngroups,
                                                     const float* __restrict__ W2, int d2, float* __restrict__ W2out, int64_t ldw2out) {
  const int64_t nq = ngroups * ldwq * 8;
  const int64_t total = nq + (W2out ? (int64_t)Hout * d2 : 0);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < nq) {
      const int e = (int)(idx & 7), h = e >> 2, i = e & 3;
      const int64_t gj = idx >> 3;
      const int64_t g = gj / ldwq, j = gj - g * ldwq;
      const int64_t k = 8 * g + 2 * i + h;
      Wq[idx] = (j < Hout && k < K) ? W[j * ldw_in + col0 + k] : 0.f;
    } else {
      const int64_t t = idx - nq;
      const int64_t j = t / d2, c = t - j * d2;
      W2out[j * ldw2out + c] = W2[j * ldw_in + c];
    }
  }
}

// R[r, c] = act(R[r, c] + Yin[r, c] + bias[c]) in ONE launch: the epilogue of the slice-sharded layer forms (dist.py) on the rows a
// rank owns after the reduce-scatter / all-to-all -- as separate torch ops (addmm_, add_, leaky_relu_) these 64 MB passes were four
// launch-bound kernels per node-range chunk.
__global__ void __launch_bounds__(256) k_add_bias_act(float* __restrict__ R, int64_t ldr, const float* __restrict__ Yin, int64_t ldyin,
                                                      const float* __restrict__ bias, int64_t rows, int H, int act, float slope) {
  const int hv = (H + 3) >> 2;   // float4 columns (callers guarantee 16-byte aligned rows when vec)
  const int64_t total = rows * hv;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / hv;
    const int c = (int)(idx - r * hv) * 4;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c + u < H) {
        float y = R[r * ldr + c + u];
        if (Yin) y += Yin[r * ldyin + c + u];
        if (bias) y += bias[c + u];
        if (act == 1) y = fmaxf(y, 0.f);
        else if (act == 2) y = y >= 0.f ? y : slope * y;
        v[u] = y;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c + u < H) R[r * ldr + c + u] = v[u];
  }
}

}  // namespace fsw

using namespace fsw;

extern "C" int fsw_add_bias_act_f32(float* R, int64_t ldr, const float* Yin, int64_t ldyin, const float* bias, int64_t rows, int H,
                                    int act, float slope, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (rows == 0) return 0;
  FSW_REQUIRE(R && rows > 0 && H >= 1 && ldr >= H && (!Yin || ldyin >= H), "fsw_add_bias_act_f32: bad arguments");
  FSW_REQUIRE(act >= 0 && act <= 2, "fsw_add_bias_act_f32: act must be 0 (none), 1 (relu) or 2 (leaky relu)");
  const int64_t total = rows * ((H + 3) / 4);
  k_add_bias_act<<<(unsigned)std::min<int64_t>(ceil_div(total, 256), 4096), 256, 0, stream>>>(R, ldr, Yin, ldyin, bias, rows, H, act, slope);
  FSW_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t fsw_packed_linear_floats(int K, int Hout) {
  return (size_t)(((K + 7) / 8) + 16) * (size_t)(((Hout + 31) / 32) * 32) * 8;
}

extern "C" int fsw_pack_linear_f32(const float* W, int64_t ldw_in, int Hout, int col0, int K, float* Wq, const float* W2, int d2,
                                   float* W2out, int64_t ldw2out, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(W && Wq && Hout >= 1 && K >= 1 && col0 >= 0 && ldw_in >= col0 + K, "fsw_pack_linear_f32: bad arguments");
  FSW_REQUIRE(((uintptr_t)Wq & 15) == 0, "fsw_pack_linear_f32: Wq must be 16-byte aligned");
  FSW_REQUIRE(!W2out || (W2 && d2 >= 1 && ldw2out >= d2), "fsw_pack_linear_f32: bad W2 block");
  const int64_t ldwq = ((Hout + 31) / 32) * 32, ngroups = (K + 7) / 8 + 16;
  const int64_t total = ngroups * ldwq * 8 + (W2out ? (int64_t)Hout * d2 : 0);
  k_pack_linear<<<(unsigned)std::min<int64_t>(ceil_div(total, 256), 2048), 256, 0, stream>>>(W, ldw_in, Hout, col0, K, Wq, ldwq, ngroups, W2,
                                                                                            d2, W2out, ldw2out);
  FSW_LAUNCH_CHECK();
  return 0;
}

#ifndef FSW_FUSED_NARROW_ROWS
#define FSW_FUSED_NARROW_ROWS 128   // tile rows of the narrow-block variant (32: the common tile)
#endif
static bool fused_wide_tile(int S, int Hout) { return FSW_FUSED_NARROW_ROWS == 128 && S <= kWave / 2 && Hout <= 128; }

extern "C" size_t fsw_conv_fused_lds_bytes(int S, int has_mass) {
  const int Kp = (has_mass + S + 7) & ~7;
  const int ldh = Kp | 1;
  if (fused_wide_tile(S, 128))   // H [128][ldh] | node ids [128] | staging tile [32][kLdT]
    return (size_t)128 * ldh * sizeof(float) + 128 * sizeof(int) + (size_t)kFusedRows * kLdT * sizeof(float);
  return (size_t)kFusedRows * (ldh > kLdT ? ldh : kLdT) * sizeof(float) + kFusedRows * sizeof(int);
}

extern "C" int fsw_conv_fused_f32(const fsw_embed_args* args, const float* Wq, int64_t ldw, const float* lin_bias, int Hout,
                                  const float* Yin, int64_t ldyin, int yin_by_node, int act, float slope, float* Y, int64_t ldy,
                                  fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(args && Wq && Y, "fsw_conv_fused_f32: null pointer");
  const fsw_embed_args& e = *args;
  FSW_REQUIRE(e.rowptr && e.col && e.perm && e.bin_start && e.Xp && e.unit_table, "fsw_conv_fused_f32: null pointer in args");
  FSW_REQUIRE(e.w == nullptr && e.tau <= 1.f, "fsw_conv_fused_f32: unit weights with tau <= 1 only");
  // rows above FSW_REG_MAX_DEG are not visited: the caller finishes them with fsw_embed_f32 + its own GEMM (fsw_conv.py)
  FSW_REQUIRE(e.S >= 1 && e.ldp >= e.S && e.ldt >= e.S && (e.has_mass == 0 || e.has_mass == 1), "fsw_conv_fused_f32: bad sizes");
  FSW_REQUIRE(!Yin || ldyin >= Hout, "fsw_conv_fused_f32: bad Yin stride");
  FSW_REQUIRE(Hout >= 1 && ldy >= Hout && ldw >= ((Hout + 31) / 32) * 32 && ldw % 32 == 0, "fsw_conv_fused_f32: bad output sizes");
  FSW_REQUIRE(((uintptr_t)Wq & 15) == 0, "fsw_conv_fused_f32: Wq must be 16-byte aligned");
  FSW_REQUIRE(act >= 0 && act <= 2, "fsw_conv_fused_f32: act must be 0 (none), 1 (relu) or 2 (leaky relu)");
  const bool wide_tile = fused_wide_tile(e.S, Hout);
  const size_t lds = wide_tile ? fsw_conv_fused_lds_bytes(e.S, e.has_mass)
                               : (size_t)kFusedRows * ((((e.has_mass + e.S + 7) & ~7) | 1) > kLdT ? (((e.has_mass + e.S + 7) & ~7) | 1) : kLdT) * sizeof(float) + kFusedRows * sizeof(int);
  FSW_REQUIRE(lds <= 64 * 1024, "fsw_conv_fused_f32: embed_dim too wide for the fused tile (%zu B of LDS)", lds);
  FusedArgs a;
  a.rowptr = e.rowptr; a.col = e.col; a.perm = e.perm; a.bin_start = e.bin_start;
  a.Xp = e.Xp; a.ldp = e.ldp; a.S = e.S; a.table = e.unit_table; a.ldt = e.ldt;
  a.bias = e.bias; a.out_scale = e.out_scale; a.has_mass = e.has_mass; a.mass_fn = e.mass_fn; a.mass_scale = e.mass_scale;
  a.Wq = Wq; a.ldw = ldw; a.lin_bias = lin_bias; a.Yin = Yin; a.ldyin = ldyin; a.yin_by_node = yin_by_node ? 1 : 0; a.Hout = Hout; a.act = act; a.slope = slope;
  a.Y = Y; a.ldy = ldy;
  a.Kp = (e.has_mass + e.S + 7) & ~7;
  a.ldh = a.Kp | 1;
  a.tile_floats = wide_tile ? 128 * a.ldh : kFusedRows * (a.ldh > kLdT ? a.ldh : kLdT);
  // two launches: long rows first (more registers per wave), then degrees 0..16 at higher occupancy
  const int64_t nblocks = ceil_div(e.num_rows, kFusedRows) + FSW_REG_MAX_DEG + 1;
#ifdef FSW_ABLATION
  const char* abl_env = getenv("FSW_FUSED_ABL");
  const int abl = abl_env ? atoi(abl_env) : 0;
#define FSW_ABL_CASE(v)                                                                   \
  if (abl == v) {                                                                         \
    k_conv_fused_unit<0, FSW_REG_MAX_DEG, 4, v><<<(unsigned)nblocks, 256, lds, stream>>>(a); \
    FSW_LAUNCH_CHECK();                                                                   \
    return 0;                                                                             \
  }
  FSW_ABL_CASE(1) FSW_ABL_CASE(2) FSW_ABL_CASE(4) FSW_ABL_CASE(6) FSW_ABL_CASE(7)
#endif
  if (wide_tile) {        // a narrow slice block, Hout <= 128: two rows per wavefront, 128-row tiles
    const int64_t nb128 = ceil_div(e.num_rows, 128) + FSW_REG_MAX_DEG + 1;
    k_conv_fused_unit<0, FSW_REG_MAX_DEG, FSW_FUSED_NARROW_WAVES, 0, 2, 128><<<(unsigned)nb128, 256, lds, stream>>>(a);
  } else if (e.S <= kWave / 2)   // a narrow slice block: two rows per wavefront (fused_embed_rows)
    k_conv_fused_unit<0, FSW_REG_MAX_DEG, FSW_FUSED_NARROW_WAVES, 0, 2><<<(unsigned)nblocks, 256, lds, stream>>>(a);
  else
    k_conv_fused_unit<0, FSW_REG_MAX_DEG, 4><<<(unsigned)nblocks, 256, lds, stream>>>(a);
  FSW_LAUNCH_CHECK();
  return 0;
}

#if FSW_FUSED_STAMPS
// timing experiment only (not in include/fsw_hip.h): copies the per-workgroup stamp records of the last k_conv_fused_unit launch
extern "C" int fsw_debug_fused_stamps(unsigned long long* out, int max_wgs) {
  FSW_CHECK_HIP(hipDeviceSynchronize());
  const int nw = max_wgs < fsw::kFusedStampWgs ? max_wgs : fsw::kFusedStampWgs;
  FSW_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fsw::g_fused_stamps), sizeof(unsigned long long) * (size_t)nw * 4 * 8));
  return 0;
}
#endif
