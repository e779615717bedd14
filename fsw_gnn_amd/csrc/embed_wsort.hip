// Fused neighbourhood kernel for long rows, wave-sort path (FSW_MID_MAX_DEG < in-degree <= FSW_LDS_MAX_DEG, and the
// weighted rows above FSW_MID_MAX_DEG_WEIGHTED).  gfx950.
//
// A workgroup takes one recipient row and walks its slices in groups of SC:
//   1. gather    Xp[col_t, k0 .. k0+SC-1] for every neighbour t (runs of SC floats along the slice axis) stored
//                transposed in LDS, one line per slice;
//   2. sort      ONE WAVEFRONT PER LINE, the line in registers: lane l holds elements l*M .. l*M+M-1.  Every lane sorts
//                its M keys with the register network of sortnet.h, then six merge levels double the sorted run
//                (2, 4, .. 64 lanes): one "flip" exchange with lane ^ (lanes-1) and mirrored registers, log2(lanes)-1
//                exchanges with lane ^ stride, log2(M) in-register half-cleaners.  No LDS traffic, no barrier inside
//                the sort (the LDS bitonic network this replaces spent a barrier and an LDS round trip per stage and
//                delivered 26 GB/s of gather on an RMAT graph, tools/exp_skew.py);
//   3. readout   from the same registers.  Unit weights: lane l owns ranks l*M.., coefficients
//                (1+xi)[sin(2 pi xi (r+1)/D) - sin(2 pi xi r/D)]/(pi xi) (reference fsw_embedding.py:1047-1075, 1109)
//                by a float64 rotation started at the lane's first rank.  General weights: the weight travels with
//                its key, cumulative weight = in-lane prefix + float64 wave scan of the lane sums (the segmented
//                cumsum of fsw_embedding.py:1031-1032), phase in float64, the reference's pad element
//                (fsw_embedding.py:787-821) is element D.
#include <algorithm>
#include "fsw_common.h"
#include "sortnet.h"

namespace fsw {

constexpr double kPiW = 3.14159265358979323846;
constexpr int kWsLdsBytes = 69 * 1024;   // two workgroups per CU
constexpr int kWsSplitY = 4;             // workgroups sharing one row (disjoint slice groups)

__device__ __forceinline__ float mass_encode_w(float m, int fn) {
  if (fn == 1) return 2.f * (m / (sqrtf(m + 1.f) + 1.f));
  if (fn == 2) return log1pf(m);
  return m;
}

__device__ __forceinline__ float sin2pi_rev_w(double x) {
  const double r = x - rint(x);
  return sinpif(2.f * (float)r);
}

__device__ __forceinline__ float wave_sum_w(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ double wave_exclusive_scan_f64(double v) {
  double inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const double t = __shfl_up(inc, off);
    if (lane_id() >= off) inc += t;
  }
  return inc - v;
}

// keys (and weights) of one line across the wave, blocked layout; ascending over element index l*M + j afterwards
template <int M, bool WEIGHTED>
struct WaveLine {
  float k[M];
  float w[WEIGHTED ? M : 1];

  // exchange with another lane: this lane keeps the smaller (lower == true) or the larger key of each pair
  template <int JREV>
  __device__ __forceinline__ void exchange(int mask, bool lower) {
    float ok[M], ow[WEIGHTED ? M : 1];
#pragma unroll
    for (int j = 0; j < M; ++j) {
      ok[j] = __shfl_xor(k[JREV ? M - 1 - j : j], mask);
      if constexpr (WEIGHTED) ow[j] = __shfl_xor(w[JREV ? M - 1 - j : j], mask);
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      if constexpr (WEIGHTED) {
        const bool take = lower ? (ok[j] < k[j]) : (ok[j] > k[j]);   // ties: both lanes keep their own element
        k[j] = take ? ok[j] : k[j];
        w[j] = take ? ow[j] : w[j];
      } else {
        k[j] = lower ? fminf(k[j], ok[j]) : fmaxf(k[j], ok[j]);
      }
    }
  }
  __device__ __forceinline__ void cx(int i, int j) {
    if constexpr (WEIGHTED) {
      const bool sw = k[j] < k[i];
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
    if constexpr (WEIGHTED) {
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
    if constexpr (LANES <= kWave) {
      exchange<1>(LANES - 1, (lane & (LANES >> 1)) == 0);          // element i against i ^ (LANES*M - 1)
      half_cleaners<(LANES >> 2)>(lane);
#pragma unroll
      for (int st = M >> 1; st >= 1; st >>= 1)
#pragma unroll
        for (int j = 0; j < M; ++j)
          if ((j & st) == 0) cx(j, j + st);
      merge_levels<LANES * 2>(lane);
    }
  }
  template <int ST>
  __device__ __forceinline__ void half_cleaners(int lane) {
    if constexpr (ST >= 1) {
      exchange<0>(ST, (lane & ST) == 0);
      half_cleaners<(ST >> 1)>(lane);
    }
  }
};

template <int M, bool WEIGHTED>
__global__ void __launch_bounds__(256) k_embed_wsort(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                     const int32_t* __restrict__ bin_start, int bin_lo, int bin_hi,
                                                     const float* __restrict__ Xp, int64_t ldp, int S,
                                                     const float* __restrict__ freqs, float tau, float* __restrict__ out,
                                                     int64_t ldo, const float* __restrict__ bias, float out_scale, int has_mass,
                                                     int mass_fn, float mass_scale, const float* __restrict__ efeat,
                                                     const float* __restrict__ Ve, int64_t ldve, int d_edge) {
  constexpr int CAP = M * kWave;          // elements a wave can hold
  constexpr int LINE = CAP + kWave + 1;   // LDS floats per line: one pad per M elements, odd stride
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* res = smem;                                    // [64] results of the current slice group
  float* wrow = smem + 64;                              // [LINE] weights of the row, padded like a line (weighted only)
  float* tile = wrow + (WEIGHTED ? LINE : 0);           // [SC][LINE]
  constexpr int kTileLines = (kWsLdsBytes / 4 - 64 - (WEIGHTED ? LINE : 0)) / LINE;
  static_assert(kTileLines >= 1, "a line must fit the LDS budget");
  constexpr int SC = kTileLines >= 64 ? 64 : (kTileLines >= 4 ? (kTileLines & ~3) : kTileLines);   // lines per group (4 waves)
  __shared__ double msum[4];
  const int pbeg = bin_start[bin_lo], pend = bin_start[bin_hi + 1];
  const int lane = lane_id(), wv = wave_id();

  for (int p = pbeg + blockIdx.x; p < pend; p += gridDim.x) {
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    const int Dtot = WEIGHTED ? D + 1 : D;              // the weighted variant always carries the pad element; Dtot <= CAP
    double m = (double)D;
    if constexpr (WEIGHTED) {
      double part = 0.0;
      for (int t = threadIdx.x; t < D; t += blockDim.x) {
        const float wt = w ? w[start + t] : 1.f;
        wrow[t + t / M] = wt;
        part += (double)wt;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
      if (lane == 0) msum[wv] = part;
      __syncthreads();
      m = msum[0] + msum[1] + msum[2] + msum[3];
      if (threadIdx.x == 0) wrow[D + D / M] = (float)fmax((double)tau - m, 0.0);   // pad element: zero weight unless deficient
      __syncthreads();
    }
    const double inv = 1.0 / (WEIGHTED ? fmax(m, (double)tau) : m);

    const int ngroups = (S + SC - 1) / SC;
    for (int g = blockIdx.y; g < ngroups; g += gridDim.y) {
      const int k0 = g * SC;
      // 1. gather + transpose: consecutive threads take consecutive slices of one neighbour
      for (int i = threadIdx.x; i < Dtot * SC; i += blockDim.x) {
        const int kk = i % SC, t = i / SC;
        float key = 0.f;                                 // t == D (weighted): the reference's pad element at x = 0
        if (t < D) {
          const int kcl = min(k0 + kk, S - 1);
          key = Xp[(int64_t)col[start + t] * ldp + kcl];
          if (efeat) {   // edge features: + <e_ij, v_k[d_in:]> (reference fsw_embedding.py:934-968)
            const float* er = efeat + (int64_t)(start + t) * d_edge;
            const float* vr = Ve + (int64_t)kcl * ldve;
            for (int q = 0; q < d_edge; ++q) key = fmaf(er[q], vr[q], key);
          }
        }
        tile[kk * LINE + t + t / M] = key;
      }
      __syncthreads();
      // 2 + 3. one wave per line
      for (int kk = wv; kk < SC; kk += 4) {
        const int k = k0 + kk;
        if (k >= S) break;
        WaveLine<M, WEIGHTED> ln;
        const float* line = tile + kk * LINE + lane * (M + 1);
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const bool valid = lane * M + j < Dtot;
          ln.k[j] = valid ? line[j] : __builtin_inff();
          if constexpr (WEIGHTED) ln.w[j] = valid ? wrow[lane * (M + 1) + j] : 0.f;
        }
        ln.sort();
        const float xif = freqs[k];
        const double xi = (double)xif;
        const bool lin = xif < 1e-30f;                   // xi == 0: Delta_t = 2 w_t
        float acc = 0.f;
        if constexpr (!WEIGHTED) {
          if (lin) {
#pragma unroll
            for (int j = 0; j < M; ++j) acc += (lane * M + j < D) ? ln.k[j] : 0.f;
            acc *= 2.f * (float)inv;
          } else {
            const double step = xi * inv;                // revolutions per rank
            double sd, cd, s, c;
            sincospi(2.0 * (step - rint(step)), &sd, &cd);
            const double x0 = step * (double)(lane * M);
            sincospi(2.0 * (x0 - rint(x0)), &s, &c);
            const double scale = (1.0 + xi) / (kPiW * xi);
#pragma unroll
            for (int j = 0; j < M; ++j) {
              const double sn = fma(s, cd, c * sd), cn = fma(c, cd, -(s * sd));
              acc += (lane * M + j < D) ? (float)(scale * (sn - s)) * ln.k[j] : 0.f;
              s = sn;
              c = cn;
            }
          }
        } else {
          double part = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) part += (double)ln.w[j];
          double c = wave_exclusive_scan_f64(part);
          float sprev = lin ? 0.f : sin2pi_rev_w(xi * (c * inv));
#pragma unroll
          for (int j = 0; j < M; ++j) {
            const bool valid = lane * M + j < Dtot;
            c += (double)ln.w[j];
            if (lin) {
              acc += valid ? ln.w[j] * ln.k[j] : 0.f;
            } else {
              const float s = sin2pi_rev_w(xi * (c * inv));
              acc += valid ? (s - sprev) * ln.k[j] : 0.f;
              sprev = s;
            }
          }
          acc *= lin ? 2.f * (float)inv : (float)((1.0 + xi) / (kPiW * xi));
        }
        acc = wave_sum_w(acc);
        if (lane == 0) res[kk] = out_scale * (acc + (bias ? bias[has_mass + k] : 0.f));
      }
      __syncthreads();
      if ((int)threadIdx.x < SC && k0 + (int)threadIdx.x < S) out[(int64_t)node * ldo + has_mass + k0 + threadIdx.x] = res[threadIdx.x];
      __syncthreads();
    }
    if (has_mass && blockIdx.y == 0 && threadIdx.x == 0)
      out[(int64_t)node * ldo] = out_scale * (mass_encode_w((float)m, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
  }
}

template <int M, bool WEIGHTED>
static int launch_wsort(const fsw_embed_args& a, int bin_lo, int bin_hi, int64_t rows_upper, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    FSW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_embed_wsort<M, WEIGHTED>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kWsLdsBytes));
    attr_set = true;
  }
  dim3 grid((unsigned)std::min<int64_t>(rows_upper, 1 << 14), kWsSplitY);
  k_embed_wsort<M, WEIGHTED><<<grid, 256, kWsLdsBytes, stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, bin_lo, bin_hi, a.Xp, a.ldp,
                                                               a.S, a.freqs, a.tau, a.out, a.ldo, a.bias, a.out_scale, a.has_mass,
                                                               a.mass_fn, a.mass_scale, a.efeat, a.Ve, a.ldve, a.d_edge);
  FSW_LAUNCH_CHECK();
  return 0;
}

// rows_upper bounds the rows in the mid + LDS bins (the per-bin counts stay on the device)
int launch_embed_lds(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  const bool unit_fast = (a.w == nullptr) && (a.tau <= 1.f);
  int rc;
  if (unit_fast) {   // a wave holds 64 M keys
    if ((rc = launch_wsort<8, false>(a, FSW_BIN_LDS0, FSW_BIN_LDS0, rows_upper, stream))) return rc;
    if ((rc = launch_wsort<16, false>(a, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1, rows_upper, stream))) return rc;
    if ((rc = launch_wsort<32, false>(a, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2, rows_upper, stream))) return rc;
  } else {           // D + 1 elements with the pad element: one size up; the mid bins above FSW_MID_MAX_DEG_WEIGHTED come here too
    constexpr int sizes[FSW_NUM_MID_BINS] = FSW_MID_SIZES;
    int bin_lo = FSW_BIN_MID0;
    while (bin_lo < FSW_BIN_LDS0 && sizes[bin_lo - FSW_BIN_MID0] <= FSW_MID_MAX_DEG_WEIGHTED) ++bin_lo;
    if (bin_lo < FSW_BIN_LDS0 && (rc = launch_wsort<8, true>(a, bin_lo, FSW_BIN_LDS0 - 1, rows_upper, stream))) return rc;
    if ((rc = launch_wsort<16, true>(a, FSW_BIN_LDS0, FSW_BIN_LDS0, rows_upper, stream))) return rc;
    if ((rc = launch_wsort<32, true>(a, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1, rows_upper, stream))) return rc;
    if ((rc = launch_wsort<64, true>(a, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2, rows_upper, stream))) return rc;
  }
  return 0;
}

}  // namespace fsw
