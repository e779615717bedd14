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
#include "wave_sort.h"

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
      // 1. gather + transpose: consecutive threads take consecutive slices of one neighbour.  kGatherDepth loads are
      //    issued before the first LDS store: the loop is otherwise one HBM round trip per iteration.
      constexpr int kGatherDepth = 8;
      const int total = Dtot * SC;
      for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * kGatherDepth) {
        float key[kGatherDepth];
        int pos[kGatherDepth];
#pragma unroll
        for (int u = 0; u < kGatherDepth; ++u) {
          const int i = min(i0 + u * (int)blockDim.x, total - 1);   // clamped: the tail re-reads the last element
          const int kk = i % SC, t = i / SC;
          pos[u] = kk * LINE + t + t / M;
          key[u] = 0.f;                                  // t == D (weighted): the reference's pad element at x = 0
          if (t < D) {
            const int kcl = min(k0 + kk, S - 1);
            key[u] = Xp[(int64_t)col[start + t] * ldp + kcl];
            if (efeat) {   // edge features: + <e_ij, v_k[d_in:]> (reference fsw_embedding.py:934-968)
              const float* er = efeat + (int64_t)(start + t) * d_edge;
              const float* vr = Ve + (int64_t)kcl * ldve;
              for (int q = 0; q < d_edge; ++q) key[u] = fmaf(er[q], vr[q], key[u]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kGatherDepth; ++u) tile[pos[u]] = key[u];   // clamped duplicates store the same value
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

// ---- rows above FSW_LDS_MAX_DEG: one wavefront per (row, slice) line, the line in a global scratch buffer --------------
// The line is sorted in chunks of 64 M elements in registers (WaveLine::sort), then merged bitonically: the exchanges
// at distances >= one chunk are element-wise min/max sweeps over the scratch line (coalesced, kSweepDepth pairs in flight
// per lane), the rest of every merge level happens in registers again (WaveLine::merge_chunk); the last level feeds the
// readout instead of going back to memory.  Any in-degree works; the scratch holds 4 (unit) or 8 bytes per element of the
// padded line and wave.  The gather is one 4-byte read per neighbour and slice here (the four waves of a workgroup take
// adjacent slices, so they share sectors): these rows are few.
#ifndef FSW_SWEEP_DEPTH
#define FSW_SWEEP_DEPTH 4
#endif
constexpr int kSweepDepth = FSW_SWEEP_DEPTH;
#ifndef FSW_WSG_ABL
#define FSW_WSG_ABL 0   // timing experiments on k_embed_wsort_global: 1 no gather, 2 no register sorts, 4 no sweeps, 8 agent-scope fences
#endif
// Orders a wave's scratch-line stores before the loads of its next pass (other lanes of the SAME wave read them).
// Workgroup scope is enough: the wave's loads and stores go through the one L1 of its CU, which stays coherent for the
// CU's own stores; an agent-scope fence also invalidates that L1 for every resident wave and made this kernel 2.6x slower.
#define FSW_WSG_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, (FSW_WSG_ABL & 8) ? "agent" : "workgroup")

template <bool WEIGHTED>
__device__ __forceinline__ void sweep_pairs(float* __restrict__ sk, float* __restrict__ sw, int Dp, int size, int st, bool flip) {
  // pairs (i, p): flip: i in the lower half of every block of `size`, p = its mirror image in the block;
  //               otherwise i and i + st inside blocks of 2 st
  const int npairs = Dp >> 1;
  const int half = size >> 1;
  for (int i0 = lane_id(); i0 < npairs; i0 += kWave * kSweepDepth) {   // npairs is a multiple of 64 * kSweepDepth
    int ia[kSweepDepth], ib[kSweepDepth];
    float ka[kSweepDepth], kb[kSweepDepth], wa[kSweepDepth], wb[kSweepDepth];
#pragma unroll
    for (int u = 0; u < kSweepDepth; ++u) {
      const int idx = i0 + u * kWave;
      if (flip) {
        const int blk = idx / half, off = idx - blk * half;
        ia[u] = blk * size + off;
        ib[u] = blk * size + size - 1 - off;
      } else {
        const int blk = idx / st, off = idx - blk * st;
        ia[u] = blk * 2 * st + off;
        ib[u] = ia[u] + st;
      }
      ka[u] = sk[ia[u]];
      kb[u] = sk[ib[u]];
      if constexpr (WEIGHTED) {
        wa[u] = sw[ia[u]];
        wb[u] = sw[ib[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < kSweepDepth; ++u) {
      if constexpr (WEIGHTED) {
        if (kb[u] < ka[u]) {
          sk[ia[u]] = kb[u];
          sk[ib[u]] = ka[u];
          sw[ia[u]] = wb[u];
          sw[ib[u]] = wa[u];
        }
      } else {
        sk[ia[u]] = fminf(ka[u], kb[u]);
        sk[ib[u]] = fmaxf(ka[u], kb[u]);
      }
    }
  }
  FSW_WSG_FENCE();   // the next sweep reads what other lanes of this wave wrote
}

template <int M, bool WEIGHTED>
__global__ void __launch_bounds__(256) k_embed_wsort_global(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                            const int32_t* __restrict__ bin_start, const float* __restrict__ Xp,
                                                            int64_t ldp, int S, const float* __restrict__ freqs, float tau,
                                                            float* __restrict__ out, int64_t ldo, const float* __restrict__ bias,
                                                            float out_scale, int has_mass, int mass_fn, float mass_scale,
                                                            const float* __restrict__ efeat, const float* __restrict__ Ve,
                                                            int64_t ldve, int d_edge, char* __restrict__ scratch, int64_t wave_bytes,
                                                            int bin_lo, int bin_hi, int dlo) {
  constexpr int CAP = M * kWave;
  const int lane = lane_id();
  // lines are numbered (row, slice) with the slice fastest; the workgroups of one XCD (blockIdx.x % 8 under round-robin
  // dispatch) take consecutive lines when the grid is a multiple of 8, so that the slices sharing a sector of an Xp row
  // are read through the same L2
  const int blk = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
  const int gw = blk * 4 + wave_id(), nwaves = gridDim.x * 4;
  float* sk = reinterpret_cast<float*>(scratch + (int64_t)gw * wave_bytes);
  float* sw = sk + (wave_bytes >> 3);                   // second half of the wave's region (weighted only)
  const int pbeg = bin_start[bin_lo], pend = bin_start[bin_hi + 1];
  const int64_t nlines = (int64_t)(pend - pbeg) * S;
  for (int64_t ln_id = gw; ln_id < nlines; ln_id += nwaves) {
    const int p = pbeg + (int)(ln_id / S), k = (int)(ln_id % S);
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    if (D <= dlo) continue;                             // done by k_embed_hub_w (launch_embed_global)
    const int Dtot = WEIGHTED ? D + 1 : D;
    const int Dp = (int)pow2ceil((uint32_t)Dtot);       // >= 2 CAP: D > FSW_LDS_MAX_DEG = CAP; <= wave_bytes / 8 by the bin's bound
    double m = (double)D;
    float padw = 0.f;
    if constexpr (WEIGHTED) {
      double part = 0.0;
      for (int t = lane; t < D; t += kWave) part += (double)(w ? w[start + t] : 1.f);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
      m = part;
      padw = (float)fmax((double)tau - m, 0.0);
    }
    const double inv = 1.0 / (WEIGHTED ? fmax(m, (double)tau) : m);
    const float xif = freqs[k];
    const double xi = (double)xif;
    const bool lin = xif < 1e-30f;
    // A. chunks: gather, sort in registers, park in the scratch line
    for (int c0 = 0; c0 < Dp; c0 += CAP) {
      WaveLine<M, WEIGHTED> ln;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int t = c0 + j * kWave + lane;              // striped: lane-contiguous col / w reads (the chunk is sorted next)
        float key = __builtin_inff(), wt = 0.f;
        if (t < D) {
          key = (FSW_WSG_ABL & 1) ? (float)((t * 2654435761u) >> 8) : Xp[(int64_t)col[start + t] * ldp + k];
          if (efeat) {   // edge features: + <e_ij, v_k[d_in:]> (reference fsw_embedding.py:934-968)
            const float* er = efeat + (int64_t)(start + t) * d_edge;
            const float* vr = Ve + (int64_t)k * ldve;
            for (int q = 0; q < d_edge; ++q) key = fmaf(er[q], vr[q], key);
          }
          if constexpr (WEIGHTED) wt = w ? w[start + t] : 1.f;
        } else if (WEIGHTED && t == D) {
          key = 0.f;                                    // the reference's pad element at x = 0
          wt = padw;
        }
        ln.k[j] = key;
        if constexpr (WEIGHTED) ln.w[j] = wt;
      }
      if (!(FSW_WSG_ABL & 2)) ln.sort();
#pragma unroll
      for (int j = 0; j < M; ++j) {
        sk[c0 + lane * M + j] = ln.k[j];
        if constexpr (WEIGHTED) sw[c0 + lane * M + j] = ln.w[j];
      }
    }
    FSW_WSG_FENCE();
    // B. merge levels
    float acc = 0.f;
    double carry = 0.0;                                 // cumulative weight before the current chunk (weighted readout)
    double sd = 0.0, cd = 1.0;
    const double step = xi * inv;                       // revolutions per rank (unit weights)
    if (!WEIGHTED && !lin) sincospi(2.0 * (step - rint(step)), &sd, &cd);
    for (int size = 2 * CAP; size <= Dp; size <<= 1) {
      if (!(FSW_WSG_ABL & 4)) {
        sweep_pairs<WEIGHTED>(sk, sw, Dp, size, 0, true);
        for (int st = size >> 2; st >= CAP; st >>= 1) sweep_pairs<WEIGHTED>(sk, sw, Dp, size, st, false);
      }
      const bool last = size == Dp;
      for (int c0 = 0; c0 < Dp; c0 += CAP) {
        if (last && c0 >= Dtot) break;                  // only +inf padding from here on
        WaveLine<M, WEIGHTED> ln;
#pragma unroll
        for (int j = 0; j < M; ++j) {
          ln.k[j] = sk[c0 + lane * M + j];
          if constexpr (WEIGHTED) ln.w[j] = sw[c0 + lane * M + j];
        }
        if (!(FSW_WSG_ABL & 2)) ln.merge_chunk();
        if (!last) {
#pragma unroll
          for (int j = 0; j < M; ++j) {
            sk[c0 + lane * M + j] = ln.k[j];
            if constexpr (WEIGHTED) sw[c0 + lane * M + j] = ln.w[j];
          }
          continue;
        }
        // readout of ranks c0 + lane*M + j
        const int r0 = c0 + lane * M;
        if constexpr (!WEIGHTED) {
          if (lin) {
#pragma unroll
            for (int j = 0; j < M; ++j) acc += (r0 + j < D) ? ln.k[j] : 0.f;
          } else {
            double s, c;
            const double x0 = step * (double)r0;
            sincospi(2.0 * (x0 - rint(x0)), &s, &c);
#pragma unroll
            for (int j = 0; j < M; ++j) {
              const double sn = fma(s, cd, c * sd), cn = fma(c, cd, -(s * sd));
              acc += (r0 + j < D) ? (float)(sn - s) * ln.k[j] : 0.f;
              s = sn;
              c = cn;
            }
          }
        } else {
          double part = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) part += (double)ln.w[j];
          double c = carry + wave_exclusive_scan_f64(part);
          double tot = part;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
          carry += tot;
          float sprev = lin ? 0.f : sin2pi_rev_w(xi * (c * inv));
#pragma unroll
          for (int j = 0; j < M; ++j) {
            const bool valid = r0 + j < Dtot;
            c += (double)ln.w[j];
            if (lin) {
              acc += valid ? ln.w[j] * ln.k[j] : 0.f;
            } else {
              const float s = sin2pi_rev_w(xi * (c * inv));
              acc += valid ? (s - sprev) * ln.k[j] : 0.f;
              sprev = s;
            }
          }
        }
      }
      FSW_WSG_FENCE();
    }
    acc = wave_sum_w(acc) * (lin ? 2.f * (float)inv : (float)((1.0 + xi) / (kPiW * xi)));
    if (lane == 0) {
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (acc + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_w((float)m, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
  }
}

int launch_embed_ws_unit(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream);   // embed_hub.hip
int launch_embed_hub_weighted_lds(const fsw_embed_args& a, int bin_lo, int64_t rows_upper, hipStream_t stream);
int launch_embed_hub_weighted_hub(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream);
int launch_embed_mergepath_w(const fsw_embed_args& a, int bin_lo, int bin_hi, int dlo, int64_t rows_upper, hipStream_t stream);
#ifndef FSW_WEIGHTED_HUB
#define FSW_WEIGHTED_HUB 1   // 0: general weights on the LDS-staged / scratch-line kernels of this file only (for comparison)
#endif

// general weights: every row above FSW_LDS_MAX_DEG (unit weights with tau <= 1 take embed_hub.hip's kernels)
int launch_embed_global(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  // without edge features the rows of up to kHubWMaxDeg neighbours keep their line in registers (embed_hub.hip: k_embed_hub_w)
  int first_bin = FSW_BIN_HUB0, dlo = 0;
  if (FSW_WEIGHTED_HUB && !a.efeat) {
    if (int rc = launch_embed_hub_weighted_hub(a, rows_upper, stream)) return rc;
    first_bin = FSW_BIN_HUB0 + 1;      // its rows of exactly 8192 neighbours (8193 elements) are all that is left of this bin
    dlo = kHubWMaxDeg;
    if (a.max_degree > 0 && a.max_degree <= kHubWMaxDeg) return 0;
    if (bin_rows_or(a, first_bin, FSW_BIN_GLOBAL, 1) <= 0) return 0;
    // above: sorted blocks + merge-path levels (embed_hub.hip); FSW_WEIGHTED_SCRATCH=1 keeps the scratch-line kernel below, for comparison
    if (!getenv("FSW_WEIGHTED_SCRATCH")) return launch_embed_mergepath_w(a, first_bin, FSW_BIN_GLOBAL, dlo, rows_upper, stream);
  }
  FSW_REQUIRE(a.max_degree > FSW_LDS_MAX_DEG, "fsw_embed_f32: max_degree (host value) is required for rows above FSW_LDS_MAX_DEG");
  FSW_REQUIRE(a.scratch, "fsw_embed_f32: these rows need a scratch buffer (fsw_embed_scratch_bytes)");
  char* scratch = reinterpret_cast<char*>(a.scratch);
  // One launch per degree bin, each with the scratch line its OWN longest row needs: sized by the graph's longest row, a single
  // 150 000-neighbour hub left 680 wavefronts (two thirds of one per SIMD) for every row above 4096 neighbours.
  for (int bin = first_bin; bin <= FSW_BIN_GLOBAL; ++bin) {
    const int64_t rows = bin_rows_or(a, bin, bin, rows_upper);
    if (rows <= 0) continue;
    const int64_t bin_max = bin == FSW_BIN_GLOBAL ? a.max_degree : std::min<int64_t>(a.max_degree, (int64_t)4096 << (bin - FSW_BIN_HUB0));
    if (bin > FSW_BIN_HUB0 && bin_max <= ((int64_t)2048 << (bin - FSW_BIN_HUB0))) continue;   // no row of the graph reaches this bin
    const int64_t wave_bytes = (int64_t)pow2ceil((uint32_t)(bin_max + 1)) * 8;
    int64_t nwaves = std::min<int64_t>((int64_t)a.scratch_bytes / wave_bytes, 2048);
    nwaves = std::min<int64_t>(nwaves, ceil_div(rows * a.S, 32) * 32);
    nwaves = nwaves >= 32 ? (nwaves & ~(int64_t)31) : (nwaves & ~(int64_t)3);   // whole workgroups; a multiple of 8 of them when possible
    FSW_REQUIRE(nwaves >= 4, "fsw_embed_f32: scratch buffer too small for rows above FSW_LDS_MAX_DEG (need fsw_embed_scratch_bytes(max_degree))");
    k_embed_wsort_global<32, true><<<(unsigned)(nwaves / 4), 256, 0, stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, a.Xp, a.ldp, a.S,
        a.freqs, a.tau, a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale, a.efeat, a.Ve, a.ldve, a.d_edge, scratch, wave_bytes,
        bin, bin, dlo);
    FSW_LAUNCH_CHECK();
  }
  return 0;
}

template <int M, bool WEIGHTED>
static int launch_wsort(const fsw_embed_args& a, int bin_lo, int bin_hi, int64_t rows_upper, hipStream_t stream) {
  FSW_SET_MAX_LDS_ONCE((k_embed_wsort<M, WEIGHTED>), kWsLdsBytes);
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
  if (unit_fast) {   // one wavefront per line straight from Xp, no LDS staging (embed_hub.hip)
    return launch_embed_ws_unit(a, rows_upper, stream);
  } else {           // D + 1 elements with the pad element: one size up; the mid bins above FSW_MID_MAX_DEG_WEIGHTED come here too
    constexpr int sizes[FSW_NUM_MID_BINS] = FSW_MID_SIZES;
    int bin_lo = FSW_BIN_MID0;
    while (bin_lo < FSW_BIN_LDS0 && sizes[bin_lo - FSW_BIN_MID0] <= FSW_MID_MAX_DEG_WEIGHTED) ++bin_lo;
    if (FSW_WEIGHTED_HUB && !a.efeat) return launch_embed_hub_weighted_lds(a, FSW_BIN_MID0 + weighted_hub_first_mid_bin(), rows_upper, stream);
    if (bin_lo < FSW_BIN_LDS0 && (rc = launch_wsort<8, true>(a, bin_lo, FSW_BIN_LDS0 - 1, rows_upper, stream))) return rc;
    if ((rc = launch_wsort<16, true>(a, FSW_BIN_LDS0, FSW_BIN_LDS0, rows_upper, stream))) return rc;
    if ((rc = launch_wsort<32, true>(a, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1, rows_upper, stream))) return rc;
    if ((rc = launch_wsort<64, true>(a, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2, rows_upper, stream))) return rc;
  }
  return 0;
}

}  // namespace fsw
