// Backward kernels for long rows (in-degree above FSW_REG_MAX_DEG): LDS bitonic variant up to FSW_LDS_MAX_DEG and
// global-scratch bitonic variant above.  gfx950.  (The forward of these rows lives in embed_mid.hip / embed_wsort.hip.)
//
// One workgroup takes one recipient row and walks its slices in groups of SC.  Per group:
//   1. gather   Xp[col_t, k0 .. k0+SC-1] for every neighbour t (row segments of SC floats, coalesced along
//               the slice axis) and store them transposed, tile[slice][t], one padded line per slice, as 64-bit
//               (orderable key bits << 32 | element index) words;
//   2. sort     every slice's line with a bitonic network (all 256 threads share each stage; padding
//               elements carry +inf keys);
//   3. walk     one wave per slice over the sorted line: cumulative weight by wave scans, coefficient and its
//               xi-derivative in float64, contributions scattered back to element order, then slice-contiguous
//               atomics into gXp.
// The global variant runs the same code on a tile in a global scratch buffer instead of LDS.
#include <algorithm>
#include "fsw_common.h"

namespace fsw {

constexpr double kPiL = 3.14159265358979323846;
constexpr int kLdsBytes = 65536 - 256;
constexpr int kLdsWeightFloats = FSW_LDS_MAX_DEG + 8;  // raw weights of one row (weighted variant)
constexpr int kGlobalSC = 16;                          // slices per group on the global path
constexpr int kSplitY = 4;                             // workgroups sharing one row (disjoint slice groups)
constexpr int kGlobalWgs = 64;                         // persistent workgroups (x kSplitY) on the global path
size_t embed_global_scratch_per_wg(int64_t max_degree);

__device__ __forceinline__ float mass_encode_l(float m, int fn) {
  if (fn == 1) return 2.f * (m / (sqrtf(m + 1.f) + 1.f));
  if (fn == 2) return log1pf(m);
  return m;
}

__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ __forceinline__ float sin2pi_rev_l(double x) {
  const double r = x - rint(x);
  return sinpif(2.f * (float)r);
}

__device__ __forceinline__ double wave_inclusive_scan_f64(double v) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const double t = __shfl_up(v, off);
    if (lane_id() >= off) v += t;
  }
  return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <bool WEIGHTED>
struct Elem {
  using type = float;
  static __device__ __forceinline__ type pad() { return __builtin_inff(); }
  static __device__ __forceinline__ type make(float key, int) { return key; }
  static __device__ __forceinline__ float key(type e) { return e; }
  static __device__ __forceinline__ int idx(type) { return 0; }
};
template <>
struct Elem<true> {
  using type = unsigned long long;
  static __device__ __forceinline__ type pad() { return ~0ull; }
  static __device__ __forceinline__ type make(float key, int i) { return ((type)orderable(key) << 32) | (uint32_t)i; }
  static __device__ __forceinline__ float key(type e) { return from_orderable((uint32_t)(e >> 32)); }
  static __device__ __forceinline__ int idx(type e) { return (int)(uint32_t)e; }
};

// key of CSR entry e for slice k: projected sender feature (+ projected edge feature, reference fsw_embedding.py:934-968)
__device__ __forceinline__ float long_key(const float* __restrict__ Xp, int64_t ldp, const int32_t* __restrict__ col, int e, int k,
                                          const float* __restrict__ efeat, const float* __restrict__ Ve, int64_t ldve, int d_edge) {
  float key = Xp[(int64_t)col[e] * ldp + k];
  if (efeat) {
    const float* er = efeat + (int64_t)e * d_edge;
    const float* vr = Ve + (int64_t)k * ldve;
    for (int q = 0; q < d_edge; ++q) key = fmaf(er[q], vr[q], key);
  }
  return key;
}

// All threads of the workgroup: bitonic sort of `nlines` lines of Dp elements each, line stride `ls`.
template <class T>
__device__ __forceinline__ void bitonic_lines(T* tile, int nlines, int Dp, int ls) {
  const int half = Dp >> 1;
  const int total = nlines * half;
  for (int size = 2; size <= Dp; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int line = i / half;
        const int j = i - line * half;
        const int pos = 2 * j - (j & (stride - 1));
        T* p = tile + (int64_t)line * ls + pos;
        const T a = p[0], b = p[stride];
        const bool up = (pos & size) == 0;
        if ((a > b) == up) {
          p[0] = b;
          p[stride] = a;
        }
      }
      __syncthreads();
    }
  }
}

// ---- backward for long rows ------------------------------------------------------------------------------------------
// value and xi-derivative of  F(xi; c) = (1 + xi) sin(2 pi xi c) / (pi xi):  (1 + xi) Delta_t = F(c_t) - F(c_{t-1}),
// so the coefficient of the element at cumulative weight c_t is F(c_t) - F(c_t - w_t) and its xi-derivative the same
// difference of dF.  Evaluated in float64 (the two terms of dF cancel to O(1) from O(c / xi)); series for tiny phases.
__device__ __forceinline__ void fsw_F_dF(double xi, double c, double& F, double& dF) {
  const double x = 2.0 * kPiL * xi * c;
  if (x < 1e-4) {
    const double q = 1.0 - x * x * (1.0 / 6.0);
    F = (1.0 + xi) * 2.0 * c * q;
    dF = 2.0 * c * q - (1.0 + xi) * 2.0 * c * (2.0 * kPiL * c) * (2.0 * kPiL * c) * xi * (1.0 / 3.0);
    return;
  }
  const double ph = xi * c;
  double s, co;
  sincospi(2.0 * (ph - rint(ph)), &s, &co);
  F = (1.0 + xi) * s / (kPiL * xi);
  dF = -s / (kPiL * xi * xi) + (1.0 + xi) * 2.0 * c * co / xi;
}

// Same structure as k_embed_long: gather + transpose (always with the element index as payload), bitonic sort of
// every slice line, then per slice one wave walks the sorted line: cumulative weight by wave scans, coefficient
// C_s = F(c_s) - F(c_{s-1}) and its xi-derivative, gfreq reduced per slice, and the neighbour's contribution
// g * C_s is scattered back to the element's ORIGINAL position in a second tile.  After a barrier every neighbour t
// receives one atomic add per slice of the group with the lanes along the slice axis: contiguous runs in gXp[col_t].
template <bool WEIGHTED, bool GLOBAL>
__global__ void __launch_bounds__(256) k_embed_long_bwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ bin_start, const float* __restrict__ Xp,
                                                        int64_t ldp, int S, const float* __restrict__ freqs, float tau,
                                                        const float* __restrict__ g, int64_t ldg, int gcol0, float out_scale,
                                                        float* __restrict__ gXp, int64_t ldgp, float* __restrict__ gfreq,
                                                        char* __restrict__ scratch, int64_t scratch_per_wg,
                                                        const float* __restrict__ efeat, const float* __restrict__ Ve, int64_t ldve,
                                                        int d_edge, float* __restrict__ gkey, int64_t ldk, int bin_lo, int bin_hi) {
  using E = Elem<true>;
  using T = unsigned long long;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wrow = reinterpret_cast<float*>(smem);
  constexpr int head = WEIGHTED ? kLdsWeightFloats * 4 : 0;
  char* tile_base = GLOBAL ? scratch + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * scratch_per_wg : smem + head;
  const int tile_bytes = GLOBAL ? 0 : kLdsBytes - head;
  const int pbeg = bin_start[bin_lo], pend = bin_start[bin_hi + 1];   // rows of the degree bins bin_lo .. bin_hi
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  __shared__ double msum[4];

  for (int p = pbeg + blockIdx.x; p < pend; p += gridDim.x) {
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    const int Dtot = WEIGHTED ? D + 1 : D;
    const int Dp = (int)pow2ceil((uint32_t)Dtot);
    const int ls = Dp + 1;
    int SC;
    if constexpr (GLOBAL) {
      SC = kGlobalSC;
    } else {
      SC = 64;
      while (SC > 1 && (int64_t)SC * ls * 12 > tile_bytes) SC >>= 1;
    }
    T* tile = reinterpret_cast<T*>(tile_base);
    float* tc = reinterpret_cast<float*>(tile_base + (int64_t)SC * ls * sizeof(T));

    double m = (double)D;
    if constexpr (WEIGHTED) {
      double part = 0.0;
      for (int t = threadIdx.x; t < D; t += blockDim.x) {
        const float wt = w ? w[start + t] : 1.f;
        if constexpr (!GLOBAL) wrow[t] = wt;
        part += (double)wt;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
      if (lane == 0) msum[wv] = part;
      __syncthreads();
      m = msum[0] + msum[1] + msum[2] + msum[3];
      if constexpr (!GLOBAL)
        if (threadIdx.x == 0) wrow[D] = (float)fmax((double)tau - m, 0.0);
      __syncthreads();
    }
    const double denom = WEIGHTED ? fmax(m, (double)tau) : m;
    const double inv = 1.0 / denom;
    const float padw = WEIGHTED ? (float)fmax((double)tau - m, 0.0) : 0.f;

    const int ngroups = (S + SC - 1) / SC;
    for (int grp = blockIdx.y; grp < ngroups; grp += gridDim.y) {
      const int k0 = grp * SC;
      for (int i = threadIdx.x; i < Dp * SC; i += blockDim.x) {
        const int kk = i % SC, t = i / SC;
        T e = E::pad();
        if (t < D) {
          e = E::make(long_key(Xp, ldp, col, start + t, min(k0 + kk, S - 1), efeat, Ve, ldve, d_edge), t);
        } else if (WEIGHTED && t == D) {
          e = E::make(0.f, t);
        }
        tile[(int64_t)kk * ls + t] = e;
      }
      __syncthreads();
      bitonic_lines<T>(tile, SC, Dp, ls);
      for (int kk = wv; kk < SC; kk += 4) {
        const int k = k0 + kk;
        if (k >= S) break;
        const double xi = (double)freqs[k];
        const float gi = out_scale * g[(int64_t)node * ldg + gcol0 + k];
        const T* line = tile + (int64_t)kk * ls;
        float* tcl = tc + (int64_t)kk * ls;
        double carry = 0.0, F_carry = 0.0, dF_carry = 0.0;
        float gf = 0.f;
        for (int t0 = 0; t0 < Dtot; t0 += kWave) {
          const int t = t0 + lane;
          const bool valid = t < Dtot;
          const T e = valid ? line[t] : E::make(0.f, 0);
          const float key = E::key(e);
          const int id = E::idx(e);
          double c;
          if constexpr (WEIGHTED) {
            float wt;
            if constexpr (GLOBAL)
              wt = valid ? (id == D ? padw : (w ? w[start + id] : 1.f)) : 0.f;
            else
              wt = valid ? wrow[id] : 0.f;
            c = wave_inclusive_scan_f64((double)wt) + carry;
          } else {
            c = (double)min(t + 1, Dtot);
          }
          double F, dF;
          fsw_F_dF(xi, c * inv, F, dF);
          double Fp = __shfl_up(F, 1), dFp = __shfl_up(dF, 1);
          if (lane == 0) {
            Fp = F_carry;
            dFp = dF_carry;
          }
          if (valid) {
            if (id < D) tcl[id] = gi * (float)(F - Fp);            // the pad element (id == D) has no source row
            gf = fmaf(gi * (float)(dF - dFp), key, gf);
          }
          carry = __shfl(c, kWave - 1);
          F_carry = __shfl(F, kWave - 1);
          dF_carry = __shfl(dF, kWave - 1);
        }
        gf = wave_sum_f32(gf);
        if (lane == 0 && gfreq) atomicAdd(gfreq + k, gf);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < D * SC; i += blockDim.x) {
        const int kk = i % SC, t = i / SC;
        if (k0 + kk < S) {
          if (gkey) gkey[(int64_t)(start + t) * ldk + k0 + kk] = tc[(int64_t)kk * ls + t];   // edge features: per-entry key gradient
          else atomicAdd(gXp + (int64_t)col[start + t] * ldgp + k0 + kk, tc[(int64_t)kk * ls + t]);
        }
      }
      __syncthreads();
    }
  }
}

int launch_embed_long_bwd(const fsw_embed_args& a, bool global, int64_t rows_upper, const float* g, int64_t ldg, float* gXp,
                          int64_t ldgp, float* gfreq, float* gkey, int64_t ldk, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  const bool unit = (a.w == nullptr) && (a.tau <= 1.f);
  char* scratch = reinterpret_cast<char*>(a.scratch);
  int64_t per_wg = 0;
  dim3 grid((unsigned)std::min<int64_t>(rows_upper, global ? kGlobalWgs : (1 << 16)), kSplitY);
  if (global) {
    per_wg = (int64_t)(a.scratch_bytes / (kGlobalWgs * kSplitY));
    FSW_REQUIRE(a.scratch && a.max_degree > FSW_LDS_MAX_DEG && per_wg >= (int64_t)embed_global_scratch_per_wg(a.max_degree),
                "fsw_embed_backward_f32: scratch buffer missing or too small for the global path");
  }
#define FSW_LAUNCH_LONG_BWD(WGT, GLB)                                                                                          \
  k_embed_long_bwd<WGT, GLB><<<grid, 256, (GLB) ? 1024 : kLdsBytes, stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, a.Xp, \
                                                                             a.ldp, a.S, a.freqs, a.tau, g, ldg, a.has_mass,   \
                                                                             a.out_scale, gXp, ldgp, gfreq, scratch, per_wg,   \
                                                                             a.efeat, a.Ve, a.ldve, a.d_edge, gkey, ldk,        \
                                                                             (GLB) ? FSW_BIN_GLOBAL : FSW_BIN_MID0, (GLB) ? FSW_BIN_GLOBAL : FSW_BIN_GLOBAL - 1)
  if (global) {
    if (unit) FSW_LAUNCH_LONG_BWD(false, true);
    else FSW_LAUNCH_LONG_BWD(true, true);
  } else {
    if (unit) FSW_LAUNCH_LONG_BWD(false, false);
    else FSW_LAUNCH_LONG_BWD(true, false);
  }
#undef FSW_LAUNCH_LONG_BWD
  FSW_LAUNCH_CHECK();
  return 0;
}

size_t embed_global_scratch_per_wg(int64_t max_degree) {
  const size_t Dp = pow2ceil((uint32_t)(max_degree + 1));
  // sorted 64-bit tile + (backward only) one float per element: 12 bytes per (slice, element)
  return (size_t)kGlobalSC * (Dp + 1) * (sizeof(unsigned long long) + sizeof(float));
}

size_t embed_global_scratch_bytes(int64_t max_degree) {
  return embed_global_scratch_per_wg(max_degree) * kGlobalWgs * kSplitY;
}

}  // namespace fsw
