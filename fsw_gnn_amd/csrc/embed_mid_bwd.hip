// Backward of the neighbourhood embedding for rows of FSW_REG_MAX_DEG < in-degree <= 128, one lane per slice.  gfx950.
//
// Mirror of embed_mid.hip: one wavefront per recipient row and 64-slice chunk, the whole neighbourhood in the lane's
// registers, here as (key, element index) pairs sorted by the network of the bin's padded size (equal keys by index = the
// reference's stable order).  The lane then walks its ranks -- unit weights: sin and cos of the rank angle by a float64
// rotation; general weights: cumulative weight in float64 + sincospi -- and drops g * [F(c_r) - F(c_r - w_r)] into the
// element's ORIGINAL position of a wave-private LDS tile [element][lane] (bank = lane: conflict-free although every lane
// scatters to its own permutation).  Then one wave-wide float atomic on 256 contiguous bytes of gXp[col_t] per neighbour
// (the full-rate shape), as on the <= 32 path (embed_bwd.hip); gfreq is reduced per lane over the workgroup's rows.
// F(xi; c) = (1 + xi) sin(2 pi xi c)/(pi xi), see embed_wsort_bwd.hip (which serves the rows above 128).
#include <algorithm>
#include "fsw_common.h"
#include "sortnet.h"

namespace fsw {

constexpr double kPiMB = 3.14159265358979323846;

struct FCoefM {
  double xi, a1, a2, a3;   // a1 = (1 + xi)/(pi xi), a2 = 1/(pi xi^2), a3 = 2 (1 + xi)/xi: no division left per element
  __device__ __forceinline__ explicit FCoefM(double x) : xi(x) {
    const double r = x > 0.0 ? 1.0 / x : 0.0;
    a1 = (1.0 + x) * r * (1.0 / kPiMB);
    a2 = r * r * (1.0 / kPiMB);
    a3 = 2.0 * (1.0 + x) * r;
  }
};
__device__ __forceinline__ void F_dF_sc_m(const FCoefM& f, double c, double s, double co, double& F, double& dF) {
  const double x = 2.0 * kPiMB * f.xi * c;
  if (x < 1e-4) {   // series: the two terms of dF cancel for tiny phases; xi == 0 gives F = dF = 2 c
    const double q = 1.0 - x * x * (1.0 / 6.0);
    F = (1.0 + f.xi) * 2.0 * c * q;
    dF = 2.0 * c * q - (1.0 + f.xi) * 2.0 * c * (2.0 * kPiMB * c) * (2.0 * kPiMB * c) * f.xi * (1.0 / 3.0);
  } else {
    F = f.a1 * s;
    dF = fma(f.a3 * c, co, -(f.a2 * s));
  }
}

// DP wires: the bin's padded degree, + 1 for the reference's pad element when WEIGHTED
template <int DP, bool WEIGHTED>
__global__ void __launch_bounds__(256) k_embed_mid_bwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ bin_start, int bin,
                                                       const float* __restrict__ Xp, int64_t ldp, int S,
                                                       const float* __restrict__ freqs, float tau, const float* __restrict__ g,
                                                       int64_t ldg, int gcol0, float out_scale, float* __restrict__ gXp,
                                                       int64_t ldgp, float* __restrict__ gfreq, const float* __restrict__ efeat,
                                                       const float* __restrict__ Ve, int64_t ldve, int d_edge,
                                                       float* __restrict__ gkey, int64_t ldk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int chunk = blockIdx.y * 4 + wave_id();
  if (chunk * kWave >= S) return;
  const int lane = lane_id();
  const int k = chunk * kWave + lane;
  const bool kvalid = k < S;
  const int kc = kvalid ? k : S - 1;
  float* tile = smem + wave_id() * (DP * kWave) + lane;   // [element][lane] of this wave
  const int pbeg = bin_start[bin], pend = bin_start[bin + 1];
  const double xi = (double)freqs[kc];
  const FCoefM fc(xi);
  const double taud = (double)tau;
  float gf = 0.f;
  for (int p = pbeg + blockIdx.x; p < pend; p += gridDim.x) {
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    const int Dtot = WEIGHTED ? D + 1 : D;
    U64Net<DP> net;
    double m = (double)D;
    if constexpr (WEIGHTED) m = 0.0;
#pragma unroll
    for (int t = 0; t < DP; ++t) {
      float key = __builtin_inff();
      if (t < D) {
        key = Xp[(int64_t)col[start + t] * ldp + kc];
        if (efeat) {   // edge features: + <e_ij, v_k[d_in:]> (reference fsw_embedding.py:934-968)
          const float* er = efeat + (int64_t)(start + t) * d_edge;
          const float* vr = Ve + (int64_t)kc * ldve;
          for (int q = 0; q < d_edge; ++q) key = fmaf(er[q], vr[q], key);
        }
        if constexpr (WEIGHTED) m += (double)(w ? w[start + t] : 1.f);
      } else if (WEIGHTED && t == D) {
        key = 0.f;   // the reference's pad element at x = 0
      }
      net.e[t] = pack_key_index(key, t);
    }
    sort_network<DP>(net);
    const double inv = 1.0 / (WEIGHTED ? fmax(m, taud) : m);
    const float padw = WEIGHTED ? (float)fmax(taud - m, 0.0) : 0.f;
    const float gi = kvalid ? out_scale * g[(int64_t)node * ldg + gcol0 + k] : 0.f;
    double Fp = 0.0, dFp = 0.0;   // F(0) = dF(0) = 0
    if constexpr (!WEIGHTED) {
      const double step = xi * inv;
      double sd, cd, s = 0.0, c = 1.0;
      sincospi(2.0 * (step - rint(step)), &sd, &cd);
#pragma unroll
      for (int r = 0; r < DP; ++r) {
        if (r < D) {
          const double sn = fma(s, cd, c * sd), cn = fma(c, cd, -(s * sd));
          s = sn;
          c = cn;
          double F, dF;
          F_dF_sc_m(fc, (double)(r + 1) * inv, s, c, F, dF);
          tile[(int)(unsigned int)net.e[r] * kWave] = gi * (float)(F - Fp);
          gf = fmaf(gi * (float)(dF - dFp), from_orderable_bits((unsigned int)(net.e[r] >> 32)), gf);
          Fp = F;
          dFp = dF;
        }
      }
    } else {
      double c = 0.0;
#pragma unroll
      for (int r = 0; r < DP; ++r) {
        if (r < Dtot) {
          const int id = (int)(unsigned int)net.e[r];
          c += (double)(id == D ? padw : (w ? w[start + id] : 1.f));
          const double ph = xi * (c * inv);
          double s, co, F, dF;
          sincospi(2.0 * (ph - rint(ph)), &s, &co);
          F_dF_sc_m(fc, c * inv, s, co, F, dF);
          if (id < D) tile[id * kWave] = gi * (float)(F - Fp);   // the pad element has no source row
          gf = fmaf(gi * (float)(dF - dFp), from_orderable_bits((unsigned int)(net.e[r] >> 32)), gf);
          Fp = F;
          dFp = dF;
        }
      }
    }
    // lane-private column of the tile: no barrier needed between the scatter above and the reads below
#pragma unroll 4
    for (int t = 0; t < D; ++t) {
      const float v = tile[t * kWave];
      if (kvalid) {
        if (gkey) gkey[(int64_t)(start + t) * ldk + k] = v;   // edge features: per-entry key gradient
        else atomicAdd(gXp + (int64_t)col[start + t] * ldgp + k, v);
      }
    }
  }
  if (gfreq && kvalid && gf != 0.f) atomicAdd(gfreq + k, gf);
}

#ifndef FSW_MID_BWD_PART
#error "compile with -DFSW_MID_BWD_PART=0|1"
#endif
int launch_mid_bwd_unit(const fsw_embed_args& a, dim3 grid, const float* g, int64_t ldg, float* gXp, int64_t ldgp, float* gfreq,
                        float* gkey, int64_t ldk, hipStream_t stream);
int launch_mid_bwd_weighted(const fsw_embed_args& a, dim3 grid, const float* g, int64_t ldg, float* gXp, int64_t ldgp, float* gfreq,
                            float* gkey, int64_t ldk, hipStream_t stream);

#define FSW_MID_BWD(i, DP, WGT)                                                                                                   \
  do {                                                                                                                            \
    constexpr int lds = 4 * (DP) * kWave * (int)sizeof(float);                                                                    \
    FSW_SET_MAX_LDS_ONCE((k_embed_mid_bwd<DP, WGT>), lds);                                                                        \
    k_embed_mid_bwd<DP, WGT><<<grid, 256, lds, stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, FSW_BIN_MID0 + i, a.Xp, a.ldp,  \
                                                         a.S, a.freqs, a.tau, g, ldg, a.has_mass, a.out_scale, gXp, ldgp, gfreq,  \
                                                         a.efeat, a.Ve, a.ldve, a.d_edge, gkey, ldk);                             \
    FSW_LAUNCH_CHECK();                                                                                                           \
  } while (0)

#if FSW_MID_BWD_PART == 0
int launch_mid_bwd_unit(const fsw_embed_args& a, dim3 grid, const float* g, int64_t ldg, float* gXp, int64_t ldgp, float* gfreq,
                        float* gkey, int64_t ldk, hipStream_t stream) {
  FSW_MID_BWD(0, 40, false); FSW_MID_BWD(1, 48, false); FSW_MID_BWD(2, 64, false); FSW_MID_BWD(3, 80, false);
  FSW_MID_BWD(4, 96, false); FSW_MID_BWD(5, 128, false);
  return 0;
}

// bins of the padded sizes 40 .. 128 (FSW_MID_MAX_DEG_WEIGHTED); rows_upper bounds the rows of all long-row bins
int launch_embed_mid_bwd(const fsw_embed_args& a, int64_t rows_upper, const float* g, int64_t ldg, float* gXp, int64_t ldgp,
                         float* gfreq, float* gkey, int64_t ldk, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  const bool unit = (a.w == nullptr) && (a.tau <= 1.f);
  dim3 grid((unsigned)std::min<int64_t>(rows_upper, 4096), (unsigned)ceil_div(a.S, 4 * kWave));
  return unit ? launch_mid_bwd_unit(a, grid, g, ldg, gXp, ldgp, gfreq, gkey, ldk, stream)
              : launch_mid_bwd_weighted(a, grid, g, ldg, gXp, ldgp, gfreq, gkey, ldk, stream);
}
#else
int launch_mid_bwd_weighted(const fsw_embed_args& a, dim3 grid, const float* g, int64_t ldg, float* gXp, int64_t ldgp, float* gfreq,
                            float* gkey, int64_t ldk, hipStream_t stream) {
  FSW_MID_BWD(0, 41, true); FSW_MID_BWD(1, 49, true); FSW_MID_BWD(2, 65, true); FSW_MID_BWD(3, 81, true);
  FSW_MID_BWD(4, 97, true); FSW_MID_BWD(5, 129, true);
  return 0;
}
#endif
#undef FSW_MID_BWD

}  // namespace fsw
