// Backward of the fused neighbourhood kernel, register path, unit weights.  gfx950.
//
// The reference differentiates its chain of sparse ops by reverse-mode autograd (ag.*.backward, reference
// fsw_embedding.py:1284-2257): sum_sparseToDense.backward expands the output gradient to E*S entries,
// sinc_cos_sparse.backward multiplies saved E*S factors (:1796-1817), cumsum_sparse.backward is a reverse segmented
// cumsum (:2158-2172), permute_sparse.backward re-sorts E*S int64 keys (:1284-1325) and sort.backward scatters
// (:2055-2070).  With unit weights the forward is  out[i,k] = sum_t C[D_i][t][k] p_(t)  with a coefficient table that
// depends on (degree, rank, slice) only, so for an output gradient g:
//     d L / d Xp[j,k]  = sum over edges j->i of  g[i,k] C[D_i][rank_ik(j)][k]
//     d L / d xi_k     = sum_i g[i,k] sum_t dC[D_i][t][k] p_(t),        dC = d C / d xi  (second float64 table)
// One wavefront = one recipient row x one 64-slice chunk, lane = slice, as in the forward.  Ranks come from
// D(D-1)/2 register compares (no second sort: rank[u] += p_t < p_u, rank[t] += otherwise), the two coefficient
// rows of the wave's chunk sit in LDS indexed by the lane's rank, and every neighbour receives ONE wave-wide
// no-return global_atomic_add_f32 on 256 contiguous bytes of gXp[col_t] -- the full-rate shape for float atomics
// on this part (atomics execute at the memory side, ~1.3 TB/s of added bytes chip-wide; this kernel is bound by that).
// gX = gXp . V and gV = gXp^T . X are two plain GEMMs left to the BLAS.
#include "fsw_common.h"

namespace fsw {

constexpr int kBwdRows = 32;
constexpr double kPiB = 3.14159265358979323846;

// d/dxi [ (1 + xi) Delta_t ] for weights 1/D:  Delta = 2 w sinc(xi w) cos(B),  B = pi xi (2c - w)
__global__ void __launch_bounds__(256) k_unit_dtable(const float* __restrict__ freqs, int S, int max_deg,
                                                     float* __restrict__ dtable, int64_t ldt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  int D = 1;
  while ((D + 1) * D / 2 <= row) ++D;
  const int t = row - D * (D - 1) / 2;
  if (k >= S || D > max_deg) return;
  const double xi = (double)freqs[k];
  const double w = 1.0 / D, c = (double)(t + 1) / D;
  const double z = xi * w;
  const double sinc = (z == 0.0) ? 1.0 : sinpi(z) / (kPiB * z);
  const double dsinc = (z == 0.0) ? 0.0 : (cospi(z) - sinc) / z;
  const double B = xi * (2.0 * c - w);  // in units of pi
  const double delta = 2.0 * w * sinc * cospi(B);
  const double ddelta = 2.0 * w * (w * dsinc * cospi(B) - sinc * kPiB * (2.0 * c - w) * sinpi(B));
  dtable[(int64_t)row * ldt + k] = (float)(delta + (1.0 + xi) * ddelta);
}

template <int D>
__device__ __forceinline__ void unit_bwd_rows(int p, int pe, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                              const int32_t* __restrict__ perm, const float* __restrict__ Xp, int64_t ldp,
                                              const float* __restrict__ table, const float* __restrict__ dtable, int64_t ldt,
                                              const float* __restrict__ g, int64_t ldg, int gcol0, float out_scale,
                                              float* __restrict__ gXp, int64_t ldgp, float* __restrict__ gfreq,
                                              float* __restrict__ cS /* [2][D][64] wave-private LDS */, int k, int kc, bool kvalid,
                                              float* __restrict__ gkey, int64_t ldk) {
  const int lane = lane_id();
  const int64_t trow = (int64_t)(D * (D - 1) / 2);
#pragma unroll
  for (int t = 0; t < D; ++t) {
    cS[t * kWave + lane] = out_scale * table[(trow + t) * ldt + kc];
    cS[(D + t) * kWave + lane] = out_scale * dtable[(trow + t) * ldt + kc];
  }
  __builtin_amdgcn_wave_barrier();
  float gf = 0.f;
  for (; p < pe; ++p) {
    const int node = perm[p];
    const int start = rowptr[node];
    const float gi = kvalid ? g[(int64_t)node * ldg + gcol0 + k] : 0.f;
    float key[D];
    int cidx[D];
#pragma unroll
    for (int t = 0; t < D; ++t) {
      cidx[t] = col[start + t];
      key[t] = Xp[(int64_t)cidx[t] * ldp + kc];
    }
    int rank[D];
#pragma unroll
    for (int t = 0; t < D; ++t) rank[t] = 0;
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
      for (int t = u + 1; t < D; ++t) {
        const int before = key[t] < key[u];   // t sorts ahead of u only when strictly smaller (ties keep edge order)
        rank[u] += before;
        rank[t] += 1 - before;
      }
#pragma unroll
    for (int t = 0; t < D; ++t) {
      const float c = cS[rank[t] * kWave + lane];
      const float dc = cS[(D + rank[t]) * kWave + lane];
      gf = fmaf(gi * dc, key[t], gf);
      if (kvalid) {
        if (gkey) gkey[(int64_t)(start + t) * ldk + k] = gi * c;   // store-and-sum form: one plain 256-byte store per neighbour
        else atomicAdd(gXp + (int64_t)cidx[t] * ldgp + k, gi * c);
      }
    }
  }
  if (kvalid && gfreq) atomicAdd(gfreq + k, gf);
}

#define FSW_BWD_CASES(X)                                                                                               \
  X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) \
  X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

template <int DLO, int DHI>
__global__ void __launch_bounds__(256) k_embed_reg_unit_bwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start,
                                                            const float* __restrict__ Xp, int64_t ldp, int S,
                                                            const float* __restrict__ table, const float* __restrict__ dtable,
                                                            int64_t ldt, const float* __restrict__ g, int64_t ldg, int gcol0,
                                                            float out_scale, float* __restrict__ gXp, int64_t ldgp,
                                                            float* __restrict__ gfreq, float* __restrict__ gkey, int64_t ldk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wv = wave_id();
  const int chunk = blockIdx.y * 4 + wv;
  if (chunk * kWave >= S) return;
  const int k = chunk * kWave + lane_id();
  const bool kvalid = k < S;
  const int kc = kvalid ? k : S - 1;
  int D, p = 0, pe = 0;
  if (!find_degree_tile<kBwdRows>(bin_start, DLO, DHI, (int)blockIdx.x, D, p, pe)) return;   // fsw_common.h
  float* cS = smem + wv * (2 * DHI * kWave);
  switch (D) {
#define X(d)                                                                                                          \
  case d:                                                                                                             \
    if constexpr (d >= DLO && d <= DHI)                                                                               \
      unit_bwd_rows<d>(p, pe, rowptr, col, perm, Xp, ldp, table, dtable, ldt, g, ldg, gcol0, out_scale, gXp, ldgp,    \
                       gfreq, cS, k, kc, kvalid, gkey, ldk);                                                          \
    break;
    FSW_BWD_CASES(X)
#undef X
    default:
      break;
  }
}

// ---- general weights (self loops, 'gcn', explicit W, tau > 1), register path -------------------------------------------
// (1 + xi) Delta_t = F(c_t) - F(c_t - w_t) with F(xi; c) = (1 + xi) sin(2 pi xi c)/(pi xi) and c_t the cumulative
// normalised weight up to and including the element: the coefficient of neighbour t needs only its own cumulative
// weight, which D(D+1)/2 compares give without sorting (cum[u] += w[t] when t sorts ahead of u).  Coefficient and
// xi-derivative in float64 (the two terms of dF/dxi cancel from O(c/xi) to O(1)).
__device__ __forceinline__ void F_dF(double xi, double c, double& F, double& dF) {
  const double x = 2.0 * kPiB * xi * c;
  if (x < 1e-4) {
    const double q = 1.0 - x * x * (1.0 / 6.0);
    F = (1.0 + xi) * 2.0 * c * q;
    dF = 2.0 * c * q - (1.0 + xi) * 2.0 * c * (2.0 * kPiB * c) * (2.0 * kPiB * c) * xi * (1.0 / 3.0);
    return;
  }
  const double ph = xi * c;
  double s, co;
  sincospi(2.0 * (ph - rint(ph)), &s, &co);
  F = (1.0 + xi) * s / (kPiB * xi);
  dF = -s / (kPiB * xi * xi) + (1.0 + xi) * 2.0 * c * co / xi;
}

template <int DEG>
__device__ __forceinline__ void weighted_bwd_rows(int p, int pe, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                  const float* __restrict__ Xp, int64_t ldp, const float* __restrict__ freqs,
                                                  float tau, const float* __restrict__ g, int64_t ldg, int gcol0, float out_scale,
                                                  float* __restrict__ gXp, int64_t ldgp, float* __restrict__ gfreq, int k, int kc,
                                                  bool kvalid, const float* __restrict__ efeat, const float* __restrict__ Ve,
                                                  int64_t ldve, int d_edge, float* __restrict__ gkey, int64_t ldk) {
  const double xi = (double)freqs[kc];
  float gf = 0.f;
  for (; p < pe; ++p) {
    const int node = perm[p];
    const int start = rowptr[node];
    const float gi = kvalid ? out_scale * g[(int64_t)node * ldg + gcol0 + k] : 0.f;
    float key[DEG + 1], wr[DEG + 1];
    int cidx[DEG];
    double m = 0.0;
#pragma unroll
    for (int t = 0; t < DEG; ++t) {
      cidx[t] = col[start + t];
      key[t] = Xp[(int64_t)cidx[t] * ldp + kc];
      if (efeat) {
        const float* er = efeat + (int64_t)(start + t) * d_edge;
        const float* vr = Ve + (int64_t)kc * ldve;
        for (int q = 0; q < d_edge; ++q) key[t] = fmaf(er[q], vr[q], key[t]);
      }
      wr[t] = w ? w[start + t] : 1.f;
      m += (double)wr[t];
    }
    const double taud = (double)tau;
    const double inv = 1.0 / fmax(m, taud);
    key[DEG] = 0.f;                               // the reference's pad element (fsw_embedding.py:787-821)
    wr[DEG] = (float)fmax(taud - m, 0.0);
    double cum[DEG + 1];
#pragma unroll
    for (int t = 0; t <= DEG; ++t) cum[t] = (double)wr[t];
#pragma unroll
    for (int u = 0; u <= DEG; ++u)
#pragma unroll
      for (int t = u + 1; t <= DEG; ++t) {
        const bool before = key[t] < key[u];      // ties keep element order, like a stable sort
        cum[u] += before ? (double)wr[t] : 0.0;
        cum[t] += before ? 0.0 : (double)wr[u];
      }
#pragma unroll
    for (int t = 0; t < DEG; ++t) {
      double F1, dF1, F0, dF0;
      F_dF(xi, cum[t] * inv, F1, dF1);
      F_dF(xi, (cum[t] - (double)wr[t]) * inv, F0, dF0);
      gf = fmaf(gi * (float)(dF1 - dF0), key[t], gf);
      if (kvalid) {
        if (gkey) gkey[(int64_t)(start + t) * ldk + k] = gi * (float)(F1 - F0);   // edge features: per-entry key gradient
        else atomicAdd(gXp + (int64_t)cidx[t] * ldgp + k, gi * (float)(F1 - F0));
      }
    }
  }
  if (kvalid && gfreq) atomicAdd(gfreq + k, gf);
}

__global__ void __launch_bounds__(256) k_embed_reg_weighted_bwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                                const int32_t* __restrict__ bin_start, const float* __restrict__ Xp,
                                                                int64_t ldp, int S, const float* __restrict__ freqs, float tau,
                                                                const float* __restrict__ g, int64_t ldg, int gcol0, float out_scale,
                                                                float* __restrict__ gXp, int64_t ldgp, float* __restrict__ gfreq,
                                                                const float* __restrict__ efeat, const float* __restrict__ Ve,
                                                                int64_t ldve, int d_edge, float* __restrict__ gkey, int64_t ldk) {
  const int chunk = blockIdx.y * 4 + wave_id();
  if (chunk * kWave >= S) return;
  const int k = chunk * kWave + lane_id();
  const bool kvalid = k < S;
  const int kc = kvalid ? k : S - 1;
  int D, p = 0, pe = 0;
  if (!find_degree_tile<kBwdRows>(bin_start, 1, FSW_REG_MAX_DEG, (int)blockIdx.x, D, p, pe)) return;
  switch (D) {
#define X(d)                                                                                                                 \
  case d:                                                                                                                    \
    weighted_bwd_rows<d>(p, pe, rowptr, col, w, perm, Xp, ldp, freqs, tau, g, ldg, gcol0, out_scale, gXp, ldgp, gfreq, k, kc, \
                         kvalid, efeat, Ve, ldve, d_edge, gkey, ldk);                                                        \
    break;
    FSW_BWD_CASES(X)
#undef X
    default:
      break;
  }
}

int launch_embed_long_bwd(const fsw_embed_args& a, bool global, int64_t rows_upper, const float* g, int64_t ldg, float* gXp,
                          int64_t ldgp, float* gfreq, float* gkey, int64_t ldk, hipStream_t stream);

}  // namespace fsw

using namespace fsw;

extern "C" int fsw_unit_dcoeff_table(const float* freqs, int S, int max_deg, float* dtable, int64_t ldt, fsw_stream_t stream) {
  FSW_REQUIRE(freqs && dtable, "fsw_unit_dcoeff_table: null pointer");
  FSW_REQUIRE(S >= 1 && ldt >= S && max_deg >= 1 && max_deg <= FSW_REG_MAX_DEG,
              "fsw_unit_dcoeff_table: need S >= 1, ldt >= S, 1 <= max_deg <= %d", FSW_REG_MAX_DEG);
  dim3 grid((unsigned)ceil_div(S, 256), (unsigned)(max_deg * (max_deg + 1) / 2));
  k_unit_dtable<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(freqs, S, max_deg, dtable, ldt);
  FSW_LAUNCH_CHECK();
  return 0;
}

static int embed_backward_impl(const fsw_embed_args* args, const float* dtable, const float* g, int64_t ldg, float* gXp,
                               int64_t ldgp, float* gfreq, float* gkey, int64_t ldk, hipStream_t stream) {
  FSW_REQUIRE(args && g && (gXp || gkey), "fsw_embed_backward: null pointer");
  const fsw_embed_args& a = *args;
  FSW_REQUIRE(a.rowptr && a.col && a.perm && a.bin_start && a.Xp && a.freqs, "fsw_embed_backward: null pointer in args");
  FSW_REQUIRE(a.S >= 1 && a.ldp >= a.S && ldg >= a.S + a.has_mass && a.tau > 0.f, "fsw_embed_backward: bad sizes");
  FSW_REQUIRE(gkey ? ldk >= a.S : ldgp >= a.S, "fsw_embed_backward: bad gradient stride");
  FSW_REQUIRE(!a.efeat || (a.w && a.Ve && a.d_edge >= 1 && a.ldve >= a.d_edge && gkey),
              "fsw_embed_backward: edge features need a coalesced weighted graph, Ve and the key-gradient form");
  // unit-weight register rows use the coefficient tables; the key-gradient form may come without them (edge-feature
  // graphs are weighted anyway) and then takes the general kernels with w = 1
  const bool unit_fast = (a.w == nullptr) && (a.tau <= 1.f) && (!gkey || (a.unit_table && dtable));
  FSW_REQUIRE(!unit_fast || (a.unit_table && dtable && a.ldt >= a.S),
              "fsw_embed_backward: unit weights with tau <= 1 need unit_table and dtable");
  const int64_t nreg = a.num_reg_rows < 0 ? a.num_rows : a.num_reg_rows;
  const int64_t nlds = a.num_lds_rows < 0 ? a.num_rows : a.num_lds_rows;
  const int64_t nglob = a.num_global_rows < 0 ? a.num_rows : a.num_global_rows;
  if (nreg > 0) {
    dim3 grid((unsigned)(ceil_div(nreg, kBwdRows) + FSW_REG_MAX_DEG), (unsigned)ceil_div(a.S, 4 * kWave));
    if (unit_fast) {
      // long rows first; the two launches differ in the LDS they need for the wave-private coefficient rows
      k_embed_reg_unit_bwd<17, 32><<<grid, 256, 4 * 2 * 32 * kWave * sizeof(float), stream>>>(
          a.rowptr, a.col, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.unit_table, dtable, a.ldt, g, ldg, a.has_mass, a.out_scale,
          gXp, ldgp, gfreq, gkey, ldk);
      FSW_LAUNCH_CHECK();
      k_embed_reg_unit_bwd<1, 16><<<grid, 256, 4 * 2 * 16 * kWave * sizeof(float), stream>>>(
          a.rowptr, a.col, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.unit_table, dtable, a.ldt, g, ldg, a.has_mass, a.out_scale,
          gXp, ldgp, gfreq, gkey, ldk);
      FSW_LAUNCH_CHECK();
    } else {
      k_embed_reg_weighted_bwd<<<grid, 256, 0, stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.freqs,
                                                         a.tau, g, ldg, a.has_mass, a.out_scale, gXp, ldgp, gfreq, a.efeat, a.Ve,
                                                         a.ldve, a.d_edge, gkey, ldk);
      FSW_LAUNCH_CHECK();
    }
  }
  int rc;
  if (nlds > 0 && (rc = launch_embed_long_bwd(a, false, nlds, g, ldg, gXp, ldgp, gfreq, gkey, ldk, stream))) return rc;
  if (nglob > 0 && (rc = launch_embed_long_bwd(a, true, nglob, g, ldg, gXp, ldgp, gfreq, gkey, ldk, stream))) return rc;
  return 0;
}

extern "C" int fsw_embed_backward_f32(const fsw_embed_args* args, const float* dtable, const float* g, int64_t ldg, float* gXp,
                                      int64_t ldgp, float* gfreq, fsw_stream_t stream) {
  FSW_REQUIRE(gXp, "fsw_embed_backward_f32: null gXp");
  return embed_backward_impl(args, dtable, g, ldg, gXp, ldgp, gfreq, nullptr, 0, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int fsw_embed_backward_keys_f32(const fsw_embed_args* args, const float* dtable, const float* g, int64_t ldg, float* gkey,
                                           int64_t ldk, float* gfreq, fsw_stream_t stream) {
  FSW_REQUIRE(gkey, "fsw_embed_backward_keys_f32: null gkey");
  return embed_backward_impl(args, dtable, g, ldg, nullptr, 0, gfreq, gkey, ldk, reinterpret_cast<hipStream_t>(stream));
}

// ---- store-and-sum: gXp[j, :] = sum of the key-gradient rows of j's out-edges -------------------------------------------------
// Second half of the atomic-free backward: the kernels above stored one contribution row per CSR entry (gkey); here every
// workgroup takes kSegLen CONSECUTIVE entries of the sender-major order (fsw_graph_transpose) -- perfectly balanced whatever the
// out-degrees -- walks the senders covering them (wave = 64-slice chunk, lane = slice: one coalesced 256-byte row gather per
// entry, 8 in flight) and writes a sender's sum with a plain store when its whole list lies inside the segment, with one
// wave-wide float atomic otherwise (out is zeroed by the caller; two partial sums commute, so results stay bitwise reproducible
// for senders of up to 2 segments; longer lists -- hubs -- add in arrival order).
constexpr int kSegLen = 256;

__global__ void __launch_bounds__(256) k_segment_sum_rows(const float* __restrict__ src, int64_t lds, const int32_t* __restrict__ ptr,
                                                          const int32_t* __restrict__ order, int64_t num_out, int64_t nnz, int S,
                                                          float* __restrict__ out, int64_t ldo) {
  const int chunk = blockIdx.y * 4 + wave_id();
  if (chunk * kWave >= S) return;
  const int k = chunk * kWave + lane_id();
  const bool kvalid = k < S;
  const int kc = kvalid ? k : S - 1;
  // entries past ptr[num_out] belong to no sender (fsw_graph_transpose sorts out-of-range columns there): never walked
  const int64_t q0 = (int64_t)blockIdx.x * kSegLen, q1 = min(q0 + kSegLen, min(nnz, (int64_t)ptr[num_out]));
  if (q0 >= q1) return;
  // sender of entry q0: the largest j with ptr[j] <= q0.  A 64-ary search -- every lane probes one of 64 evenly spaced entries of
  // ptr, a ballot counts how many are <= q0 -- takes four rounds of one vector load at a million senders; the binary search it
  // replaces was twenty DEPENDENT scalar loads before the workgroup's first gather (cf. find_degree_tile, fsw_common.h).
  int64_t lo = 0, hi = num_out;                            // ptr[lo] <= q0 < ptr[hi]
  while (hi - lo > 1) {
    const int64_t step = (hi - lo + kWave - 1) / kWave;
    const int64_t idx = min(lo + (int64_t)(lane_id() + 1) * step, hi);
    const bool le = idx < hi && (int64_t)ptr[idx] <= q0;   // monotone over the lanes
    const int cnt = __popcll(__ballot(le));
    const int64_t nlo = lo + (int64_t)cnt * step;
    hi = min(nlo + step, hi);
    lo = nlo;
  }
  int64_t j = lo;
  int64_t nextb = ptr[j + 1];
  while (nextb <= q0) {          // ptr[j] == ptr[j + 1] == q0 runs of empty senders: take the last j with ptr[j] <= q0 < ptr[j + 1]
    ++j;
    nextb = ptr[j + 1];
  }
  float acc = 0.f;
  auto flush = [&](int64_t jj) {
    const bool complete = (int64_t)ptr[jj] >= q0 && (int64_t)ptr[jj + 1] <= q1;
    if (kvalid) {
      if (complete) out[jj * ldo + k] = acc;
      else atomicAdd(out + jj * ldo + k, acc);
    }
    acc = 0.f;
  };
  constexpr int kDepth = 8;
  for (int64_t qb = q0; qb < q1; qb += kDepth) {
    float v[kDepth];
#pragma unroll
    for (int u = 0; u < kDepth; ++u) {
      const int64_t q = min(qb + u, q1 - 1);
      v[u] = src[(int64_t)order[q] * lds + kc];
    }
#pragma unroll
    for (int u = 0; u < kDepth; ++u) {
      const int64_t q = qb + u;
      if (q < q1) {
        while (q == nextb) {
          flush(j);
          ++j;
          nextb = ptr[j + 1];
        }
        acc += v[u];
      }
    }
  }
  flush(j);
}

extern "C" int fsw_segment_sum_rows_f32(const float* src, int64_t lds, const int32_t* ptr, const int32_t* order, int64_t num_out,
                                        int64_t nnz, int S, float* out, int64_t ldo, fsw_stream_t stream) {
  FSW_REQUIRE(src && ptr && order && out, "fsw_segment_sum_rows_f32: null pointer");
  FSW_REQUIRE(S >= 1 && lds >= S && ldo >= S && num_out >= 1 && nnz >= 0,
              "fsw_segment_sum_rows_f32: need S >= 1, lds >= S, ldo >= S, num_out >= 1, nnz >= 0 (got S %d, lds %lld, ldo %lld, num_out %lld, nnz %lld)", S,
              (long long)lds, (long long)ldo, (long long)num_out, (long long)nnz);
  if (nnz == 0) return 0;
  dim3 grid((unsigned)ceil_div(nnz, kSegLen), (unsigned)ceil_div(S, 4 * kWave));
  k_segment_sum_rows<<<grid, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(src, lds, ptr, order, num_out, nnz, S, out, ldo);
  FSW_LAUNCH_CHECK();
  return 0;
}
