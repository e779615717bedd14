// Fused neighbourhood kernels, register path (1 <= in-degree <= FSW_REG_MAX_DEG).  gfx950.
//
// One wavefront handles one recipient row and one 64-slice chunk: lane = slice.  For each neighbour the
// wave reads Xp[col, k0 .. k0+63] -- one coalesced 256-byte run, the neighbour index is wave-uniform
// (scalar loads) -- so the E*S gather, which is >80 % of the algorithmic bytes, is a stream of full
// cache lines.  The neighbourhood then lives in registers: an exact-size sorting network (sortnet.h)
// replaces the reference's two E*S int64 key sorts (sp.permute / sp.get_slice_info, reference
// fsw_embedding.py:2586-2678, 2721-2758), the cumulative weights replace segcumsum
// (fsw_embedding.py:1031-1032, 2795-3012) and the readout replaces the five elementwise COO passes and the
// second segcumsum (fsw_embedding.py:1047-1105).  Rows are visited in degree order (perm / bin_start from
// fsw_graph_build) so that consecutive rows run the same specialisation.
//
//   unit weights, tau <= 1 : c_t = t/D exactly, so Delta_t depends on (D, t, slice) only; the
//                            coefficients come from a float64-evaluated table and the inner loop is
//                            gather -> min/max network -> D FMAs, no transcendental per element.
//   general weights        : (key, weight) network with the reference's pad element (x = 0, weight
//                            max(tau - m, 0), fsw_embedding.py:787-821) always present; cumulative weight
//                            and phase in float64 (v_fma_f64 is full rate per instruction class on gfx950),
//                            Delta_t = [sin(2 pi xi c_t) - sin(2 pi xi c_{t-1})] (1+xi)/(pi xi), which equals
//                            the reference's 2 w sinc(xi w) cos(pi xi (2c - w)) by sum-to-product.
#include "fsw_common.h"
#include "sortnet.h"
#include "row_pipeline.h"
#ifndef FSW_REG_PIPE_BARRIER
#define FSW_REG_PIPE_BARRIER 1   // measured: tools/exp_variants.sh
#endif

namespace fsw {

constexpr int kRowsPerBlock = 32;  // perm positions per workgroup
constexpr double kPi = 3.14159265358979323846;

__device__ __forceinline__ float mass_encode(float m, int fn) {
  // reference fsw_embedding.py:857-865
  if (fn == 1) return 2.f * (m / (sqrtf(m + 1.f) + 1.f));
  if (fn == 2) return log1pf(m);
  return m;
}

// ---- unit-weight coefficient table ------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_unit_table(const float* __restrict__ freqs, int S, int max_deg,
                                                    float* __restrict__ table, int64_t ldt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;  // D*(D-1)/2 + t
  int D = 1;
  while ((D + 1) * D / 2 <= row) ++D;
  const int t = row - D * (D - 1) / 2;
  if (k >= S || D > max_deg) return;
  const double xi = (double)freqs[k];
  const double w = 1.0 / D;
  const double c = (double)(t + 1) / D;
  const double x = xi * w;
  const double sinc = (x == 0.0) ? 1.0 : sinpi(x) / (kPi * x);
  // reference fsw_embedding.py:1047-1075 (Delta) and :1109 (1 + xi)
  table[(int64_t)row * ldt + k] = (float)((1.0 + xi) * 2.0 * w * sinc * cospi(xi * (2.0 * c - w)));
}

// ---- zero in-degree rows ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_zero_rows(const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start,
                                                   int S, float* __restrict__ out, int64_t ldo, const float* __restrict__ bias,
                                                   float out_scale, int has_mass) {
  // one wavefront per row, lanes along the columns: contiguous stores, no integer division per element
  const int p0 = bin_start[0], n0 = bin_start[1] - p0;
  const int width = S + has_mass;
  const int wave = blockIdx.x * (blockDim.x / kWave) + wave_id(), nwaves = gridDim.x * (blockDim.x / kWave);
  // four rows per step: their perm entries are loaded together, so four rows' stores are in flight instead of one row's
  for (int i = wave * 4; i < n0; i += nwaves * 4) {
    int node[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) node[u] = i + u < n0 ? perm[p0 + i + u] : -1;
    for (int c = lane_id(); c < width; c += kWave) {
      const float v = bias ? out_scale * bias[c] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (node[u] >= 0) out[(int64_t)node[u] * ldo + c] = v;
    }
  }
}

// Workgroup -> (degree bin, perm range).  Every workgroup works on rows of ONE degree, highest degrees
// first (longest rows start earliest); bin D owns ceil(count_D / kRowsPerBlock) consecutive workgroups.
// Returns false for the surplus workgroups of the upper-bound grid.
__device__ __forceinline__ bool block_range(const int32_t* __restrict__ bin_start, int& D, int& p, int& pe) {
  return find_degree_tile<kRowsPerBlock>(bin_start, 1, FSW_REG_MAX_DEG, (int)blockIdx.x, D, p, pe);   // fsw_common.h
}

// ---- unit weights ---------------------------------------------------------------------------------------
// RPW = rows per wavefront: 1 = lane is a slice of ONE row (col indices wave-uniform, scalar loads); 2 = a slice block of at
// most 32 slices (one rank's share of a slice-sharded layer, dist.py): the two halves of the wavefront work on two different
// rows of the same degree, so that every lane gathers (per-lane col loads, two 128-byte runs per gather instruction).
template <int D, int RPW = 1>
__device__ __forceinline__ void unit_run(int p, int pe, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                         const int32_t* __restrict__ perm, const float* __restrict__ Xp, int64_t ldp,
                                         const float* __restrict__ table, int64_t ldt, float* __restrict__ out, int64_t ldo,
                                         const float* __restrict__ bias, float out_scale, int has_mass, int mass_fn,
                                         float mass_scale, int kc, bool mass_lane_wave) {
  float coef[D];
  const float* tab = table + (int64_t)(D * (D - 1) / 2) * ldt + kc;
#pragma unroll
  for (int t = 0; t < D; ++t) coef[t] = out_scale * tab[(int64_t)t * ldt];
  const float b = bias ? out_scale * bias[has_mass + kc] : 0.f;
  const int nrows = pe - p;
  const int lane = lane_id();
  const int sub = RPW == 1 ? 0 : lane / (kWave / RPW);
  const int nodev = perm[p + min(lane, nrows - 1)];   // lane r: node id and CSR offset of the block's row r
  const int startv = rowptr[nodev];
  if (mass_lane_wave && lane < nrows)
    out[(int64_t)nodev * ldo] = out_scale * (mass_encode((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
  // lanes past the last slice recompute slice S-1 and store the same value to the same address (row_pipeline.h); with RPW = 2 a
  // step past the last row recomputes the last row in the same way
  float* ok = out + has_mass + kc;
  const int nsteps = (nrows + RPW - 1) / RPW;
  pipelined_rows<D, pipeline_depth<D>(), FSW_REG_PIPE_BARRIER>(
      nsteps, col, Xp + kc, ldp,
      [&](int r) {
        if constexpr (RPW == 1) return __builtin_amdgcn_readlane(startv, r);
        else return __shfl(startv, min(r * RPW + sub, nrows - 1));
      },
      [&](KeyNet<D>& net, int r) {
        sort_network<D>(net);
        float acc = b;
#pragma unroll
        for (int t = 0; t < D; ++t) acc = fmaf(coef[t], net.k[t], acc);
        if constexpr (RPW == 1) ok[(int64_t)__builtin_amdgcn_readlane(nodev, r) * ldo] = acc;
        else ok[(int64_t)__shfl(nodev, min(r * RPW + sub, nrows - 1)) * ldo] = acc;
      });
}

#define FSW_CASES_1_32(X)                                                                                              \
  X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) \
  X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

__global__ void __launch_bounds__(256) k_embed_reg_unit(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start,
                                                        const float* __restrict__ Xp, int64_t ldp, int S,
                                                        const float* __restrict__ table, int64_t ldt, float* __restrict__ out,
                                                        int64_t ldo, const float* __restrict__ bias, float out_scale,
                                                        int has_mass, int mass_fn, float mass_scale) {
  const int chunk = blockIdx.y * 4 + wave_id();
  if (chunk * kWave >= S) return;
  const int k = chunk * kWave + lane_id();
  const bool kvalid = k < S;
  const int kc = kvalid ? k : S - 1;
  int D, p, pe;
  if (!block_range(bin_start, D, p, pe)) return;
  switch (D) {
#define X(d)                                                                                                          \
  case d:                                                                                                             \
    unit_run<d>(p, pe, rowptr, col, perm, Xp, ldp, table, ldt, out, ldo, bias, out_scale, has_mass, mass_fn,          \
                mass_scale, kc, has_mass && chunk == 0);                                                              \
    break;
    FSW_CASES_1_32(X)
#undef X
    default:
      break;
  }
}

// Narrow slice blocks (S <= 64): ONE 64-slice chunk exists, so instead of three waves of every workgroup leaving at once the four
// waves split the workgroup's 32 rows; at S <= 32 two rows per wavefront step (unit_run<D, 2>).
// ROWS: rows per workgroup (a wavefront takes ROWS / 4 <= 64 of them): 128 for the two-rows-per-step form, so that a workgroup's
// start-up chain (tile search, perm -> rowptr -> col -> first gather) is paid once per 128 rows (conv_fused.hip: TR)
template <int RPW, int ROWS>
__global__ void __launch_bounds__(256) k_embed_reg_unit_narrow(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                               const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start,
                                                               const float* __restrict__ Xp, int64_t ldp, int S,
                                                               const float* __restrict__ table, int64_t ldt, float* __restrict__ out,
                                                               int64_t ldo, const float* __restrict__ bias, float out_scale,
                                                               int has_mass, int mass_fn, float mass_scale) {
  const int kc = min(lane_id() % (kWave / RPW), S - 1);
  int D, p, pe;
  if (!find_degree_tile<ROWS>(bin_start, 1, FSW_REG_MAX_DEG, (int)blockIdx.x, D, p, pe)) return;
  constexpr int kRowsPerWave = ROWS / 4;
  static_assert(kRowsPerWave <= kWave, "a lane per row of the wavefront's share");
  p += wave_id() * kRowsPerWave;
  pe = min(pe, p + kRowsPerWave);
  if (p >= pe) return;
  switch (D) {
#define X(d)                                                                                                          \
  case d:                                                                                                             \
    unit_run<d, RPW>(p, pe, rowptr, col, perm, Xp, ldp, table, ldt, out, ldo, bias, out_scale, has_mass, mass_fn,     \
                     mass_scale, kc, has_mass != 0);                                                                  \
    break;
    FSW_CASES_1_32(X)
#undef X
    default:
      break;
  }
}

// ---- general weights ------------------------------------------------------------------------------------
// sin(2 pi x) for a float64 phase x in revolutions: reduce in float64, evaluate in float32 with
// relative accuracy (sinpif on |r| <= 1).
__device__ __forceinline__ float sin2pi_rev(double x) {
  const double r = x - rint(x);
  return sinpif(2.f * (float)r);
}

template <int DEG>
__device__ __forceinline__ void weighted_run(int p, int pe, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                             const float* __restrict__ w, const int32_t* __restrict__ perm,
                                             const float* __restrict__ Xp, int64_t ldp, const float* __restrict__ freqs,
                                             float tau, float* __restrict__ out, int64_t ldo, const float* __restrict__ bias,
                                             float out_scale, int has_mass, int mass_fn, float mass_scale, int k, int kc,
                                             bool kvalid, bool mass_lane, const float* __restrict__ efeat,
                                             const float* __restrict__ Ve, int64_t ldve, int d_edge) {
  const float xif = freqs[kc];
  const double xi = (double)xif;
  const bool lin = xif < 1e-30f;  // xi == 0: sinc(0) = 1 and cos(0) = 1, Delta_t = 2 w_t
  const float scale = lin ? 2.f : (float)((1.0 + xi) / (kPi * xi));
  const float b = bias ? bias[has_mass + kc] : 0.f;
  const float b0 = bias ? bias[0] : 0.f;
  for (; p < pe; ++p) {
    const int node = perm[p];
    const int start = rowptr[node];
    PairNet<DEG + 1> net;
    double m = 0.0;
#pragma unroll
    for (int t = 0; t < DEG; ++t) {
      const int c = col[start + t];
      const float wt = w ? w[start + t] : 1.f;
      float key = Xp[(int64_t)c * ldp + kc];
      if (efeat) {   // edge features: + <e_ij, v_k[d_in:]>, the feature row is wave-uniform (reference fsw_embedding.py:934-968)
        const float* er = efeat + (int64_t)(start + t) * d_edge;
        const float* vr = Ve + (int64_t)kc * ldve;
        for (int q = 0; q < d_edge; ++q) key = fmaf(er[q], vr[q], key);
      }
      net.k[t] = key;
      net.w[t] = wt;
      m += (double)wt;
    }
    const double taud = (double)tau;
    const double denom = fmax(m, taud);
    net.k[DEG] = 0.f;                             // the reference's pad element at x = 0
    net.w[DEG] = (float)fmax(taud - m, 0.0);      // zero weight when the row is not deficient
    const double inv = 1.0 / denom;
    sort_network<DEG + 1>(net);
    double c = 0.0;
    float sprev = 0.f, acc = 0.f, acc0 = 0.f;
#pragma unroll
    for (int t = 0; t <= DEG; ++t) {
      c += (double)net.w[t];
      const float s = sin2pi_rev(xi * (c * inv));
      acc = fmaf(s - sprev, net.k[t], acc);
      acc0 = fmaf(net.w[t], net.k[t], acc0);
      sprev = s;
    }
    const float val = lin ? acc0 * (float)inv : acc;
    float* orow = out + (int64_t)node * ldo;
    if (kvalid) orow[has_mass + k] = out_scale * (scale * val + b);
    if (mass_lane) orow[0] = out_scale * (mass_encode((float)m, mass_fn) * mass_scale + b0);
  }
}

__global__ void __launch_bounds__(256) k_embed_reg_weighted(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                            const int32_t* __restrict__ bin_start, const float* __restrict__ Xp,
                                                            int64_t ldp, int S, const float* __restrict__ freqs, float tau,
                                                            float* __restrict__ out, int64_t ldo, const float* __restrict__ bias,
                                                            float out_scale, int has_mass, int mass_fn, float mass_scale,
                                                            const float* __restrict__ efeat, const float* __restrict__ Ve,
                                                            int64_t ldve, int d_edge) {
  const int chunk = blockIdx.y * 4 + wave_id();
  if (chunk * kWave >= S) return;
  const int k = chunk * kWave + lane_id();
  const bool kvalid = k < S;
  const int kc = kvalid ? k : S - 1;
  const bool mass_lane = has_mass && k == 0;
  int D, p, pe;
  if (!block_range(bin_start, D, p, pe)) return;
  switch (D) {
#define X(d)                                                                                                          \
  case d:                                                                                                             \
    weighted_run<d>(p, pe, rowptr, col, w, perm, Xp, ldp, freqs, tau, out, ldo, bias, out_scale, has_mass, mass_fn,   \
                    mass_scale, k, kc, kvalid, mass_lane, efeat, Ve, ldve, d_edge);                                   \
    break;
    FSW_CASES_1_32(X)
#undef X
    default:
      break;
  }
}

// host-side launchers used by embed_api.hip
int launch_unit_table(const float* freqs, int S, int max_deg, float* table, int64_t ldt, hipStream_t stream) {
  dim3 grid((unsigned)ceil_div(S, 256), (unsigned)(max_deg * (max_deg + 1) / 2));
  k_unit_table<<<grid, 256, 0, stream>>>(freqs, S, max_deg, table, ldt);
  FSW_LAUNCH_CHECK();
  return 0;
}

int launch_zero_rows(const fsw_embed_args& a, hipStream_t stream) {
  k_zero_rows<<<2048, 256, 0, stream>>>(a.perm, a.bin_start, a.S, a.out, a.ldo, a.bias, a.out_scale, a.has_mass);
  FSW_LAUNCH_CHECK();
  return 0;
}

int launch_embed_reg(const fsw_embed_args& a, bool unit_fast, int64_t rows_upper, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  dim3 grid((unsigned)(ceil_div(rows_upper, kRowsPerBlock) + FSW_REG_MAX_DEG), (unsigned)ceil_div(a.S, 4 * kWave));
  if (unit_fast && a.S <= kWave / 2)
    k_embed_reg_unit_narrow<2, 128><<<(unsigned)(ceil_div(rows_upper, 128) + FSW_REG_MAX_DEG), 256, 0, stream>>>(
        a.rowptr, a.col, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.unit_table, a.ldt, a.out, a.ldo, a.bias, a.out_scale, a.has_mass,
        a.mass_fn, a.mass_scale);
  else if (unit_fast && a.S <= kWave)
    k_embed_reg_unit_narrow<1, kRowsPerBlock><<<grid.x, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.unit_table, a.ldt,
                                                          a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale);
  else if (unit_fast)
    k_embed_reg_unit<<<grid, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.unit_table, a.ldt,
                                               a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale);
  else
    k_embed_reg_weighted<<<grid, 256, 0, stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.freqs, a.tau,
                                                   a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale, a.efeat, a.Ve, a.ldve,
                                                   a.d_edge);
  FSW_LAUNCH_CHECK();
  return 0;
}

}  // namespace fsw
