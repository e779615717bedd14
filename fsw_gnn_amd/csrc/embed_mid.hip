// Fused neighbourhood kernels, register path for mid-degree rows (FSW_REG_MAX_DEG < in-degree <= FSW_MID_MAX_DEG).  gfx950.
//
// Same mapping as embed_reg.hip -- one wavefront per recipient row and 64-slice chunk, lane = slice, one coalesced
// 256-byte gather of Xp[col, k0..k0+63] per neighbour, the whole neighbourhood in registers -- extended to rows of up
// to 256 neighbours: at one wave per SIMD a lane owns 512 registers, and a register-resident Batcher network costs
// ~11 min/max pairs per key at 128 keys, which the matrix-free VALU retires in about the time HBM needs to deliver
// the 256 bytes of that key's gather.  (The LDS bitonic path, which these rows took before, spends a barrier and an
// LDS round trip per network stage: 72 GB/s of gather against ~4.7 TB/s here on an RMAT graph, tools/exp_skew.py.)
//   * rows are binned by padded network size (FSW_MID_SIZES); a kernel instance sorts DP wires, wires >= D hold +inf
//     and never receive a coefficient;
//   * unit weights, tau <= 1: the coefficients (1+xi)[sin(2 pi xi (r+1)/D) - sin(2 pi xi r/D)]/(pi xi) (reference
//     fsw_embedding.py:1047-1075, 1109) come from a float64 rotation recurrence per (row, slice) -- a (D, r, slice)
//     table as on the <= 32 path would not fit -- so there is still no transcendental per element;
//   * general weights: (key, weight) network, the reference's pad element (fsw_embedding.py:787-821) at wire D,
//     cumulative weight and phase in float64 as in embed_reg.hip.
#include "fsw_common.h"
#include <algorithm>
#include <stdlib.h>
#include "sortnet.h"

#ifndef FSW_MID_ROW_BARRIER
#define FSW_MID_ROW_BARRIER 0
#endif

namespace fsw {

constexpr double kPiM = 3.14159265358979323846;

__device__ __forceinline__ float mass_encode_m(float m, int fn) {
  // reference fsw_embedding.py:857-865
  if (fn == 1) return 2.f * (m / (sqrtf(m + 1.f) + 1.f));
  if (fn == 2) return log1pf(m);
  return m;
}

__device__ __forceinline__ float sin2pi_rev_m(double x) {
  const double r = x - rint(x);
  return sinpif(2.f * (float)r);
}

template <int DP>
__global__ void __launch_bounds__(256) k_embed_mid_unit(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start,
                                                        int bin, const float* __restrict__ Xp, int64_t ldp, int S,
                                                        const float* __restrict__ freqs, float* __restrict__ out, int64_t ldo,
                                                        const float* __restrict__ bias, float out_scale, int has_mass,
                                                        int mass_fn, float mass_scale) {
  const int chunk = blockIdx.y * 4 + wave_id();
  if (chunk * kWave >= S) return;
  const int kc = min(chunk * kWave + lane_id(), S - 1);   // lanes past the last slice recompute slice S-1 (same value, same address)
  const int pbeg = bin_start[bin], pend = bin_start[bin + 1];
  const float xif = freqs[kc];
  const double xi = (double)xif;
  const bool lin = xif < 1e-30f;   // xi == 0: Delta_t = 2 w_t
  const float b = bias ? out_scale * bias[has_mass + kc] : 0.f;
  const float* xk = Xp + kc;
  // A row's descriptors arrive ahead of it: its node id (perm) is requested before the previous row's gather, its CSR range (rowptr)
  // before the previous row's sort -- read in place they were two dependent scalar round trips at the top of every row.
  int p = pbeg + blockIdx.x;
  int node = p < pend ? perm[p] : 0;
  int start = rowptr[node], end = rowptr[node + 1];
  for (; p < pend; p += gridDim.x) {
#if FSW_MID_ROW_BARRIER
    if (DP >= FSW_MID_ROW_BARRIER) __builtin_amdgcn_s_barrier();   // the four wavefronts walk the unrolled network together (instruction cache)
#endif
    const int pn = p + (int)gridDim.x;
    const int node_n = pn < pend ? perm[pn] : 0;
    const int D = end - start;   // FSW_REG_MAX_DEG < D <= DP
    KeyNet<DP> net;
#pragma unroll
    for (int t = 0; t < DP; ++t) {
      net.k[t] = __builtin_inff();
      if (t < D) net.k[t] = xk[(int64_t)col[start + t] * ldp];
    }
    const int start_n = rowptr[node_n], end_n = rowptr[node_n + 1];
    sort_network<DP>(net);
    // D in a vector register the compiler cannot prove uniform: `r < D` with a scalar D becomes one 64-bit lane mask per wire, all
    // of them computed up front and spilled (v_writelane + 2-3 v_readlane per wire); a vector compare is one instruction
    int Dv = D;
    asm volatile("" : "+v"(Dv));
    float acc = b;
    if (lin) {
      const float coef = out_scale * 2.f / (float)D;
#pragma unroll
      for (int r = 0; r < DP; ++r)
        acc = fmaf(coef, r < Dv ? net.k[r] : 0.f, acc);
    } else {
      // coefficient of rank r = B cos(2 pi xi (r + 1/2) / D): one float64 FMA per rank (UnitCoef, fsw_common.h)
      UnitCoef uc;
      uc.start(xi, D, 0);
      float a2 = 0.f;
#pragma unroll
      for (int r = 0; r < DP; ++r) a2 = fmaf(uc.next(), r < Dv ? net.k[r] : 0.f, a2);  // the recurrence itself runs unconditionally:
                                                                                     // under `if (r < D)` its four state registers
                                                                                     // were re-selected at every wire
      acc = fmaf(out_scale * uc.B, a2, acc);
    }
    float* orow = out + (int64_t)node * ldo;
    orow[has_mass + kc] = acc;
    if (has_mass && chunk == 0 && lane_id() == 0)
      orow[0] = out_scale * (mass_encode_m((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    node = node_n;
    start = start_n;
    end = end_n;
  }
}

// The same kernel with the NEXT row's gather in flight while this row is sorted (rows of 97+ neighbours): at one or two wavefronts per
// SIMD nothing else covers the 2-3 us between issuing a row's gathers and their arrival -- a quarter to a third of a row's time.
//   * the column indices arrive two rows ahead as ceil(DP / 64) coalesced VECTOR loads (lane l holds col[start + 64 q + l]) and are
//     handed to the address arithmetic by v_readlane -- D scalar loads in batches of 16 with a wait per batch were a serial chain of
//     their own;
//   * the keys of row i + 1 (the first PF wires; PF = DP except for the 256-wire network, whose line leaves 128 registers free) are
//     gathered into a second register set right before row i is sorted and copied into the network's wires at the top of the next
//     iteration (PF moves against ~25 DP network instructions).
// One wavefront per SIMD (2 DP + ~40 registers) -- the unpipelined form runs DP = 96 / 128 at two.
template <int DP, int PF>
__global__ void __launch_bounds__(256, 1) k_embed_mid_unit_pf(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start,
                                                           int bin, const float* __restrict__ Xp, int64_t ldp, int S,
                                                           const float* __restrict__ freqs, float* __restrict__ out, int64_t ldo,
                                                           const float* __restrict__ bias, float out_scale, int has_mass,
                                                           int mass_fn, float mass_scale) {
  static_assert(PF >= 0 && PF <= DP, "prefetched wires");
  constexpr int NCV = (DP + kWave - 1) / kWave;
  const int chunk = blockIdx.y * 4 + wave_id();
  if (chunk * kWave >= S) return;
  const int lane = lane_id();
  const int kc = min(chunk * kWave + lane, S - 1);
  const int pbeg = bin_start[bin], pend = bin_start[bin + 1];
  const float xif = freqs[kc];
  const double xi = (double)xif;
  const bool lin = xif < 1e-30f;
  const float b = bias ? out_scale * bias[has_mass + kc] : 0.f;
  const float* xk = Xp + kc;
  const int stride = gridDim.x;
  struct Row {
    int node, start, D;
  };
  auto describe = [&](int p) {
    Row r{-1, 0, 0};
    if (p < pend) {
      r.node = perm[p];
      r.start = rowptr[r.node];
      r.D = rowptr[r.node + 1] - r.start;
    }
    return r;
  };
  auto load_cols = [&](const Row& r, int (&cv)[NCV]) {
#pragma unroll
    for (int q = 0; q < NCV; ++q) cv[q] = col[r.start + min(q * kWave + lane, max(r.D - 1, 0))];   // wires >= D: the last neighbour again
  };
  int p = pbeg + blockIdx.x;
  Row cur = describe(p), nxt = describe(p + stride);
  int cvc[NCV], cvn[NCV];
  load_cols(cur, cvc);
  load_cols(nxt, cvn);
  // Every wire is gathered (wires >= D re-read the last neighbour's line and are overwritten by +inf): a branch per wire puts
  // every load into its own basic block, and the wait-count pass then drains the memory pipe (vmcnt(0)) at every one of them.
  float nk[PF > 0 ? PF : 1];
#pragma unroll
  for (int t = 0; t < PF; ++t) {
    nk[t] = xk[(int64_t)__builtin_amdgcn_readlane(cvc[t / kWave], t % kWave) * ldp];
    if ((t & 15) == 15) __builtin_amdgcn_sched_barrier(0);
  }
  for (; p < pend; p += stride) {
    KeyNet<DP> net;
    const int D = cur.D;
    int Dm = D;
    asm volatile("" : "+v"(Dm));                          // vector compares for the padding masks (see the readout)
#pragma unroll
    for (int t = 0; t < PF; ++t) net.k[t] = t < Dm ? nk[t] : __builtin_inff();
    // wires beyond the prefetch set (PF = 0: all of them) are gathered now.  A scheduling barrier every 16 loads: left alone the
    // scheduler computes all DP scalar addresses first and spills them (v_writelane / v_readlane, two per load)
#pragma unroll
    for (int t = PF; t < DP; ++t) {
      net.k[t] = xk[(int64_t)__builtin_amdgcn_readlane(cvc[t / kWave], t % kWave) * ldp];
      if ((t & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
    // row i + 1: keys on their way while row i is sorted; row i + 2: its column indices
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      nk[t] = xk[(int64_t)__builtin_amdgcn_readlane(cvn[t / kWave], t % kWave) * ldp];
      if ((t & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = PF; t < DP; ++t) net.k[t] = t < Dm ? net.k[t] : __builtin_inff();
    const Row nn = describe(p + 2 * stride);
#pragma unroll
    for (int q = 0; q < NCV; ++q) cvc[q] = cvn[q];
    load_cols(nn, cvn);
    sort_network<DP>(net);
    // D in a vector register the compiler cannot prove uniform: `r < D` with a scalar D becomes one 64-bit lane mask per wire, all
    // of them computed up front and spilled (v_writelane + 2-3 v_readlane per wire); a vector compare is one instruction
    int Dv = D;
    asm volatile("" : "+v"(Dv));
    float acc = b;
    if (lin) {
      const float coef = out_scale * 2.f / (float)D;
#pragma unroll
      for (int r = 0; r < DP; ++r)
        acc = fmaf(coef, r < Dv ? net.k[r] : 0.f, acc);
    } else {
      UnitCoef uc;
      uc.start(xi, D, 0);
      float a2 = 0.f;
#pragma unroll
      for (int r = 0; r < DP; ++r) a2 = fmaf(uc.next(), r < Dv ? net.k[r] : 0.f, a2);  // the recurrence itself runs unconditionally:
                                                                                     // under `if (r < D)` its four state registers
                                                                                     // were re-selected at every wire
      acc = fmaf(out_scale * uc.B, a2, acc);
    }
    float* orow = out + (int64_t)cur.node * ldo;
    orow[has_mass + kc] = acc;
    if (has_mass && chunk == 0 && lane == 0)
      orow[0] = out_scale * (mass_encode_m((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    cur = nxt;
    nxt = nn;
  }
}

// DP = padded degree + 1: wire D carries the reference's pad element
template <int DP>
__global__ void __launch_bounds__(256) k_embed_mid_weighted(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                            const int32_t* __restrict__ bin_start, int bin,
                                                            const float* __restrict__ Xp, int64_t ldp, int S,
                                                            const float* __restrict__ freqs, float tau, float* __restrict__ out,
                                                            int64_t ldo, const float* __restrict__ bias, float out_scale,
                                                            int has_mass, int mass_fn, float mass_scale,
                                                            const float* __restrict__ efeat, const float* __restrict__ Ve,
                                                            int64_t ldve, int d_edge) {
  const int chunk = blockIdx.y * 4 + wave_id();
  if (chunk * kWave >= S) return;
  const int kc = min(chunk * kWave + lane_id(), S - 1);
  const int pbeg = bin_start[bin], pend = bin_start[bin + 1];
  const float xif = freqs[kc];
  const double xi = (double)xif;
  const bool lin = xif < 1e-30f;
  const float scale = lin ? 2.f : (float)((1.0 + xi) / (kPiM * xi));
  const float b = bias ? bias[has_mass + kc] : 0.f;
  const double taud = (double)tau;
  for (int p = pbeg + blockIdx.x; p < pend; p += gridDim.x) {
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;   // D + 1 <= DP
    PairNet<DP> net;
    double m = 0.0;
#pragma unroll
    for (int t = 0; t < DP; ++t) {
      net.k[t] = __builtin_inff();
      net.w[t] = 0.f;
      if (t < D) {
        const float wt = w ? w[start + t] : 1.f;
        float key = Xp[(int64_t)col[start + t] * ldp + kc];
        if (efeat) {   // edge features: + <e_ij, v_k[d_in:]> (reference fsw_embedding.py:934-968)
          const float* er = efeat + (int64_t)(start + t) * d_edge;
          const float* vr = Ve + (int64_t)kc * ldve;
          for (int q = 0; q < d_edge; ++q) key = fmaf(er[q], vr[q], key);
        }
        net.k[t] = key;
        net.w[t] = wt;
        m += (double)wt;
      }
    }
    const float padw = (float)fmax(taud - m, 0.0);   // zero weight when the row is not deficient
#pragma unroll
    for (int t = FSW_REG_MAX_DEG + 1; t < DP; ++t)
      if (t == D) {
        net.k[t] = 0.f;                              // the reference's pad element at x = 0
        net.w[t] = padw;
      }
    const double inv = 1.0 / fmax(m, taud);
    sort_network<DP>(net);
    double c = 0.0;
    float sprev = 0.f, acc = 0.f, acc0 = 0.f;
#pragma unroll
    for (int t = 0; t < DP; ++t) {
      if (t <= D) {
        c += (double)net.w[t];
        const float s = sin2pi_rev_m(xi * (c * inv));
        acc = fmaf(s - sprev, net.k[t], acc);
        acc0 = fmaf(net.w[t], net.k[t], acc0);
        sprev = s;
      }
    }
    const float val = lin ? acc0 * (float)inv : acc;
    float* orow = out + (int64_t)node * ldo;
    orow[has_mass + kc] = out_scale * (scale * val + b);
    if (has_mass && chunk == 0 && lane_id() == 0)
      orow[0] = out_scale * (mass_encode_m((float)m, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
  }
}

// The instantiations are split over three translation units (FSW_MID_PART = 0, 1, 2; see the Makefile): the
// 256-wire network alone takes minutes to compile.
#ifndef FSW_MID_PART
#error "compile with -DFSW_MID_PART=0|1|2"
#endif
int launch_mid_unit_small(const fsw_embed_args& a, dim3 grid, hipStream_t stream);
int launch_mid_unit_large(const fsw_embed_args& a, dim3 grid, hipStream_t stream);
int launch_mid_weighted(const fsw_embed_args& a, dim3 grid, hipStream_t stream);
int launch_embed_mid_split(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream);   // embed_hub.hip
int launch_embed_mid_lds(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream);     // embed_hub.hip

#define FSW_MID_UNIT(i, DP)                                                                                               \
  if (bin_rows_or(a, FSW_BIN_MID0 + i, FSW_BIN_MID0 + i, 1) > 0)                                                            \
  k_embed_mid_unit<DP><<<grid, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, FSW_BIN_MID0 + i, a.Xp, a.ldp, a.S,  \
                                                 a.freqs, a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn,       \
                                                 a.mass_scale);                                                           \
  FSW_LAUNCH_CHECK()
#ifndef FSW_MID_PF_MIN
#define FSW_MID_PF_MIN 192    // networks of at least this many wires prefetch the next row's keys (k_embed_mid_unit_pf<DP, PF > 0>)
#endif
#ifndef FSW_MID_VCOL_MIN
#define FSW_MID_VCOL_MIN 1000 // ... of at least this many: vector column loads + branch-free gathers, no key prefetch (PF = 0)
#endif
#define FSW_MID_UNIT_PF(i, DP, PF)                                                                                        \
  if (bin_rows_or(a, FSW_BIN_MID0 + i, FSW_BIN_MID0 + i, 1) > 0) {                                                          \
    if constexpr (DP >= FSW_MID_PF_MIN || DP >= FSW_MID_VCOL_MIN)                                                         \
      k_embed_mid_unit_pf<DP, (DP >= FSW_MID_PF_MIN ? PF : 0)><<<grid, 256, 0, stream>>>(                                 \
          a.rowptr, a.col, a.perm, a.bin_start, FSW_BIN_MID0 + i, a.Xp, a.ldp, a.S, a.freqs, a.out, a.ldo, a.bias,        \
          a.out_scale, a.has_mass, a.mass_fn, a.mass_scale);                                                              \
    else                                                                                                                  \
      k_embed_mid_unit<DP><<<grid, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, FSW_BIN_MID0 + i, a.Xp, a.ldp,  \
                                                     a.S, a.freqs, a.out, a.ldo, a.bias, a.out_scale, a.has_mass,         \
                                                     a.mass_fn, a.mass_scale);                                            \
  }                                                                                                                       \
  FSW_LAUNCH_CHECK()
#define FSW_MID_WEIGHTED(i, DP)                                                                                           \
  if (bin_rows_or(a, FSW_BIN_MID0 + i, FSW_BIN_MID0 + i, 1) > 0)                                                            \
  k_embed_mid_weighted<DP + 1><<<grid, 256, 0, stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, FSW_BIN_MID0 + i, a.Xp,  \
                                                         a.ldp, a.S, a.freqs, a.tau, a.out, a.ldo, a.bias, a.out_scale,   \
                                                         a.has_mass, a.mass_fn, a.mass_scale, a.efeat, a.Ve, a.ldve,      \
                                                         a.d_edge);                                                       \
  FSW_LAUNCH_CHECK()

#if FSW_MID_PART == 0
int launch_mid_unit_small(const fsw_embed_args& a, dim3 grid, hipStream_t stream) {
  FSW_MID_UNIT_PF(0, 40, 40); FSW_MID_UNIT_PF(1, 48, 48); FSW_MID_UNIT_PF(2, 64, 64); FSW_MID_UNIT_PF(3, 80, 80); FSW_MID_UNIT_PF(4, 96, 96);
  FSW_MID_UNIT_PF(5, 128, 128);
  return 0;
}

// rows_upper: upper bound of the rows in the mid bins (the per-bin counts stay on the device); surplus workgroups of
// an instance whose bin is short or empty leave at once.
int launch_embed_mid(const fsw_embed_args& a, bool unit_fast, int64_t rows_upper, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  dim3 grid((unsigned)std::min<int64_t>(rows_upper, 8192), (unsigned)ceil_div(a.S, 4 * kWave));
  int rc;
  if (unit_fast) {
    if ((rc = launch_mid_unit_small(a, grid, stream))) return rc;
    // 129..256 neighbours: FSW_MID_SPLIT=1 in the environment runs the line split over four lanes instead (embed_hub.hip:
    // k_embed_rowlines<M, 4>, three waves per SIMD).  Measured SLOWER on the populated bin of the RMAT graphs (129..160 neighbours:
    // 373 against 524 G keys/s; LL = 2 / 8: 379 / 392): four 64-byte pieces per gather instruction instead of one 256-byte run
    // cost more than the occupancy gains -- kept for comparison (tools/exp_skew.py --fine)
    static const int split = [] { const char* e = getenv("FSW_MID_SPLIT"); return e ? atoi(e) : 0; }();
    if (split == 2) return launch_embed_mid_lds(a, rows_upper, stream);   // whole-row gathers through LDS (k_embed_mid_lds)
    return split ? launch_embed_mid_split(a, rows_upper, stream) : launch_mid_unit_large(a, grid, stream);
  }
  return launch_mid_weighted(a, grid, stream);   // bins above FSW_MID_MAX_DEG_WEIGHTED go to the wave-sort path (embed_wsort.hip)
}
#elif FSW_MID_PART == 1
int launch_mid_unit_large(const fsw_embed_args& a, dim3 grid, hipStream_t stream) {
  FSW_MID_UNIT_PF(6, 160, 160); FSW_MID_UNIT_PF(7, 192, 192); FSW_MID_UNIT_PF(8, 256, 128);
  return 0;
}
#else
int launch_mid_weighted(const fsw_embed_args& a, dim3 grid, hipStream_t stream) {
  // without edge features the bins from weighted_hub_first_mid_bin() on run on k_embed_hub_w (embed_wsort.hip: launch_embed_lds)
  const int end = a.efeat ? 6 : weighted_hub_first_mid_bin();
  if (end > 0) { FSW_MID_WEIGHTED(0, 40); }
  if (end > 1) { FSW_MID_WEIGHTED(1, 48); }
  if (end > 2) { FSW_MID_WEIGHTED(2, 64); }
  if (end > 3) { FSW_MID_WEIGHTED(3, 80); }
  if (end > 4) { FSW_MID_WEIGHTED(4, 96); }
  if (end > 5) { FSW_MID_WEIGHTED(5, 128); }
  return 0;
}
#endif
#undef FSW_MID_UNIT
#undef FSW_MID_UNIT_PF
#undef FSW_MID_WEIGHTED

}  // namespace fsw
