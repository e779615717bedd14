// Backward of the neighbourhood embedding for rows above FSW_REG_MAX_DEG, wave-sort form.  gfx950.
//
//   out[i, k] = sum_r C_k(r) x_(r)           ->   gXp[col_t, k] += g[i, k] C_k(rank_k(t)),
//   C = F(c_r) - F(c_r - w_r)                      gfreq[k]     += g[i, k] sum_r dC/dxi x_(r)
// with F(xi; c) = (1 + xi) sin(2 pi xi c)/(pi xi) (reference
// fsw_embedding.py:1047-1109 differentiated by hand, verified against the reference's autograd in
// tests/test_hip_parity.py).  Same structure as the forward (embed_wsort.hip): the neighbourhood transposed through
// LDS, one wavefront per (row, slice) line held across its registers -- here every key carries its ELEMENT INDEX as
// payload.  After the sort lane l owns ranks l*M..; it evaluates F and dF/dxi (unit weights: sin and cos of the
// rank angle by a float64 rotation; general weights: cumulative weight by in-lane prefix + float64 wave scan, then
// sincospi), reduces gfreq over the wave and drops g*C into the element's ORIGINAL position of the same LDS line.
// After a barrier all threads add the tile to gXp with the lanes along the slice axis (runs of SC floats per
// neighbour: the shape float atomics run fastest in), or store it to gkey (edge features).
// Rows above FSW_LDS_MAX_DEG: one wave per line in a global scratch region, chunks sorted in registers and merged by
// sweeps as in the forward; the contribution is scattered to element order inside the wave's scratch and then added
// with one 4-byte atomic per neighbour (these rows are few).
#include <algorithm>
#include <stdlib.h>
#include "fsw_common.h"
#include "sortnet.h"
#include "wave_sort.h"

namespace fsw {

constexpr double kPiB = 3.14159265358979323846;
// LDS per workgroup: two workgroups per CU, except where the registers of a wave (one wave per SIMD) already allow only one
// workgroup per CU -- those instances take twice the lines per slice group, i.e. twice as long atomic runs per neighbour
template <int M, bool WEIGHTED>
constexpr int wb_lds_bytes() { return (WEIGHTED ? M >= 64 : M >= 32) ? 140 * 1024 : 69 * 1024; }
constexpr int kWbSplitY = 4;
#ifndef FSW_WSB_ABL
#define FSW_WSB_ABL 0   // timing experiments on k_embed_wsort_bwd: 1 no atomics, 2 no sort, 4 no coefficient walk
#endif

__device__ __forceinline__ float wave_sum_b(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_sum_b64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_exclusive_scan_b64(double v) {
  double inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const double t = __shfl_up(inc, off);
    if (lane_id() >= off) inc += t;
  }
  return inc - v;
}

// F and dF/dxi at normalised cumulative weight c, given sin and cos of 2 pi xi c; series for tiny phases (the two terms
// of dF cancel there); xi == 0: F = 2 c, dF = 2 c.  FCoef holds the per-slice factors so that no division is left per element.
struct FCoef {
  double xi, a1, a2, a3;   // a1 = (1 + xi)/(pi xi), a2 = 1/(pi xi^2), a3 = 2 (1 + xi)/xi
  __device__ __forceinline__ explicit FCoef(double x) : xi(x) {
    const double r = x > 0.0 ? 1.0 / x : 0.0;
    a1 = (1.0 + x) * r * (1.0 / kPiB);
    a2 = r * r * (1.0 / kPiB);
    a3 = 2.0 * (1.0 + x) * r;
  }
};
__device__ __forceinline__ void F_dF_sc(const FCoef& f, double c, double s, double co, double& F, double& dF) {
  const double x = 2.0 * kPiB * f.xi * c;
  if (x < 1e-4) {
    const double q = 1.0 - x * x * (1.0 / 6.0);
    F = (1.0 + f.xi) * 2.0 * c * q;
    dF = 2.0 * c * q - (1.0 + f.xi) * 2.0 * c * (2.0 * kPiB * c) * (2.0 * kPiB * c) * f.xi * (1.0 / 3.0);
  } else {
    F = f.a1 * s;
    dF = fma(f.a3 * c, co, -(f.a2 * s));
  }
}
__device__ __forceinline__ void F_dF(const FCoef& f, double c, double& F, double& dF) {
  const double ph = f.xi * c;
  double s, co;
  sincospi(2.0 * (ph - rint(ph)), &s, &co);
  F_dF_sc(f, c, s, co, F, dF);
}

// Walks the sorted line held by `ln` (ranks r0 + j, j < M, of a line of Dtot elements of which the first D are
// neighbours): calls emit(element index, g * C) for every neighbour and returns this lane's share of the slice's gfreq.
// WEIGHTED: weight_of(index) returns the element's weight, `cbase` is the cumulative weight before this lane's first
// element on entry (only used when WEIGHTED).
template <int M, bool WEIGHTED, class WeightFn, class EmitFn>
__device__ __forceinline__ float walk_line(const WaveLine64<M>& ln, int r0, int D, int Dtot, double xi, double inv, float gi,
                                           double cbase, WeightFn weight_of, EmitFn emit) {
  float gf = 0.f;
  double Fp, dFp;
  const FCoef fc(xi);
  if constexpr (!WEIGHTED) {
    const double step = xi * inv;   // revolutions per rank
    double sd, cd, s, c;
    sincospi(2.0 * (step - rint(step)), &sd, &cd);
    const double x0 = step * (double)r0;
    sincospi(2.0 * (x0 - rint(x0)), &s, &c);
    F_dF_sc(fc, (double)r0 * inv, s, c, Fp, dFp);
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const double sn = fma(s, cd, c * sd), cn = fma(c, cd, -(s * sd));
      s = sn;
      c = cn;
      double F, dF;
      F_dF_sc(fc, (double)min(r0 + j + 1, D) * inv, s, c, F, dF);
      if (r0 + j < D) {
        emit(ln.index(j), gi * (float)(F - Fp));
        gf = fmaf(gi * (float)(dF - dFp), ln.key(j), gf);
      }
      Fp = F;
      dFp = dF;
    }
  } else {
    double c = cbase;
    F_dF(fc, c * inv, Fp, dFp);
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int id = ln.index(j);
      const bool valid = r0 + j < Dtot;
      c += valid ? (double)weight_of(id) : 0.0;
      double F, dF;
      F_dF(fc, c * inv, F, dF);
      if (valid) {
        if (id < D) emit(id, gi * (float)(F - Fp));   // the pad element (id == D) has no source row
        gf = fmaf(gi * (float)(dF - dFp), ln.key(j), gf);
      }
      Fp = F;
      dFp = dF;
    }
  }
  return gf;
}

template <int M, bool WEIGHTED>
__global__ void __launch_bounds__(256, (M <= 16 ? 2 : 1)) k_embed_wsort_bwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                         const int32_t* __restrict__ bin_start, int bin_lo, int bin_hi,
                                                         const float* __restrict__ Xp, int64_t ldp, int S,
                                                         const float* __restrict__ freqs, float tau, const float* __restrict__ g,
                                                         int64_t ldg, int gcol0, float out_scale, float* __restrict__ gXp,
                                                         int64_t ldgp, float* __restrict__ gfreq, const float* __restrict__ efeat,
                                                         const float* __restrict__ Ve, int64_t ldve, int d_edge,
                                                         float* __restrict__ gkey, int64_t ldk) {
  constexpr int CAP = M * kWave;
  constexpr int LINE = CAP + kWave + 1;   // one pad per M elements, odd stride
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wrow = smem;                                   // [LINE] weights of the row, padded like a line (weighted only)
  float* tile = smem + (WEIGHTED ? LINE : 0);           // [SC][LINE]
  constexpr int kTileLines = (wb_lds_bytes<M, WEIGHTED>() / 4 - (WEIGHTED ? LINE : 0)) / LINE;
  static_assert(kTileLines >= 1, "a line must fit the LDS budget");
  constexpr int SC = kTileLines >= 64 ? 64 : (kTileLines >= 4 ? (kTileLines & ~3) : kTileLines);
  __shared__ double msum[4];
  const int pbeg = bin_start[bin_lo], pend = bin_start[bin_hi + 1];
  const int lane = lane_id(), wv = wave_id();

  for (int p = pbeg + blockIdx.x; p < pend; p += gridDim.x) {
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    const int Dtot = WEIGHTED ? D + 1 : D;
    double m = (double)D;
    if constexpr (WEIGHTED) {
      double part = 0.0;
      for (int t = threadIdx.x; t < D; t += blockDim.x) {
        const float wt = w ? w[start + t] : 1.f;
        wrow[t + t / M] = wt;
        part += (double)wt;
      }
      part = wave_sum_b64(part);
      if (lane == 0) msum[wv] = part;
      __syncthreads();
      m = msum[0] + msum[1] + msum[2] + msum[3];
      if (threadIdx.x == 0) wrow[D + D / M] = (float)fmax((double)tau - m, 0.0);
      __syncthreads();
    }
    const double inv = 1.0 / (WEIGHTED ? fmax(m, (double)tau) : m);

    const int ngroups = (S + SC - 1) / SC;
    for (int grp = blockIdx.y; grp < ngroups; grp += gridDim.y) {
      const int k0 = grp * SC;
      constexpr int kGatherDepth = 8;
      const int total = Dtot * SC;
      for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * kGatherDepth) {
        float key[kGatherDepth];
        int pos[kGatherDepth];
#pragma unroll
        for (int u = 0; u < kGatherDepth; ++u) {
          const int i = min(i0 + u * (int)blockDim.x, total - 1);
          const int kk = i % SC, t = i / SC;
          pos[u] = kk * LINE + t + t / M;
          key[u] = 0.f;                                  // t == D (weighted): the reference's pad element at x = 0
          if (t < D) {
            const int kcl = min(k0 + kk, S - 1);
            key[u] = Xp[(int64_t)col[start + t] * ldp + kcl];
            if (efeat) {
              const float* er = efeat + (int64_t)(start + t) * d_edge;
              const float* vr = Ve + (int64_t)kcl * ldve;
              for (int q = 0; q < d_edge; ++q) key[u] = fmaf(er[q], vr[q], key[u]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kGatherDepth; ++u) tile[pos[u]] = key[u];
      }
      __syncthreads();
      for (int kk = wv; kk < SC; kk += 4) {
        const int k = k0 + kk;
        if (k >= S) break;
        WaveLine64<M> ln;
        float* line = tile + kk * LINE;
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const int t = lane * M + j;
          ln.e[j] = pack_key_index(t < Dtot ? line[lane * (M + 1) + j] : __builtin_inff(), t);
        }
        if (!(FSW_WSB_ABL & 2)) ln.sort();
        const double xi = (double)freqs[k];
        const float gi = out_scale * g[(int64_t)node * ldg + gcol0 + k];
        double cbase = 0.0;
        if constexpr (WEIGHTED) {
          double part = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) {
            const int id = ln.index(j);
            part += (lane * M + j < Dtot) ? (double)wrow[id + id / M] : 0.0;
          }
          cbase = wave_exclusive_scan_b64(part);
        }
        float gf = (FSW_WSB_ABL & 4) ? 0.f : walk_line<M, WEIGHTED>(
            ln, lane * M, D, Dtot, xi, inv, gi, cbase, [&](int id) { return wrow[id + id / M]; },
            [&](int id, float v) { line[id + id / M] = v; });   // every lane has read its keys: the line is free
        gf = wave_sum_b(gf);
        if (lane == 0 && gfreq) atomicAdd(gfreq + k, gf);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < D * SC; i += blockDim.x) {
        const int kk = i % SC, t = i / SC;
        if (k0 + kk < S) {
          const float v = tile[kk * LINE + t + t / M];
          if (gkey) gkey[(int64_t)(start + t) * ldk + k0 + kk] = v;   // edge features: per-entry key gradient
          else if (!(FSW_WSB_ABL & 1)) atomicAdd(gXp + (int64_t)col[start + t] * ldgp + k0 + kk, v);
        }
      }
      __syncthreads();
    }
  }
}

// ---- 129 .. 2048 neighbours, unit weights, key gradients STORED (the store-and-sum backward): the forward's quad structure ----
// k_embed_wsort_bwd above stages a [slices][elements] tile of up to 140 KB in LDS (one workgroup of four wavefronts per CU) and
// gathers 4 bytes per lane; on the RMAT-20 graph its three instances took 33 ms of a 75 ms training step, 7-13 x their forward
// counterparts.  Here, as in k_embed_hub_quad (embed_hub.hip): the four wavefronts of a workgroup own four ADJACENT slices of one
// row (S % 4 == 0), wavefront w gathers a quarter of the row as float4 = slices k0 .. k0 + 3 and the lines change hands through
// LDS; every wavefront sorts its line as packed (key, element index) words in registers, walks it (walk_line: coefficients and
// their frequency derivatives by float64 rotation) and drops g * C into the element's place of ITS line of the same LDS buffer;
// the workgroup then stores the four slices of every element as ONE 16-byte piece of gkey[entry, k0 .. k0 + 3].  4 .. 32 KB of LDS
// and two to four waves per SIMD.  The atomics form (gkey == NULL), general weights and edge features stay on the kernel above.
template <int M>
__global__ void __launch_bounds__(256, (M >= 32 ? 2 : (M >= 16 ? 3 : 4))) k_embed_quad_bwd(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ bin_start, int bin_lo, int bin_hi, const float* __restrict__ Xp, int64_t ldp, int S,
    const float* __restrict__ freqs, const float* __restrict__ g, int64_t ldg, int gcol0, float out_scale, float* __restrict__ gfreq,
    float* __restrict__ gkey, int64_t ldk) {
  constexpr int CAP = M * kWave;
  constexpr int NQ = M >= 4 ? M / 4 : 1;            // float4 gathers per lane
  static_assert(M % 4 == 0, "keys per lane: a multiple of 4");
  __shared__ __attribute__((aligned(16))) float xq[4][CAP];
  const int pbeg = bin_start[bin_lo], nrows = bin_start[bin_hi + 1] - pbeg;
  const int lane = lane_id(), w = wave_id();
  const int xcd = blockIdx.x & 7;
  for (int64_t vb = blockIdx.x;; vb += gridDim.x) {
    const int64_t i = (vb >> 3) * 4;              // the workgroup's first slice; S % 4 == 0: four slices of one row
    const int64_t rl = i / S;
    const int k0 = (int)(i - rl * S);
    const int64_t r = rl * 8 + xcd;
    if (r >= nrows) return;                       // the whole workgroup
    const int node = perm[pbeg + r];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    const int32_t* colrow = col + start;
    const float* xr = Xp + k0;
    int c[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) c[q] = colrow[min((q * 4 + w) * kWave + lane, D - 1)];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int t = (q * 4 + w) * kWave + lane;
      float4 v = *reinterpret_cast<const float4*>(xr + (int64_t)c[q] * ldp);
      if (t >= D) v = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());
      xq[0][t] = v.x;
      xq[1][t] = v.y;
      xq[2][t] = v.z;
      xq[3][t] = v.w;
    }
    __syncthreads();
    WaveLine64<M> ln;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int t = j * kWave + lane;               // striped: conflict-free reads; the index travels with the key
      ln.e[j] = pack_key_index(xq[w][t], t);
    }
    ln.sort();
    const int k = k0 + w;
    const double xi = (double)freqs[k];
    const float gi = out_scale * g[(int64_t)node * ldg + gcol0 + k];
    // every lane of THIS wavefront has read the line (the sort needed all keys): its places may take the results
    float gf = walk_line<M, false>(
        ln, lane * M, D, D, xi, 1.0 / (double)D, gi, 0.0, [](int) { return 0.f; }, [&](int id, float v) { xq[w][id] = v; });
    gf = wave_sum_b(gf);
    if (lane == 0 && gfreq) atomicAdd(gfreq + k, gf);
    __syncthreads();
    for (int e = threadIdx.x; e < D; e += blockDim.x)
      *reinterpret_cast<float4*>(gkey + (int64_t)(start + e) * ldk + k0) = make_float4(xq[0][e], xq[1][e], xq[2][e], xq[3][e]);
    __syncthreads();                               // the next row overwrites the buffer
  }
}

template <int M>
static int launch_quad_bwd(const fsw_embed_args& a, int bin_lo, int bin_hi, int64_t rows_upper, const float* g, int64_t ldg, float* gfreq,
                           float* gkey, int64_t ldk, hipStream_t stream) {
  rows_upper = bin_rows_or(a, bin_lo, bin_hi, rows_upper);
  if (rows_upper <= 0) return 0;
  const int64_t nvirtual = ceil_div(rows_upper, 8) * (a.S / 4) * 8;
  k_embed_quad_bwd<M><<<(unsigned)std::min<int64_t>(nvirtual, 1ll << 20), 256, 0, stream>>>(
      a.rowptr, a.col, a.perm, a.bin_start, bin_lo, bin_hi, a.Xp, a.ldp, a.S, a.freqs, g, ldg, a.has_mass, a.out_scale, gfreq, gkey, ldk);
  FSW_LAUNCH_CHECK();
  return 0;
}

// ---- rows above FSW_LDS_MAX_DEG ------------------------------------------------------------------------------------------
// fences below: workgroup scope orders a wave's scratch stores before its own later loads (same CU, same L1; see embed_wsort.hip)
constexpr int kSweepDepthB = 4;

__device__ __forceinline__ void sweep_pairs_b(unsigned long long* __restrict__ se, int Dp, int size, int st, bool flip) {
  const int npairs = Dp >> 1;
  const int half = size >> 1;
  for (int i0 = lane_id(); i0 < npairs; i0 += kWave * kSweepDepthB) {
    int ia[kSweepDepthB], ib[kSweepDepthB];
    unsigned long long a[kSweepDepthB], b[kSweepDepthB];
#pragma unroll
    for (int u = 0; u < kSweepDepthB; ++u) {
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
      a[u] = se[ia[u]];
      b[u] = se[ib[u]];
    }
#pragma unroll
    for (int u = 0; u < kSweepDepthB; ++u) {
      if (b[u] < a[u]) {   // packed words: by key, equal keys by element index
        se[ia[u]] = b[u];
        se[ib[u]] = a[u];
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

template <int M, bool WEIGHTED>
__global__ void __launch_bounds__(256) k_embed_wsort_global_bwd(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const float* __restrict__ w, const int32_t* __restrict__ perm,
                                                                const int32_t* __restrict__ bin_start, const float* __restrict__ Xp,
                                                                int64_t ldp, int S, const float* __restrict__ freqs, float tau,
                                                                const float* __restrict__ g, int64_t ldg, int gcol0, float out_scale,
                                                                float* __restrict__ gXp, int64_t ldgp, float* __restrict__ gfreq,
                                                                const float* __restrict__ efeat, const float* __restrict__ Ve,
                                                                int64_t ldve, int d_edge, float* __restrict__ gkey, int64_t ldk,
                                                                char* __restrict__ scratch, int64_t wave_bytes, int bin_lo, int bin_hi) {
  constexpr int CAP = M * kWave;
  const int lane = lane_id();
  const int gw = blockIdx.x * 4 + wave_id(), nwaves = gridDim.x * 4;
  unsigned long long* se = reinterpret_cast<unsigned long long*>(scratch + (int64_t)gw * wave_bytes);   // packed (key, index) words
  float* sc = reinterpret_cast<float*>(se + wave_bytes / 12);                                           // contributions, element order
  const int pbeg = bin_start[bin_lo], pend = bin_start[bin_hi + 1];   // rows above FSW_LDS_MAX_DEG, one launch per degree bin
  const int64_t nlines = (int64_t)(pend - pbeg) * S;
  for (int64_t ln_id = gw; ln_id < nlines; ln_id += nwaves) {
    const int p = pbeg + (int)(ln_id / S), k = (int)(ln_id % S);
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    const int Dtot = WEIGHTED ? D + 1 : D;
    const int Dp = (int)pow2ceil((uint32_t)Dtot);
    double m = (double)D;
    float padw = 0.f;
    if constexpr (WEIGHTED) {
      double part = 0.0;
      for (int t = lane; t < D; t += kWave) part += (double)(w ? w[start + t] : 1.f);
      m = wave_sum_b64(part);
      padw = (float)fmax((double)tau - m, 0.0);
    }
    const double inv = 1.0 / (WEIGHTED ? fmax(m, (double)tau) : m);
    const double xi = (double)freqs[k];
    const float gi = out_scale * g[(int64_t)node * ldg + gcol0 + k];
    for (int c0 = 0; c0 < Dp; c0 += CAP) {
      WaveLine64<M> ln;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const int t = c0 + lane * M + j;
        float key = __builtin_inff();
        if (t < D) {
          key = Xp[(int64_t)col[start + t] * ldp + k];
          if (efeat) {
            const float* er = efeat + (int64_t)(start + t) * d_edge;
            const float* vr = Ve + (int64_t)k * ldve;
            for (int q = 0; q < d_edge; ++q) key = fmaf(er[q], vr[q], key);
          }
        } else if (WEIGHTED && t == D) {
          key = 0.f;
        }
        ln.e[j] = pack_key_index(key, t);
      }
      ln.sort();
#pragma unroll
      for (int j = 0; j < M; ++j) se[c0 + lane * M + j] = ln.e[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    float gf = 0.f;
    double carry = 0.0;
    auto weight_of = [&](int id) { return id == D ? padw : (w ? w[start + id] : 1.f); };
    for (int size = 2 * CAP; size <= Dp; size <<= 1) {
      sweep_pairs_b(se, Dp, size, 0, true);
      for (int st = size >> 2; st >= CAP; st >>= 1) sweep_pairs_b(se, Dp, size, st, false);
      const bool last = size == Dp;
      for (int c0 = 0; c0 < Dp; c0 += CAP) {
        if (last && c0 >= Dtot) break;
        WaveLine64<M> ln;
#pragma unroll
        for (int j = 0; j < M; ++j) ln.e[j] = se[c0 + lane * M + j];
        ln.merge_chunk();
        if (!last) {
#pragma unroll
          for (int j = 0; j < M; ++j) se[c0 + lane * M + j] = ln.e[j];
          continue;
        }
        double cbase = 0.0;
        if constexpr (WEIGHTED) {
          double part = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) part += (c0 + lane * M + j < Dtot) ? (double)weight_of(ln.index(j)) : 0.0;
          cbase = carry + wave_exclusive_scan_b64(part);
          carry += wave_sum_b64(part);
        }
        gf += walk_line<M, WEIGHTED>(ln, c0 + lane * M, D, Dtot, xi, inv, gi, cbase, weight_of, [&](int id, float v) { sc[id] = v; });
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    gf = wave_sum_b(gf);
    if (lane == 0 && gfreq) atomicAdd(gfreq + k, gf);
    for (int t = lane; t < D; t += kWave) {
      const float v = sc[t];
      if (gkey) gkey[(int64_t)(start + t) * ldk + k] = v;
      else atomicAdd(gXp + (int64_t)col[start + t] * ldgp + k, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the next line reuses the scratch
  }
}

// Scratch of the rows above FSW_LDS_MAX_DEG (forward: 4 or 8 bytes per element of the padded line and wave, backward 12):
// room for up to 2048 resident waves, capped at 2 GiB -- fewer waves then share the lines of a very long row.
size_t embed_global_scratch_bytes(int64_t max_degree) {
  const size_t wave_bytes = (size_t)pow2ceil((uint32_t)(max_degree + 1)) * 12;
  const size_t cap = (size_t)2 << 30;   // (6 GB = 2048 wavefronts on a 150 000-neighbour line measured slower than 680: 70.7 vs 61.6 ms)
  size_t waves = std::min<size_t>(2048, cap / wave_bytes);
  waves = std::max<size_t>(waves & ~(size_t)3, 4);
  return waves * wave_bytes;
}

template <int M, bool WEIGHTED>
static int launch_wsort_bwd(const fsw_embed_args& a, int bin_lo, int bin_hi, int64_t rows_upper, const float* g, int64_t ldg,
                            float* gXp, int64_t ldgp, float* gfreq, float* gkey, int64_t ldk, hipStream_t stream) {
  FSW_SET_MAX_LDS_ONCE((k_embed_wsort_bwd<M, WEIGHTED>), (wb_lds_bytes<M, WEIGHTED>()));
  dim3 grid((unsigned)std::min<int64_t>(rows_upper, 1 << 14), kWbSplitY);
  k_embed_wsort_bwd<M, WEIGHTED><<<grid, 256, wb_lds_bytes<M, WEIGHTED>(), stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, bin_lo, bin_hi, a.Xp, a.ldp,
                                                                   a.S, a.freqs, a.tau, g, ldg, a.has_mass, a.out_scale, gXp, ldgp, gfreq,
                                                                   a.efeat, a.Ve, a.ldve, a.d_edge, gkey, ldk);
  FSW_LAUNCH_CHECK();
  return 0;
}

int launch_embed_mid_bwd(const fsw_embed_args& a, int64_t rows_upper, const float* g, int64_t ldg, float* gXp, int64_t ldgp,
                         float* gfreq, float* gkey, int64_t ldk, hipStream_t stream);

// rows of FSW_REG_MAX_DEG < degree <= FSW_LDS_MAX_DEG (rows_upper bounds their number) or, global == true, above
int launch_embed_long_bwd(const fsw_embed_args& a, bool global, int64_t rows_upper, const float* g, int64_t ldg, float* gXp,
                          int64_t ldgp, float* gfreq, float* gkey, int64_t ldk, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  const bool unit = (a.w == nullptr) && (a.tau <= 1.f);
  int rc;
  if (global) {
    FSW_REQUIRE(a.scratch && a.max_degree > FSW_LDS_MAX_DEG, "fsw_embed_backward: rows above FSW_LDS_MAX_DEG need the scratch buffer and max_degree");
    char* scratch = reinterpret_cast<char*>(a.scratch);
    // one launch per degree bin, the scratch line sized by the bin's own longest row (see launch_embed_global, embed_wsort.hip)
    for (int bin = FSW_BIN_HUB0; bin <= FSW_BIN_GLOBAL; ++bin) {
      const int64_t rows = bin_rows_or(a, bin, bin, rows_upper);
      if (rows <= 0) continue;
      const int64_t bin_max = bin == FSW_BIN_GLOBAL ? a.max_degree : std::min<int64_t>(a.max_degree, (int64_t)4096 << (bin - FSW_BIN_HUB0));
      if (bin > FSW_BIN_HUB0 && bin_max <= ((int64_t)2048 << (bin - FSW_BIN_HUB0))) continue;   // no row of the graph reaches this bin
      const int64_t wave_bytes = (int64_t)pow2ceil((uint32_t)(bin_max + 1)) * 12;
      int64_t nwaves = std::min<int64_t>((int64_t)a.scratch_bytes / wave_bytes, 2048);
      nwaves = std::min<int64_t>(nwaves, ceil_div(rows * a.S, 4) * 4) & ~(int64_t)3;
      FSW_REQUIRE(nwaves >= 4, "fsw_embed_backward: scratch buffer too small (need fsw_embed_scratch_bytes(max_degree))");
      if (unit)
        k_embed_wsort_global_bwd<32, false><<<(unsigned)(nwaves / 4), 256, 0, stream>>>(
            a.rowptr, a.col, a.w, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.freqs, a.tau, g, ldg, a.has_mass, a.out_scale, gXp, ldgp,
            gfreq, a.efeat, a.Ve, a.ldve, a.d_edge, gkey, ldk, scratch, wave_bytes, bin, bin);
      else
        k_embed_wsort_global_bwd<32, true><<<(unsigned)(nwaves / 4), 256, 0, stream>>>(
            a.rowptr, a.col, a.w, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.freqs, a.tau, g, ldg, a.has_mass, a.out_scale, gXp, ldgp,
            gfreq, a.efeat, a.Ve, a.ldve, a.d_edge, gkey, ldk, scratch, wave_bytes, bin, bin);
      FSW_LAUNCH_CHECK();
    }
    return 0;
  }
#define FSW_WB(M, WGT, LO, HI) \
  if ((rc = launch_wsort_bwd<M, WGT>(a, LO, HI, rows_upper, g, ldg, gXp, ldgp, gfreq, gkey, ldk, stream))) return rc
  // 33 .. 128: one lane per slice (embed_mid_bwd.hip)
  if ((rc = launch_embed_mid_bwd(a, rows_upper, g, ldg, gXp, ldgp, gfreq, gkey, ldk, stream))) return rc;
  constexpr int kFirst = FSW_BIN_MID0 + 6;                     // first bin above FSW_MID_MAX_DEG_WEIGHTED = 128
  // unit weights with stored key gradients (the store-and-sum backward) and four-slice alignment: the quad kernels
  const bool quad = unit && gkey && !a.efeat && a.S % 4 == 0 && ldk % 4 == 0 && a.ldp % 4 == 0 && ((uintptr_t)gkey & 15) == 0 &&
                    ((uintptr_t)a.Xp & 15) == 0 && !getenv("FSW_BWD_QUAD_OFF");
  if (quad) {
    if ((rc = launch_quad_bwd<4>(a, kFirst, FSW_BIN_LDS0 - 1, rows_upper, g, ldg, gfreq, gkey, ldk, stream))) return rc;   // 129 .. 256
    if ((rc = launch_quad_bwd<8>(a, FSW_BIN_LDS0, FSW_BIN_LDS0, rows_upper, g, ldg, gfreq, gkey, ldk, stream))) return rc;
    if ((rc = launch_quad_bwd<16>(a, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1, rows_upper, g, ldg, gfreq, gkey, ldk, stream))) return rc;
    if ((rc = launch_quad_bwd<32>(a, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2, rows_upper, g, ldg, gfreq, gkey, ldk, stream))) return rc;
  } else if (unit) {   // a wave holds 64 M elements
    FSW_WB(4, false, kFirst, FSW_BIN_LDS0 - 1);                // 129 .. 256
    FSW_WB(8, false, FSW_BIN_LDS0, FSW_BIN_LDS0);
    FSW_WB(16, false, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1);
    FSW_WB(32, false, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2);
  } else {      // D + 1 elements with the pad element: the bin of 256 and the LDS bins go one size up
    FSW_WB(4, true, kFirst, FSW_BIN_LDS0 - 2);                 // 129 .. 192
    FSW_WB(8, true, FSW_BIN_LDS0 - 1, FSW_BIN_LDS0 - 1);       // 193 .. 256
    FSW_WB(16, true, FSW_BIN_LDS0, FSW_BIN_LDS0);
    FSW_WB(32, true, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1);
    FSW_WB(64, true, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2);
  }
#undef FSW_WB
  return 0;
}

}  // namespace fsw
