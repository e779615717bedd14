// Generic neighbourhood kernels, any in-degree, float32 or float64 storage: the float64 build of the path and the
// gradients with respect to the weights.  gfx950.
//
// The tuned kernels (embed_reg / embed_mid / embed_hub / embed_wsort) are float32, specialised per degree class and treat the
// weights as constants.  This file restates the same per-neighbourhood computation once, for every degree and both value
// types, with all arithmetic in float64:
//   T = double   FSW_embedding / FSW_conv(dtype=torch.float64) (the reference's test_conv.py runs the layer in float64,
//                test_conv.py:24): forward and backward, pinned to the reference's float64 goldens at 1e-12;
//   T = float    d loss / d W for the float32 path (reference ag.div_sparse_dense.backward fsw_embedding.py:1656,
//                ag.cumsum_sparse.backward :2160 = reverse segmented cumsum, ag.permute_sparse.backward :1286).
// One workgroup per recipient row, looping over the slices.  Per (row, slice): (key, element index) pairs -- the reference's
// pad element x = 0 (fsw_embedding.py:787-821) is element D -- sorted by a bitonic network in LDS (up to 2048 elements) or in
// the workgroup's global scratch (any degree), ties by element index (= the reference's stable order); cumulative weights by a
// workgroup scan (the segmented cumsum of fsw_embedding.py:1031-1032); readout
//   Delta_t = 2 w_t sinc(xi w_t) cos(pi xi (2 c_t - w_t))      (fsw_embedding.py:1047-1075, the product form: no cancellation)
//   out     = (1 + xi) sum_t Delta_t p_(t)                      (fsw_embedding.py:1084-1109).
// Backward: d out / d p_(t) = (1 + xi) Delta_t stored per CSR entry (gkey), d out / d xi summed per slice, and for the weights
//   d out / d c_t = 2 (1 + xi) cos(2 pi xi c_t) (p_(t) - p_(t+1)),   c_t = A_t / M,  A_t = raw cumulative weight,  M = max(m, tau)
//   d out / d a_j = [ R(rank_j) - [m <= tau] R(rank_pad) - [m >= tau] sum_t H_t c_t ] / M,   R(r) = sum_{t >= r} H_t
// (a reverse cumulative sum over the sorted order; the two clamps pass gradients like the reference's custom_lowclamp, :1735-1744).
#include <algorithm>
#include "fsw_common.h"

namespace fsw {

constexpr int kGenThreads = 256;
constexpr int kGenLdsElems = 2048;   // lines up to this many elements are sorted in LDS
constexpr double kPiG = 3.14159265358979323846;

template <class T>
struct GenArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const T* w;        // [nnz] raw weights or null (unit)
  int64_t num_rows;
  const T* Xp;       // [num_cols, ldp]
  int64_t ldp;
  const T* Ke;       // [nnz, ldke] edge-feature term of every key, or null
  int64_t ldke;
  const T* freqs;
  int S;
  double tau;
  // forward
  T* out;
  int64_t ldo;
  const T* bias;
  double out_scale;
  int has_mass, mass_fn;
  double mass_scale;
  // backward (g != null): gkey [nnz, ldk] stored, gfreq [S] and gw [nnz] accumulated with atomics (zeroed by the caller)
  const T* g;
  int64_t ldg;
  T* gkey;
  int64_t ldk;
  T* gfreq;
  T* gw;
  // scratch: per workgroup line_elems * kGenScratchBytesPerElem bytes
  char* scratch;
  int64_t line_elems;
};
constexpr int kGenScratchBytesPerElem = 8 + 4 + 8 + 8;   // key, index, cumulative weight, H / reverse sum

__device__ __forceinline__ double mass_encode_g(double m, int fn) {
  if (fn == 1) return 2.0 * (m / (sqrt(m + 1.0) + 1.0));
  if (fn == 2) return log1p(m);
  return m;
}
__device__ __forceinline__ double sinc_g(double z) { return z == 0.0 ? 1.0 : sinpi(z) / (kPiG * z); }
__device__ __forceinline__ double dsinc_g(double z) { return z == 0.0 ? 0.0 : (cospi(z) - sinc_g(z)) / z; }   // reference sp.dsinc :2760-2774

// workgroup-wide sum of a double (all threads get the result)
__device__ __forceinline__ double block_sum(double v, double* red /* LDS [4] */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if (lane_id() == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// inclusive scan of one double per thread over the workgroup; returns this thread's inclusive value, *total = sum of all
__device__ __forceinline__ double block_inclusive_scan(double v, double* red /* LDS [4] */, double* total) {
  double inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const double t = __shfl_up(inc, off);
    if (lane_id() >= off) inc += t;
  }
  __syncthreads();
  if (lane_id() == kWave - 1) red[threadIdx.x >> 6] = inc;
  __syncthreads();
  double base = 0.0, tot = 0.0;
#pragma unroll
  for (int i = 0; i < kGenThreads / kWave; ++i) {
    if (i < (int)(threadIdx.x >> 6)) base += red[i];
    tot += red[i];
  }
  *total = tot;
  return base + inc;
}

__device__ __forceinline__ void atomic_add_t(float* p, double v) { atomicAdd(p, (float)v); }
__device__ __forceinline__ void atomic_add_t(double* p, double v) { atomicAdd(p, v); }

template <class T>
__global__ void __launch_bounds__(kGenThreads) k_embed_generic(const GenArgs<T> a) {
  __shared__ double lkey[kGenLdsElems];
  __shared__ int lidx[kGenLdsElems];
  __shared__ double red[4];
  char* myscr = a.scratch + (int64_t)blockIdx.x * a.line_elems * kGenScratchBytesPerElem;
  double* gkeyb = reinterpret_cast<double*>(myscr);                       // [line_elems] keys (rows above the LDS size)
  double* cw = gkeyb + a.line_elems;                                      // [line_elems] cumulative normalised weight
  double* hr = cw + a.line_elems;                                         // [line_elems] H_t, then reverse sums
  int* gidx = reinterpret_cast<int*>(hr + a.line_elems);                  // [line_elems]
  const bool backward = a.g != nullptr;
  const int tid = threadIdx.x;
  for (int64_t row = blockIdx.x; row < a.num_rows; row += gridDim.x) {
    const int start = a.rowptr[row];
    const int D = a.rowptr[row + 1] - start;
    const int Dtot = D + 1;                                                // + the pad element
    int Dp = 1;
    while (Dp < Dtot) Dp <<= 1;
    const bool in_lds = Dp <= kGenLdsElems;
    double* keys = in_lds ? lkey : gkeyb;
    int* idx = in_lds ? lidx : gidx;
    // total mass, pad weight, normalisation (fsw_embedding.py:778-829)
    double part = 0.0;
    for (int t = tid; t < D; t += kGenThreads) part += a.w ? (double)a.w[start + t] : 1.0;
    const double m = block_sum(part, red);
    const double M = fmax(m, a.tau);
    const double padw = fmax(a.tau - m, 0.0);
    const double invM = 1.0 / M;
    auto raw_weight = [&](int e) -> double { return e < D ? (a.w ? (double)a.w[start + e] : 1.0) : (e == D ? padw : 0.0); };
    if (!backward && a.has_mass && tid == 0)
      a.out[row * a.ldo] = (T)(a.out_scale * (mass_encode_g(m, a.mass_fn) * a.mass_scale + (a.bias ? (double)a.bias[0] : 0.0)));
    for (int k = 0; k < a.S; ++k) {
      // A. keys
      for (int t = tid; t < Dp; t += kGenThreads) {
        double key = __builtin_inf();
        if (t < D) {
          key = (double)a.Xp[(int64_t)a.col[start + t] * a.ldp + k];
          if (a.Ke) key += (double)a.Ke[(int64_t)(start + t) * a.ldke + k];
        } else if (t == D) {
          key = 0.0;
        }
        keys[t] = key;
        idx[t] = t;
      }
      __syncthreads();
      // B. bitonic sort by (key, index)
      for (int size = 2; size <= Dp; size <<= 1) {
        for (int st = size >> 1; st >= 1; st >>= 1) {
          for (int i = tid; i < Dp; i += kGenThreads) {
            const int j = i ^ st;
            if (j > i) {
              const double ki = keys[i], kj = keys[j];
              const int ii = idx[i], ij = idx[j];
              const bool up = (i & size) == 0;
              const bool gt = ki > kj || (ki == kj && ii > ij);
              if (gt == up) {
                keys[i] = kj;
                keys[j] = ki;
                idx[i] = ij;
                idx[j] = ii;
              }
            }
          }
          __syncthreads();
        }
      }
      // C. cumulative normalised weights in sorted order
      double run = 0.0;
      for (int b0 = 0; b0 < Dtot; b0 += kGenThreads) {
        const int t = b0 + tid;
        const double wv = t < Dtot ? raw_weight(idx[t]) * invM : 0.0;
        double tot;
        const double inc = block_inclusive_scan(wv, red, &tot);
        if (t < Dtot) cw[t] = run + inc;
        run += tot;
      }
      __syncthreads();
      const double xi = (double)a.freqs[k];
      // D. readout (and the per-key coefficients for the backward)
      const double gk = backward ? a.out_scale * (double)a.g[row * a.ldg + a.has_mass + k] : 0.0;
      double acc = 0.0, dacc = 0.0;
      for (int t = tid; t < Dtot; t += kGenThreads) {
        const int e = idx[t];
        const double wv = raw_weight(e) * invM;
        const double c = cw[t];
        const double B = xi * (2.0 * c - wv);            // phase / pi
        const double sc = sinc_g(xi * wv);
        const double cb = cospi(B);
        const double delta = 2.0 * wv * sc * cb;
        const double key = keys[t];
        acc += delta * key;
        if (backward) {
          const double ddelta = 2.0 * wv * (wv * dsinc_g(xi * wv) * cb - sc * kPiG * (2.0 * c - wv) * sinpi(B));
          dacc += (delta + (1.0 + xi) * ddelta) * key;
          if (e < D && a.gkey) a.gkey[(int64_t)(start + e) * a.ldk + k] = (T)(gk * (1.0 + xi) * delta);
          if (a.gw) {
            const double knext = t + 1 < Dtot ? keys[t + 1] : 0.0;
            hr[t] = 2.0 * (1.0 + xi) * cospi(2.0 * xi * c) * (key - knext);
          }
        }
      }
      if (!backward) {
        const double val = block_sum(acc, red);
        if (tid == 0)
          a.out[row * a.ldo + a.has_mass + k] = (T)(a.out_scale * ((1.0 + xi) * val + (a.bias ? (double)a.bias[a.has_mass + k] : 0.0)));
      } else {
        if (a.gfreq) {
          const double dv = block_sum(dacc, red);
          if (tid == 0 && gk != 0.0) atomic_add_t(&a.gfreq[k], gk * dv);
        }
        if (a.gw) {
          // E. weights: reverse cumulative sums of H over the sorted order, sum_t H_t c_t, the pad element's rank
          __syncthreads();
          double hc = 0.0;
          for (int t = tid; t < Dtot; t += kGenThreads) hc += hr[t] * cw[t];
          const double HC = block_sum(hc, red);
          double runr = 0.0;
          const int nchunk = (Dtot + kGenThreads - 1) / kGenThreads;
          for (int cix = nchunk - 1; cix >= 0; --cix) {      // chunks from the end; inside a chunk the scan runs over mirrored threads
            const int t = cix * kGenThreads + (kGenThreads - 1 - tid);
            const double hv = t < Dtot ? hr[t] : 0.0;
            double tot;
            const double inc = block_inclusive_scan(hv, red, &tot);
            __syncthreads();
            if (t < Dtot) hr[t] = runr + inc;                // R(t) = sum_{s >= t} H_s
            runr += tot;
          }
          __syncthreads();
          double rp = 0.0;
          for (int t = tid; t < Dtot; t += kGenThreads)
            if (idx[t] == D) rp = hr[t];
          const double Rpad = block_sum(rp, red);
          const double corr = (m <= a.tau ? Rpad : 0.0) + (m >= a.tau ? HC : 0.0);
          for (int t = tid; t < Dtot; t += kGenThreads) {
            const int e = idx[t];
            if (e < D && gk != 0.0) atomic_add_t(&a.gw[start + e], gk * (hr[t] - corr) * invM);
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---- float64 projection on the matrix cores: Xp [n, ldp] = X [n, ldx] . V [S, ldv]^T ---------------------------------------
// v_mfma_f64_16x16x4_f64: lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16] of a 16 x 4 by 4 x 16 product and holds
// C[l / 16 + 4 i][l % 16], i < 4.  One wavefront per 16 x 16 output tile, operands straight from global memory (this is the
// float64 build for tests, reference fsw_embedding.py:909-913 in float64; the measured path is project.hip).
using f64x4 = __attribute__((ext_vector_type(4))) double;

__global__ void __launch_bounds__(256) k_project_f64(const double* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                     const double* __restrict__ V, int S, int64_t ldv, double* __restrict__ Xp,
                                                     int64_t ldp, int32_t* __restrict__ stats) {
  const int lane = lane_id();
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nct = (S + 15) / 16;
  const int64_t rt = tile / nct;
  const int ct = (int)(tile - rt * nct);
  if (rt * 16 >= n) return;
  const int64_t r = rt * 16 + (lane & 15);
  const int c = ct * 16 + (lane & 15);
  const int kq = lane >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  int nonfinite = 0;
  for (int k0 = 0; k0 < d; k0 += 4) {
    const int k = k0 + kq;
    const double av = (r < n && k < d) ? X[r * ldx + k] : 0.0;
    const double bv = (c < S && k < d) ? V[(int64_t)c * ldv + k] : 0.0;
    nonfinite |= !(fabs(av) <= 1.7976931348623157e308);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t orow = rt * 16 + kq + 4 * i;      // C/D map of the f64 MFMA: row = (lane >> 4) + 4 reg, col = lane & 15
    if (orow < n && c < S) Xp[orow * ldp + c] = acc[i];
  }
  if (stats && nonfinite && ct == 0) atomicOr(&stats[FSW_STAT_FLAGS], FSW_FLAG_X_NONFINITE);
}

template <class T>
static int run_generic(const fsw_generic_args* g, hipStream_t stream) {
  GenArgs<T> a;
  a.rowptr = g->rowptr; a.col = g->col; a.w = (const T*)g->w; a.num_rows = g->num_rows;
  a.Xp = (const T*)g->Xp; a.ldp = g->ldp; a.Ke = (const T*)g->Ke; a.ldke = g->ldke; a.freqs = (const T*)g->freqs; a.S = g->S;
  a.tau = g->tau; a.out = (T*)g->out; a.ldo = g->ldo; a.bias = (const T*)g->bias; a.out_scale = g->out_scale;
  a.has_mass = g->has_mass; a.mass_fn = g->mass_fn; a.mass_scale = g->mass_scale;
  a.g = (const T*)g->g; a.ldg = g->ldg; a.gkey = (T*)g->gkey; a.ldk = g->ldk; a.gfreq = (T*)g->gfreq; a.gw = (T*)g->gw;
  int64_t line = 1;
  while (line < g->max_degree + 1) line <<= 1;
  a.line_elems = line;
  a.scratch = (char*)g->scratch;
  const int64_t per_wg = line * kGenScratchBytesPerElem;
  int64_t nwg = std::min<int64_t>(g->num_rows, 2048);
  nwg = std::min<int64_t>(nwg, (int64_t)(g->scratch_bytes / (size_t)per_wg));
  FSW_REQUIRE(nwg >= 1, "fsw_embed_generic: scratch buffer too small (need fsw_embed_generic_scratch_bytes)");
  k_embed_generic<T><<<(unsigned)nwg, kGenThreads, 0, stream>>>(a);
  FSW_LAUNCH_CHECK();
  return 0;
}

}  // namespace fsw

using namespace fsw;

extern "C" size_t fsw_embed_generic_scratch_bytes(int64_t max_degree, int64_t num_rows) {
  int64_t line = 1;
  while (line < max_degree + 1) line <<= 1;
  const size_t per_wg = (size_t)line * kGenScratchBytesPerElem;
  const size_t cap = (size_t)1 << 30;
  size_t nwg = (size_t)std::max<int64_t>(1, std::min<int64_t>(num_rows, 2048));
  nwg = std::max<size_t>(1, std::min<size_t>(nwg, cap / per_wg));
  return nwg * per_wg;
}

extern "C" int fsw_embed_generic(const fsw_generic_args* g, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(g, "fsw_embed_generic: null args");
  FSW_REQUIRE(g->value_dtype == 0 || g->value_dtype == 1, "fsw_embed_generic: value_dtype must be 0 (float32) or 1 (float64)");
  FSW_REQUIRE(g->rowptr && g->Xp && g->freqs && g->scratch && (g->num_rows == 0 || g->col || g->max_degree == 0), "fsw_embed_generic: null pointer");
  FSW_REQUIRE(g->num_rows >= 0 && g->S >= 1 && g->ldp >= g->S && g->max_degree >= 0 && g->tau > 0.0, "fsw_embed_generic: bad sizes");
  FSW_REQUIRE(!g->Ke || g->ldke >= g->S, "fsw_embed_generic: bad edge-term stride");
  if (g->g) {
    FSW_REQUIRE(g->ldg >= g->S + g->has_mass && (!g->gkey || g->ldk >= g->S), "fsw_embed_generic: bad gradient strides");
  } else {
    FSW_REQUIRE(g->out && g->ldo >= g->S + g->has_mass, "fsw_embed_generic: bad output");
  }
  if (g->num_rows == 0) return 0;
  return g->value_dtype == 0 ? run_generic<float>(g, stream) : run_generic<double>(g, stream);
}

extern "C" int fsw_project_f64(const double* X, int64_t n, int d, int64_t ldx, const double* V, int S, int64_t ldv, double* Xp,
                               int64_t ldp, int32_t* stats, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(X && V && Xp && n >= 1 && d >= 1 && S >= 1 && ldx >= d && ldv >= d && ldp >= S, "fsw_project_f64: bad arguments");
  const int64_t tiles = ceil_div(n, 16) * ceil_div(S, 16);
  FSW_REQUIRE(ceil_div(tiles, 4) < (1ll << 31), "fsw_project_f64: grid too large");
  k_project_f64<<<(unsigned)ceil_div(tiles, 4), 256, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats);
  FSW_LAUNCH_CHECK();
  return 0;
}
