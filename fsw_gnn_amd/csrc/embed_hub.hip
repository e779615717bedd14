// Fused neighbourhood kernel for hub rows, unit weights (FSW_LDS_MAX_DEG < in-degree <= FSW_HUB_MAX_DEG).  gfx950.
//
// One workgroup of NW wavefronts (2, 4, 8 or 16) takes ONE (recipient row, slice) line of up to NW * 2048 keys and keeps
// the whole line in registers: wavefront w holds elements w * 2048 .. w * 2048 + 2047 (32 per lane, WaveLine of
// wave_sort.h).  Every wavefront gathers and sorts its own chunk; the bitonic merge levels above one chunk -- a mirrored
// "flip" with wavefront w ^ (size - 1), half-cleaners with w ^ stride -- exchange registers through an LDS buffer (one
// conflict-free store and one load per key and exchange, then ONE v_min or v_max per key: which side of the pair a
// wavefront is on is wave-uniform), the rest of every level stays inside the wavefront (WaveLine::merge_chunk).  Nothing of
// the line ever goes to global memory: the scratch-line kernel this replaces for these degrees (embed_wsort.hip,
// k_embed_wsort_global) spent a third of its time in min/max sweeps over a global scratch line and ran a 64M-edge RMAT
// graph's hub class at 26 G keys/s.
// Gather: 4 bytes per lane from 64 different rows of Xp (one key per neighbour for ONE slice), so the kernel leans on the
// caches: blocks are dealt to the 8 XCDs round-robin, and the block index is decoded so that the blocks of one XCD walk
// the slices of the same row one after the other -- the 16 slices that share a 64-byte sector of an Xp row are then read by
// co-resident workgroups of one XCD out of its L2.  The column indices are read striped (lane-contiguous), which any
// initial arrangement allows because the line is sorted afterwards.
// Replaces, for these rows, the reference's global sort + sparse permutation + segmented cumsum (fsw_embedding.py:917-1032)
// and the readout (:1047-1109).
#include <algorithm>
#include "fsw_common.h"
#include "sortnet.h"
#include "wave_sort.h"

namespace fsw {

constexpr double kPiH = 3.14159265358979323846;
constexpr int kHubM = 32;                  // keys per lane
#ifndef FSW_HUB_ABL
#define FSW_HUB_ABL 0   // timing experiments (tools/exp_hub.sh): 1 no gather, 2 no wave sort, 4 no cross-wave merge
#endif

__device__ __forceinline__ float mass_encode_h(float m, int fn) {
  if (fn == 1) return 2.f * (m / (sqrtf(m + 1.f) + 1.f));
  if (fn == 2) return log1pf(m);
  return m;
}

// NW wavefronts per line, M keys per lane; NW == 1: the workgroup is four independent wavefronts on four lines (adjacent
// slices of one row) and never synchronises -- the wave-sort classes 257..2048 (M = 8 / 16 / 32) run this way
template <int NW, int M>
__global__ void __launch_bounds__(NW == 1 ? 256 : NW * kWave) k_embed_hub(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ bin_start, int bin, const float* __restrict__ Xp, int64_t ldp, int S,
    const float* __restrict__ freqs, float* __restrict__ out, int64_t ldo, const float* __restrict__ bias, float out_scale,
    int has_mass, int mass_fn, float mass_scale) {
  constexpr int CAP = M * kWave;
  constexpr int LPB = NW == 1 ? 4 : 1;   // lines per block
  __shared__ float xbuf[NW > 1 ? NW : 1][NW > 1 ? CAP : 1];   // exchange buffer: element (lane, j) of wavefront w at xbuf[w][j * 64 + lane]
  __shared__ float red[NW];
  const int pbeg = bin_start[bin], nrows = bin_start[bin + 1] - pbeg;
  const int lane = lane_id();
  const int w = NW == 1 ? 0 : wave_id();
  // block -> (row, slice): the blocks b, b + 8, b + 16, ... (one XCD under round-robin dispatch) take slices 0, 1, 2, ... of
  // row xcd, then of row xcd + 8, ...
  const int xcd = blockIdx.x & 7;
  const int64_t i = (int64_t)(blockIdx.x >> 3) * LPB + (NW == 1 ? wave_id() : 0);
  const int64_t rl = i / S;
  const int k = (int)(i - rl * S);
  const int64_t r = rl * 8 + xcd;
  if (r >= nrows) return;          // NW > 1: the whole workgroup leaves; NW == 1: no barrier below
  const int node = perm[pbeg + r];
  const int start = rowptr[node];
  const int D = rowptr[node + 1] - start;

  // 1. gather this wavefront's chunk (striped: element j * 64 + lane of the chunk) and sort it
  WaveLine<M, false> ln;
  {
    int c[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int t = w * CAP + j * kWave + lane;
      c[j] = t < D ? col[start + t] : -1;
    }
#pragma unroll
    for (int j = 0; j < M; ++j)
      ln.k[j] = c[j] >= 0 ? ((FSW_HUB_ABL & 1) ? (float)((c[j] * 2654435761u) >> 8) : Xp[(int64_t)c[j] * ldp + k]) : __builtin_inff();
  }
  if (!(FSW_HUB_ABL & 2)) ln.sort();

  // 2. merge levels above one wavefront: `size` sorted chunks -> one sorted run
  if constexpr (NW > 1) {
    auto exchange = [&](int partner, bool mirrored, bool lower) {
      float* mine = xbuf[w];
#pragma unroll
      for (int j = 0; j < M; ++j) mine[j * kWave + lane] = ln.k[j];
      __syncthreads();
      const float* theirs = xbuf[partner];
      float o[M];
#pragma unroll
      for (int j = 0; j < M; ++j) o[j] = mirrored ? theirs[(M - 1 - j) * kWave + (kWave - 1 - lane)] : theirs[j * kWave + lane];
      if (lower) {
#pragma unroll
        for (int j = 0; j < M; ++j) ln.k[j] = fminf(ln.k[j], o[j]);
      } else {
#pragma unroll
        for (int j = 0; j < M; ++j) ln.k[j] = fmaxf(ln.k[j], o[j]);
      }
      __syncthreads();   // everybody has read: the buffer may be overwritten by the next exchange
    };
#pragma unroll
    for (int size = 2; size <= ((FSW_HUB_ABL & 4) ? 0 : NW); size <<= 1) {
      exchange(w ^ (size - 1), true, (w & (size >> 1)) == 0);        // element E against E ^ (size * CAP - 1)
      for (int st = size >> 2; st >= 1; st >>= 1) exchange(w ^ st, false, (w & st) == 0);
      ln.merge_chunk();
    }
  }

  // 3. readout: element (lane, j) of wavefront w has rank w * CAP + lane * M + j; coefficients
  //    (1 + xi) [sin(2 pi xi (r + 1) / D) - sin(2 pi xi r / D)] / (pi xi) (reference fsw_embedding.py:1047-1075, 1109 with
  //    weights 1 / D) by a float64 rotation started at the lane's first rank
  const float xif = freqs[k];
  const double xi = (double)xif;
  const bool lin = xif < 1e-30f;            // xi == 0: Delta_t = 2 w_t
  const double inv = 1.0 / (double)D;
  const int r0 = w * CAP + lane * M;
  float acc = 0.f;
  if (lin) {
#pragma unroll
    for (int j = 0; j < M; ++j) acc += (r0 + j < D) ? ln.k[j] : 0.f;
    acc *= 2.f * (float)inv;
  } else {
    const double step = xi * inv;           // revolutions per rank
    double sd, cd, s, c;
    sincospi(2.0 * (step - rint(step)), &sd, &cd);
    const double x0 = step * (double)r0;
    sincospi(2.0 * (x0 - rint(x0)), &s, &c);
    const double scale = (1.0 + xi) / (kPiH * xi);
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const double sn = fma(s, cd, c * sd), cn = fma(c, cd, -(s * sd));
      acc += (r0 + j < D) ? (float)(scale * (sn - s)) * ln.k[j] : 0.f;
      s = sn;
      c = cn;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  float tot = acc;
  if constexpr (NW > 1) {
    if (lane == 0) red[w] = acc;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) tot += red[q];
  }
  if (lane == 0 && w == 0) {
    float* orow = out + (int64_t)node * ldo;
    orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
    if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
  }
}

template <int NW, int M>
static int launch_hub(const fsw_embed_args& a, int bin, int64_t rows_upper, hipStream_t stream) {
  constexpr int LPB = NW == 1 ? 4 : 1;
  const int64_t nblocks = ceil_div(ceil_div(rows_upper, 8) * a.S, LPB) * 8;
  FSW_REQUIRE(nblocks < (1ll << 31), "fsw_embed_f32: too many long rows x slices for one launch");
  k_embed_hub<NW, M><<<(unsigned)nblocks, NW == 1 ? 256 : NW * kWave, 0, stream>>>(
      a.rowptr, a.col, a.perm, a.bin_start, bin, a.Xp, a.ldp, a.S, a.freqs, a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn,
      a.mass_scale);
  FSW_LAUNCH_CHECK();
  return 0;
}

// unit weights, tau <= 1: rows of the four hub bins.  rows_upper bounds the rows above FSW_LDS_MAX_DEG (the per-bin counts
// stay on the device: surplus blocks exit at once).
int launch_embed_hub(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  int rc;
  const int64_t md = a.max_degree;   // host value, <= 0 when unknown
  if ((rc = launch_hub<2, kHubM>(a, FSW_BIN_HUB0, rows_upper, stream))) return rc;
  if ((md <= 0 || md > 4096) && (rc = launch_hub<4, kHubM>(a, FSW_BIN_HUB0 + 1, rows_upper, stream))) return rc;
  if ((md <= 0 || md > 8192) && (rc = launch_hub<8, kHubM>(a, FSW_BIN_HUB0 + 2, rows_upper, stream))) return rc;
  if ((md <= 0 || md > 16384) && (rc = launch_hub<16, kHubM>(a, FSW_BIN_HUB0 + 3, rows_upper, stream))) return rc;
  return 0;
}

// unit weights, tau <= 1: the wave-sort classes 257..512 / ..1024 / ..2048 (one wavefront per line, 8 / 16 / 32 keys per lane).
// rows_upper bounds the rows of FSW_REG_MAX_DEG < degree <= FSW_LDS_MAX_DEG.
int launch_embed_ws_unit(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  int rc;
  const int64_t md = a.max_degree;
  if ((md <= 0 || md > FSW_MID_MAX_DEG) && (rc = launch_hub<1, 8>(a, FSW_BIN_LDS0, rows_upper, stream))) return rc;
  if ((md <= 0 || md > 512) && (rc = launch_hub<1, 16>(a, FSW_BIN_LDS0 + 1, rows_upper, stream))) return rc;
  if ((md <= 0 || md > 1024) && (rc = launch_hub<1, 32>(a, FSW_BIN_LDS0 + 2, rows_upper, stream))) return rc;
  return 0;
}

}  // namespace fsw
