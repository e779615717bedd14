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
#include "merge_path.h"

namespace fsw {

constexpr double kPiH = 3.14159265358979323846;
[[maybe_unused]] constexpr int kHubM = 32;                  // keys per lane
#ifndef FSW_HUB_ABL
#define FSW_HUB_ABL 0   // timing experiments (tools/exp_hub.sh): 1 no gather, 2 no wave sort, 4 no cross-wave merge, 8 no readout
#endif

__device__ __forceinline__ float mass_encode_h(float m, int fn) {
  if (fn == 1) return 2.f * (m / (sqrtf(m + 1.f) + 1.f));
  if (fn == 2) return log1pf(m);
  return m;
}

__device__ __forceinline__ float wave_sum_h(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// The instantiations are split over three translation units (FSW_HUB_PART = 0: unit weights, 1 / 2: general weights up to 2048 /
// above; see the Makefile): the (key, weight) networks take minutes to compile.
#ifndef FSW_HUB_PART
#error "compile with -DFSW_HUB_PART=0|1|2"
#endif

#if FSW_HUB_PART == 0
// ---- building blocks shared by the kernels below ---------------------------------------------------------------------
// register exchange between the wavefronts of a workgroup through xbuf [NW][CAP]: this wavefront keeps, element by
// element, the smaller (lower) or larger key of (its own, wavefront `partner`'s -- same element, or mirrored)
template <int M>
__device__ __forceinline__ void wave_exchange(WaveLine<M, false>& ln, float* __restrict__ xbuf, int w, int lane, int partner,
                                              bool mirrored, bool lower) {
  constexpr int CAP = M * kWave;
  // one base register per side and immediate offsets j * 256 B: the asm statements keep the compiler from folding the lane
  // term into 32 separate per-element addresses (v_bitop3 of lane ^ constant), which it then hoists out of loops and spills.
  // The OFFSET is laundered, not the pointer: an opaque pointer loses its LDS address space and every access became a
  // flat_load / flat_store (72 of each in k_embed_hub<4, 24>: the flat path counts on vmcnt AND lgkmcnt, so each exchange also
  // waited for the wavefront's outstanding gathers)
  int moff = w * CAP + lane;
  asm volatile("" : "+v"(moff));
  float* mine = xbuf + moff;
#pragma unroll
  for (int j = 0; j < M; ++j) mine[j * kWave] = ln.k[j];
  __syncthreads();
  int toff = partner * CAP + (mirrored ? kWave - 1 - lane : lane);
  asm volatile("" : "+v"(toff));
  const float* theirs = xbuf + toff;
  const float lim = lower ? -__builtin_inff() : __builtin_inff();   // wave-uniform: min below the partner, max above it
#pragma unroll
  for (int j = 0; j < M; ++j) ln.k[j] = minmax_by_limit(ln.k[j], theirs[(mirrored ? M - 1 - j : j) * kWave], lim);
  __syncthreads();   // everybody has read: the buffer may be overwritten by the next exchange
}

// NW sorted chunks (one per wavefront, after WaveLine::sort) -> the workgroup's NW * CAP keys sorted; element (lane, j) of
// wavefront w then has rank w * CAP + lane * M + j
template <int NW, int M>
__device__ __forceinline__ void workgroup_merge_levels(WaveLine<M, false>& ln, float* __restrict__ xbuf, int w, int lane) {
#pragma unroll
  for (int size = 2; size <= ((FSW_HUB_ABL & 4) ? 0 : NW); size <<= 1) {
    wave_exchange<M>(ln, xbuf, w, lane, w ^ (size - 1), true, (w & (size >> 1)) == 0);        // element E against E ^ (size * CAP - 1)
    for (int st = size >> 2; st >= 1; st >>= 1) wave_exchange<M>(ln, xbuf, w, lane, w ^ st, false, (w & st) == 0);
    ln.merge_chunk();
  }
}

// the workgroup's NW * CAP keys form a bitonic sequence whose halves were separated elsewhere: finish the merge
template <int NW, int M>
__device__ __forceinline__ void workgroup_merge_block(WaveLine<M, false>& ln, float* __restrict__ xbuf, int w, int lane) {
#pragma unroll
  for (int st = NW >> 1; st >= 1; st >>= 1) wave_exchange<M>(ln, xbuf, w, lane, w ^ st, false, (w & st) == 0);
  ln.merge_chunk();
}

// unit-weight readout of the lane's M keys of ranks r0 .. r0 + M - 1 in a neighbourhood of D: coefficients
// (1 + xi) [sin(2 pi xi (r + 1) / D) - sin(2 pi xi r / D)] / (pi xi) (reference fsw_embedding.py:1047-1075, 1109 with
// weights 1 / D) = B cos(2 pi xi (r + 1/2) / D) by the one-FMA float64 recurrence of UnitCoef (fsw_common.h) started at the
// lane's first rank.  Returns the lane's partial sum.
template <int M, class Line>
__device__ __forceinline__ float unit_readout(const Line& ln, int r0, int D, float xif) {
  const double xi = (double)xif;
  const double inv = 1.0 / (double)D;
  float acc = 0.f;
  if (xif < 1e-30f) {                       // xi == 0: Delta_t = 2 w_t
#pragma unroll
    for (int j = 0; j < M; ++j) acc += (r0 + j < D) ? ln.k[j] : 0.f;
    return acc * 2.f * (float)inv;
  }
  if constexpr (FSW_HUB_ABL & 8) {
#pragma unroll
    for (int j = 0; j < M; ++j) acc += (r0 + j < D) ? ln.k[j] : 0.f;
    return acc;
  }
  UnitCoef uc;
  uc.start(xi, D, r0);
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const float cj = uc.next();
    acc += (r0 + j < D) ? cj * ln.k[j] : 0.f;
  }
  return acc * uc.B;
}

// gather elements t0 + j * 64 + lane (j < M; striped: lane-contiguous col reads -- the chunk is sorted next) of slice k
template <int M>
__device__ __forceinline__ void gather_chunk(WaveLine<M, false>& ln, const int32_t* __restrict__ colrow, int t0, int D,
                                             const float* __restrict__ Xp, int64_t ldp, int k, int lane) {
  int c[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int t = t0 + j * kWave + lane;
    c[j] = t < D ? colrow[t] : -1;
  }
#pragma unroll
  for (int j = 0; j < M; ++j)
    ln.k[j] = c[j] >= 0 ? ((FSW_HUB_ABL & 1) ? (float)((c[j] * 2654435761u) >> 8) : Xp[(int64_t)c[j] * ldp + k]) : __builtin_inff();
}

// one line: gather, sort inside the wavefronts, merge across them, readout.  Returns the wavefront's partial sum.
template <int NW, int M>
__device__ __forceinline__ float hub_line(const int32_t* __restrict__ colrow, int D, const float* __restrict__ Xp, int64_t ldp, int k,
                                          float xif, float* __restrict__ xbuf, int w, int lane) {
  constexpr int CAP = M * kWave;
  WaveLine<M, false> ln;
  gather_chunk<M>(ln, colrow, w * CAP, D, Xp, ldp, k, lane);
  if (!(FSW_HUB_ABL & 2)) ln.sort();
  if constexpr (NW > 1) workgroup_merge_levels<NW, M>(ln, xbuf, w, lane);
  return wave_sum_h(unit_readout<M>(ln, w * CAP + lane * M, D, xif));
}

#ifndef FSW_HUB_MINWAVES
#define FSW_HUB_MINWAVES 4   // waves per SIMD the hub kernels are compiled for (128 registers)
#endif
#ifndef FSW_ROWLINES_SPLIT
#define FSW_ROWLINES_SPLIT 0
#endif
#ifndef FSW_HUB_ROWLINES
#define FSW_HUB_ROWLINES 1   // 0: class 257..512 as one 64-lane line of 8 keys per lane (for comparison)
#endif
#ifndef FSW_HUB_SPLIT
#define FSW_HUB_SPLIT 1   // 0: every line on the full class size (for comparison)
#endif

// NW wavefronts per line, M keys per lane; NW == 1: the workgroup is four independent wavefronts on four lines (adjacent
// slices of one row) and never synchronises -- the wave-sort classes 257..2048 (M = 8 / 16 / 32) run this way
template <int NW, int M>
__global__ void __launch_bounds__(NW == 1 ? 256 : NW * kWave, M > 32 ? 2 : FSW_HUB_MINWAVES) k_embed_hub(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ bin_start, int bin, const float* __restrict__ Xp, int64_t ldp, int S,
    const float* __restrict__ freqs, float* __restrict__ out, int64_t ldo, const float* __restrict__ bias, float out_scale,
    int has_mass, int mass_fn, float mass_scale, int dmin, int dmax) {
  constexpr int CAP = M * kWave;
  constexpr int LPB = NW == 1 ? 4 : 1;   // lines per block
  __shared__ float xbuf[NW > 1 ? NW * CAP : 1];   // exchange buffer: element (lane, j) of wavefront w at xbuf[w * CAP + j * 64 + lane]
  __shared__ float red[NW];
  const int pbeg = bin_start[bin], nrows = bin_start[bin + 1] - pbeg;
  const int lane = lane_id();
  const int w = NW == 1 ? 0 : wave_id();
  // virtual block -> (row, slice): the blocks b, b + 8, b + 16, ... (one XCD under round-robin dispatch) take slices 0, 1, 2,
  // ... of row xcd, then of row xcd + 8, ...  The grid is capped (a dispatch holds < 2^32 work-items) and strides over the
  // virtual blocks; gridDim.x is a multiple of 8, so a workgroup stays on its residue and leaves at its first row past the bin.
  const int xcd = blockIdx.x & 7;
  for (int64_t vb = blockIdx.x;; vb += gridDim.x) {
    const int64_t i = (vb >> 3) * LPB + (NW == 1 ? wave_id() : 0);
    const int64_t rl = i / S;
    const int k = (int)(i - rl * S);
    const int64_t r = rl * 8 + xcd;
    if (r >= nrows) return;          // NW > 1: the whole workgroup leaves; NW == 1: no barrier below
    const int node = perm[pbeg + r];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    if (D < dmin || D > dmax) continue;   // multi-wavefront classes: the 3/4-size instantiation and the full one share a bin

    // one wavefront per line: a line of at most 3/4 of the class size runs on 3/4 of the keys per lane (24 / 12 / 6) -- the same
    // network, a quarter fewer comparators; rows spread over (2^k, 2^(k+1)], so about half of them qualify (wave-uniform
    // branch).  Measured on the 64M-edge RMAT graph: class 1025..2048 13.2 -> 11.6 ms.  The multi-wavefront classes gain
    // nothing from it (4097..8192: 12.3 ms either way -- they wait on the exchange barriers, not on comparators) and the
    // second code path costs them registers (20..50 spilled), so they always run the full size.
    constexpr int MS = (M * 3) / 4;
    float tot;
    if constexpr (FSW_HUB_SPLIT && NW == 1) {
      if (D <= kWave * MS) tot = hub_line<NW, MS>(col + start, D, Xp, ldp, k, freqs[k], xbuf, w, lane);
      else tot = hub_line<NW, M>(col + start, D, Xp, ldp, k, freqs[k], xbuf, w, lane);
    } else {
      tot = hub_line<NW, M>(col + start, D, Xp, ldp, k, freqs[k], xbuf, w, lane);
    }
    if constexpr (NW > 1) {
      if (lane == 0) red[w] = tot;
      __syncthreads();
      tot = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) tot += red[q];
      __syncthreads();               // red is rewritten by the next virtual block
    }
    if (lane == 0 && w == 0) {
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
  }
}

// ---- 2049..32768 neighbours with 16-byte gathers: a workgroup takes FOUR adjacent slices of one row, one after the other ----------
// The line of slice k0 (a multiple of 4: S % 4 == 0) is gathered as float4 = slices k0 .. k0 + 3; the first component goes into the
// registers, the other three into the lane's own places of a stash in global memory (lane-contiguous 4-byte stores), from where the
// SAME lane reads them back (coalesced) when the workgroup turns to slices k0 + 1 .. k0 + 3: no synchronisation, 6 bytes of streamed
// traffic per key in exchange for a quarter of the scattered requests -- which are what bounds these kernels (gather alone: 0.37 of
// 0.41 ms / 0.70 of 0.90 / 0.44 of 0.72 on the three populated classes of the RMAT-20 graph, tools/exp_hub.sh).
// stash: 3 * NW * 64 * M floats per workgroup of the grid.
template <int NW, int M>
__global__ void __launch_bounds__(NW* kWave, M > 32 ? 2 : FSW_HUB_MINWAVES) k_embed_hub_q4(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ bin_start, int bin, const float* __restrict__ Xp, int64_t ldp, int S,
    const float* __restrict__ freqs, float* __restrict__ out, int64_t ldo, const float* __restrict__ bias, float out_scale,
    int has_mass, int mass_fn, float mass_scale, int dmin, int dmax, float* __restrict__ stash_all) {
  static_assert(NW > 1, "one line across several wavefronts");
  constexpr int CAP = M * kWave;
  constexpr int G8 = 8;      // stash places per immediate-offset window (8 x 256 bytes)
  __shared__ float xbuf[NW * CAP];
  __shared__ float red[NW];
  const int pbeg = bin_start[bin], nrows = bin_start[bin + 1] - pbeg;
  const int lane = lane_id(), w = wave_id();
  // wave-uniform base of the wavefront's part of the stash (scalar registers); element (line, j, lane) at
  // swave[line * NW * CAP + j * 64 + lane]: the lane is the only vector term of every stash address
  float* swave = stash_all + (int64_t)blockIdx.x * (3 * NW * CAP) + w * CAP;
  const int xcd = blockIdx.x & 7;
  // ONE loop over (slice quad, slice of the quad): step n works on line sl = n & 3 of the workgroup's (n >> 2)-th quad
  for (int64_t n = 0;; ++n) {
    const int64_t vb = blockIdx.x + (n >> 2) * (int64_t)gridDim.x;
    const int sl = (int)(n & 3);
    const int64_t i = (vb >> 3) * 4;
    const int64_t rl = i / S;
    const int k0 = (int)(i - rl * S);
    const int64_t r = rl * 8 + xcd;
    if (r >= nrows) return;
    const int node = perm[pbeg + r];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    if (D < dmin || D > dmax) continue;
    WaveLine<M, false> ln;
    if (sl == 0) {
      const int32_t* colrow = col + start;
      const float* xr = Xp + k0;
      // batches of G elements: G column indices, then G 16-byte gathers (4 G registers in flight), stashed before the next batch
      constexpr int G = 8;
      static_assert(M % G == 0, "keys per lane: a multiple of 8");
      int c[2][G];
#pragma unroll
      for (int q = 0; q < G; ++q) {
        const int t = w * CAP + q * kWave + lane;
        c[0][q] = colrow[min(t, D - 1)];             // past the end: the last neighbour again (its key is replaced by +inf below)
      }
      // three running pointers into the stash (lines k0 + 1 .. k0 + 3), bumped once per batch, immediate offsets inside a batch.
      // The empty asm statements make them opaque: left alone the compiler forms a 64-bit vector address for every one of the
      // 3 M stash places, hoists them out of the loop over the lines and spills them (250 registers).
      // (running OFFSETS into the wave's stash, not pointers: an opaque pointer is a generic one, and its stores became flat_store --
      // counted on lgkmcnt as well, so every LDS exchange of the sort waited for them)
      int o0 = lane, o1 = lane + NW * CAP, o2 = lane + 2 * NW * CAP;
      asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2));
#pragma unroll
      for (int g = 0; g < M / G; ++g) {
        if (g + 1 < M / G) {
#pragma unroll
          for (int q = 0; q < G; ++q) {
            const int t = w * CAP + ((g + 1) * G + q) * kWave + lane;
            c[(g + 1) & 1][q] = colrow[min(t, D - 1)];
          }
        }
        float4 v[G];
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int cc = c[g & 1][q];                  // always a valid row: no branch around the load
          if constexpr (FSW_HUB_ABL & 1) v[q].x = v[q].y = v[q].z = v[q].w = (float)((cc * 2654435761u) >> 8);
          else v[q] = *reinterpret_cast<const float4*>(xr + (int64_t)cc * ldp);
        }
#pragma unroll
        for (int q = 0; q < G; ++q) {
          if (w * CAP + (g * G + q) * kWave + lane >= D) v[q] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());
          ln.k[g * G + q] = v[q].x;
          swave[o0 + q * kWave] = v[q].y;
          swave[o1 + q * kWave] = v[q].z;
          swave[o2 + q * kWave] = v[q].w;
        }
        o0 += G * kWave;
        o1 += G * kWave;
        o2 += G * kWave;
        asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2));   // also keeps the batches apart (all M gathers at once: 4 M registers)
      }
    } else {
      int so = (sl - 1) * NW * CAP + lane;
      asm volatile("" : "+v"(so));
#pragma unroll
      for (int j = 0; j < M; ++j) {
        ln.k[j] = swave[so + (j % G8) * kWave];
        if ((j % G8) == G8 - 1) {
          so += G8 * kWave;
          asm volatile("" : "+v"(so));
        }
      }
    }
    if (!(FSW_HUB_ABL & 2)) ln.sort();
    workgroup_merge_levels<NW, M>(ln, xbuf, w, lane);
    float tot = wave_sum_h(unit_readout<M>(ln, w * CAP + lane * M, D, freqs[k0 + sl]));
    if (lane == 0) red[w] = tot;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) tot += red[q];
    __syncthreads();
    if (lane == 0 && w == 0) {
      const int k = k0 + sl;
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
  }
}

// ---- 513..2048 neighbours with 16-byte gathers: the four wavefronts of a workgroup take four ADJACENT slices of one row ----------
// k_embed_hub<1, M> reads 4 bytes per lane from 64 different rows of Xp per instruction, and that -- the number of requests, not the
// bytes -- is what bounds it (gather alone 1.18 of the class's 1.73 ms on the RMAT-20 graph, tools/exp_hub.sh).  Here wavefront w
// gathers a QUARTER of the row's elements as float4 = slices k0 .. k0 + 3 (k0 a multiple of 4: S % 4 == 0) and the four lines are
// dealt to their wavefronts through LDS: one conflict-free ds_write_b32 and one ds_read_b32 per key, a quarter of the global load
// instructions with 16 bytes per request.  Sort and readout are those of k_embed_hub<1, M> (one line in the registers of one wavefront).
template <int M>
__device__ __forceinline__ float hub_quad_line(const float* __restrict__ xline, int D, float xif, int lane) {
  WaveLine<M, false> ln;
#pragma unroll
  for (int j = 0; j < M; ++j) ln.k[j] = xline[j * kWave + lane];     // striped: any arrangement will do, the line is sorted next
  if (!(FSW_HUB_ABL & 2)) ln.sort();
  return wave_sum_h(unit_readout<M>(ln, lane * M, D, xif));
}

template <int M>
__global__ void __launch_bounds__(256, FSW_HUB_MINWAVES) k_embed_hub_quad(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ bin_start, int bin, const float* __restrict__ Xp, int64_t ldp, int S,
    const float* __restrict__ freqs, float* __restrict__ out, int64_t ldo, const float* __restrict__ bias, float out_scale,
    int has_mass, int mass_fn, float mass_scale, int dmin, int dmax) {
  constexpr int CAP = M * kWave;
  constexpr int MS = (M * 3) / 4;
  __shared__ float xq[4][CAP];
  const int pbeg = bin_start[bin], nrows = bin_start[bin + 1] - pbeg;
  const int lane = lane_id(), w = wave_id();
  const int xcd = blockIdx.x & 7;
  for (int64_t vb = blockIdx.x;; vb += gridDim.x) {
    const int64_t i = (vb >> 3) * 4;              // the workgroup's first line; S % 4 == 0: its four lines are one row's
    const int64_t rl = i / S;
    const int k0 = (int)(i - rl * S);
    const int64_t r = rl * 8 + xcd;
    if (r >= nrows) return;                       // the whole workgroup
    const int node = perm[pbeg + r];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    if (D < dmin || D > dmax) continue;
    const int32_t* colrow = col + start;
    const float* xr = Xp + k0;
    const bool small = FSW_HUB_SPLIT && D <= kWave * MS;      // a line of at most 3/4 of the class size: 3/4 of the keys per lane
    const int nq = small ? MS / 4 : M / 4;
    static_assert(M % 16 == 0, "3/4 of the keys per lane must still be a multiple of 4");
    int c[M / 4];
#pragma unroll
    for (int q = 0; q < M / 4; ++q) {
      const int t = (q * 4 + w) * kWave + lane;
      c[q] = colrow[min(t, D - 1)];                  // past the end: the last neighbour again (replaced by +inf below)
    }
#pragma unroll
    for (int q = 0; q < M / 4; ++q) {
      if (q < nq) {                                 // wave-uniform
        const int t = (q * 4 + w) * kWave + lane;
        float4 v;
        if constexpr (FSW_HUB_ABL & 1) v.x = v.y = v.z = v.w = (float)((c[q] * 2654435761u) >> 8);
        else v = *reinterpret_cast<const float4*>(xr + (int64_t)c[q] * ldp);
        if (t >= D) v = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());
        xq[0][t] = v.x;
        xq[1][t] = v.y;
        xq[2][t] = v.z;
        xq[3][t] = v.w;
      }
    }
    __syncthreads();
    const float xif = freqs[k0 + w];
    const float tot = small ? hub_quad_line<MS>(xq[w], D, xif, lane) : hub_quad_line<M>(xq[w], D, xif, lane);
    __syncthreads();                               // every line has been read: the next row may overwrite the buffer
    if (lane == 0) {
      const int k = k0 + w;
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
  }
}

// ---- rows above FSW_HUB_MAX_DEG, unit weights: blocks of 16384 keys sorted in the registers of an 8-wavefront workgroup --
// A line (row, slice) of any length is cut into blocks of kGiantBlk = 8 * 2048 keys (8 wavefronts: 16 would leave 128 registers per lane and spill).  Every block is gathered and sorted
// like a hub row and parked in the workgroup's scratch line (global memory, 4 bytes per key); the bitonic merge levels
// above one block are element-wise min/max sweeps over pairs of blocks (coalesced, by all 512 threads) followed by the
// in-workgroup tail of the level (workgroup_merge_block) on every block, back in registers; the last level feeds the readout
// instead of going back to memory.  Blocks past the end of the row hold only +inf and are never touched: a pair with such a
// block on its upper side is a no-op (the scratch-line kernel of embed_wsort.hip, which this replaces for unit weights,
// swept every level at the granularity of one wavefront's 2048 keys: 28 sweeps for a 150 000-neighbour hub against 10 here).
#ifndef FSW_GIANT_NW
#define FSW_GIANT_NW 16
#endif
constexpr int kGiantNW = FSW_GIANT_NW;
constexpr int kGiantBlk = kGiantNW * kHubM * kWave;   // 16384

__global__ void __launch_bounds__(kGiantNW* kWave, 4) k_embed_giant(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                const int32_t* __restrict__ perm,
                                                                const int32_t* __restrict__ bin_start, const float* __restrict__ Xp,
                                                                int64_t ldp, int S, const float* __restrict__ freqs,
                                                                float* __restrict__ out, int64_t ldo, const float* __restrict__ bias,
                                                                float out_scale, int has_mass, int mass_fn, float mass_scale,
                                                                float* __restrict__ scratch, int64_t line_floats) {
  constexpr int NW = kGiantNW, M = kHubM, CAP = M * kWave, BLK = kGiantBlk, NT = NW * kWave;
  __shared__ float xbuf[NW * CAP];
  __shared__ float red[NW];
  const int pbeg = bin_start[FSW_BIN_GLOBAL], nrows = bin_start[FSW_BIN_GLOBAL + 1] - pbeg;
  const int lane = lane_id(), w = wave_id();
  // the workgroups of one XCD take consecutive lines (slices of the same row) when the grid is a multiple of 8
  const int blk = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
  float* sl = scratch + (int64_t)blk * line_floats;
  const int64_t nlines = (int64_t)nrows * S;
  // order within the workgroup: every wavefront's scratch stores performed, then a barrier.  Workgroup scope is enough: the
  // wavefronts of a workgroup share their CU's L1, which the CU's own stores write through (an agent-scope fence would write
  // back the whole XCD's L2 at every one of the ~10 synchronisation points of a line: measured 3x the kernel time)
  auto sync_scratch = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
  };
  for (int64_t line = blk; line < nlines; line += gridDim.x) {
    const int p = pbeg + (int)(line / S), k = (int)(line % S);
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    const int nb = (D + BLK - 1) / BLK;                   // blocks that hold keys
    const int nbp = (int)pow2ceil((uint32_t)nb);
    const float xif = freqs[k];
    WaveLine<M, false> ln;
    // A. blocks: gather, sort in the workgroup's registers, park in the scratch line
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
      gather_chunk<M>(ln, col + start, b * BLK + w * CAP, D, Xp, ldp, k, lane);
      ln.sort();
      workgroup_merge_levels<NW, M>(ln, xbuf, w, lane);
      float* dst = sl + (int64_t)b * BLK + w * CAP + lane * M;
#pragma unroll
      for (int j = 0; j < M; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(ln.k[j], ln.k[j + 1], ln.k[j + 2], ln.k[j + 3]);
    }
    sync_scratch();
    // B. merge levels above one block
    float acc = 0.f;
#pragma unroll 1
    for (int size = 2; size <= nbp; size <<= 1) {
      // element-wise exchanges between blocks: the flip (b against b ^ (size - 1), mirrored), then strides size / 4 .. 1
      auto sweep = [&](bool flip, int st) {
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
          const int b2 = flip ? (b ^ (size - 1)) : (b ^ st);
          if (b2 <= b || b2 >= nb) continue;              // each pair once, from its lower block; all-+inf partners: no-op
          float* lo = sl + (int64_t)b * BLK;
          float* hi = sl + (int64_t)b2 * BLK;
#pragma unroll 2
          for (int e = threadIdx.x * 4; e < BLK; e += NT * 4) {
            const float4 x = *reinterpret_cast<const float4*>(lo + e);
            float4 y;
            if (flip) {                                    // lo[e] against hi[BLK - 1 - e]
              const float4 t = *reinterpret_cast<const float4*>(hi + (BLK - 4 - e));
              y = make_float4(t.w, t.z, t.y, t.x);
            } else {
              y = *reinterpret_cast<const float4*>(hi + e);
            }
            const float4 mn = make_float4(fminf(x.x, y.x), fminf(x.y, y.y), fminf(x.z, y.z), fminf(x.w, y.w));
            const float4 mx = make_float4(fmaxf(x.x, y.x), fmaxf(x.y, y.y), fmaxf(x.z, y.z), fmaxf(x.w, y.w));
            *reinterpret_cast<float4*>(lo + e) = mn;
            if (flip) *reinterpret_cast<float4*>(hi + (BLK - 4 - e)) = make_float4(mx.w, mx.z, mx.y, mx.x);
            else *reinterpret_cast<float4*>(hi + e) = mx;
          }
        }
        sync_scratch();
      };
      sweep(true, 0);
      for (int st = size >> 2; st >= 1; st >>= 1) sweep(false, st);
      const bool last = size == nbp;
#pragma unroll 1
      for (int b = 0; b < nb; ++b) {
        const float* src = sl + (int64_t)b * BLK + w * CAP + lane * M;
#pragma unroll
        for (int j = 0; j < M; j += 4) {
          const float4 v = *reinterpret_cast<const float4*>(src + j);
          ln.k[j] = v.x; ln.k[j + 1] = v.y; ln.k[j + 2] = v.z; ln.k[j + 3] = v.w;
        }
        workgroup_merge_block<NW, M>(ln, xbuf, w, lane);
        if (last) {
          acc += unit_readout<M>(ln, b * BLK + w * CAP + lane * M, D, xif);
        } else {
          float* dst = sl + (int64_t)b * BLK + w * CAP + lane * M;
#pragma unroll
          for (int j = 0; j < M; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(ln.k[j], ln.k[j + 1], ln.k[j + 2], ln.k[j + 3]);
        }
      }
      sync_scratch();
    }
    acc = wave_sum_h(acc);
    if (lane == 0) red[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) tot += red[q];
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
    __syncthreads();
  }
}

// unit weights, tau <= 1: the rows above FSW_HUB_MAX_DEG.  scratch: fsw_embed_scratch_bytes(max_degree).
int launch_embed_mergepath(const fsw_embed_args& a, int bin_lo, int bin_hi, int dlo, int64_t rows_upper, hipStream_t stream);
int launch_embed_giant(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  // FSW_GIANT_MERGEPATH=1: sorted blocks of 8192 + merge-path levels (k_embed_mergepath) instead of the block sweeps below.  Measured
  // on the 64M-edge RMAT graph's 23 rows above 32768 neighbours: 6.33 ms against 6.07 ms here -- for bare keys the sweeps are cheap
  // (one v_min / v_max per element) and a merge-path tile is a chain of dependent LDS reads at two wavefronts per SIMD; with a
  // payload it is the other way round (k_embed_mergepath_w: 11.6 ms against 61.6 ms for the scratch-line kernel)
  if (getenv("FSW_GIANT_MERGEPATH")) return launch_embed_mergepath(a, FSW_BIN_GLOBAL, FSW_BIN_GLOBAL, FSW_HUB_MAX_DEG, rows_upper, stream);
  rows_upper = bin_rows_or(a, FSW_BIN_GLOBAL, FSW_BIN_GLOBAL, rows_upper);
  if (rows_upper <= 0 || (a.max_degree > 0 && a.max_degree <= FSW_HUB_MAX_DEG)) return 0;
  FSW_REQUIRE(a.max_degree > FSW_HUB_MAX_DEG, "fsw_embed_f32: max_degree (host value) is required for rows above FSW_HUB_MAX_DEG");
  FSW_REQUIRE(a.scratch, "fsw_embed_f32: rows above FSW_HUB_MAX_DEG need a scratch buffer (fsw_embed_scratch_bytes)");
  const int64_t line_floats = ceil_div(a.max_degree, kGiantBlk) * kGiantBlk;
  int64_t nwg = std::min<int64_t>((int64_t)(a.scratch_bytes / (size_t)(line_floats * 4)), 256 * (16 / kGiantNW));   // every CU full
  nwg = std::min<int64_t>(nwg, ceil_div(rows_upper * a.S, 8) * 8);
  if (nwg >= 8) nwg &= ~(int64_t)7;
  FSW_REQUIRE(nwg >= 1, "fsw_embed_f32: scratch buffer too small for rows above FSW_HUB_MAX_DEG (need fsw_embed_scratch_bytes(max_degree))");
  FSW_REQUIRE(((uintptr_t)a.scratch & 15) == 0, "fsw_embed_f32: scratch must be 16-byte aligned");
  k_embed_giant<<<(unsigned)nwg, kGiantNW * kWave, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, a.Xp, a.ldp, a.S, a.freqs, a.out, a.ldo,
                                                                a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale,
                                                                reinterpret_cast<float*>(a.scratch), line_floats);
  FSW_LAUNCH_CHECK();
  return 0;
}


// ---- rows above FSW_HUB_MAX_DEG, unit weights: sorted blocks of 8192 keys + merge-path levels (merge_path.h) -------------------
// Phase A is k_embed_hub<4, 32>'s line (four wavefronts x 32 keys per lane, sorted in registers, two bitonic levels through LDS)
// for every block of 8192 neighbours, parked in the workgroup's scratch line; the levels above are one merge-path pass each, the
// last one straight into the readout.  A 150 000-neighbour hub: 19 blocks, 5 passes (k_embed_giant: 10 blocks of 16384, 14 passes).
struct MpKeys {
  float k[kMpVT];
};

__global__ void __launch_bounds__(kMpNT, 2) k_embed_mergepath(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start,
                                                            int bin_lo, int bin_hi, int dlo, const float* __restrict__ Xp, int64_t ldp,
                                                            int S, const float* __restrict__ freqs, float* __restrict__ out, int64_t ldo,
                                                            const float* __restrict__ bias, float out_scale, int has_mass, int mass_fn,
                                                            float mass_scale, float* __restrict__ scratch, int64_t line_cap) {
  constexpr int NW = 4, M = 32, CAP = M * kWave;
  static_assert(NW * CAP == kMpBlk, "block = one workgroup's registers");
  extern __shared__ __attribute__((aligned(16))) float xsm[];   // phase A: exchange buffer [NW][CAP]; levels: tile keys | tile boundaries
  __shared__ float red[NW];
  float* tk = xsm;
  int* part = reinterpret_cast<int*>(xsm + kMpTileLds);
  const int pbeg = bin_start[bin_lo], nrows = bin_start[bin_hi + 1] - pbeg;
  const int lane = lane_id(), w = wave_id();
  const int blk = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
  float* k0 = scratch + (int64_t)blk * 2 * line_cap;
  float* k1 = k0 + line_cap;
  const int64_t nlines = (int64_t)nrows * S;
  for (int64_t line = blk; line < nlines; line += gridDim.x) {
    const int p = pbeg + (int)(line / S), k = (int)(line % S);
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    if (D <= dlo) continue;
    const int nb = (D + kMpBlk - 1) / kMpBlk;
    const float xif = freqs[k];
    float acc = 0.f;
    WaveLine<M, false> ln;
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
      gather_chunk<M>(ln, col + start, b * kMpBlk + w * CAP, D, Xp, ldp, k, lane);
      ln.sort();
      workgroup_merge_levels<NW, M>(ln, xsm, w, lane);
      if (nb == 1) break;                                  // the whole line is in registers
      float* dst = k0 + (int64_t)b * kMpBlk + w * CAP + lane * M;
#pragma unroll
      for (int j = 0; j < M; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(ln.k[j], ln.k[j + 1], ln.k[j + 2], ln.k[j + 3]);
    }
    if (nb == 1) {
      acc = unit_readout<M>(ln, w * CAP + lane * M, D, xif);
    } else {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __syncthreads();
      MpStamps st{};
      merge_path_levels<false>(k0, k1, nullptr, nullptr, nb, tk, nullptr, part, [&](int r0, const float* ok, const float*) {
        MpKeys t;
#pragma unroll
        for (int j = 0; j < kMpVT; ++j) t.k[j] = ok[j];
        acc += unit_readout<kMpVT>(t, r0, D, xif);
      }, st);
    }
    acc = wave_sum_h(acc);
    if (lane == 0) red[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) tot += red[q];
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
    __syncthreads();
  }
}

// unit weights, tau <= 1: rows of the bins bin_lo .. bin_hi with more than dlo neighbours.  scratch: fsw_embed_scratch_bytes(max_degree).
int launch_embed_mergepath(const fsw_embed_args& a, int bin_lo, int bin_hi, int dlo, int64_t rows_upper, hipStream_t stream) {
  rows_upper = bin_rows_or(a, bin_lo, bin_hi, rows_upper);
  if (rows_upper <= 0 || (a.max_degree > 0 && a.max_degree <= dlo)) return 0;
  FSW_REQUIRE(a.max_degree > dlo, "fsw_embed_f32: max_degree (host value) is required for rows above FSW_LDS_MAX_DEG");
  FSW_REQUIRE(a.scratch, "fsw_embed_f32: these rows need a scratch buffer (fsw_embed_scratch_bytes)");
  FSW_REQUIRE(((uintptr_t)a.scratch & 15) == 0, "fsw_embed_f32: scratch must be 16-byte aligned");
  const int64_t line_cap = ceil_div(a.max_degree, kMpBlk) * kMpBlk;
  int64_t nwg = std::min<int64_t>((int64_t)(a.scratch_bytes / (size_t)(2 * line_cap * 4)), 512);   // two workgroups per CU
  nwg = std::min<int64_t>(nwg, ceil_div(rows_upper * a.S, 8) * 8);
  if (nwg >= 8) nwg &= ~(int64_t)7;
  FSW_REQUIRE(nwg >= 1, "fsw_embed_f32: scratch buffer too small for rows above FSW_HUB_MAX_DEG (need fsw_embed_scratch_bytes(max_degree))");
  const size_t lds = sizeof(float) * 4 * 32 * kWave;      // phase A's exchange buffer; the tile + boundaries of the levels fit inside
  static_assert(sizeof(float) * kMpTileLds + sizeof(int) * (kMpParts + 1) <= sizeof(float) * 4 * 32 * kWave, "LDS of the merge levels");
  k_embed_mergepath<<<(unsigned)nwg, kMpNT, lds, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, bin_lo, bin_hi, dlo, a.Xp, a.ldp, a.S, a.freqs,
                                                          a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale,
                                                          reinterpret_cast<float*>(a.scratch), line_cap);
  FSW_LAUNCH_CHECK();
  return 0;
}

template <int NW, int M>
static int launch_hub(const fsw_embed_args& a, int bin, int64_t rows_upper, hipStream_t stream, int dmin = 0, int dmax = 0x7fffffff) {
  constexpr int LPB = NW == 1 ? 4 : 1;
  rows_upper = bin_rows_or(a, bin, bin, rows_upper);   // exact when the host knows the bins: an empty bin is not launched
  if (rows_upper <= 0) return 0;
  // virtual blocks = (rows rounded up to 8) x slices / lines per block; the launched grid is capped at 2^20 workgroups
  // (a dispatch holds < 2^32 work-items: 17.7M x 256 threads were silently truncated on a 64M-edge graph) and strides
  const int64_t nvirtual = ceil_div(ceil_div(rows_upper, 8) * a.S, LPB) * 8;
  const int64_t nblocks = std::min<int64_t>(nvirtual, 1ll << 20);
  if constexpr (NW == 1 && M % 16 == 0) {
    // 16-byte gathers dealt through LDS when the workgroup's four lines are four slices of one row (k_embed_hub_quad)
    if (a.S % 4 == 0 && a.ldp % 4 == 0 && ((uintptr_t)a.Xp & 15) == 0) {
      k_embed_hub_quad<M><<<(unsigned)nblocks, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, bin, a.Xp, a.ldp, a.S, a.freqs, a.out,
                                                                 a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale, dmin, dmax);
      FSW_LAUNCH_CHECK();
      return 0;
    }
  }
#ifndef FSW_HUB_Q4_MAXNW
#define FSW_HUB_Q4_MAXNW 2   // widest line (wavefronts) that takes the 16-byte gather + stash form
#endif
  if constexpr (NW >= 2 && NW <= FSW_HUB_Q4_MAXNW) {
    // 2049..4096 neighbours only: measured on the RMAT graphs 0.41 -> 0.36 ms (scale 20) for two wavefronts per line, but 10.2 -> 12.8 ms
    // (scale 22, 4097..8192) and no change (8193..16384) for four and eight -- there the barrier-separated exchanges between the
    // wavefronts bound the kernel, not the gather, and the stash traffic is pure cost.
    // 16-byte gathers with the other three slices stashed in the caller's scratch buffer (k_embed_hub_q4): the grid is bounded by
    // the stash (3 * NW * 64 * M floats per workgroup) and by what is resident anyway (the workgroups stride over the lines)
    const size_t per_wg = (size_t)3 * NW * kWave * M * sizeof(float);
    if (a.S % 4 == 0 && a.ldp % 4 == 0 && ((uintptr_t)a.Xp & 15) == 0 && a.scratch && ((uintptr_t)a.scratch & 15) == 0 &&
        a.scratch_bytes >= 8 * per_wg) {
      int64_t nq = ceil_div(ceil_div(rows_upper, 8) * (a.S / 4), 1) * 8;                 // virtual blocks: 8 rows x S / 4 slice quads
      nq = std::min<int64_t>(nq, std::min<int64_t>((int64_t)(a.scratch_bytes / per_wg), 256 * 16 / NW));
      nq &= ~(int64_t)7;
      k_embed_hub_q4<NW, M><<<(unsigned)nq, NW * kWave, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, bin, a.Xp, a.ldp, a.S, a.freqs,
                                                                   a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale,
                                                                   dmin, dmax, reinterpret_cast<float*>(a.scratch));
      FSW_LAUNCH_CHECK();
      return 0;
    }
  }
  k_embed_hub<NW, M><<<(unsigned)nblocks, NW == 1 ? 256 : NW * kWave, 0, stream>>>(
      a.rowptr, a.col, a.perm, a.bin_start, bin, a.Xp, a.ldp, a.S, a.freqs, a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn,
      a.mass_scale, dmin, dmax);
  FSW_LAUNCH_CHECK();
  return 0;
}

// a multi-wavefront class as two launches over its bin: rows of at most 3/4 of the class size on 24 keys per lane, the rest on 32
// (the single-wavefront classes branch per line inside one kernel; here the second code path would spill)
template <int NW>
static int launch_hub_pair(const fsw_embed_args& a, int bin, int64_t rows_upper, hipStream_t stream) {
  constexpr int kSmall = NW * kWave * 24;
  int rc;
  if (FSW_HUB_SPLIT && (rc = launch_hub<NW, 24>(a, bin, rows_upper, stream, 0, kSmall))) return rc;
  return launch_hub<NW, kHubM>(a, bin, rows_upper, stream, FSW_HUB_SPLIT ? kSmall + 1 : 0, 0x7fffffff);
}


// ---- 129..512 neighbours: SEVERAL lines per wavefront, LL lanes x M keys each ---------------------------------------------------
// With LL <= 16 lanes per line every cross-lane exchange of the merge levels is one DPP move inside a row of 16 lanes, and more of
// the network runs inside a lane (one instruction per key and comparator instead of two).  The 64 / LL lines of a wavefront are
// adjacent slices of one row, so a gather instruction reads 64 / LL consecutive floats from each of LL rows of Xp.
//   LL = 16, M = 32  : 257..512 neighbours (four lines per wavefront; a 64-lane line of 8 keys per lane ran 11.5 ms against 9.7)
//   LL = 4, M = 48/64: 129..256 neighbours (16 lines per wavefront).  These rows used to run lane = slice with the whole row in one
//                      lane's registers (embed_mid.hip: 160..256 keys -> one wave per SIMD, which can issue a vector instruction
//                      only every 4 cycles and has nothing to overlap its gather with); split over four lanes the line takes
//                      48..64 registers per lane and 3..4 waves per SIMD cover each other's gathers.
// 4 x 4 transpose across the four rows of 16 lanes: lane (s, j) holds v[c] = slice c of ITS element and receives v[c] = slice s of
// the element of lane (c, j).  Two v_permlane32_swap (rows {0,1} <-> {2,3}) and two v_permlane16_swap (even <-> odd rows): one
// instruction per key, no LDS crossbar.
__device__ __forceinline__ void transpose_rows_4x4(float (&v)[4]) {
  auto swap32 = [](float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
  };
  auto swap16 = [](float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
  };
  swap32(v[0], v[2]);
  swap32(v[1], v[3]);
  swap16(v[0], v[1]);
  swap16(v[2], v[3]);
}

// VEC4 (LL = 16, S % 4 == 0): the wavefront's four lines are slices k0 .. k0 + 3 of ONE row, k0 a multiple of 4.  Lane (s, j) then
// loads 16 bytes -- the four slices of its neighbour -- for a quarter of the line's elements and the values are dealt to the four
// lines in registers (transpose_rows_4x4).  A quarter of the load instructions, each request 16 bytes instead of 4: the 4-byte
// gathers of the long-row kernels are bound by the number of requests, not by bytes (gather alone: 531 G keys/s here, 322 G keys/s
// in the 64-lane lines of k_embed_hub, whatever the sort costs -- tools/exp_hub.sh).
template <int M, int LL, bool VEC4 = false>   // line = LL * M keys
__global__ void __launch_bounds__(256, M <= 32 ? 4 : M <= 64 ? 3 : 2) k_embed_rowlines(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start, int bin,
                                                           const float* __restrict__ Xp, int64_t ldp, int S, const float* __restrict__ freqs,
                                                           float* __restrict__ out, int64_t ldo, const float* __restrict__ bias,
                                                           float out_scale, int has_mass, int mass_fn, float mass_scale, int dmin,
                                                           int dmax) {
  constexpr int LPW = kWave / LL, LPB = 4 * LPW;   // lines per wavefront / per block
  const int pbeg = bin_start[bin], nrows = bin_start[bin + 1] - pbeg;
  const int lane = lane_id(), sub = lane & (LL - 1);
  const int xcd = blockIdx.x & 7;
  for (int64_t vb = blockIdx.x;; vb += gridDim.x) {
    const int64_t i = (vb >> 3) * LPB + wave_id() * LPW + lane / LL;
    const int64_t rl = i / S;
    const int k = (int)(i - rl * S);
    const int64_t r = rl * 8 + xcd;
    if (r >= nrows) return;          // per group of LL lanes; nothing below synchronises or leaves the group
    const int node = perm[pbeg + r];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    if (D < dmin || D > dmax) continue;   // instantiations of different line sizes may share a bin
    WaveLine<M, false, false, LL> ln;
    const int32_t* colrow = col + start;
    const float* xk = Xp + k;
    if constexpr (VEC4) {
      static_assert(LL == 16 && M % 4 == 0, "four lines of 16 lanes");
      const int sline = lane >> 4;
      const float* xq = Xp + (k - sline);           // slice k0 of the row: 16-byte aligned
      int c[M / 4];
#pragma unroll
      for (int i = 0; i < M / 4; ++i) {
        const int t = (i * 4 + sline) * LL + sub;
        c[i] = colrow[min(t, D - 1)];                // past the end: the last neighbour again (replaced by +inf below)
      }
#pragma unroll
      for (int i = 0; i < M / 4; ++i) {
        float v[4];
        if constexpr (FSW_HUB_ABL & 1) {
          v[0] = v[1] = v[2] = v[3] = (float)((c[i] * 2654435761u) >> 8);
        } else {
          const float4 q = *reinterpret_cast<const float4*>(xq + (int64_t)c[i] * ldp);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
        if ((i * 4 + sline) * LL + sub >= D) v[0] = v[1] = v[2] = v[3] = __builtin_inff();
        transpose_rows_4x4(v);
#pragma unroll
        for (int u = 0; u < 4; ++u) ln.k[i * 4 + u] = v[u];
      }
    } else {
    // striped element order (lane-contiguous col reads; the line is sorted next).  Column indices one batch ahead of their gathers.
    constexpr int G = M % 16 == 0 ? 16 : 8;
    static_assert(M % G == 0, "keys per lane: a multiple of 8");
    int c[2][G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int t = q * LL + sub;
      c[0][q] = t < D ? colrow[t] : -1;
    }
#pragma unroll
    for (int g = 0; g < M / G; ++g) {
      if (g + 1 < M / G) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int t = ((g + 1) * G + q) * LL + sub;
          c[(g + 1) & 1][q] = t < D ? colrow[t] : -1;
        }
      }
#pragma unroll
      for (int q = 0; q < G; ++q) {
        const int cc = c[g & 1][q];
        ln.k[g * G + q] = cc >= 0 ? ((FSW_HUB_ABL & 1) ? (float)((cc * 2654435761u) >> 8) : xk[(int64_t)cc * ldp]) : __builtin_inff();
      }
    }
    }
    if (!(FSW_HUB_ABL & 2)) ln.sort();
    float tot = unit_readout<M>(ln, sub * M, D, freqs[k]);
    if constexpr (LL >= 16) tot += xor_lane<8>(tot);
    if constexpr (LL >= 8) tot += xor_lane<4>(tot);
    if constexpr (LL >= 4) tot += xor_lane<2>(tot);
    if constexpr (LL >= 2) tot += xor_lane<1>(tot);
    if (sub == 0) {
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)D, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
  }
}

template <int M, int LL>
static int launch_rowlines_one(const fsw_embed_args& a, int bin, int64_t rows_upper, hipStream_t stream, int dmin, int dmax) {
  constexpr int LPB = 4 * (kWave / LL);
  rows_upper = bin_rows_or(a, bin, bin, rows_upper);
  if (rows_upper <= 0) return 0;
  const int64_t nvirtual = ceil_div(ceil_div(rows_upper, 8) * a.S, LPB) * 8;
  const int64_t nblocks = std::min<int64_t>(nvirtual, 1ll << 20);
  // 16-byte gathers: the four lines of a wavefront must be four slices of one row starting at a multiple of 4
  const bool vec4 = LL == 16 && a.S % 4 == 0 && a.ldp % 4 == 0 && ((uintptr_t)a.Xp & 15) == 0;
  if constexpr (LL == 16) {
    if (vec4) {
      k_embed_rowlines<M, LL, true><<<(unsigned)nblocks, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, bin, a.Xp, a.ldp, a.S,
                                                                          a.freqs, a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn,
                                                                          a.mass_scale, dmin, dmax);
      FSW_LAUNCH_CHECK();
      return 0;
    }
  }
  k_embed_rowlines<M, LL><<<(unsigned)nblocks, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, bin, a.Xp, a.ldp, a.S, a.freqs, a.out,
                                                                a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale, dmin, dmax);
  FSW_LAUNCH_CHECK();
  return 0;
}

static int launch_rowlines(const fsw_embed_args& a, int bin, int64_t rows_upper, hipStream_t stream) {
  // rows of at most 384 neighbours on 24 keys per lane: off by default -- on a graph whose class sits near its upper end (the 64M-edge
  // RMAT graph: fill 0.93) the extra pass over the bin costs 0.8 ms and finds nothing
  constexpr int kSmall = 16 * 24;
  constexpr bool kSplit = FSW_HUB_SPLIT && FSW_ROWLINES_SPLIT;
  int rc;
  if (kSplit && (rc = launch_rowlines_one<24, 16>(a, bin, rows_upper, stream, 0, kSmall))) return rc;
  return launch_rowlines_one<32, 16>(a, bin, rows_upper, stream, kSplit ? kSmall + 1 : 0, 0x7fffffff);
}

// ---- 129..256 neighbours, whole-row gathers through LDS ------------------------------------------------------------------------------
// A workgroup of four wavefronts takes (row, chunk of 64 slices).  The neighbours' 256-byte runs Xp[col, k0 .. k0 + 63] -- the
// access shape that reads HBM best -- go STRAIGHT into LDS (global_load_lds_dword: no register destination, row t of the tile =
// neighbour t, 64 slices), and they are issued for the NEXT row of the workgroup before the current row is sorted, so the gather runs
// under the sort.  The 64 lines of the tile are then read transposed, 4 lanes x DP / 4 keys per line (WaveLine<DP / 4, .., 4>: every
// exchange of the merge levels a DPP move), 16 lines per wavefront; row stride 72 floats makes both the DMA writes (64 consecutive
// floats) and the transposed reads (bank = 8 sub + line) conflict-free.  Against one lane per slice with the whole row in its
// registers (embed_mid.hip: one or two waves per SIMD, nothing to overlap the gather with) the line takes a quarter of the registers.
constexpr int kMidLdsStride = 72;

template <int DP>   // padded line: 192 or 256 keys
__global__ void __launch_bounds__(256, 2) k_embed_mid_lds(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const int32_t* __restrict__ perm, const int32_t* __restrict__ bin_start, int bin,
                                                          const float* __restrict__ Xp, int64_t ldp, int S, const float* __restrict__ freqs,
                                                          float* __restrict__ out, int64_t ldo, const float* __restrict__ bias,
                                                          float out_scale, int has_mass, int mass_fn, float mass_scale) {
  constexpr int M = DP / 4, LL = 4;
  __shared__ float xs[DP * kMidLdsStride];
  const int lane = lane_id(), w = wave_id();
  const int sub = lane & (LL - 1), line = lane >> 2;
  const int k0 = blockIdx.y * kWave;                       // first slice of the chunk
  const int kl = min(k0 + lane, S - 1);                    // the slice this lane GATHERS (past the end: slice S - 1 again)
  const int ks = min(k0 + w * 16 + line, S - 1);           // the slice this lane SORTS
  const bool ks_ok = k0 + w * 16 + line < S;
  const float xif = freqs[ks];
  const int pbeg = bin_start[bin], pend = bin_start[bin + 1];
  // this wavefront gathers neighbours t = w, w + 4, ...: lane i holds the column index of its i-th neighbour (one load per row)
  auto load_cols = [&](int p, int& D) {
    int c = 0;
    D = 0;
    if (p < pend) {
      const int node = perm[p];
      const int start = rowptr[node];
      D = rowptr[node + 1] - start;
      const int t = w + 4 * lane;
      c = col[start + min(t, D - 1)];
    }
    return c;
  };
  auto issue_gather = [&](int c, int D) {                  // D wave-uniform; neighbours past the end are not loaded (masked at the read)
    const int mine = (D - w + 3) >> 2;                     // this wavefront's neighbours
#pragma unroll 8
    for (int i = 0; i < DP / 4; ++i) {
      if (i < mine) {                                      // uniform
        const int ci = __builtin_amdgcn_readlane(c, i);
        const float* src = Xp + (int64_t)ci * ldp + kl;
        __builtin_amdgcn_global_load_lds(src, xs + (w + 4 * i) * kMidLdsStride, 4, 0, 0);
      }
    }
  };
  int p = pbeg + blockIdx.x;
  int Dn;
  int cn = load_cols(p, Dn);
  if (p < pend) issue_gather(cn, Dn);
  int D = Dn;
  int node = p < pend ? perm[p] : 0;
  cn = load_cols(p + gridDim.x, Dn);
  for (; p < pend; p += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wavefront's DMAs of row p (and the column indices of the next row)
    __syncthreads();                                       // ... and everybody else's
    WaveLine<M, false, false, LL> ln;
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const int t = i * LL + sub;
      const float v = xs[t * kMidLdsStride + w * 16 + line];
      ln.k[i] = t < D ? v : __builtin_inff();
    }
    __syncthreads();                                       // the tile has been read: the next row may land in it
    const int pn = p + gridDim.x;
    const int Dcur = D, nodecur = node;
    if (pn < pend) {
      issue_gather(cn, Dn);
      D = Dn;
      node = perm[pn];
    }
    cn = load_cols(pn + gridDim.x, Dn);
    ln.sort();
    float tot = unit_readout<M>(ln, sub * M, Dcur, xif);
    tot += xor_lane<2>(tot);
    tot += xor_lane<1>(tot);
    if (sub == 0 && ks_ok) {
      float* orow = out + (int64_t)nodecur * ldo;
      orow[has_mass + ks] = out_scale * (tot + (bias ? bias[has_mass + ks] : 0.f));
      if (has_mass && ks == 0) orow[0] = out_scale * (mass_encode_h((float)Dcur, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
  }
}

int launch_embed_mid_lds(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  constexpr int sizes[FSW_NUM_MID_BINS] = FSW_MID_SIZES;
  for (int i = 0; i < FSW_NUM_MID_BINS; ++i) {
    if (sizes[i] <= 128) continue;
    const int64_t rows = bin_rows_or(a, FSW_BIN_MID0 + i, FSW_BIN_MID0 + i, rows_upper);
    if (rows <= 0) continue;
    dim3 grid((unsigned)std::min<int64_t>(rows, 256), (unsigned)ceil_div(a.S, kWave));
    if (sizes[i] <= 192)
      k_embed_mid_lds<192><<<grid, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, FSW_BIN_MID0 + i, a.Xp, a.ldp, a.S, a.freqs, a.out,
                                                     a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale);
    else
      k_embed_mid_lds<256><<<grid, 256, 0, stream>>>(a.rowptr, a.col, a.perm, a.bin_start, FSW_BIN_MID0 + i, a.Xp, a.ldp, a.S, a.freqs, a.out,
                                                     a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn, a.mass_scale);
    FSW_LAUNCH_CHECK();
  }
  return 0;
}

// unit weights, mid bins of 129..256 neighbours (FSW_MID_SIZES 160 / 192 / 256): lines of 192 and 256 keys over LL lanes
#ifndef FSW_MIDSPLIT_LL
#define FSW_MIDSPLIT_LL 4
#endif
int launch_embed_mid_split(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  constexpr int LL = FSW_MIDSPLIT_LL;
  constexpr int sizes[FSW_NUM_MID_BINS] = FSW_MID_SIZES;
  int rc;
  for (int i = 0; i < FSW_NUM_MID_BINS; ++i) {
    if (sizes[i] <= 128) continue;
    if (sizes[i] <= 192) rc = launch_rowlines_one<192 / LL, LL>(a, FSW_BIN_MID0 + i, rows_upper, stream, 0, 0x7fffffff);
    else rc = launch_rowlines_one<256 / LL, LL>(a, FSW_BIN_MID0 + i, rows_upper, stream, 0, 0x7fffffff);
    if (rc) return rc;
  }
  return 0;
}

#endif   // FSW_HUB_PART == 0

// ---- general weights (and tau > 1): the same line-in-registers structure with the weight as payload --------------------------
// Replaces, for rows of 129 .. 4096 neighbours without edge features, the LDS-staged wave-sort kernels and the scratch-line kernel
// of embed_wsort.hip (13 .. 43 G keys/s on the 64M-edge RMAT graph).  A line holds D + 1 elements: the neighbours (key, weight)
// and the reference's pad element (key 0, weight max(tau - m, 0), fsw_embedding.py:1000-1017), so a class runs one size up from
// the unit kernels.  Readout as in k_embed_wsort: float64 cumulative weight (lane sum -> wave scan -> wavefront offsets), phase
// reduced in float64, sine in float32, coefficient = difference of consecutive sines.
__device__ __forceinline__ double wave_exclusive_scan_f64_h(double v, int lane) {
  double inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const double t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  return inc - v;
}

__device__ __forceinline__ float sin2pi_rev_h(double x) {
  const double r = x - rint(x);
  return sinpif(2.f * (float)r);
}

template <int M>
__device__ __forceinline__ void wave_exchange_w(WaveLine<M, true>& ln, float* __restrict__ xk, float* __restrict__ xw, int w, int lane,
                                                int partner, bool mirrored, bool lower) {
  constexpr int CAP = M * kWave;
  int moff = w * CAP + lane;                               // the offset is laundered, not the pointers (see wave_exchange)
  asm volatile("" : "+v"(moff));
  float* mk = xk + moff;
  float* mw = xw + moff;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    mk[j * kWave] = ln.k[j];
    mw[j * kWave] = ln.w[j];
  }
  __syncthreads();
  int off = partner * CAP + (mirrored ? kWave - 1 - lane : lane);
  asm volatile("" : "+v"(off));
  const float* tk = xk + off;
  const float* tw = xw + off;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int jj = (mirrored ? M - 1 - j : j) * kWave;
    const float ok = tk[jj], ow = tw[jj];
    const bool take = lower ? (ok < ln.k[j]) : (ok > ln.k[j]);   // ties: both wavefronts keep their own element
    ln.k[j] = take ? ok : ln.k[j];
    ln.w[j] = take ? ow : ln.w[j];
  }
  __syncthreads();
}

template <int NW, int M>
__global__ void __launch_bounds__(NW == 1 ? 256 : NW * kWave, M >= 24 ? 2 : 3) k_embed_hub_w(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ wgt, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ bin_start, int bin_lo, int bin_hi, const float* __restrict__ Xp, int64_t ldp, int S,
    const float* __restrict__ freqs, float tau, float* __restrict__ out, int64_t ldo, const float* __restrict__ bias, float out_scale,
    int has_mass, int mass_fn, float mass_scale, int dlo, int dhi) {
  constexpr int CAP = M * kWave;
  constexpr int LPB = NW == 1 ? 4 : 1;
  extern __shared__ __attribute__((aligned(16))) float xsm[];   // NW > 1: keys [NW][CAP] | weights [NW][CAP]
  float* xk = xsm;
  float* xw = xsm + (NW > 1 ? NW * CAP : 0);
  __shared__ double redd[NW];
  __shared__ float red[NW];
  const int pbeg = bin_start[bin_lo], nrows = bin_start[bin_hi + 1] - pbeg;
  const int lane = lane_id();
  const int w = NW == 1 ? 0 : wave_id();
  const int xcd = blockIdx.x & 7;
  for (int64_t vb = blockIdx.x;; vb += gridDim.x) {
    const int64_t i = (vb >> 3) * LPB + (NW == 1 ? wave_id() : 0);
    const int64_t rl = i / S;
    const int k = (int)(i - rl * S);
    const int64_t r = rl * 8 + xcd;
    if (r >= nrows) return;
    const int node = perm[pbeg + r];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    if (D <= dlo || D > dhi) continue;                       // another capacity class of this bin (launch_embed_hub_weighted_*)
    const int Dtot = D + 1;                                  // with the pad element; Dtot <= NW * CAP by the class bounds

    // gather (striped: element t0 + j * 64 + lane), total mass, pad element
    WaveLine<M, true> ln;
    const int t0 = w * CAP;
    int c[M];
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int t = t0 + j * kWave + lane;
      c[j] = t < D ? col[start + t] : -1;
      ln.w[j] = t < D ? (wgt ? wgt[start + t] : 1.f) : 0.f;
      part += (double)ln.w[j];
    }
#pragma unroll
    for (int j = 0; j < M; ++j) ln.k[j] = c[j] >= 0 ? Xp[(int64_t)c[j] * ldp + k] : __builtin_inff();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    double m = part;
    if constexpr (NW > 1) {
      if (lane == 0) redd[w] = part;
      __syncthreads();
      m = 0.0;
#pragma unroll
      for (int q = 0; q < NW; ++q) m += redd[q];
      __syncthreads();
    }
    const double taud = (double)tau;
    const double inv = 1.0 / fmax(m, taud);
    {
      const float padw = (float)fmax(taud - m, 0.0);         // zero weight unless the row is deficient
#pragma unroll
      for (int j = 0; j < M; ++j)
        if (t0 + j * kWave + lane == D) {
          ln.k[j] = 0.f;
          ln.w[j] = padw;
        }
    }
    ln.sort();
    if constexpr (NW > 1) {
#pragma unroll
      for (int size = 2; size <= NW; size <<= 1) {
        wave_exchange_w<M>(ln, xk, xw, w, lane, w ^ (size - 1), true, (w & (size >> 1)) == 0);
        for (int st = size >> 2; st >= 1; st >>= 1) wave_exchange_w<M>(ln, xk, xw, w, lane, w ^ st, false, (w & st) == 0);
        ln.merge_chunk();
      }
    }
    // readout: element (w, lane, j) has rank w * CAP + lane * M + j
    const float xif = freqs[k];
    const double xi = (double)xif;
    const bool lin = xif < 1e-30f;
    double lsum = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) lsum += (double)ln.w[j];
    double cw = wave_exclusive_scan_f64_h(lsum, lane);
    if constexpr (NW > 1) {
      const double wtot = __shfl(cw + lsum, kWave - 1);      // this wavefront's total
      if (lane == 0) redd[w] = wtot;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NW; ++q)
        if (q < w) cw += redd[q];
      __syncthreads();
    }
    float acc = 0.f;
    float sprev = lin ? 0.f : sin2pi_rev_h(xi * (cw * inv));
    const int r0 = w * CAP + lane * M;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const bool valid = r0 + j < Dtot;
      cw += (double)ln.w[j];
      if (lin) {
        acc += valid ? ln.w[j] * ln.k[j] : 0.f;
      } else {
        const float sn = sin2pi_rev_h(xi * (cw * inv));
        acc += valid ? (sn - sprev) * ln.k[j] : 0.f;
        sprev = sn;
      }
    }
    acc *= lin ? 2.f * (float)inv : (float)((1.0 + xi) / (kPiH * xi));
    float tot = wave_sum_h(acc);
    if constexpr (NW > 1) {
      if (lane == 0) red[w] = tot;
      __syncthreads();
      tot = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) tot += red[q];
      __syncthreads();
    }
    if (lane == 0 && w == 0) {
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)m, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
  }
}

// rows of the bins bin_lo .. bin_hi with dlo < degree <= dhi (the pad element must fit: dhi + 1 <= NW * M * 64)
template <int NW, int M>
static int launch_hub_w(const fsw_embed_args& a, int bin_lo, int bin_hi, int64_t rows_upper, hipStream_t stream, int dlo = 0,
                        int dhi = NW * M * kWave - 1) {
  constexpr int LPB = NW == 1 ? 4 : 1;
  static_assert(NW * M * kWave >= 2, "line capacity");
  dhi = std::min(dhi, NW * M * kWave - 1);
  rows_upper = bin_rows_or(a, bin_lo, bin_hi, rows_upper);
  if (rows_upper <= 0 || dlo >= dhi || (a.max_degree > 0 && a.max_degree <= dlo)) return 0;
  const size_t lds = NW > 1 ? sizeof(float) * 2 * NW * M * kWave : 0;
  if (lds + 1024 > 64 * 1024) FSW_SET_MAX_LDS_ONCE((&k_embed_hub_w<NW, M>), lds);   // the kernel's static LDS comes on top of the 64 KB default
  const int64_t nvirtual = ceil_div(ceil_div(rows_upper, 8) * a.S, LPB) * 8;
  const int64_t nblocks = std::min<int64_t>(nvirtual, 1ll << 20);
  k_embed_hub_w<NW, M><<<(unsigned)nblocks, NW == 1 ? 256 : NW * kWave, lds, stream>>>(
      a.rowptr, a.col, a.w, a.perm, a.bin_start, bin_lo, bin_hi, a.Xp, a.ldp, a.S, a.freqs, a.tau, a.out, a.ldo, a.bias, a.out_scale,
      a.has_mass, a.mass_fn, a.mass_scale, dlo, dhi);
  FSW_LAUNCH_CHECK();
  return 0;
}


#if FSW_HUB_PART == 2
#if FSW_MP_STAMPS
__device__ unsigned long long g_mp_stamps[512][4][16];   // [workgroup][wavefront][phase]: s_memtime ticks; [..][15]: lines
#endif
// ---- general weights, rows above kHubWMaxDeg: sorted (key, weight) blocks of 8192 + merge-path levels (merge_path.h) --------------
// Phase A is k_embed_hub_w<4, 32>'s line for every block of 8192 elements (the pad element is element D of the line), parked in the
// workgroup's scratch lines; one merge-path pass per level; the last pass feeds the readout, whose cumulative weight runs on from
// tile to tile (float64, as in k_embed_hub_w).  Replaces the scratch-line kernel of embed_wsort.hip for rows without edge features:
// 28 bitonic sweeps of a 150 000-neighbour line by ONE wavefront became 5 passes by a workgroup.
//
// readout of N consecutive ranks per thread, threads in rank order (wavefront w, lane, j): returns the thread's partial sum and
// advances `carry` (the cumulative weight before the workgroup's first element) by the workgroup's total
template <int N>
__device__ __forceinline__ float weighted_readout(const float* __restrict__ kk, const float* __restrict__ ww, int r0, int Dtot, double xi,
                                                  double inv, bool lin, double& carry, double* __restrict__ redd, int w, int lane) {
  constexpr int NW = kMpNT / kWave;
  double lsum = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) lsum += (double)ww[j];
  double cw = wave_exclusive_scan_f64_h(lsum, lane);
  const double wtot = __shfl(cw + lsum, kWave - 1);
  if (lane == 0) redd[w] = wtot;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int q = 0; q < NW; ++q) {
    if (q < w) cw += redd[q];
    tot += redd[q];
  }
  __syncthreads();
  cw += carry;
  carry += tot;
  float acc = 0.f;
  float sprev = lin ? 0.f : sin2pi_rev_h(xi * (cw * inv));
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const bool valid = r0 + j < Dtot;
    cw += (double)ww[j];
    if (lin) {
      acc += valid ? ww[j] * kk[j] : 0.f;
    } else {
      const float sn = sin2pi_rev_h(xi * (cw * inv));
      acc += valid ? (sn - sprev) * kk[j] : 0.f;
      sprev = sn;
    }
  }
  return acc;
}

__global__ void __launch_bounds__(kMpNT, 2) k_embed_mergepath_w(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ wgt, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ bin_start, int bin_lo, int bin_hi, int dlo, const float* __restrict__ Xp, int64_t ldp, int S,
    const float* __restrict__ freqs, float tau, float* __restrict__ out, int64_t ldo, const float* __restrict__ bias, float out_scale,
    int has_mass, int mass_fn, float mass_scale, float* __restrict__ scratch, int64_t line_cap) {
  constexpr int NW = 4, M = 32, CAP = M * kWave;
  static_assert(NW * CAP == kMpBlk && NW * kWave == kMpNT, "block = one workgroup's registers");
  extern __shared__ __attribute__((aligned(16))) float xsm[];   // phase A: keys [NW][CAP] | weights [NW][CAP]; levels: tiles | boundaries
  __shared__ double redd[NW];
  __shared__ float red[NW];
  float* xk = xsm;
  float* xw = xsm + NW * CAP;
  float* tk = xsm;
  float* tw = xsm + kMpTileLds;
  int* part = reinterpret_cast<int*>(xsm + 2 * kMpTileLds);
  const int pbeg = bin_start[bin_lo], nrows = bin_start[bin_hi + 1] - pbeg;
  const int lane = lane_id(), w = wave_id();
  const int blk = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
  float* k0 = scratch + (int64_t)blk * 4 * line_cap;
  float* k1 = k0 + line_cap;
  float* w0 = k1 + line_cap;
  float* w1 = w0 + line_cap;
  const int64_t nlines = (int64_t)nrows * S;
  MpStamps st{};
  st.start();
  for (int64_t line = blk; line < nlines; line += gridDim.x) {
    const int p = pbeg + (int)(line / S), k = (int)(line % S);
    const int node = perm[p];
    const int start = rowptr[node];
    const int D = rowptr[node + 1] - start;
    if (D <= dlo) continue;
    FSW_MP_MARK(st, 0);                                      // row header
    const int Dtot = D + 1;                                 // with the reference's pad element (fsw_embedding.py:1000-1017)
    const int nb = (Dtot + kMpBlk - 1) / kMpBlk;
    // total mass of the row: summed from the weights as the blocks load them (the pad element, whose weight needs it, is element
    // D of the line = in the LAST block: one workgroup reduction there instead of a pass of its own over the row's weights)
    double pm = 0.0, m = 0.0, inv = 0.0;
    float padw = 0.f;
    const double taud = (double)tau;
    const float xif = freqs[k];
    const double xi = (double)xif;
    const bool lin = xif < 1e-30f;
    WaveLine<M, true> ln;
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
      const int t0 = b * kMpBlk + w * CAP;
#pragma unroll
      for (int h = 0; h < M; h += 16) {                     // two batches of 16 gathers: the index registers are the budget
        int c[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int t = t0 + (h + j) * kWave + lane;
          c[j] = t < D ? col[start + t] : -1;
          ln.w[h + j] = t < D ? (wgt ? wgt[start + t] : 1.f) : 0.f;
          pm += (double)ln.w[h + j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) ln.k[h + j] = c[j] >= 0 ? Xp[(int64_t)c[j] * ldp + k] : __builtin_inff();
      }
      FSW_MP_MARK(st, 1);                                  // block gathered
      if (b == nb - 1) {                                   // every weight of the row has been read: its mass, then the pad element
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pm += __shfl_xor(pm, off);
        if (lane == 0) redd[w] = pm;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NW; ++q) m += redd[q];
        __syncthreads();
        inv = 1.0 / fmax(m, taud);
        padw = (float)fmax(taud - m, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j)
          if (t0 + j * kWave + lane == D) {
            ln.k[j] = 0.f;
            ln.w[j] = padw;
          }
      }
      ln.sort();
#pragma unroll
      for (int size = 2; size <= NW; size <<= 1) {
        wave_exchange_w<M>(ln, xk, xw, w, lane, w ^ (size - 1), true, (w & (size >> 1)) == 0);
        for (int st = size >> 2; st >= 1; st >>= 1) wave_exchange_w<M>(ln, xk, xw, w, lane, w ^ st, false, (w & st) == 0);
        ln.merge_chunk();
      }
      FSW_MP_MARK(st, 2);                                  // block sorted (registers + two exchange levels)
      if (nb == 1) break;
      const int64_t o = (int64_t)b * kMpBlk + w * CAP + lane * M;
#pragma unroll
      for (int j = 0; j < M; j += 4) {
        *reinterpret_cast<float4*>(k0 + o + j) = make_float4(ln.k[j], ln.k[j + 1], ln.k[j + 2], ln.k[j + 3]);
        *reinterpret_cast<float4*>(w0 + o + j) = make_float4(ln.w[j], ln.w[j + 1], ln.w[j + 2], ln.w[j + 3]);
      }
    }
    float acc = 0.f;
    double carry = 0.0;
    FSW_MP_MARK(st, 3);                                    // blocks parked in the scratch line
    if (nb == 1) {
      acc = weighted_readout<M>(ln.k, ln.w, w * CAP + lane * M, Dtot, xi, inv, lin, carry, redd, w, lane);
    } else {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __syncthreads();
      FSW_MP_MARK(st, 4);                                  // fence + barrier
      merge_path_levels<true>(k0, k1, w0, w1, nb, tk, tw, part, [&](int r0, const float* ok, const float* ow) {
        acc += weighted_readout<kMpVT>(ok, ow, r0, Dtot, xi, inv, lin, carry, redd, w, lane);
      }, st);
    }
    acc *= lin ? 2.f * (float)inv : (float)((1.0 + xi) / (kPiH * xi));
    acc = wave_sum_h(acc);
    if (lane == 0) red[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) tot += red[q];
      float* orow = out + (int64_t)node * ldo;
      orow[has_mass + k] = out_scale * (tot + (bias ? bias[has_mass + k] : 0.f));
      if (has_mass && k == 0) orow[0] = out_scale * (mass_encode_h((float)m, mass_fn) * mass_scale + (bias ? bias[0] : 0.f));
    }
    __syncthreads();
    FSW_MP_MARK(st, 10);                                   // workgroup sum + store
#if FSW_MP_STAMPS
    st.sum[15] += 1;
#endif
  }
#if FSW_MP_STAMPS
  if (lane == 0 && blockIdx.x < 512 && w < 4)
    for (int i = 0; i < 16; ++i) g_mp_stamps[blockIdx.x][w][i] += st.sum[i];
#endif
}

// general weights without edge features: rows of the bins bin_lo .. bin_hi with more than dlo neighbours
int launch_embed_mergepath_w(const fsw_embed_args& a, int bin_lo, int bin_hi, int dlo, int64_t rows_upper, hipStream_t stream) {
  rows_upper = bin_rows_or(a, bin_lo, bin_hi, rows_upper);
  if (rows_upper <= 0 || (a.max_degree > 0 && a.max_degree <= dlo)) return 0;
  FSW_REQUIRE(a.max_degree > dlo, "fsw_embed_f32: max_degree (host value) is required for rows above FSW_LDS_MAX_DEG");
  FSW_REQUIRE(a.scratch, "fsw_embed_f32: these rows need a scratch buffer (fsw_embed_scratch_bytes)");
  FSW_REQUIRE(((uintptr_t)a.scratch & 15) == 0, "fsw_embed_f32: scratch must be 16-byte aligned");
  const int64_t line_cap = ceil_div(a.max_degree + 1, kMpBlk) * kMpBlk;
  int64_t nwg = std::min<int64_t>((int64_t)(a.scratch_bytes / (size_t)(4 * line_cap * 4)), 512);   // two workgroups per CU
  nwg = std::min<int64_t>(nwg, ceil_div(rows_upper * a.S, 8) * 8);
  if (nwg >= 8) nwg &= ~(int64_t)7;
  FSW_REQUIRE(nwg >= 1, "fsw_embed_f32: scratch buffer too small for rows above FSW_LDS_MAX_DEG (need fsw_embed_scratch_bytes(max_degree))");
  const size_t lds = sizeof(float) * 2 * 4 * 32 * kWave;   // phase A's (key, weight) exchange buffers; the tiles + boundaries fit inside
  static_assert(sizeof(float) * 2 * kMpTileLds + sizeof(int) * (kMpParts + 1) <= sizeof(float) * 2 * 4 * 32 * kWave, "LDS of the merge levels");
  FSW_SET_MAX_LDS_ONCE((&k_embed_mergepath_w), lds);   // 64 KB of dynamic LDS + the kernel's static words
  k_embed_mergepath_w<<<(unsigned)nwg, kMpNT, lds, stream>>>(a.rowptr, a.col, a.w, a.perm, a.bin_start, bin_lo, bin_hi, dlo, a.Xp, a.ldp, a.S,
                                                            a.freqs, a.tau, a.out, a.ldo, a.bias, a.out_scale, a.has_mass, a.mass_fn,
                                                            a.mass_scale, reinterpret_cast<float*>(a.scratch), line_cap);
  FSW_LAUNCH_CHECK();
  return 0;
}
#if FSW_MP_STAMPS
}  // namespace fsw
// timing experiment only (not in include/fsw_hip.h): copies and clears the stamp sums of k_embed_mergepath_w ([512][4][16] words)
extern "C" int fsw_debug_mergepath_stamps(unsigned long long* out) {
  FSW_CHECK_HIP(hipDeviceSynchronize());
  FSW_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fsw::g_mp_stamps), sizeof(unsigned long long) * 512 * 4 * 16));
  void* p = nullptr;
  FSW_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(fsw::g_mp_stamps)));
  FSW_CHECK_HIP(hipMemset(p, 0, sizeof(unsigned long long) * 512 * 4 * 16));
  return 0;
}
namespace fsw {
#endif
#endif   // FSW_HUB_PART == 2

#define FSW_HW(NW, M, LO, HI, DLO, DHI) \
  if ((rc = launch_hub_w<NW, M>(a, LO, HI, rows_upper, stream, DLO, DHI))) return rc
#if FSW_HUB_PART == 1
// general weights without edge features: rows of FSW_MID_MAX_DEG_WEIGHTED < degree <= 4096 (bins bin_lo .. FSW_BIN_HUB0).
// bin_lo: the first mid bin above FSW_MID_MAX_DEG_WEIGHTED.  lds_rows / hub_rows bound the rows of the two ranges.
int launch_embed_hub_weighted_lds(const fsw_embed_args& a, int bin_lo, int64_t rows_upper, hipStream_t stream) {
  // A line holds D + 1 elements, and a bitonic line costs its CAPACITY, not its fill: every bin is split over the capacities
  // 64 * {3, 4, 6, 8, 12, 16, 24} (x 2 wavefronts) by a degree window inside the kernel -- a row of 300 neighbours runs on 384
  // wires instead of 768, the one row of exactly 512 (513 elements) on 768.  Rows outside a launch's window cost two loads.
  int rc;
  constexpr int B160 = FSW_BIN_LDS0 - 3, B192 = FSW_BIN_LDS0 - 2, B256 = FSW_BIN_LDS0 - 1;   // mid bins ..160, ..192, ..256
  constexpr int B40 = FSW_BIN_MID0, B64 = FSW_BIN_MID0 + 2, B128 = FSW_BIN_MID0 + 5;          // ..40, ..64, ..128
  if (bin_lo <= B64) FSW_HW(1, 1, std::max(bin_lo, B40), B64, 0, 63);
  if (bin_lo <= B128) FSW_HW(1, 2, std::max(bin_lo, B64), B128, 63, 127);
  if (bin_lo <= B128) FSW_HW(1, 3, B128, B128, 127, 191);
  if (bin_lo <= B160) FSW_HW(1, 3, std::max(bin_lo, B160), B160, 0, 191);
  if (bin_lo <= B256) FSW_HW(1, 4, std::max(bin_lo, B192), B256, 0, 255);
  if (bin_lo <= B256) FSW_HW(1, 6, B256, B256, 255, 383);
  FSW_HW(1, 6, FSW_BIN_LDS0, FSW_BIN_LDS0, 0, 383);               // 257 .. 512
  FSW_HW(1, 8, FSW_BIN_LDS0, FSW_BIN_LDS0, 383, 511);
  FSW_HW(1, 12, FSW_BIN_LDS0, FSW_BIN_LDS0, 511, 767);
  FSW_HW(1, 12, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1, 0, 767);      // 513 .. 1024
  FSW_HW(1, 16, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1, 767, 1023);
  FSW_HW(1, 24, FSW_BIN_LDS0 + 1, FSW_BIN_LDS0 + 1, 1023, 1535);
  FSW_HW(1, 24, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2, 0, 1535);     // 1025 .. 2048
  FSW_HW(2, 16, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2, 1535, 2047);
  FSW_HW(2, 24, FSW_BIN_LDS0 + 2, FSW_BIN_LDS0 + 2, 2047, 3071);
  return 0;
}
#elif FSW_HUB_PART == 2
// general weights, 2049 .. kHubWMaxDeg (fsw_common.h) neighbours: two and four wavefronts per line.  Above, the scratch-line kernel of
// embed_wsort.hip stays: sixteen wavefronts x 12 keys (or eight x 24) per line measured 124 (159) ms on the RMAT graph's class
// 4097..8192 against its 97 ms -- one workgroup per CU at 96 KB of exchange buffer does not cover the barrier-separated exchanges
int launch_embed_hub_weighted_hub(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  int rc;
  FSW_HW(2, 24, FSW_BIN_HUB0, FSW_BIN_HUB0, 0, 3071);             // 2049 .. 4096
  FSW_HW(4, 16, FSW_BIN_HUB0, FSW_BIN_HUB0, 3071, 4095);
  FSW_HW(4, 24, FSW_BIN_HUB0, FSW_BIN_HUB0, 4095, 6143);
  FSW_HW(4, 24, FSW_BIN_HUB0 + 1, FSW_BIN_HUB0 + 1, 0, 6143);     // 4097 .. 8191 (8192 itself: scratch-line kernel)
  FSW_HW(4, 32, FSW_BIN_HUB0 + 1, FSW_BIN_HUB0 + 1, 6143, kHubWMaxDeg);
  return 0;
}
#undef FSW_HW

#else
// unit weights, tau <= 1: rows of the four hub bins.  rows_upper bounds the rows above FSW_LDS_MAX_DEG (the per-bin counts
// stay on the device: surplus blocks exit at once).
int launch_embed_hub(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  int rc;
  const int64_t md = a.max_degree;   // host value, <= 0 when unknown
  if ((rc = launch_hub_pair<2>(a, FSW_BIN_HUB0, rows_upper, stream))) return rc;
#ifndef FSW_HUB_WIDE2
#define FSW_HUB_WIDE2 0   // 1: 4097..8192 neighbours on TWO wavefronts x 48 / 64 keys per lane (16-byte gathers + stash, one exchange level)
#endif
  if (FSW_HUB_WIDE2) {
    if (md <= 0 || md > 4096) {
      if ((rc = launch_hub<2, 48>(a, FSW_BIN_HUB0 + 1, rows_upper, stream, 0, 2 * kWave * 48))) return rc;
      if ((rc = launch_hub<2, 64>(a, FSW_BIN_HUB0 + 1, rows_upper, stream, 2 * kWave * 48 + 1, 0x7fffffff))) return rc;
    }
  } else if ((md <= 0 || md > 4096) && (rc = launch_hub_pair<4>(a, FSW_BIN_HUB0 + 1, rows_upper, stream))) return rc;
  if ((md <= 0 || md > 8192) && (rc = launch_hub_pair<8>(a, FSW_BIN_HUB0 + 2, rows_upper, stream))) return rc;
  if ((md <= 0 || md > 16384) && (rc = launch_hub_pair<16>(a, FSW_BIN_HUB0 + 3, rows_upper, stream))) return rc;
  return 0;
}

// unit weights, tau <= 1: the wave-sort classes 257..512 / ..1024 / ..2048 (one wavefront per line, 8 / 16 / 32 keys per lane).
// rows_upper bounds the rows of FSW_REG_MAX_DEG < degree <= FSW_LDS_MAX_DEG.
int launch_embed_ws_unit(const fsw_embed_args& a, int64_t rows_upper, hipStream_t stream) {
  if (rows_upper <= 0) return 0;
  int rc;
  const int64_t md = a.max_degree;
  if ((md <= 0 || md > FSW_MID_MAX_DEG) &&
      (rc = FSW_HUB_ROWLINES ? launch_rowlines(a, FSW_BIN_LDS0, rows_upper, stream) : launch_hub<1, 8>(a, FSW_BIN_LDS0, rows_upper, stream)))
    return rc;
  if ((md <= 0 || md > 512) && (rc = launch_hub<1, 16>(a, FSW_BIN_LDS0 + 1, rows_upper, stream))) return rc;
  if ((md <= 0 || md > 1024) && (rc = launch_hub<1, 32>(a, FSW_BIN_LDS0 + 2, rows_upper, stream))) return rc;
  return 0;
}

#endif

}  // namespace fsw
