// Merge levels above one sorted block for the longest neighbourhoods: merge path.  gfx950.
//
// A line longer than one workgroup's registers is sorted in blocks of kMpBlk elements (phase A of the callers: four wavefronts x
// 32 keys per lane, bitonic inside the workgroup) that are parked in a scratch line.  The levels above a block used to be bitonic
// too: every level re-reads and re-writes the whole line once per stride (element-wise min / max sweeps) plus once for the
// in-block tail -- O(n log n) work and log(n / block) + 1 passes PER LEVEL (k_embed_giant: 14 passes for a 150 000-neighbour hub,
// the (key, weight) scratch-line kernel 28).  Merging two SORTED runs is O(n): here a level is ONE pass.  Per level
//   1. one thread per tile boundary finds, by binary search along its diagonal of the (A, B) merge grid, how many elements of A
//      precede output position d (merge_path_split: A[i] goes before B[j] iff A[i] <= B[j], i.e. ties keep A first);
//   2. per tile of kMpTile = 256 threads x 16 consecutive outputs: the A-part and the B-part the tile needs (together exactly
//      kMpTile elements) are staged in LDS with coalesced loads, every thread splits the tile at its own diagonal (the same search
//      in LDS) and merges 16 outputs serially out of LDS;
//   3. the outputs leave as 64 contiguous bytes per thread -- or, on the LAST level, never leave: the caller's `consume` receives
//      16 consecutive ranks per thread and folds them into the readout.
// Runs ping-pong between two scratch lines; a run without a partner at some level is copied.  Everything is keyed on element
// VALUES only, so equal keys may leave in any order of their original positions -- the readout is invariant under that (tied keys
// contribute key * (sum of their coefficients), whatever the assignment).
#pragma once
#include "fsw_common.h"

#ifndef FSW_MP_STAMPS
#define FSW_MP_STAMPS 0   // 1: s_memtime stamps of the merge-path kernels' phases (tools/exp_mergepath_stamps.py)
#endif

namespace fsw {

#if FSW_MP_STAMPS
struct MpStamps {
  unsigned long long sum[16];
  unsigned long long last;
  __device__ __forceinline__ void start() { last = clock64(); }
  __device__ __forceinline__ void mark(int i) {
    const unsigned long long now = clock64();
    sum[i] += now - last;
    last = now;
  }
};
#define FSW_MP_MARK(st, i) (st).mark(i)
#else
struct MpStamps {
  __device__ __forceinline__ void start() {}
};
#define FSW_MP_MARK(st, i) do { } while (0)
#endif

constexpr int kMpNT = 256;                  // threads per workgroup
constexpr int kMpVT = 16;                   // outputs per thread and tile
constexpr int kMpTile = kMpNT * kMpVT;      // 4096
constexpr int kMpBlk = 8192;                // elements of one sorted block (a multiple of kMpTile)
constexpr int kMpParts = 512;               // tile boundaries held in LDS at a time
constexpr int kMpTileLds = kMpTile + kMpTile / 16;   // floats of LDS per staged tile (mp_pad)
static_assert(kMpBlk % kMpTile == 0, "tiles must not straddle runs");

// elements of A among the first d outputs of merge(A[0..nA), B[0..nB)), ties: A first.  0 <= d <= nA + nB.
__device__ __forceinline__ int merge_path_split(const float* A, int nA, const float* B, int nB, int d) {
  int lo = max(0, d - nB), hi = min(d, nA);
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (A[mid] <= B[d - 1 - mid]) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// LDS index of tile element i: one spare word per 16, so that threads whose read positions are ~16 (or ~8) elements apart -- thread t
// starts its serial merge near element 16 t -- fall into different banks (unpadded: 16-way conflicts on every read of the merge)
__device__ __forceinline__ int mp_pad(int i) { return i + (i >> 4); }

// merge_path_split on a staged tile: A = elements 0 .. na - 1, B = elements na .. na + nbb - 1
__device__ __forceinline__ int merge_path_split_tile(const float* tk, int na, int nbb, int d) {
  int lo = max(0, d - nbb), hi = min(d, na);
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tk[mp_pad(mid)] <= tk[mp_pad(na + d - 1 - mid)]) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// nb sorted blocks in k0 (weights in w0 when WEIGHTED) -> one sorted line, handed to consume(rank0, keys[kMpVT], weights[kMpVT])
// sixteen consecutive ranks per thread (every thread of the workgroup calls it once per tile: it may synchronise).  k1 / w1: the
// second line of the ping-pong.  tk / tw: kMpTileLds floats of LDS each, part: kMpParts + 1 ints.  nb >= 2.
template <bool WEIGHTED, class Consume>
__device__ __forceinline__ void merge_path_levels(float* k0, float* k1, float* w0, float* w1, int nb, float* tk, float* tw, int* part,
                                                  Consume&& consume, MpStamps& st) {
  const int tid = threadIdx.x;
  const int total = nb * kMpBlk;
  float *sk = k0, *dk = k1, *sw = w0, *dw = w1;
  for (int R = kMpBlk;; R <<= 1) {   // R <= total / 2 < 2^30 at the top of every iteration (the last level leaves by `break`)
    const int nruns = (total + R - 1) / R;              // >= 2
    const bool last = nruns == 2;
    const int covered = (int)min((int64_t)total, (int64_t)(nruns >> 1) * 2 * R);   // elements that belong to a pair of runs
    const unsigned pair_mask = 2u * (unsigned)R - 1u;   // a pair of runs is 2 R elements, R a power of two: start of a position's pair
    const int ntiles = covered / kMpTile;
    for (int g0 = 0; g0 < ntiles; g0 += kMpParts) {
      const int cnt = min(kMpParts, ntiles - g0);
      // elements of A before the start of tiles g0 .. g0 + cnt (0 at the first tile of a pair)
      for (int i = tid; i <= cnt; i += kMpNT) {
        const int pos = (g0 + i) * kMpTile;
        int v = 0;
        if (pos < covered) {
          const int pb = (int)((unsigned)pos & ~pair_mask), d = pos - pb;
          if (d) v = merge_path_split(sk + pb, R, sk + pb + R, min(R, total - pb - R), d);
        }
        part[i] = v;
      }
      __syncthreads();
      FSW_MP_MARK(st, 5);                                 // tile boundaries of the level (binary searches in the scratch line)
      // geometry of tile i of this chunk: start of its pair, offset in the pair, A- and B-parts
      struct TileGeo {
        int pos, pb, a0, b0, na;
      };
      auto geometry = [&](int i) {
        TileGeo t;
        t.pos = (g0 + i) * kMpTile;
        t.pb = (int)((unsigned)t.pos & ~pair_mask);
        const int d0 = t.pos - t.pb;
        const int nB = min(R, total - t.pb - R);
        t.a0 = part[i];
        const int a1 = (d0 + kMpTile >= R + nB) ? R : part[i + 1];   // the pair ends with this tile: all of A is before its end
        t.b0 = d0 - t.a0;
        t.na = a1 - t.a0;
        return t;
      };
      // the tile's elements (A-part, then B-part), element tid + u * 256 in register u: the loads of tile i + 1 are issued before
      // tile i is merged out of LDS, so the round trip to the scratch line hides behind the merge
      float pk[kMpVT], pw[WEIGHTED ? kMpVT : 1];
      auto fetch = [&](const TileGeo& t) {
#pragma unroll
        for (int u = 0; u < kMpVT; ++u) {
          const int e = tid + u * kMpNT;
          const int src = e < t.na ? t.pb + t.a0 + e : t.pb + R + t.b0 + (e - t.na);
          pk[u] = sk[src];
          if constexpr (WEIGHTED) pw[u] = sw[src];
        }
      };
      fetch(geometry(0));
      for (int i = 0; i < cnt; ++i) {
        const TileGeo g = geometry(i);
        const int pos = g.pos, na = g.na, nbb = kMpTile - g.na;
#pragma unroll
        for (int u = 0; u < kMpVT; ++u) {
          tk[mp_pad(tid + u * kMpNT)] = pk[u];
          if constexpr (WEIGHTED) tw[mp_pad(tid + u * kMpNT)] = pw[u];
        }
        __syncthreads();
        FSW_MP_MARK(st, 6);                               // tile staged in LDS (waited for its loads)
        if (i + 1 < cnt) fetch(geometry(i + 1));
        const int dd = tid * kMpVT;
        int ia = merge_path_split_tile(tk, na, nbb, dd);
        int ib = dd - ia;
        float ok[kMpVT], ow[WEIGHTED ? kMpVT : 1];
        float ka = tk[mp_pad(min(ia, kMpTile - 1))], kb = tk[mp_pad(min(na + ib, kMpTile - 1))];
        float wa = 0.f, wb = 0.f;
        if constexpr (WEIGHTED) {
          wa = tw[mp_pad(min(ia, kMpTile - 1))];
          wb = tw[mp_pad(min(na + ib, kMpTile - 1))];
        }
#pragma unroll
        for (int j = 0; j < kMpVT; ++j) {
          const bool ta = ia < na && (ib >= nbb || ka <= kb);     // a value read past the end of its part is never compared
          ok[j] = ta ? ka : kb;
          if constexpr (WEIGHTED) ow[j] = ta ? wa : wb;
          ia += ta ? 1 : 0;
          ib += ta ? 0 : 1;
          const int nx = mp_pad(min(ta ? ia : na + ib, kMpTile - 1));   // the next element of the part that gave this output
          const float kn = tk[nx];
          ka = ta ? kn : ka;
          kb = ta ? kb : kn;
          if constexpr (WEIGHTED) {
            const float wn = tw[nx];
            wa = ta ? wn : wa;
            wb = ta ? wb : wn;
          }
        }
        FSW_MP_MARK(st, 7);                               // split at the thread's diagonal + serial merge
        if (last) {
          consume(pos + dd, ok, ow);                             // one pair is left: pb == 0, pos is the rank of the tile's first output
        } else {
          float4* ok4 = reinterpret_cast<float4*>(dk + pos + dd);
#pragma unroll
          for (int j = 0; j < kMpVT; j += 4) ok4[j >> 2] = make_float4(ok[j], ok[j + 1], ok[j + 2], ok[j + 3]);
          if constexpr (WEIGHTED) {
            float4* ow4 = reinterpret_cast<float4*>(dw + pos + dd);
#pragma unroll
            for (int j = 0; j < kMpVT; j += 4) ow4[j >> 2] = make_float4(ow[j], ow[j + 1], ow[j + 2], ow[j + 3]);
          }
        }
        __syncthreads();                                          // the tile buffers are overwritten next
        FSW_MP_MARK(st, 8);                               // readout / stores of the outputs + barrier
      }
    }
    if (last) break;
    if (nruns & 1) {                                              // a run without a partner moves on unchanged
      for (int e = covered + tid * 4; e < total; e += kMpNT * 4) {
        *reinterpret_cast<float4*>(dk + e) = *reinterpret_cast<const float4*>(sk + e);
        if constexpr (WEIGHTED) *reinterpret_cast<float4*>(dw + e) = *reinterpret_cast<const float4*>(sw + e);
      }
    }
    // the next level reads what other wavefronts of this workgroup wrote: workgroup scope is enough (one CU, one L1, stores
    // write through), an agent-scope fence would write back the XCD's L2
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    float* t = sk; sk = dk; dk = t;
    t = sw; sw = dw; dw = t;
    FSW_MP_MARK(st, 9);                                   // copy of a run without a partner, fence, barrier
  }
}

}  // namespace fsw
