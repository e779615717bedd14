// Shared host/device helpers for libfsw_hip.so (gfx950 only).
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/fsw_hip.h"

namespace fsw {

void set_error(const char* fmt, ...);

#define FSW_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      fsw::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)

#define FSW_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      fsw::set_error(__VA_ARGS__);        \
      return 1;                           \
    }                                     \
  } while (0)

#define FSW_LAUNCH_CHECK() FSW_CHECK_HIP(hipGetLastError())

// Function attributes (the dynamic-LDS ceiling of a kernel) belong to a DEVICE: one flag per device ordinal and call site, so
// a process that uses several GPUs sets them on each (racing host threads at worst set the same value twice).
struct PerDeviceOnce {
  std::atomic<bool> done[64];
};
#define FSW_SET_MAX_LDS_ONCE(kernel, bytes)                                                                          \
  do {                                                                                                               \
    static fsw::PerDeviceOnce once_;                                                                                 \
    int dev_ = 0;                                                                                                    \
    FSW_CHECK_HIP(hipGetDevice(&dev_));                                                                              \
    if (dev_ < 0 || dev_ >= 64 || !once_.done[dev_].load(std::memory_order_relaxed)) {                               \
      FSW_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      if (dev_ >= 0 && dev_ < 64) once_.done[dev_].store(true, std::memory_order_relaxed);                           \
    }                                                                                                                \
  } while (0)

constexpr int kWave = 64;

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// wave index inside the workgroup as a provably wave-uniform (SGPR) value
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__host__ __device__ inline int degree_bin(int deg) {
  if (deg <= FSW_REG_MAX_DEG) return deg;
  if (deg > FSW_HUB_MAX_DEG) return FSW_BIN_GLOBAL;
  if (deg > FSW_LDS_MAX_DEG) return FSW_BIN_HUB0 + (deg > 4096) + (deg > 8192) + (deg > 16384);
  if (deg > FSW_MID_MAX_DEG) return FSW_BIN_LDS0 + (deg > 512) + (deg > 1024);
  constexpr int sizes[FSW_NUM_MID_BINS] = FSW_MID_SIZES;
  int i = 0;
#pragma unroll
  for (int j = 0; j < FSW_NUM_MID_BINS - 1; ++j) i += deg > sizes[j];
  return FSW_BIN_MID0 + i;
}

// general weights without edge features: rows of up to this many neighbours (+ the pad element = 8192 = four wavefronts x 32 keys per
// lane) keep their (key, weight) line in registers (embed_hub.hip: k_embed_hub_w); above, the scratch-line kernel of embed_wsort.hip
constexpr int kHubWMaxDeg = 8191;

// general weights without edge features: index (into FSW_MID_SIZES) of the first mid bin that runs on the (key, weight) lines of
// embed_hub.hip (k_embed_hub_w) instead of the per-lane (key, weight) network of embed_mid.hip.  FSW_W_HUB_FROM = 33 | 65 | 129 in
// the environment overrides the default (timing experiments).
#ifndef FSW_W_HUB_FROM_DEFAULT
#define FSW_W_HUB_FROM_DEFAULT 129
#endif
inline int weighted_hub_first_mid_bin() {
  static const int idx = [] {
    const char* e = getenv("FSW_W_HUB_FROM");
    const int from = e ? atoi(e) : FSW_W_HUB_FROM_DEFAULT;
    constexpr int sizes[FSW_NUM_MID_BINS] = FSW_MID_SIZES;
    int i = 0;
    while (i < FSW_NUM_MID_BINS && sizes[i] < from) ++i;   // first bin whose rows can have `from` neighbours or more
    return std::min(i, 6);                                  // bins above FSW_MID_MAX_DEG_WEIGHTED never take the per-lane network
  }();
  return idx;
}

// rows of the degree bins lo .. hi when the caller passed the host copy of bin_start, `upper` (a bound) otherwise
inline int64_t bin_rows_or(const fsw_embed_args& a, int lo, int hi, int64_t upper) {
  return a.bin_start_host ? (int64_t)a.bin_start_host[hi + 1] - a.bin_start_host[lo] : upper;
}

__host__ __device__ inline uint32_t pow2ceil(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// sin and cos of 2 pi x for x in REVOLUTIONS (|x| < 2^40), absolute error ~1e-15: reduction to [-1/8, 1/8] revolutions around the
// nearest quarter turn (exact in float64), Taylor polynomials on |angle| <= pi/4, quadrant fix-up.  About a third of the
// instructions of sincospi() -- the long-row kernels evaluate two of these per lane and row to start a recurrence.
__device__ __forceinline__ void sincos_rev(double x, double* s, double* c) {
  const double r = x - rint(x);                 // [-1/2, 1/2]
  const double qd = rint(4.0 * r);              // -2 .. 2
  const double a = (r - 0.25 * qd) * 6.283185307179586476925;
  const double a2 = a * a;
  double sp = 1.0 / 6227020800.0;
  sp = fma(sp, a2, -1.0 / 39916800.0);
  sp = fma(sp, a2, 1.0 / 362880.0);
  sp = fma(sp, a2, -1.0 / 5040.0);
  sp = fma(sp, a2, 1.0 / 120.0);
  sp = fma(sp, a2, -1.0 / 6.0);
  const double sa = fma(sp * a2, a, a);
  double cp = -1.0 / 87178291200.0;
  cp = fma(cp, a2, 1.0 / 479001600.0);
  cp = fma(cp, a2, -1.0 / 3628800.0);
  cp = fma(cp, a2, 1.0 / 40320.0);
  cp = fma(cp, a2, -1.0 / 720.0);
  cp = fma(cp, a2, 1.0 / 24.0);
  cp = fma(cp, a2, -0.5);
  const double ca = fma(cp, a2, 1.0);
  const int q = (int)qd & 3;                    // angle = a + q pi / 2
  const double s1 = (q & 1) ? ca : sa, c1 = (q & 1) ? sa : ca;
  *s = (q & 2) ? -s1 : s1;
  *c = (q == 1 || q == 2) ? -c1 : c1;
}

// Unit weights (c_t = t / D): coefficient of rank r in a neighbourhood of D at frequency xi,
//   (1 + xi) [sin(2 pi xi (r + 1) / D) - sin(2 pi xi r / D)] / (pi xi)  =  B cos(2 pi step (r + 1/2)),   step = xi / D,
//   B = (1 + xi) / (pi xi) * 2 sin(pi step)
// (reference fsw_embedding.py:1047-1075, 1109 with weights 1 / D, by sum-to-product).  The cosines of consecutive ranks obey
// c_{r+1} = 2 cos(2 pi step) c_r - c_{r-1}: ONE float64 FMA per rank (rounding error after m steps <= m^2 2^-53; every lane
// restarts from exact values at its first rank).
struct UnitCoef {
  double twoc, cur, prev;   // 2 cos(2 pi step), cos(phi_r), cos(phi_{r-1}) for the next rank r
  float B;
  __device__ __forceinline__ void start(double xi, int D, int r0) {
    const double step = xi / (double)D;          // revolutions per rank
    double sh, ch, s0, c0;
    sincos_rev(0.5 * step, &sh, &ch);
    sincos_rev(step * ((double)r0 + 0.5), &s0, &c0);
    const double sd = 2.0 * sh * ch, cd = fma(-2.0 * sh, sh, 1.0);
    twoc = 2.0 * cd;
    cur = c0;
    prev = fma(c0, cd, s0 * sd);                 // cos(phi - theta)
    B = (float)((1.0 + xi) / (3.14159265358979323846 * xi) * 2.0 * sh);
  }
  __device__ __forceinline__ float next() {     // cos(phi_r) of the current rank, then advance
    const float v = (float)cur;
    const double n = fma(twoc, cur, -prev);
    prev = cur;
    cur = n;
    return v;
  }
};

// Workgroup index b -> (degree D, perm range [p, pe)) for the kernels that walk the degree bins DHI, DHI - 1, .. DLO in tiles of ROWS
// rows (one degree per workgroup, the longest rows first).  ONE vector load of the bin table (lane l: bin_start[l]) and a wave-level
// suffix sum.  The loop this replaces read bin_start[D], bin_start[D + 1] with scalar loads, bin after bin until it found the tile:
// ~20 DEPENDENT loads, and under the kernels' col-index stream (every neighbour index is a scalar load) they miss the scalar cache --
// the s_memtime stamps of tools/exp_fused_stamps.py put 19 us of a workgroup's 91 us before its first gather.
template <int ROWS>
__device__ __forceinline__ bool find_degree_tile(const int32_t* __restrict__ bin_start, int DLO, int DHI, int b, int& D, int& p, int& pe) {
  static_assert(FSW_REG_MAX_DEG + 1 < 64, "one lane per degree bin");
  const int lane = (int)__lane_id();
  const int lo = bin_start[min(lane, FSW_REG_MAX_DEG + 1)];
  const int hi = __shfl_down(lo, 1);
  const bool mine = lane >= DLO && lane <= DHI;
  const int nb = mine ? (hi - lo + ROWS - 1) / ROWS : 0;
  int suf = nb;                                            // tiles of the bins >= lane
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_down(suf, off);
    if (lane + off < 64) suf += t;
  }
  const int before = suf - nb;                             // tiles of the bins above this one: they come first
  const unsigned long long hit = __ballot(mine && b >= before && b < suf);
  if (!hit) return false;
  D = __ffsll((long long)hit) - 1;
  const int lo_d = __builtin_amdgcn_readlane(lo, D), hi_d = __builtin_amdgcn_readlane(hi, D);
  p = lo_d + (b - __builtin_amdgcn_readlane(before, D)) * ROWS;
  pe = min(p + ROWS, hi_d);
  return true;
}

static_assert(FSW_NUM_LDS_BINS == 3 && FSW_MID_MAX_DEG == 256 && FSW_LDS_MAX_DEG == 2048, "degree_bin assumes LDS bins 512 / 1024 / 2048");
static_assert(FSW_NUM_HUB_BINS == 4 && FSW_HUB_MAX_DEG == 32768, "degree_bin assumes hub bins 4096 / 8192 / 16384 / 32768");

}  // namespace fsw
