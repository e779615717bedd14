// Shared host/device helpers for libfsw_hip.so (gfx950 only).
#pragma once
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

// rows of the degree bins lo .. hi when the caller passed the host copy of bin_start, `upper` (a bound) otherwise
inline int64_t bin_rows_or(const fsw_embed_args& a, int lo, int hi, int64_t upper) {
  return a.bin_start_host ? (int64_t)a.bin_start_host[hi + 1] - a.bin_start_host[lo] : upper;
}

__host__ __device__ inline uint32_t pow2ceil(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

static_assert(FSW_NUM_LDS_BINS == 3 && FSW_MID_MAX_DEG == 256 && FSW_LDS_MAX_DEG == 2048, "degree_bin assumes LDS bins 512 / 1024 / 2048");
static_assert(FSW_NUM_HUB_BINS == 4 && FSW_HUB_MAX_DEG == 32768, "degree_bin assumes hub bins 4096 / 8192 / 16384 / 32768");

}  // namespace fsw
