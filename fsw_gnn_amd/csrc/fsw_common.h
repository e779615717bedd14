// Shared host/device helpers for libfsw_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

constexpr int kWave = 64;

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// wave index inside the workgroup as a provably wave-uniform (SGPR) value
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int degree_bin(int deg) {
  return deg <= FSW_REG_MAX_DEG ? deg : (deg <= FSW_LDS_MAX_DEG ? FSW_BIN_LDS : FSW_BIN_GLOBAL);
}

__host__ __device__ inline uint32_t pow2ceil(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace fsw
