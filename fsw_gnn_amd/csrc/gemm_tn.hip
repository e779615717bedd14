// C[M, N] = A^T . B for two tall row-major matrices A [K, M], B [K, N] with K in the millions and M, N in the hundreds: the
// weight-gradient shape of the backward pass (gV = gXp^T . X, reference fsw_embedding.py:909-913 differentiated; gW of FSW_conv's
// first Linear layer).  BLAS libraries tile over M x N and walk K inside a workgroup, which leaves a 256 x 128 result on a handful
// of CUs (or a slow split-K); here the K axis is what the grid splits.  gfx950.
//   * a persistent workgroup of 8 wavefronts takes a contiguous range of k rows; a wavefront owns blocks of 64 x 64 outputs (2 x 2
//     tiles of v_mfma_f32_32x32x2_f32: exact fp32 products and sums) and feeds the MFMA operands STRAIGHT from global memory: for
//     the k pair (k, k + 1) lane (r, h) reads A[k + h][m0 + r] and B[k + h][n0 + r] -- two 128-byte runs per instruction, and the
//     eight wavefronts of the workgroup re-read the same two rows out of L1;
//   * the workgroups' partial results go to a [G, M, N] buffer and a second kernel sums them in a fixed order: the result does not
//     depend on how the K axis was split or scheduled (no float atomics).
#include <algorithm>
#include <stdlib.h>
#include "fsw_common.h"

namespace fsw {

using f32x16g = __attribute__((ext_vector_type(16))) float;
constexpr int kTnWaves = 8;
constexpr int kTnMaxBlocks = 2;   // 64 x 64 blocks per wavefront (64 accumulator registers each)

template <int NB>
__global__ void __launch_bounds__(kTnWaves* kWave) k_gemm_tn_partial(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                                     int64_t ldb, int64_t K, int M, int N, float* __restrict__ P) {
  const int lane = lane_id(), w = wave_id();
  const int fr = lane & 31, fh = lane >> 5;
  const int nbn = (N + 63) / 64, nblocks = ((M + 63) / 64) * nbn;
  // k range of this workgroup: equal pieces of an even number of rows
  const int64_t per = ((K + gridDim.x - 1) / gridDim.x + 1) & ~(int64_t)1;
  const int64_t k0 = (int64_t)blockIdx.x * per, k1 = min(k0 + per, K);
  f32x16g acc[NB][4];
  int m0[NB], n0[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int blk = w + b * kTnWaves;
    m0[b] = blk < nblocks ? (blk / nbn) * 64 : -1;
    n0[b] = blk < nblocks ? (blk % nbn) * 64 : 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
  }
  constexpr int UN = 4;   // k pairs in flight
  for (int64_t k = k0; k < k1; k += 2 * UN) {
    float a[NB][UN][2], bb[NB][UN][2];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t kr = k + 2 * u + fh;
        const bool kok = kr < k1 && m0[b] >= 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int m = m0[b] + 32 * t + fr, n = n0[b] + 32 * t + fr;
          a[b][u][t] = (kok && m < M) ? A[kr * lda + m] : 0.f;
          bb[b][u][t] = (kok && n < N) ? B[kr * ldb + n] : 0.f;
        }
      }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[b][tm * 2 + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][u][tm], bb[b][u][tn], acc[b][tm * 2 + tn], 0, 0, 0);
  }
  float* Pg = P + (int64_t)blockIdx.x * M * N;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (m0[b] < 0) continue;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0[b] + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * fh;   // C/D map of the 32x32 MFMA: row from (r, fh), column fr
          const int n = n0[b] + 32 * tn + fr;
          if (m < M && n < N) Pg[(int64_t)m * N + n] = acc[b][tm * 2 + tn][r];
        }
  }
}

// ---- the same product on the bf16 matrix cores, fp32-accurate (bf16 x 3, project.hip) ------------------------------------------------
// x = x1 + x2 + x3 (three bf16 pieces = 24 significand bits); the six products x_i y_j with i + j <= 4 carry everything down to 2^-24,
// each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16: 24 matrix instructions of 32 cycles per 16 k and 64 x 64 block
// against 32 of 64 cycles on the fp32 instruction.  Operands straight from global memory as above: lane (r, h) holds
// A[k + 8 h + j][m0 + 32 t + r], j = 0..7 -- the instruction's A layout for A^T -- and the same for B.
typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3t(const float (&v)[8], bf16x8t& p1, bf16x8t& p2, bf16x8t& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 a = (__bf16)v[j];
    const float r = v[j] - (float)a;
    const __bf16 b = (__bf16)r;
    p1[j] = a;
    p2[j] = b;
    p3[j] = (__bf16)(r - (float)b);
  }
}

template <int NB>
__global__ void __launch_bounds__(kTnWaves* kWave) k_gemm_tn_partial_bf3(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                                         int64_t ldb, int64_t K, int M, int N, float* __restrict__ P) {
  const int lane = lane_id(), w = wave_id();
  const int fr = lane & 31, fh = lane >> 5;
  const int nbn = (N + 63) / 64, nblocks = ((M + 63) / 64) * nbn;
  // k range of this workgroup: equal pieces of a multiple of 16 rows
  const int64_t per = (((K + gridDim.x - 1) / gridDim.x) + 15) & ~(int64_t)15;
  const int64_t k0 = (int64_t)blockIdx.x * per, k1 = min(k0 + per, K);
  f32x16g acc[NB][4];
  int m0[NB], n0[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int blk = w + b * kTnWaves;
    m0[b] = blk < nblocks ? (blk / nbn) * 64 : -1;
    n0[b] = blk < nblocks ? (blk % nbn) * 64 : 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
  }
  // (the loads of a step are consumed right away: a version that keeps the NEXT step's 32 loads in flight during the split and the
  // matrix instructions measured slower -- 0.86 against 0.80 ms at 256 x 128 -- at 200 registers)
  for (int64_t k = k0; k < k1; k += 16) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (m0[b] < 0) continue;                              // uniform
      float a[2][8], bb[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t kr = k + 8 * fh + j;
        const bool kok = kr < k1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int m = m0[b] + 32 * t + fr, n = n0[b] + 32 * t + fr;
          a[t][j] = (kok && m < M) ? A[kr * lda + m] : 0.f;
          bb[t][j] = (kok && n < N) ? B[kr * ldb + n] : 0.f;
        }
      }
      bf16x8t ap[2][3], bp[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        split3t(a[t], ap[t][0], ap[t][1], ap[t][2]);
        split3t(bb[t], bp[t][0], bp[t][1], bp[t][2]);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          f32x16g c = acc[b][tm * 2 + tn];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[tm][2], bp[tn][0], c, 0, 0, 0);   // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[tm][1], bp[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[tm][0], bp[tn][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[tm][1], bp[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[tm][0], bp[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[tm][0], bp[tn][0], c, 0, 0, 0);
          acc[b][tm * 2 + tn] = c;
        }
    }
  }
  float* Pg = P + (int64_t)blockIdx.x * M * N;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (m0[b] < 0) continue;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0[b] + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * fh;
          const int n = n0[b] + 32 * tn + fr;
          if (m < M && n < N) Pg[(int64_t)m * N + n] = acc[b][tm * 2 + tn][r];
        }
  }
}

__global__ void __launch_bounds__(256) k_gemm_tn_reduce(const float* __restrict__ P, int G, int64_t MN, int N, float* __restrict__ C,
                                                        int64_t ldc, float beta) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < MN; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += P[(int64_t)g * MN + i];
    const int64_t m = i / N, n = i - m * N;
    float* c = C + m * ldc + n;
    *c = beta != 0.f ? beta * *c + s : s;
  }
}

}  // namespace fsw

using namespace fsw;

constexpr int kTnMaxGrid = 512;   // two workgroups per CU
extern "C" size_t fsw_gemm_tn_workspace_bytes(int M, int N) { return (size_t)kTnMaxGrid * (size_t)M * (size_t)N * sizeof(float); }

extern "C" int fsw_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, int M, int N, float* C, int64_t ldc,
                               float beta, void* workspace, size_t workspace_bytes, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(A && B && C && workspace, "fsw_gemm_tn_f32: null pointer");
  FSW_REQUIRE(K >= 1 && M >= 1 && N >= 1 && lda >= M && ldb >= N && ldc >= N, "fsw_gemm_tn_f32: bad sizes");
  const int nblocks = ((M + 63) / 64) * ((N + 63) / 64);
  FSW_REQUIRE(nblocks <= kTnWaves * kTnMaxBlocks, "fsw_gemm_tn_f32: M x N above %d blocks of 64 x 64", kTnWaves * kTnMaxBlocks);
  const int G = (int)std::min<int64_t>(kTnMaxGrid, std::max<int64_t>(1, K / 512));
  FSW_REQUIRE(workspace_bytes >= (size_t)G * M * N * sizeof(float), "fsw_gemm_tn_f32: workspace too small (fsw_gemm_tn_workspace_bytes)");
  float* P = reinterpret_cast<float*>(workspace);
  // FSW_GEMM_TN_EXACT_FP32=1: the fp32 matrix instruction instead of bf16 x 3 (same accuracy class, 2.7x the matrix cycles)
  static const bool exact = [] { const char* e = getenv("FSW_GEMM_TN_EXACT_FP32"); return e && atoi(e) != 0; }();
  if (exact) {
    if (nblocks <= kTnWaves) k_gemm_tn_partial<1><<<G, kTnWaves * kWave, 0, stream>>>(A, lda, B, ldb, K, M, N, P);
    else k_gemm_tn_partial<2><<<G, kTnWaves * kWave, 0, stream>>>(A, lda, B, ldb, K, M, N, P);
  } else {
    if (nblocks <= kTnWaves) k_gemm_tn_partial_bf3<1><<<G, kTnWaves * kWave, 0, stream>>>(A, lda, B, ldb, K, M, N, P);
    else k_gemm_tn_partial_bf3<2><<<G, kTnWaves * kWave, 0, stream>>>(A, lda, B, ldb, K, M, N, P);
  }
  FSW_LAUNCH_CHECK();
  const int64_t MN = (int64_t)M * N;
  k_gemm_tn_reduce<<<(unsigned)std::min<int64_t>(ceil_div(MN, 256), 1024), 256, 0, stream>>>(P, G, MN, N, C, ldc, beta);
  FSW_LAUNCH_CHECK();
  return 0;
}
