// Per-slice projection  Xp[n, S] = X[n, d] . V[S, d]^T  on the fp32 matrix cores.  gfx950.
//
// Replaces torch.tensordot(X, projVecs, dims=((-1,),(1,))) (reference fsw_embedding.py:909-913), the one
// dense contraction of the path.  v_mfma_f32_32x32x2_f32 keeps exact fp32 products and accumulation
// (bit-for-bit a k-ordered fmaf chain), which the 1e-5 parity target needs; gfx950 has no xf32.
// Xp is written row-major with the slice index minor so that the neighbourhood kernels gather one
// contiguous 256-byte run per (neighbour, 64-slice chunk).
//
// Tiling: 256 threads = 4 waves as 2x2; workgroup tile 128 rows x 128 slices, K step 32 through LDS;
// each wave owns a 64x64 sub-tile = 2x2 MFMA blocks of 32x32 (64 accumulator registers).
#include "fsw_common.h"

namespace fsw {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PBM = 128, PBN = 128, PBK = 32, PLD = PBK + 1;

template <bool VEC>
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, int64_t ld, int64_t row0, int64_t nrows, int k0,
                                           int kmax, float (*dst)[PLD], int& nonfinite, float* __restrict__ copy_dst = nullptr,
                                           int64_t ld_copy = 0) {
  // 128 rows x 32 k; thread t covers k-quad (t & 7) of rows (t >> 3) + 32 i
  const int kq = (threadIdx.x & 7) * 4;
  const int r0 = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + 32 * i;
    const int64_t gr = row0 + r;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gr < nrows) {
      const float* p = src + gr * ld + k0 + kq;
      if (VEC && k0 + kq + 3 < kmax) {
        float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + kq + j < kmax) v[j] = p[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      nonfinite |= !(fabsf(v[j]) <= 3.402823466e38f);
      dst[r][kq + j] = v[j];
      if (copy_dst && gr < nrows && k0 + kq + j < kmax) copy_dst[gr * ld_copy + k0 + kq + j] = v[j];
    }
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256) k_project(const float* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                 const float* __restrict__ V, int S, int64_t ldv, float* __restrict__ Xp,
                                                 int64_t ldp, int32_t* __restrict__ stats, int nct,
                                                 float* __restrict__ x_copy, int64_t ld_copy) {
  __shared__ float As[PBM][PLD];
  __shared__ float Bs[PBN][PLD];
  const int ct = blockIdx.x % nct;
  const int64_t rt = blockIdx.x / nct;
  const int64_t row0 = rt * PBM;
  const int col0 = ct * PBN;
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int fr = lane & 31, fh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int nonfinite = 0, dummy = 0;
  for (int k0 = 0; k0 < d; k0 += PBK) {
    stage_tile<VEC>(X, ldx, row0, n, k0, d, As, nonfinite, ct == 0 ? x_copy : nullptr, ld_copy);
    stage_tile<VEC>(V, ldv, col0, S, k0, d, Bs, dummy);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < PBK; kk += 2) {
      const float a0 = As[wr * 64 + fr][kk + fh];
      const float a1 = As[wr * 64 + 32 + fr][kk + fh];
      const float b0 = Bs[wc * 64 + fr][kk + fh];
      const float b1 = Bs[wc * 64 + 32 + fr][kk + fh];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = col0 + wc * 64 + j * 32 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t gr = row0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (gr < n && c < S) Xp[gr * ldp + c] = acc[i][j][r];
      }
    }
  if (stats && nonfinite) atomicOr(&stats[FSW_STAT_FLAGS], FSW_FLAG_X_NONFINITE);
}

}  // namespace fsw

using namespace fsw;

extern "C" int fsw_project_f32(const float* X, int64_t n, int d, int64_t ldx, const float* V, int S, int64_t ldv,
                               float* Xp, int64_t ldp, float* x_copy, int64_t ld_copy, int32_t* stats, fsw_stream_t stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  FSW_REQUIRE(X && V && Xp, "fsw_project_f32: null pointer");
  FSW_REQUIRE(!x_copy || ld_copy >= d, "fsw_project_f32: ld_copy must be >= d");
  FSW_REQUIRE(n >= 1 && d >= 1 && S >= 1 && ldx >= d && ldv >= d && ldp >= S, "fsw_project_f32: bad sizes n=%lld d=%d S=%d",
              (long long)n, d, S);
  const int nct = (int)ceil_div(S, PBN);
  const int64_t nblocks = ceil_div(n, PBM) * nct;
  FSW_REQUIRE(nblocks < (1ll << 31), "fsw_project_f32: grid too large");
  const bool vec = (ldx % 4 == 0) && (ldv % 4 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)V % 16 == 0);
  if (vec)
    k_project<true><<<(unsigned)nblocks, 256, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, nct, x_copy, ld_copy);
  else
    k_project<false><<<(unsigned)nblocks, 256, 0, stream>>>(X, n, d, ldx, V, S, ldv, Xp, ldp, stats, nct, x_copy, ld_copy);
  FSW_LAUNCH_CHECK();
  return 0;
}
