// Probe (not part of the product library): an fp32 product from TWO fp16 planes per operand -- three
// v_mfma_f32_16x16x32_f16 plane products (hi*hi, hi*lo, lo*hi) instead of the six bf16 ones of precision "f32x3".
//   x * 2^s = hi + lo * 2^-L,   hi = fp16(x 2^s),   lo = fp16((x 2^s - hi) 2^L)     (s per tensor, a power of two: exact)
//   MODE 0 (L = 11): the cross terms hi*lo + lo*hi in an accumulator of their own, scaled by 2^-11 at the end;
//   MODE 1 (L = 0):  lo unscaled (small lo values are fp16 subnormals), all three products in ONE accumulator.
// A deliberately plain kernel -- fragments straight from global memory, one 32 x 32 block per wave -- whose only job is
// to produce the ARITHMETIC of such a path (MFMA accumulation order included) for an error table against fp64; its rate
// means nothing.  C[M][N] = out_scale * sum_k A[m][k] B[n][k], operands k-contiguous; M, N multiples of 64, K of 32.
#include <hip/hip_runtime.h>
#include <stdint.h>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ void __launch_bounds__(256)
k_f16x2_nt(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al, const _Float16 *__restrict__ Bh,
           const _Float16 *__restrict__ Bl, int64_t lda, int64_t ldb, float *__restrict__ C, int64_t ldc, int K,
           float cross_scale, float out_scale) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
  const int l15 = lane & 15, q = lane >> 4;
  f32x4 acc[2][2], accx[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = accx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t oa = (int64_t)(m0 + i * 16 + l15) * lda + k0 + 8 * q;
      const int64_t ob = (int64_t)(n0 + i * 16 + l15) * ldb + k0 + 8 * q;
      ah[i] = *reinterpret_cast<const half8 *>(Ah + oa);
      al[i] = *reinterpret_cast<const half8 *>(Al + oa);
      bh[i] = *reinterpret_cast<const half8 *>(Bh + ob);
      bl[i] = *reinterpret_cast<const half8 *>(Bl + ob);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        if (MODE == 0) {
          accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
        } else if (MODE == 1) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }                         // MODE 2: hi*hi alone (what one fp16 plane gives: the size of what the cross terms repair)
      }
  }
  // lane (l15, q) of a 16 x 16 block holds column l15, rows 4 q .. 4 q + 3
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(int64_t)(m0 + i * 16 + 4 * q + r) * ldc + n0 + j * 16 + l15] = (acc[i][j][r] + cross_scale * accx[i][j][r]) * out_scale;
}

extern "C" int f16x2_gemm_nt(int mode, const void *Ah, const void *Al, const void *Bh, const void *Bl, int64_t lda, int64_t ldb,
                             float *C, int64_t ldc, int M, int N, int K, float cross_scale, float out_scale, void *stream) {
  if (M % 64 || N % 64 || K % 32 || mode < 0 || mode > 2) return 1;
  const dim3 grid(N / 64, M / 64), block(256);
  hipStream_t s = (hipStream_t)stream;
  const _Float16 *ah = (const _Float16 *)Ah, *al = (const _Float16 *)Al, *bh = (const _Float16 *)Bh, *bl = (const _Float16 *)Bl;
  if (mode == 0) hipLaunchKernelGGL(k_f16x2_nt<0>, grid, block, 0, s, ah, al, bh, bl, lda, ldb, C, ldc, K, cross_scale, out_scale);
  else if (mode == 1) hipLaunchKernelGGL(k_f16x2_nt<1>, grid, block, 0, s, ah, al, bh, bl, lda, ldb, C, ldc, K, cross_scale, out_scale);
  else hipLaunchKernelGGL(k_f16x2_nt<2>, grid, block, 0, s, ah, al, bh, bl, lda, ldb, C, ldc, K, cross_scale, out_scale);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
