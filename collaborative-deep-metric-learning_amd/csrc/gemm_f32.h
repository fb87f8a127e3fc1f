// Shared between gemm_f32.hip (128x128 / 64x128 / 64x64 kernels, entry points,
// dispatch) and gemm_f32_pp.hip (128x256 ping-pong kernel).
#pragma once
#include "common.h"

namespace cdml {

enum { EPI_BIAS_LRELU = 1, EPI_LRELU_MASK = 2, EPI_SLAB_COLSUM = 3 };

struct GemmArgs {
  const float *A; int64_t lda;
  const float *B; int64_t ldb;
  float *C; int64_t ldc;
  const float *bias;             // EPI_BIAS_LRELU
  const float *aux; int64_t ldaux;  // EPI_LRELU_MASK (may be null)
  float *colsum;                 // EPI_SLAB_COLSUM: [chunks][N] partial column sums of B (may be null)
  float alpha;
  int M, N, K;                   // output M x N, contraction K
  int k_per_split;               // multiple of the K-tile; blockIdx.y = split
  int64_t slab_stride;           // elements between split outputs
  int tiles_m, tiles_n;
};

// ---- 128x256x32 ping-pong kernel (gemm_f32_pp.hip) ----
// form 0 = NN (A[M][K], B[K][N]; FC forward), 1 = NT (A[M][K], B[N][K]; data gradient),
// 2 = TN (A[K][M], B[K][N]; weight gradient)
bool gemm_f32_pp_usable(int form, int M, int N, int K, int64_t lda, int64_t ldb);
int gemm_f32_pp_splits(int M, int N, int K);          // TN only
int gemm_f32_pp_colsum_chunks(int M, int splits);     // TN: partial rows the kernel writes to g.colsum
int launch_gemm_f32_pp(int form, GemmArgs g, int splits, hipStream_t stream);

}  // namespace cdml
