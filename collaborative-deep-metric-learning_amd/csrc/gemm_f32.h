// Argument block of the fp32 MFMA GEMM kernels (gemm_f32.hip).
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

}  // namespace cdml
