// Shared between gemm_bf16.hip (128x128 kernel, transposes, dispatch) and
// gemm_bf16_256.hip (256x256 ping-pong kernel).
#pragma once
#include "common.h"

namespace cdml {

using bf16 = __bf16;

// 4 = epilogue 0 that ALSO writes the sign bitmask of its output (bit j of byte b of row r = C[r][8b+j] > 0)
// to `aux` (uint8 [M][ldaux bytes]); 5 = epilogue 2 reading that bitmask instead of the bf16 activations:
// the leaky-relu derivative needs one bit per element, not the 2-byte value (252 MB -> 16 MB per step at
// config 4's shape).  4 runs on the 256x256 kernel only, 5 on the K = 256 streaming kernel only.
enum { BE_BIAS_LRELU_BF16 = 0, BE_BIAS_LRELU_F32 = 1, BE_MASK_BF16 = 2, BE_F32 = 3, BE_BIAS_LRELU_BF16_BITS = 4,
       BE_MASKBITS_BF16 = 5,
       // fp32 products on the bf16 MFMA (gemm_bf16x3.hip): the result, bias + leaky-relu / times the mask
       // applied, split into three bf16 planes hi | mid | lo (C = bf16, planes x3_plane_c elements apart)
       BE_BIAS_LRELU_X3 = 6, BE_MASK_X3 = 7,
       // 6 with the bias indexed by the output ROW (a layer computed transposed: C = W^T-planes . x-planes^T)
       BE_ROWBIAS_LRELU_X3 = 8,
       // ABI-level ids of cdml_gemm_bf16x3_nt (mapped to 6 / 7 with BArgs::mask_out / aux_bits set): 6 that also writes the
       // sign bitmask of its result to `aux`, 7 reading that bitmask
       BE_BIAS_LRELU_X3_BITS = 9, BE_MASKBITS_X3 = 10,
       // semi-hard negative mining (BASELINE config 2) as the epilogue of the B x 2B score product S = E_anchor . E^T:
       // nothing of S is written -- every (tile, column strip) hands back, per anchor row, its best candidates
       // (BArgs::mine_*; gemm_bf16_256.hip "mine tail", loss.hip cdml_semihard_mine_x3)
       BE_MINE_X3 = 11,
       // ABI-level id of cdml_gemm_bf16x3_nt: 10 (or 7 without a mask) with the planes of the result k8-interleaved
       BE_MASKBITS_X3_KI = 12,
       // exact kNN (faiss_knn.py:82-131) as the epilogue of the query x catalogue score product: nothing of the scores is
       // written -- an element whose distance is within its query's current k-th best (BArgs::knn_*) is appended to the
       // query's candidate list (gemm_bf16_256.hip "knn filter", knn.hip cdml_knn_filter_x3)
       BE_KNN_X3 = 13 };

// one candidate pair of an anchor over some set of columns: the closest eligible column with d > d_p ("outside"; ties ->
// smaller column) and the farthest eligible one; c = 0x7fffffff: none
struct alignas(16) MineCand { float out_d; int out_c; float in_d; int in_c; };

struct BArgs {
  const bf16 *A; int64_t lda;
  const bf16 *B; int64_t ldb;
  void *C; int64_t ldc;
  const float *bias;
  const bf16 *aux; int64_t ldaux;
  float alpha;
  int M, N, K;
  int k_per_split;
  int64_t slab_stride;
  int tiles_m, tiles_n;
  float *colsum_partial;   // k-strided form only: [splits*tiles_m*2][N] column sums of B (nullable)
  uint8_t *mask_out; int64_t ldmask;   // epilogue 0 on the 256x256 kernel: sign bitmask of C (nullable), bytes per row
  int aux_bits;            // K = 256 streaming kernel: aux is that bitmask (ldaux in bytes), not bf16 values
  // Split-fp32 form (X3 instantiations of the 256x256 kernel): every operand is three bf16 planes hi | mid | lo
  // with hi + mid + lo = the fp32 value; K counts the K-tiles of ALL plane products, x3_tpp K-tiles each, walked
  // as (A, B) = (hi,hi) (hi,mid) (mid,hi) (hi,lo) (lo,hi) (mid,mid).  Plane p of A starts x3_plane_a elements
  // (k-contiguous form: along k; k-strided form: along the columns) after plane p - 1; B likewise.
  int x3_tpp;
  int x3_products;         // K-major walk (> 0): K-tile v = product v % x3_products of K-tile v / x3_products (0: product-major)
  int64_t x3_plane_a, x3_plane_b, x3_plane_c;
  // A launch over a SUBSET of the tile grid (gemm_bf16_256.hip, "the last round in half tiles"): grid_tiles = the tile
  // count the block -> tile map is computed for (0: the launch's own grid), narrow_first = the first block index (in
  // that map) of the tiles a NARROW launch computes as two 128 x 256 halves each
  int grid_tiles, narrow_first;
  // k_gemm_x3_rounds: this many of the half-tile blocks are dispatched FIRST (before the full tiles), the rest last: the CUs
  // that start with a half tile then run half a tile out of phase with the others for the whole launch, so the tiles'
  // plane stores (393 KB each) reach HBM in two bursts of half the chip instead of one of the whole chip (0: all last)
  int stagger_lead;
  // plane-output epilogue BE_MASK_X3: write the planes k8-interleaved ([plane][row / 8][column][8 rows]; ldc = columns per
  // row group, x3_plane_c = elements per plane) for the weight gradient that contracts over the rows
  int c_kint;
  // BE_MINE_X3: A = the anchors' planes (row i = embedded row 2 i), B = every embedded row's planes, C unused.
  // mine_sqn[c] = |e_c|^2, mine_ids[c] = the video id of row c, mine_dp[i] = d(anchor i, its positive);
  // mine_out[(tn * 4 + strip) * mine_ld + i] = anchor i's candidates over the 64 columns of strip `strip` of tile column tn
  const float *mine_sqn; const int32_t *mine_ids; const float *mine_dp; MineCand *mine_out; int64_t mine_ld;
  // BE_KNN_X3: A = the queries' planes, B = a block of the catalogue's planes (column c of the launch = catalogue row
  // knn_col0 + c; rows >= knn_n_valid are padding), C unused.  d = knn_qsq[row] + knn_bsq[c] - 2 <q, b> (clamped at 0); an
  // element with d <= knn_tau[row] takes slot atomicAdd(knn_cnt + row, 1) of the row's knn_cap-slot list knn_cand
  // ((d, id) pairs; a slot beyond the capacity is dropped -- the count says so)
  const float *knn_qsq, *knn_bsq, *knn_tau; int32_t *knn_cnt; uint2 *knn_cand; int knn_cap, knn_col0, knn_n_valid;
  // Two-plane fp16 form ("f16x2": gemm_f16x2_256.hip = this kernel compiled with CDML_F16X2): the operands hold
  // a 2^sa and b 2^sb as fp16 planes hi | lo, so the accumulator holds 2^(sa + sb) a.b: out_scale = 2^-(sa + sb) multiplies it
  // before anything else of the epilogue; a plane output is written as the fp16 planes of (value * c_scale), clamped to fp16's range
  float out_scale, c_scale;
};

// 256x256x64 kernel: true if the shape can use it (N % 256 == 0, K per split a
// multiple of 128, operands inside the 2 GiB buffer-descriptor window).
bool gemm_bf16_256_usable(int M, int N, int K, int64_t lda, int64_t ldb);
int gemm_bf16_256_splits(int M, int N, int K);
// g.tiles_m / g.tiles_n / g.k_per_split / g.C (slabs when splits > 1) set by the caller
int launch_gemm_bf16_256(const BArgs &g, int epilogue, int splits, hipStream_t stream);
// the split-fp32 forms (g.x3_* set; epilogues 1, 3, 6, 7 k-contiguous, 3 k-strided)
int launch_gemm_bf16_256_x3(const BArgs &g, bool tn, int epilogue, int splits, hipStream_t stream);
// the same on two fp16 planes per operand, three plane products (hi.hi + hi.lo + lo.hi) on v_mfma_f32_16x16x32_f16
// (gemm_f16x2_256.hip; g.x3_products = 3, g.out_scale / g.c_scale set; epilogues 1, 3, 6, 7 k-contiguous, 3 k-strided)
int launch_gemm_f16x2_256(const BArgs &g, bool tn, int epilogue, int splits, hipStream_t stream);
int launch_gemm_f16x2_mine(const BArgs &g, hipStream_t stream);     // BE_MINE_X3 on fp16 planes (g.out_scale set)
int launch_gemm_f16x2_knn(const BArgs &g, hipStream_t stream);      // BE_KNN_X3 on fp16 planes
// the score product of semi-hard mining with the selection as its epilogue (BE_MINE_X3; six products, resident-plane walk)
int launch_gemm_x3_mine(const BArgs &g, hipStream_t stream);
// the query x catalogue score product of the kNN export with the threshold filter as its epilogue (BE_KNN_X3)
int launch_gemm_x3_knn(const BArgs &g, hipStream_t stream);
// the k-strided product on k8-INTERLEAVED operands ([plane][k / 8][column][8 k]; g.lda / g.ldb = elements per k-group,
// g.x3_plane_* = elements per plane): the resident-plane walk with one 16-B LDS read per fragment
int launch_gemm_x3_tnk(const BArgs &g, int splits, hipStream_t stream);

// streaming kernel for the mask / plain-bf16 epilogue (BE_MASK_BF16) at K == 256: N % 256 == 0,
// lda / ldb / ldc / ldaux multiples of 8, A inside the 2 GiB buffer-descriptor window
bool gemm_bf16_k256_usable(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, bool has_aux);
int launch_gemm_bf16_k256(const BArgs &g, hipStream_t stream);

// k-strided form C[M][N] = sum_k A[k][M] * B[k][N] (fp32 slabs): M, N % 256 == 0, K % 128 == 0
bool gemm_bf16_tn_usable(int M, int N, int K, int64_t lda, int64_t ldb);
int launch_gemm_bf16_tn(const BArgs &g, int splits, hipStream_t stream);

// two k-strided products over the same K in one stream-K launch + one fix-up pass (gemm_bf16_256.hip);
// g.C / g.ldc = the outputs, db1 / db2 (nullable) = column sums of the two B operands; 0 = shapes not taken
size_t gemm_bf16_tn2_workspace(int M1, int N1, int M2, int N2, int K);
int launch_gemm_bf16_tn2(const BArgs &g1, const BArgs &g2, float *db1, float *db2, void *workspace,
                         size_t workspace_bytes, hipStream_t stream);

}  // namespace cdml
