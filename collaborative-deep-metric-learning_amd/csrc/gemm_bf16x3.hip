// fp32 products of the tower on the bf16 matrix cores ("split-fp32", precision "f32x3").
//
// gfx950 multiplies bf16 sixteen times faster than fp32 (2.5 PFLOP/s against 157 TFLOP/s dense).  An fp32 value
// is EXACTLY the sum of three bf16 values -- hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): three
// roundings to nearest hold 24 significant bits, and bf16 has fp32's exponent range -- and a product of two
// bf16 values is exact in the fp32 accumulator of the MFMA.  So
//     a b = (ah + am + al)(bh + bm + bl) = ah bh + ah bm + am bh + ah bl + al bh + am bm  [+ am bl + al bm + al bl]
// with the three dropped terms below 2^-26 |a b|: six bf16 MFMAs give an fp32 product to better than one fp32
// rounding, at 6 / 16 of the fp32 MFMA's cost.  Measured at FC1's shape (tools/f32x3_probe.py): 0.556 ms against
// 0.936 ms for the fp32 kernel, max error against fp64 2.6e-6 of max|C| against 2.2e-6 for the fp32 kernel.
// (`products` = 3 keeps ah bh + ah bm + am bh only: 16-bit operands, error 4.7e-6, half the time again.)
//
// Operands live in memory as three planes side by side: a k-contiguous operand A[M][K] as [M][hi K | mid K | lo K]
// (plane stride along k), a k-strided one A[K][M] as [K][hi M | mid M | lo M] (plane stride along the columns).
// The GEMMs are the 256x256 ping-pong kernel of gemm_bf16_256.hip with a K loop that walks the plane products
// (BArgs::x3_*); epilogues 6 / 7 write their result as planes again, so the chain FC1 -> FC2 / dH1 -> dW1 never
// holds an activation in fp32.  Replaces the same reference lines as gemm_f32.hip (models.py:59-60, train.py:141).
#include "gemm_bf16.h"
#include <stdlib.h>

namespace cdml {
namespace {

using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kThreads = 256;

__device__ __forceinline__ void split3(float v, bf16 &h, bf16 &m, bf16 &l) {
  h = (bf16)v;
  const float r = v - (float)h;
  m = (bf16)r;
  l = (bf16)(r - (float)m);
}

// dst[r][p * plane + c] = plane p of src[r][c]; 4 columns per thread
__global__ void __launch_bounds__(kThreads)
k_split3(const float *__restrict__ src, int64_t lds_, int rows, int cols, bf16 *__restrict__ dst, int64_t ldd,
         int64_t plane) {
  const int c4n = cols >> 2;
  const int64_t total = (int64_t)rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(src + r * lds_ + c);
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16 a, b, c_;
      split3(v[j], a, b, c_);
      h[j] = a; m[j] = b; l[j] = c_;
    }
    bf16 *d = dst + r * ldd + c;
    *reinterpret_cast<bf16x4 *>(d) = h;
    *reinterpret_cast<bf16x4 *>(d + plane) = m;
    *reinterpret_cast<bf16x4 *>(d + 2 * plane) = l;
  }
}

// dst[c][p * plane + r] = plane p of src[r][c]: 64 x 64 tiles through LDS
__global__ void __launch_bounds__(kThreads)
k_split3_transpose(const float *__restrict__ src, int64_t lds_, int rows, int cols, bf16 *__restrict__ dst,
                   int64_t ldd, int64_t plane) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int c = c0 + j, r = r0 + tx;
    if (c < cols && r < rows) {
      bf16 h, m, l;
      split3(tile[tx][j], h, m, l);
      bf16 *d = dst + (int64_t)c * ldd + r;
      d[0] = h; d[plane] = m; d[2 * plane] = l;
    }
  }
}

// ---- two fp16 planes (precision "f16x2"): hi = fp16(v * scale), lo = fp16(v * scale - hi); scale a power of two, the
// scaled value clamped to fp16's range (a tensor that outgrows its scale saturates instead of turning into infinities:
// engine_f16x2.py watches the maxima and moves the scales) ----
using half4v = __attribute__((ext_vector_type(4))) _Float16;
__device__ __forceinline__ void split2h(float v, float scale, _Float16 &h, _Float16 &l) {
  const float s = __builtin_amdgcn_fmed3f(v * scale, -65504.f, 65504.f);
  h = (_Float16)s;
  l = (_Float16)(s - (float)h);
}

__global__ void __launch_bounds__(kThreads)
k_split2h(const float *__restrict__ src, int64_t lds_, int rows, int cols, _Float16 *__restrict__ dst, int64_t ldd,
          int64_t plane, float scale) {
  const int c4n = cols >> 2;
  const int64_t total = (int64_t)rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(src + r * lds_ + c);
    half4v h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      _Float16 a, b;
      split2h(v[j], scale, a, b);
      h[j] = a; l[j] = b;
    }
    _Float16 *d = dst + r * ldd + c;
    *reinterpret_cast<half4v *>(d) = h;
    *reinterpret_cast<half4v *>(d + plane) = l;
  }
}

__global__ void __launch_bounds__(kThreads)
k_split2h_transpose(const float *__restrict__ src, int64_t lds_, int rows, int cols, _Float16 *__restrict__ dst,
                    int64_t ldd, int64_t plane, float scale) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? src[(int64_t)r * lds_ + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int c = c0 + j, r = r0 + tx;
    if (c < cols && r < rows) {
      _Float16 h, l;
      split2h(tile[tx][j], scale, h, l);
      _Float16 *d = dst + (int64_t)c * ldd + r;
      d[0] = h; d[plane] = l;
    }
  }
}

int grid1d(int64_t n) {
  const int64_t b = (n + kThreads - 1) / kThreads;
  return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

// walk order of the plane products: K-major unless CDML_X3_ORDER=product (A/B runs; read per call)
bool x3_kmajor() {
  const char *e = getenv("CDML_X3_ORDER");
  return !(e && e[0] == 'p');
}

// split-K for the k-contiguous forms whose output is ONE tile column wide (N = 256: the narrow forward layer, whose
// M / 256 tiles alone would leave most of the chip idle; fp32 slab output only).  The K partition is a function of K
// ALONE -- slabs of twenty six-step periods (120 K-tile steps), at most 16 of them, summed in slab order by k_x3_sum_slabs
// -- so the result does not depend on how many rows the call carries: a batch computed whole and the same batch computed
// in row blocks (two ranks against one) give the same bits.  (Round 3 chose the slab count from the tile count: 8
// slabs at 8 192 rows, none at 49 152, and leaky-relu' flips near zero then loosened the two-rank gradient bound.)
// Round 5: 120 steps instead of 60 -- four slabs at K = 5 120 -- which halves the slab round trip (the forward layer at
// 16 384 rows: 64 tiles x 4 slabs = one round of the chip instead of two rounds of half-length blocks); where four slabs
// leave the chip under-filled (8 192 rows: 128 blocks) the SAME slabs run as 128 x 256 half tiles (gemm_bf16_256.hip),
// so the partition stays a function of K alone.
// Wider outputs are never split: their tiles fill the chip at every batch size the step uses.
// Round 6: the slab length is a function of K and of the ROW-TILE CLASS.  The 120-step slabs were chosen for BASELINE's
// batches; the reference's own recipe (B = 1 024: 3 072 rows = 12 row tiles) fills 96 half-tile blocks with them and ran
// 0.622 against 0.643 ms/step with 60-step slabs (profiles/r05_reference_recipe_slab_steps_ab.txt).  Rule: where the
// 120-step slabs give at most 64 full tiles (128 half-tile blocks: half the chip) the slabs are 60 steps long -- twice the
// blocks -- else 120.  Within a class the partition is still a function of K alone (a batch whole or in row blocks of the
// same class: the same bits); ACROSS the class boundary the last bits of z differ.  A caller that needs one partition for
// every call size pins it: cdml_x3_slab_steps(120) (thread-local; catalogue inference does, so an embedding has the same
// bits in 4 096- and 49 152-row chunks: test_embedding_bits_do_not_depend_on_the_chunk).
thread_local int tl_slab_steps = 0;                        // 0 = by class
// (half_steps: the two-plane fp16 form walks THREE steps per K-tile where the bf16 form walks six -- its slabs are half as
// many steps long, so that the partition of K itself is the same: 20 or 10 K-tiles of every plane per slab)
int x3_nt_splits(int N, int ktiles, int tiles_m, bool half_steps = false) {
  if (N / 256 != 1) return 1;
  if (half_steps) ktiles *= 2;
  const char *e = getenv("CDML_X3_SLAB_STEPS");            // (A/B runs: one length for everything)
  int steps = e && atoi(e) >= 12 ? atoi(e) / 6 * 6 : tl_slab_steps;
  if (steps < 12) {
    int s120 = ktiles / 120;
    s120 = s120 < 1 ? 1 : (s120 > 16 ? 16 : s120);
    steps = tiles_m * s120 <= 64 ? 60 : 120;
  }
  const int s = ktiles / steps;
  return s < 1 ? 1 : (s > 16 ? 16 : s);
}

// K-tile steps per split, rounded up to whole walk periods (`period` = 6 for the six-product K-major walk, else 2), and
// the number of splits that then actually carry K-tiles: rounding `per` up can leave the last of the requested splits
// empty (ktiles = 288 in 11 splits: per = 30, ten splits cover it) -- an empty split would still launch its blocks, write
// a zero slab and be read back by the combine pass
void x3_split_geometry(int ktiles, int requested, int period, int &per, int &splits) {
  per = (ktiles + requested - 1) / requested;
  per = (per + period - 1) / period * period;
  splits = (ktiles + per - 1) / per;
}

}  // namespace
}  // namespace cdml

using namespace cdml;

extern "C" int cdml_split_f32_bf16x3(const float *src, int64_t ld_src, int rows, int cols, uint16_t *dst,
                                     int64_t ld_dst, int64_t plane, int transpose, cdml_stream_t stream) {
  CDML_REQUIRE(src && dst && rows > 0 && cols > 0, CDML_E_BADARG, "split_f32_bf16x3: bad argument");
  const int out_cols = transpose ? rows : cols;
  CDML_REQUIRE(plane >= out_cols && ld_dst >= 2 * plane + out_cols && ld_src >= cols, CDML_E_BADARG,
               "split_f32_bf16x3: planes of %d columns need plane >= %d and ld_dst >= 2 plane + %d", out_cols, out_cols, out_cols);
  hipStream_t s = (hipStream_t)stream;
  if (transpose) {
    hipLaunchKernelGGL(k_split3_transpose, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(kThreads), 0, s, src, ld_src,
                       rows, cols, reinterpret_cast<bf16 *>(dst), ld_dst, plane);
  } else {
    CDML_REQUIRE((cols & 3) == 0 && (ld_src & 3) == 0 && (ld_dst & 3) == 0 && (plane & 3) == 0 && aligned16(src) &&
                     (reinterpret_cast<uintptr_t>(dst) & 7) == 0,
                 CDML_E_ALIGN, "split_f32_bf16x3: columns, strides and plane must be multiples of 4");
    hipLaunchKernelGGL(k_split3, dim3(grid1d((int64_t)rows * (cols / 4))), dim3(kThreads), 0, s, src, ld_src, rows, cols,
                       reinterpret_cast<bf16 *>(dst), ld_dst, plane);
  }
  return check_launch("split_f32_bf16x3");
}

// dst = the two fp16 planes hi | lo of src * scale (scale > 0, a power of two for an exact scaling; values beyond fp16's
// range saturate); layout and `transpose` as cdml_split_f32_bf16x3 with two planes: ld_dst >= plane + columns.
extern "C" int cdml_split_f32_f16x2(const float *src, int64_t ld_src, int rows, int cols, uint16_t *dst, int64_t ld_dst,
                                    int64_t plane, int transpose, float scale, cdml_stream_t stream) {
  CDML_REQUIRE(src && dst && rows > 0 && cols > 0 && scale > 0.f, CDML_E_BADARG, "split_f32_f16x2: bad argument");
  const int out_cols = transpose ? rows : cols;
  CDML_REQUIRE(plane >= out_cols && ld_dst >= plane + out_cols && ld_src >= cols, CDML_E_BADARG,
               "split_f32_f16x2: planes of %d columns need plane >= %d and ld_dst >= plane + %d", out_cols, out_cols, out_cols);
  hipStream_t s = (hipStream_t)stream;
  if (transpose) {
    hipLaunchKernelGGL(k_split2h_transpose, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(kThreads), 0, s, src, ld_src,
                       rows, cols, reinterpret_cast<_Float16 *>(dst), ld_dst, plane, scale);
  } else {
    CDML_REQUIRE((cols & 3) == 0 && (ld_src & 3) == 0 && (ld_dst & 3) == 0 && (plane & 3) == 0 && aligned16(src) &&
                     (reinterpret_cast<uintptr_t>(dst) & 7) == 0,
                 CDML_E_ALIGN, "split_f32_f16x2: columns, strides and plane must be multiples of 4");
    hipLaunchKernelGGL(k_split2h, dim3(grid1d((int64_t)rows * (cols / 4))), dim3(kThreads), 0, s, src, ld_src, rows, cols,
                       reinterpret_cast<_Float16 *>(dst), ld_dst, plane, scale);
  }
  return check_launch("split_f32_f16x2");
}

extern "C" int cdml_x3_slab_steps(int steps) {
  const int prev = tl_slab_steps;
  tl_slab_steps = steps >= 12 ? steps / 6 * 6 : 0;
  return prev;
}

extern "C" size_t cdml_gemm_bf16x3_workspace(int tn, int M, int N, int K, int products) {
  if (M <= 0 || N <= 0 || K <= 0 || (products != 3 && products != 6) || K % 64) return 0;
  const int ktiles = products * (K / 64);
  // (k-contiguous form: the larger of the two slab lengths' counts -- a workspace serves any pin of cdml_x3_slab_steps)
  int requested = tn ? gemm_bf16_256_splits(M, N, ktiles * 64) : x3_nt_splits(N, ktiles, (M + 255) / 256);
  if (!tn && N / 256 == 1) {
    const int s60 = ktiles / 60 < 1 ? 1 : (ktiles / 60 > 16 ? 16 : ktiles / 60);
    if (s60 > requested) requested = s60;
  }
  int splits = requested, per = 0, s2 = requested;
  x3_split_geometry(ktiles, requested, products == 6 ? 6 : 2, per, splits);
  x3_split_geometry(ktiles, requested, 2, per, s2);          // (the product-major A/B walk rounds to pairs: never fewer slabs)
  if (s2 > splits) splits = s2;
  const size_t cs = (size_t)splits * ((M + 255) / 256) * 2 * N * sizeof(float) + (size_t)N * sizeof(float);   // bias-gradient partials
  return (size_t)splits * M * N * sizeof(float) + cs;       // (a slab even unsplit: the k-strided form's bias pass reads one)
}

// workspace of cdml_gemm_f16x2_nt / _tn: its slabs partition K as the six-product form's do (k-contiguous form) or as the
// three-product walk's length asks (k-strided form) -- the larger of the two serves both
extern "C" size_t cdml_gemm_f16x2_workspace(int tn, int M, int N, int K) {
  const size_t a = cdml_gemm_bf16x3_workspace(tn, M, N, K, 6), b = cdml_gemm_bf16x3_workspace(tn, M, N, K, 3);
  return a > b ? a : b;
}

namespace cdml {
namespace {
// colsum[n] = the partial rows added in a fixed order: 16 columns x 16 row lanes per block, four loads in flight per lane
__device__ __forceinline__ void x3_colsum_block(const float *__restrict__ partial, int n_rows, int N, float *__restrict__ out,
                                                int blk, float scale) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blk * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < N) {
    int r = rl;
    for (; r + 48 < n_rows; r += 64) {
      s0 += partial[(int64_t)r * N + c];
      s1 += partial[(int64_t)(r + 16) * N + c];
      s2 += partial[(int64_t)(r + 32) * N + c];
      s3 += partial[(int64_t)(r + 48) * N + c];
    }
    for (; r < n_rows; r += 16) s0 += partial[(int64_t)r * N + c];
  }
  red[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < N) {
    float t = red[0][cl];
#pragma unroll
    for (int j = 1; j < 16; ++j) t += red[j][cl];
    out[c] = t * scale;                                  // (1 on the bf16 planes; 2^-s on fp16 planes holding values times 2^s)
  }
}

// sums `splits` fp32 slabs (+ bias, leaky-relu when bias != null) into out -- blocks [0, slab_blocks) -- and, in the SAME
// launch, finishes the bias gradient from its per-(split, tile, row group) partial rows -- the blocks after them (round 3
// ran k_x3_colsum_final as a launch of its own: 2 x 10 us per step for a few KB of sums)
__global__ void __launch_bounds__(kThreads)
k_x3_sum_slabs(const float *__restrict__ slabs, int64_t slab_stride, int splits, int rows, int N,
               const float *__restrict__ bias, float alpha, float *__restrict__ out, int64_t ldo, int slab_blocks,
               const float *__restrict__ cs_partial, int cs_rows, float *__restrict__ cs_out, float cs_scale) {
  if ((int)blockIdx.x >= slab_blocks) {
    x3_colsum_block(cs_partial, cs_rows, N, cs_out, blockIdx.x - slab_blocks, cs_scale);
    return;
  }
  const int n4 = N >> 2;
  const int64_t total = (int64_t)rows * n4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)slab_blocks * blockDim.x) {
    const int64_t r = i / n4;
    const int c = (int)(i - r * n4);
    f32x4 s = reinterpret_cast<const f32x4 *>(slabs)[i];
    for (int z = 1; z < splits; ++z) s += reinterpret_cast<const f32x4 *>(slabs + (int64_t)z * slab_stride)[i];
    if (bias) {
      s += reinterpret_cast<const f32x4 *>(bias)[c];
      s.x = fmaxf(s.x, s.x * alpha); s.y = fmaxf(s.y, s.y * alpha);
      s.z = fmaxf(s.z, s.z * alpha); s.w = fmaxf(s.w, s.w * alpha);
    }
    reinterpret_cast<f32x4 *>(out + r * ldo)[c] = s;
  }
}
}  // namespace
}  // namespace cdml

// C = epilogue(A . B^T) for fp32 A[M][K], B[N][K] given as planes [rows][hi K | mid K | lo K] (plane strides
// plane_a / plane_b >= K along k).  epilogue 1: fp32 C = lrelu(. + bias); 3: fp32 C; 6: C = planes of lrelu(. + bias)
// (bf16 [M][ldc], planes plane_c apart); 7: C = planes of (. times (aux > 0 ? 1 : alpha)), aux = bf16 [M][ldaux];
// 9 = 6 that ALSO writes the sign bitmask of its result to aux (uint8 [M][ldaux BYTES]: bit j of byte b of row r =
// C[r][8 b + j] > 0); 10 = 7 reading that bitmask instead of bf16 values (one bit per element instead of two bytes).
// (f16: the two-plane fp16 form -- three products, planes hi | lo, out_scale / c_scale as BArgs says; cdml_gemm_f16x2_nt)
static int gemm_x3_nt_impl(bool f16, float out_scale, float c_scale, int epilogue, const uint16_t *A, int64_t lda, int64_t plane_a,
                           const uint16_t *B, int64_t ldb, int64_t plane_b, int M, int N, int K, int products, void *C,
                           int64_t ldc, int64_t plane_c, const float *bias, const uint16_t *aux,
                           int64_t ldaux, float alpha, float *colsum, void *workspace,
                           size_t workspace_bytes, cdml_stream_t stream) {
  CDML_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, CDML_E_BADARG, "gemm_bf16x3_nt: bad argument");
  const int np1 = f16 ? 1 : 2;                                // planes after the first
  CDML_REQUIRE(!f16 || (products == 3 && epilogue != BE_MASKBITS_X3_KI && epilogue != BE_ROWBIAS_LRELU_X3 && !colsum &&
                        out_scale > 0.f && c_scale > 0.f),
               CDML_E_UNSUPPORTED, "gemm_f16x2_nt: epilogues 1, 3, 6, 7, 9, 10, positive scales, no column sums");
  const bool kint_out = epilogue == BE_MASKBITS_X3_KI;      // 12 = 10 with the result's planes k8-interleaved
  const bool bits_out = epilogue == BE_BIAS_LRELU_X3_BITS, bits_in = epilogue == BE_MASKBITS_X3 || (kint_out && aux);
  if (bits_out) epilogue = BE_BIAS_LRELU_X3;
  if (bits_in || kint_out) epilogue = BE_MASK_X3;
  CDML_REQUIRE(epilogue == BE_BIAS_LRELU_F32 || epilogue == BE_F32 || epilogue == BE_BIAS_LRELU_X3 || epilogue == BE_MASK_X3 ||
                   epilogue == BE_ROWBIAS_LRELU_X3,
               CDML_E_BADARG, "gemm_bf16x3_nt: epilogue must be 1, 3, 6, 7, 8, 9, 10 or 12");
  CDML_REQUIRE(!(bits_out || bits_in) || (aux && N % 8 == 0 && ldaux >= N / 8), CDML_E_BADARG,
               "gemm_bf16x3_nt: epilogues 9 / 10 need the bitmask in aux (ldaux >= N / 8 bytes)");
  CDML_REQUIRE(products == 3 || products == 6, CDML_E_BADARG, "gemm_bf16x3_nt: products must be 3 or 6");
  CDML_REQUIRE(N % 256 == 0 && K % 64 == 0, CDML_E_UNSUPPORTED,
               "gemm_bf16x3_nt: N must be a multiple of 256 and K of 64, got N=%d K=%d", N, K);
  CDML_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C) && !(lda & 7) && !(ldb & 7) && !(plane_a & 7) && !(plane_b & 7) &&
                   plane_a >= K && plane_b >= K && lda >= np1 * plane_a + K && ldb >= np1 * plane_b + K,
               CDML_E_ALIGN, "gemm_bf16x3_nt: 16-B aligned bases, strides multiples of 8, ld >= 2 plane + K (fp16 form: plane + K)");
  const bool planes_out = epilogue == BE_BIAS_LRELU_X3 || epilogue == BE_MASK_X3 || epilogue == BE_ROWBIAS_LRELU_X3;
  CDML_REQUIRE(kint_out ? (M % 8 == 0 && !(ldc & 7) && ldc >= N && plane_c >= (int64_t)(M / 8) * ldc * 8 &&
                           (int64_t)3 * plane_c * 2 < ((int64_t)1 << 32))
                        : planes_out ? (!(ldc & 7) && !(plane_c & 7) && plane_c >= N && ldc >= np1 * plane_c + N) : (!(ldc & 3) && ldc >= N),
               CDML_E_ALIGN, "gemm_bf16x3_nt: ldc (plane outputs: ldc and plane_c multiples of 8, ldc >= 2 plane_c + N; epilogue 12: M a "
               "multiple of 8, ldc = columns per row group >= N, plane_c >= M ldc)");
  CDML_REQUIRE(epilogue != BE_MASK_X3 || !aux || bits_in || (aligned16(aux) && !(ldaux & 7) && ldaux >= N), CDML_E_ALIGN,
               "gemm_bf16x3_nt: aux must be 16-B aligned with ldaux a multiple of 8");
  CDML_REQUIRE((epilogue != BE_BIAS_LRELU_F32 && epilogue != BE_BIAS_LRELU_X3 && epilogue != BE_ROWBIAS_LRELU_X3) || bias,
               CDML_E_BADARG, "gemm_bf16x3_nt: bias required");
  CDML_REQUIRE(!colsum || (M % 256 == 0 && products == 6 && epilogue == BE_F32), CDML_E_UNSUPPORTED,
               "gemm_bf16x3_nt: colsum (sum over k of B[n][k]) goes with epilogue 3, M a multiple of 256 and six products");
  CDML_REQUIRE(((int64_t)M + 256) * lda * 2 < ((int64_t)1 << 31) && (int64_t)N * ldb * 2 < ((int64_t)1 << 31), CDML_E_UNSUPPORTED,
               "gemm_bf16x3_nt: an operand exceeds the 2 GiB buffer-descriptor range");
  BArgs g{};
  g.A = reinterpret_cast<const bf16 *>(A); g.lda = lda;
  g.B = reinterpret_cast<const bf16 *>(B); g.ldb = ldb;
  g.C = C; g.ldc = ldc; g.bias = bias; g.alpha = alpha;
  g.aux = reinterpret_cast<const bf16 *>(aux); g.ldaux = ldaux;
  if (bits_out) { g.mask_out = reinterpret_cast<uint8_t *>(const_cast<uint16_t *>(aux)); g.ldmask = ldaux; g.aux = nullptr; g.ldaux = 0; }
  g.aux_bits = bits_in ? 1 : 0;
  g.c_kint = kint_out ? 1 : 0;
  g.M = M; g.N = N;
  g.x3_tpp = K / 64; g.x3_plane_a = plane_a; g.x3_plane_b = plane_b; g.x3_plane_c = plane_c;
  g.x3_products = (f16 || x3_kmajor()) ? products : 0;
  g.out_scale = out_scale; g.c_scale = c_scale;
  const int ktiles = products * g.x3_tpp;
  CDML_REQUIRE(ktiles % 2 == 0, CDML_E_UNSUPPORTED, "gemm_bf16x3_nt: products * K / 64 must be even");
  g.K = ktiles * 64; g.k_per_split = g.K;
  g.tiles_m = (M + 255) / 256; g.tiles_n = N / 256;
  hipStream_t s = (hipStream_t)stream;
  auto launch = [&](int epi, int n_splits) {
    return f16 ? launch_gemm_f16x2_256(g, false, epi, n_splits, s) : launch_gemm_bf16_256_x3(g, false, epi, n_splits, s);
  };
  int splits = planes_out ? 1 : x3_nt_splits(N, ktiles, g.tiles_m, f16), per = ktiles;
  if (splits > 1) x3_split_geometry(ktiles, splits, 6, per, splits);   // whole six-step periods of the K-major walk (and an even count; fp16 form: two K-tiles)
  if (splits > 1) {
    // The slab form needs the workspace.  workspace == NULL is the caller's explicit choice of ONE pass over K (fewer,
    // longer blocks; no slab round trip) -- its sums are associated differently, so the result's last bits differ from
    // the slab form's and the "same bits whole or in row blocks" property holds only among calls that all pass a workspace.
    // A workspace that is given but too small is an error, not a silent change of arithmetic.
    if (!workspace) {
      splits = 1;
    } else {
      const size_t need = (size_t)splits * M * N * sizeof(float) + (colsum ? (size_t)splits * g.tiles_m * 2 * N * sizeof(float) : 0);
      CDML_REQUIRE(workspace_bytes >= need && aligned16(workspace), CDML_E_BADARG,
                   "gemm_bf16x3_nt: the slab form needs a 16-B aligned workspace of %zu bytes (cdml_gemm_bf16x3_workspace); pass "
                   "NULL for the single-pass form", need);
    }
  }
  const size_t slab_bytes = splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
  const size_t cs_rows = (size_t)splits * g.tiles_m * 2;
  if (colsum) {      // colsum[n] = sum_k B[n][k] (the bias gradient when B holds a transposed activation gradient)
    CDML_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= slab_bytes + cs_rows * N * sizeof(float), CDML_E_BADARG,
                 "gemm_bf16x3_nt: colsum needs a workspace (cdml_gemm_bf16x3_workspace)");
    g.colsum_partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + slab_bytes);
  }
  int rc;
  if (splits == 1) {
    rc = launch(epilogue, 1);
  } else {
    g.k_per_split = per * 64;
    g.slab_stride = (int64_t)M * N;
    g.C = workspace; g.ldc = N;
    rc = launch(BE_F32, splits);                            // (fp16 form: the slabs are written times out_scale already)
    if (rc) return rc;
    const int sb = grid1d((int64_t)M * N / 4), cb = colsum ? (N + 15) / 16 : 0;
    hipLaunchKernelGGL(k_x3_sum_slabs, dim3(sb + cb), dim3(kThreads), 0, s,
                       static_cast<const float *>(workspace), g.slab_stride, splits, M, N,
                       epilogue == BE_BIAS_LRELU_F32 ? bias : nullptr, alpha, static_cast<float *>(C), ldc, sb,
                       g.colsum_partial, (int)cs_rows, colsum, 1.0f);
    return check_launch("gemm_bf16x3_nt combine");
  }
  if (rc || !colsum) return rc;
  hipLaunchKernelGGL(k_x3_sum_slabs, dim3((N + 15) / 16), dim3(kThreads), 0, s, static_cast<const float *>(nullptr), (int64_t)0, 0, 0,
                     N, static_cast<const float *>(nullptr), 0.f, static_cast<float *>(nullptr), (int64_t)0, 0, g.colsum_partial,
                     (int)cs_rows, colsum, 1.0f);
  return check_launch("gemm_bf16x3_nt bias gradient");
}

extern "C" int cdml_gemm_bf16x3_nt(int epilogue, const uint16_t *A, int64_t lda, int64_t plane_a, const uint16_t *B,
                                   int64_t ldb, int64_t plane_b, int M, int N, int K, int products, void *C,
                                   int64_t ldc, int64_t plane_c, const float *bias, const uint16_t *aux,
                                   int64_t ldaux, float alpha, float *colsum, void *workspace,
                                   size_t workspace_bytes, cdml_stream_t stream) {
  return gemm_x3_nt_impl(false, 1.0f, 1.0f, epilogue, A, lda, plane_a, B, ldb, plane_b, M, N, K, products, C, ldc, plane_c, bias, aux,
                         ldaux, alpha, colsum, workspace, workspace_bytes, stream);
}

// The same product on TWO fp16 planes per operand (precision "f16x2"; gemm_f16x2_256.hip): A = planes hi | lo of a * 2^sa
// [M][lda] (plane_a apart), B likewise of b * 2^sb; out_scale = 2^-(sa + sb) multiplies the accumulator before the epilogue;
// epilogues 6 / 7 / 9 / 10 write C as the two fp16 planes of (result * c_scale), clamped to fp16's range.  Three plane
// products (hi.hi + hi.lo + lo.hi) on v_mfma_f32_16x16x32_f16.  Epilogues 1, 3, 6, 7, 9, 10; workspace as
// cdml_gemm_bf16x3_workspace(0, M, N, K, 3).
extern "C" int cdml_gemm_f16x2_nt(int epilogue, const uint16_t *A, int64_t lda, int64_t plane_a, const uint16_t *B,
                                  int64_t ldb, int64_t plane_b, int M, int N, int K, void *C, int64_t ldc, int64_t plane_c,
                                  const float *bias, const uint16_t *aux, int64_t ldaux, float alpha, float out_scale,
                                  float c_scale, void *workspace, size_t workspace_bytes, cdml_stream_t stream) {
  return gemm_x3_nt_impl(true, out_scale, c_scale, epilogue, A, lda, plane_a, B, ldb, plane_b, M, N, K, 3, C, ldc, plane_c, bias, aux,
                         ldaux, alpha, nullptr, workspace, workspace_bytes, stream);
}

// C[M][N] (fp32) = sum_k A[k][M] B[k][N] for fp32 operands given as planes [K][hi | mid | lo] (plane strides along
// the columns); colsum[n] = sum_k B[k][n] on request (the bias gradient).
static int gemm_x3_tn_impl(bool f16, float out_scale, float cs_scale, const uint16_t *A, int64_t lda, int64_t plane_a,
                           const uint16_t *B, int64_t ldb, int64_t plane_b, int M, int N, int K, int products, float *C, int64_t ldc,
                           const float *bias, float alpha, float *colsum, void *workspace,
                           size_t workspace_bytes, cdml_stream_t stream) {
  CDML_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, CDML_E_BADARG, "gemm_bf16x3_tn: bad argument");
  const int np1 = f16 ? 1 : 2;
  CDML_REQUIRE(!f16 || (products == 3 && out_scale > 0.f && cs_scale > 0.f), CDML_E_BADARG, "gemm_f16x2_tn: positive scales");
  CDML_REQUIRE(products == 3 || products == 6, CDML_E_BADARG, "gemm_bf16x3_tn: products must be 3 or 6");
  CDML_REQUIRE(M % 256 == 0 && N % 256 == 0 && K % 128 == 0, CDML_E_UNSUPPORTED,
               "gemm_bf16x3_tn: M, N must be multiples of 256 and K of 128, got M=%d N=%d K=%d", M, N, K);
  CDML_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C) && !(lda & 7) && !(ldb & 7) && !(plane_a & 7) && !(plane_b & 7) &&
                   !(ldc & 3) && plane_a >= M && plane_b >= N && lda >= np1 * plane_a + M && ldb >= np1 * plane_b + N && ldc >= N,
               CDML_E_ALIGN, "gemm_bf16x3_tn: 16-B aligned bases, strides multiples of 8, ld >= 2 plane + columns");
  CDML_REQUIRE((int64_t)K * lda * 2 < ((int64_t)1 << 31) && (int64_t)K * ldb * 2 < ((int64_t)1 << 31), CDML_E_UNSUPPORTED,
               "gemm_bf16x3_tn: an operand exceeds the 2 GiB buffer-descriptor range");
  BArgs g{};
  g.A = reinterpret_cast<const bf16 *>(A); g.lda = lda;
  g.B = reinterpret_cast<const bf16 *>(B); g.ldb = ldb;
  g.M = M; g.N = N;
  g.x3_tpp = K / 64; g.x3_plane_a = plane_a; g.x3_plane_b = plane_b;
  g.x3_products = (f16 || x3_kmajor()) ? products : 0;
  g.out_scale = out_scale; g.c_scale = 1.0f;
  const int ktiles = products * g.x3_tpp;
  g.K = ktiles * 64;
  g.tiles_m = M / 256; g.tiles_n = N / 256;
  int splits = gemm_bf16_256_splits(M, N, g.K), per = 0;
  // whole six-step periods of the K-major walk per block (the unrolled loops need them); three products: an even count
  // (the fp16 form: whole three-step periods, an even count for the general loop)
  x3_split_geometry(ktiles, splits, ((products == 6 && g.x3_products == 6) || f16) ? 6 : 2, per, splits);
  const bool slabs = splits > 1 || bias != nullptr;       // (bias + leaky-relu are applied by the combine pass)
  const size_t slab_bytes = slabs ? (size_t)splits * M * N * sizeof(float) : 0;
  const size_t cs_rows = (size_t)splits * g.tiles_m * 2;
  const size_t need = slab_bytes + (colsum ? cs_rows * N * sizeof(float) : 0);
  CDML_REQUIRE(need == 0 || (workspace && workspace_bytes >= need && aligned16(workspace)), CDML_E_BADARG,
               "gemm_bf16x3_tn: workspace of %zu bytes required (cdml_gemm_bf16x3_workspace)", need);
  hipStream_t s = (hipStream_t)stream;
  g.k_per_split = per * 64;
  g.slab_stride = (int64_t)M * N;
  g.C = slabs ? workspace : static_cast<void *>(C);
  g.ldc = slabs ? N : ldc;
  g.colsum_partial = colsum ? reinterpret_cast<float *>(static_cast<char *>(workspace) + slab_bytes) : nullptr;
  int rc = f16 ? launch_gemm_f16x2_256(g, true, BE_F32, splits, s) : launch_gemm_bf16_256_x3(g, true, BE_F32, splits, s);
  if (rc) return rc;
  if (slabs || colsum) {                                   // one launch: the slab sum and, in extra blocks, the bias gradient
    const int sb = slabs ? grid1d((int64_t)M * N / 4) : 0, cb = colsum ? (N + 15) / 16 : 0;
    hipLaunchKernelGGL(k_x3_sum_slabs, dim3(sb + cb), dim3(kThreads), 0, s, static_cast<const float *>(workspace), g.slab_stride,
                       splits, M, N, bias, alpha, C, ldc, sb, g.colsum_partial, (int)cs_rows, colsum, cs_scale);
    rc = check_launch("gemm_bf16x3_tn combine");
  }
  return rc;
}

extern "C" int cdml_gemm_bf16x3_tn(const uint16_t *A, int64_t lda, int64_t plane_a, const uint16_t *B, int64_t ldb,
                                   int64_t plane_b, int M, int N, int K, int products, float *C, int64_t ldc,
                                   const float *bias, float alpha, float *colsum, void *workspace,
                                   size_t workspace_bytes, cdml_stream_t stream) {
  return gemm_x3_tn_impl(false, 1.0f, 1.0f, A, lda, plane_a, B, ldb, plane_b, M, N, K, products, C, ldc, bias, alpha, colsum, workspace,
                         workspace_bytes, stream);
}

// The k-strided product on two fp16 planes per operand (cdml_gemm_f16x2_nt's form): C = out_scale * sum_k A[k][M] B[k][N],
// colsum[n] = colsum_scale * sum_k B[k][n] (colsum_scale = 2^-sb).  Workspace: cdml_gemm_bf16x3_workspace(1, M, N, K, 3).
extern "C" int cdml_gemm_f16x2_tn(const uint16_t *A, int64_t lda, int64_t plane_a, const uint16_t *B, int64_t ldb,
                                  int64_t plane_b, int M, int N, int K, float *C, int64_t ldc, float out_scale, float *colsum,
                                  float colsum_scale, void *workspace, size_t workspace_bytes, cdml_stream_t stream) {
  return gemm_x3_tn_impl(true, out_scale, colsum_scale, A, lda, plane_a, B, ldb, plane_b, M, N, K, 3, C, ldc, nullptr, 0.f, colsum,
                         workspace, workspace_bytes, stream);
}

// ---- semi-hard negative mining fused with its score product (BASELINE config 2; build-defined, spec oracle/tower.py
// semihard_select) ------------------------------------------------------------------------------------------------------
// Round 2 wrote S = E_anchor . E^T (B x 2B fp32: 537 MB at B = 8192) with the fp32-MFMA data-gradient kernel and scanned it
// with k_semihard_select (0.54 + 0.14 ms of a 3.66-ms step).  Here the product runs on the plane kernels (six bf16 plane
// products per fp32 product) and the selection is its EPILOGUE (gemm_bf16_256.hip, BE_MINE_X3): S never exists in memory.
//   1. k_mine_prep: one wave per triplet -- |e|^2 of the anchor and the positive row, d_p = |a|^2 + |p|^2 - 2 <a, p>, both
//      rows as planes hi | mid | lo;
//   2. the score product, 256 anchors x 256 rows per tile: per (tile column, 64-column strip) and anchor the closest
//      eligible row with d > d_p and the farthest eligible row (16 B);
//   3. k_semihard_finish: per anchor the merge of its 4 tiles_n records -> neg_row.
namespace cdml {
namespace {

// z != null (round 6): the rows arrive UN-normalised (the output layer's z) -- the wave normalises both of its rows first, with
// cdml_l2norm_fwd's arithmetic (models.py:61: z / sqrt(max(sum z^2, 1e-12))), and writes them to e: one launch and a 2 x 16 MB
// round trip fewer in BASELINE config 2's step; e as cdml_l2norm_fwd writes it to an ulp (which multiply-adds become fmas differs
// from kernel to kernel), planes, norms and distances those of the e written here.
__global__ void __launch_bounds__(kThreads)
k_mine_prep(const float *__restrict__ e_in, float *__restrict__ e_out, int64_t lde, const float *__restrict__ z, int64_t ldz,
            int B, int D, bf16 *__restrict__ e3, int64_t ld3, int64_t plane, float *__restrict__ sqn, float *__restrict__ dp,
            float h2_scale) {      // h2_scale > 0: the rows as TWO fp16 planes of e * h2_scale instead of three bf16 planes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  for (int i = blockIdx.x * (kThreads / 64) + wave; i < B; i += gridDim.x * (kThreads / 64)) {
    if (z) {
      const float *za = z + (int64_t)(2 * i) * ldz, *zp = za + ldz;
      float *ea = e_out + (int64_t)(2 * i) * lde, *ep = ea + lde;
      float s0 = 0.f, s1 = 0.f;
      for (int q = lane; q < nq; q += 64) {
        const f32x4 x = reinterpret_cast<const f32x4 *>(za)[q], y = reinterpret_cast<const f32x4 *>(zp)[q];
        s0 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        s1 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
      }
      s0 = wave_sum(s0);
      s1 = wave_sum(s1);
      const float i0 = 1.0f / sqrtf(fmaxf(s0, 1e-12f)), i1 = 1.0f / sqrtf(fmaxf(s1, 1e-12f));
      for (int q = lane; q < nq; q += 64) {           // (each lane reads back below exactly what it stores here)
        const f32x4 x = reinterpret_cast<const f32x4 *>(za)[q], y = reinterpret_cast<const f32x4 *>(zp)[q];
        reinterpret_cast<f32x4 *>(ea)[q] = f32x4{x.x * i0, x.y * i0, x.z * i0, x.w * i0};
        reinterpret_cast<f32x4 *>(ep)[q] = f32x4{y.x * i1, y.y * i1, y.z * i1, y.w * i1};
      }
    }
    const float *e = z ? e_out : e_in;
    const float *a = e + (int64_t)(2 * i) * lde, *p = a + lde;
    float sa = 0.f, sp = 0.f, ap = 0.f;
    for (int q = lane; q < nq; q += 64) {
      const f32x4 x = reinterpret_cast<const f32x4 *>(a)[q], y = reinterpret_cast<const f32x4 *>(p)[q];
      sa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      sp += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
      ap += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      bf16x4 h, m, l;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const f32x4 v = r ? y : x;
        if (h2_scale > 0.f) {                              // (uniform)
          half4v hh, hl;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            _Float16 a, b;
            split2h(v[j], h2_scale, a, b);
            hh[j] = a; hl[j] = b;
          }
          _Float16 *d = reinterpret_cast<_Float16 *>(e3) + (int64_t)(2 * i + r) * ld3 + 4 * q;
          *reinterpret_cast<half4v *>(d) = hh;
          *reinterpret_cast<half4v *>(d + plane) = hl;
          continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16 b0, b1, b2;
          split3(v[j], b0, b1, b2);
          h[j] = b0; m[j] = b1; l[j] = b2;
        }
        bf16 *d = e3 + (int64_t)(2 * i + r) * ld3 + 4 * q;
        *reinterpret_cast<bf16x4 *>(d) = h;
        *reinterpret_cast<bf16x4 *>(d + plane) = m;
        *reinterpret_cast<bf16x4 *>(d + 2 * plane) = l;
      }
    }
    sa = wave_sum(sa); sp = wave_sum(sp); ap = wave_sum(ap);
    if (lane == 0) {
      sqn[2 * i] = sa;
      sqn[2 * i + 1] = sp;
      dp[i] = (sa + sp) - 2.0f * ap;
    }
  }
}

// neg_row[i] = the closest "outside" candidate over all strips, else the farthest eligible one, else -1 (masked).
// 256 threads = 64 anchors x 4 groups of strips; records of one strip are contiguous over the anchors (coalesced).
__global__ void __launch_bounds__(kThreads)
k_semihard_finish(const MineCand *__restrict__ part, int64_t ld, int n_strips, int B, int32_t *__restrict__ neg_row) {
  __shared__ MineCand red[4][64];
  const int a = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  const float inf = __builtin_huge_valf();
  MineCand best{inf, 0x7fffffff, -inf, 0x7fffffff};
  auto merge = [&](const MineCand &o) {
    if (o.out_d < best.out_d || (o.out_d == best.out_d && o.out_c < best.out_c)) { best.out_d = o.out_d; best.out_c = o.out_c; }
    if (o.in_d > best.in_d || (o.in_d == best.in_d && o.in_c < best.in_c)) { best.in_d = o.in_d; best.in_c = o.in_c; }
  };
  if (a < B) {
    int s = grp;
    for (; s + 12 < n_strips; s += 16) {             // four records in flight
      const MineCand c0 = part[(int64_t)s * ld + a], c1 = part[(int64_t)(s + 4) * ld + a];
      const MineCand c2 = part[(int64_t)(s + 8) * ld + a], c3 = part[(int64_t)(s + 12) * ld + a];
      merge(c0); merge(c1); merge(c2); merge(c3);
    }
    for (; s < n_strips; s += 4) merge(part[(int64_t)s * ld + a]);
  }
  red[grp][threadIdx.x & 63] = best;
  __syncthreads();
  if (grp == 0 && a < B) {
    merge(red[1][threadIdx.x]); merge(red[2][threadIdx.x]); merge(red[3][threadIdx.x]);
    neg_row[a] = best.out_c != 0x7fffffff ? best.out_c : (best.in_c != 0x7fffffff ? best.in_c : -1);
  }
}

}  // namespace
}  // namespace cdml

extern "C" size_t cdml_semihard_mine_x3_workspace(int B) {
  if (B < 1 || (2 * (int64_t)B) % 256) return 0;
  return (size_t)(2 * B / 256) * 4 * (size_t)B * sizeof(MineCand);
}

// e[2B][lde] fp32 (row 2i = anchor i, 2i+1 = its positive; l2-normalised or not), rows[2B] = video ids.  Scratch the
// caller owns: e_planes bf16 [2B][ldp] (planes `plane` apart), sqn float[2B], dp float[B], workspace
// (cdml_semihard_mine_x3_workspace).  neg_row_out[i] as cdml_semihard_select.  2B % 256 == 0, D % 64 == 0.
static int semihard_mine_x3_impl(const float *e, float *e_out, int64_t lde, const float *z, int64_t ldz, const int32_t *rows, int B,
                                 int D, uint16_t *e_planes, int64_t ldp, int64_t plane, float *sqn, float *dp, void *workspace,
                                 size_t workspace_bytes, int32_t *neg_row_out, cdml_stream_t stream, float h2_scale = 0.f) {
  CDML_REQUIRE(e && rows && e_planes && sqn && dp && workspace && neg_row_out && B >= 1 && D > 0, CDML_E_BADARG,
               "semihard_mine_x3: bad argument");
  CDML_REQUIRE(!z || (aligned16(z) && !(ldz & 3) && ldz >= D), CDML_E_ALIGN, "semihard_mine_x3_z: z 16-B aligned, ldz a multiple of 4 and >= D");
  CDML_REQUIRE((2 * (int64_t)B) % 256 == 0 && D % 64 == 0, CDML_E_UNSUPPORTED,
               "semihard_mine_x3: 2 B must be a multiple of 256 and D of 64, got B=%d D=%d", B, D);
  CDML_REQUIRE(aligned16(e) && !(lde & 3) && lde >= D && aligned16(e_planes) && !(ldp & 7) && !(plane & 7) && plane >= D &&
                   ldp >= (h2_scale > 0.f ? 1 : 2) * plane + D && aligned16(workspace) && aligned16(rows) && aligned16(sqn),
               CDML_E_ALIGN, "semihard_mine_x3: 16-B aligned bases, lde a multiple of 4, ldp / plane multiples of 8, ldp >= 2 plane + D");
  CDML_REQUIRE(workspace_bytes >= cdml_semihard_mine_x3_workspace(B), CDML_E_BADARG,
               "semihard_mine_x3: workspace of %zu bytes required", cdml_semihard_mine_x3_workspace(B));
  CDML_REQUIRE(((int64_t)B + 256) * 2 * ldp * 2 < ((int64_t)1 << 31), CDML_E_UNSUPPORTED,
               "semihard_mine_x3: the embedded rows exceed the 2 GiB buffer-descriptor range");
  hipStream_t s = (hipStream_t)stream;
  bf16 *e3 = reinterpret_cast<bf16 *>(e_planes);
  hipLaunchKernelGGL(k_mine_prep, dim3(grid1d((int64_t)B * 64)), dim3(kThreads), 0, s, e, e_out, lde, z, ldz, B, D, e3, ldp, plane, sqn, dp,
                     h2_scale);
  int rc = check_launch("semihard_mine_x3 prep");
  if (rc) return rc;
  BArgs g{};
  g.A = e3; g.lda = 2 * ldp;                                // anchors = even rows
  g.B = e3; g.ldb = ldp;
  g.M = B; g.N = 2 * B;
  const int prod = h2_scale > 0.f ? 3 : 6;
  g.x3_tpp = D / 64; g.x3_plane_a = plane; g.x3_plane_b = plane; g.x3_products = prod;
  g.K = prod * g.x3_tpp * 64; g.k_per_split = g.K;
  g.tiles_m = (B + 255) / 256; g.tiles_n = 2 * B / 256;
  g.mine_sqn = sqn; g.mine_ids = rows; g.mine_dp = dp;
  g.mine_out = static_cast<MineCand *>(workspace); g.mine_ld = B;
  g.out_scale = h2_scale > 0.f ? 1.0f / (h2_scale * h2_scale) : 1.0f; g.c_scale = 1.0f;
  rc = h2_scale > 0.f ? launch_gemm_f16x2_mine(g, s) : launch_gemm_x3_mine(g, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_semihard_finish, dim3((B + 63) / 64), dim3(kThreads), 0, s, static_cast<const MineCand *>(workspace),
                     (int64_t)B, g.tiles_n * 4, B, neg_row_out);
  return check_launch("semihard_mine_x3 finish");
}

extern "C" int cdml_semihard_mine_x3(const float *e, int64_t lde, const int32_t *rows, int B, int D, uint16_t *e_planes,
                                     int64_t ldp, int64_t plane, float *sqn, float *dp, void *workspace,
                                     size_t workspace_bytes, int32_t *neg_row_out, cdml_stream_t stream) {
  return semihard_mine_x3_impl(e, nullptr, lde, nullptr, 0, rows, B, D, e_planes, ldp, plane, sqn, dp, workspace, workspace_bytes,
                               neg_row_out, stream);
}

// The same from the output layer's UN-normalised rows z: the prep launch normalises them (cdml_l2norm_fwd's arithmetic,
// models.py:61) and writes e as well -- the first 2 B rows of e are an OUTPUT here.
extern "C" int cdml_semihard_mine_x3_z(const float *z, int64_t ldz, float *e, int64_t lde, const int32_t *rows, int B, int D,
                                       uint16_t *e_planes, int64_t ldp, int64_t plane, float *sqn, float *dp,
                                       void *workspace, size_t workspace_bytes, int32_t *neg_row_out, cdml_stream_t stream) {
  CDML_REQUIRE(z, CDML_E_BADARG, "semihard_mine_x3_z: z required");
  return semihard_mine_x3_impl(e, e, lde, z, ldz, rows, B, D, e_planes, ldp, plane, sqn, dp, workspace, workspace_bytes,
                               neg_row_out, stream);
}

// cdml_semihard_mine_x3 / _z (z nullable: e given when it is) with the score product on TWO fp16 planes per row -- the rows
// times `scale` (a power of two; 2^14 for unit rows), three plane products on the fp16 MFMA (precision "f16x2"): e_planes =
// fp16 [2B][ldp], ldp >= plane + D.
extern "C" int cdml_semihard_mine_h2(const float *z, int64_t ldz, float *e, int64_t lde, const int32_t *rows, int B, int D,
                                     uint16_t *e_planes, int64_t ldp, int64_t plane, float scale, float *sqn, float *dp,
                                     void *workspace, size_t workspace_bytes, int32_t *neg_row_out, cdml_stream_t stream) {
  CDML_REQUIRE(scale > 0.f, CDML_E_BADARG, "semihard_mine_h2: a positive scale");
  return semihard_mine_x3_impl(e, z ? e : nullptr, lde, z, ldz, rows, B, D, e_planes, ldp, plane, sqn, dp, workspace, workspace_bytes,
                               neg_row_out, stream, scale);
}

// ---- k8-interleaved operands for the weight gradients (round 5) ---------------------------------------------------------
// A weight gradient contracts over the batch rows, so its operands -- stored row-major, one row per batch row -- are
// k-STRIDED, and a fragment (8 consecutive k of one column) takes two transposed LDS reads (ds_read_b64_tr_b16): twice the LDS
// instructions of the k-contiguous products per phase, and the k-strided kernel's phases are bound by their read part (MFMA
// busy 0.76-0.79 against 0.84-0.85).  Stored [plane][row / 8][column][8 rows] the same fragment is ONE aligned 16-B read and an
// LDS image is filled by contiguous 1-KiB pieces: cdml_gemm_bf16x3_tnk.  cdml_interleave8_bf16x3 converts row-major planes.
namespace cdml {
namespace {
// dst[(p * (rows / 8) + r / 8) * cols * 8 + c * 8 + r % 8] = src[r][p * plane_src + c]; one thread per (k-group, 8-column run)
__global__ void __launch_bounds__(kThreads)
k_interleave8(const bf16 *__restrict__ src, int64_t ld_src, int64_t plane_src, int rows, int cols, bf16 *__restrict__ dst) {
  using bf16x8v = __attribute__((ext_vector_type(8))) __bf16;
  const int c8n = cols >> 3, kgs = rows >> 3;
  const int64_t total = (int64_t)3 * kgs * c8n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % c8n);
    const int64_t t = i / c8n;
    const int kg = (int)(t % kgs), p = (int)(t / kgs);
    bf16x8v in[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      in[r] = *reinterpret_cast<const bf16x8v *>(src + (int64_t)(kg * 8 + r) * ld_src + p * plane_src + c8 * 8);
    bf16 *d = dst + ((int64_t)(p * kgs + kg) * cols + c8 * 8) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      bf16x8v o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = in[r][c];
      *reinterpret_cast<bf16x8v *>(d + c * 8) = o;
    }
  }
}
}  // namespace
}  // namespace cdml

// src: bf16 planes row-major [rows][ld_src], plane p at columns p * plane_src (what the gather / the epilogues write);
// dst: bf16 [3][rows / 8][cols][8].  rows % 8 == 0, cols % 8 == 0.
extern "C" int cdml_interleave8_bf16x3(const uint16_t *src, int64_t ld_src, int64_t plane_src, int rows, int cols,
                                       uint16_t *dst, cdml_stream_t stream) {
  CDML_REQUIRE(src && dst && rows > 0 && cols > 0, CDML_E_BADARG, "interleave8_bf16x3: bad argument");
  CDML_REQUIRE(rows % 8 == 0 && cols % 8 == 0 && !(ld_src & 7) && !(plane_src & 7) && plane_src >= cols && ld_src >= 2 * plane_src + cols &&
                   aligned16(src) && aligned16(dst), CDML_E_ALIGN,
               "interleave8_bf16x3: rows, cols, ld_src, plane_src multiples of 8, 16-B aligned bases");
  hipLaunchKernelGGL(k_interleave8, dim3(grid1d((int64_t)3 * (rows / 8) * (cols / 8))), dim3(kThreads), 0, (hipStream_t)stream,
                     reinterpret_cast<const bf16 *>(src), ld_src, plane_src, rows, cols, reinterpret_cast<bf16 *>(dst));
  return check_launch("interleave8_bf16x3");
}

// C[M][N] (fp32) = sum_k A[k][M] B[k][N] like cdml_gemm_bf16x3_tn, the operands k8-INTERLEAVED: A = bf16 [3][K / 8][ma][8]
// with ma >= M columns per k-group (the product takes columns [a_col0, a_col0 + M)), B = bf16 [3][K / 8][nb][8] likewise.
// Six products; M, N % 256 == 0, K % 128 == 0; same split-K, slab combine, colsum and workspace as the row-major form
// (cdml_gemm_bf16x3_workspace(1, M, N, K, 6)); bit-identical results.
extern "C" int cdml_gemm_bf16x3_tnk(const uint16_t *A, int ma, int a_col0, const uint16_t *B, int nb, int b_col0, int M, int N,
                                    int K, float *C, int64_t ldc, float *colsum, void *workspace, size_t workspace_bytes,
                                    cdml_stream_t stream) {
  CDML_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, CDML_E_BADARG, "gemm_bf16x3_tnk: bad argument");
  CDML_REQUIRE(M % 256 == 0 && N % 256 == 0 && K % 128 == 0, CDML_E_UNSUPPORTED,
               "gemm_bf16x3_tnk: M, N must be multiples of 256 and K of 128, got M=%d N=%d K=%d", M, N, K);
  CDML_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C) && !(ldc & 3) && ldc >= N && a_col0 >= 0 && b_col0 >= 0 &&
                   a_col0 + M <= ma && b_col0 + N <= nb && !(ma & 7) && !(nb & 7),
               CDML_E_ALIGN, "gemm_bf16x3_tnk: 16-B aligned bases, column windows inside the operands");
  CDML_REQUIRE((int64_t)3 * (K / 8) * ma * 16 < ((int64_t)1 << 31) && (int64_t)3 * (K / 8) * nb * 16 < ((int64_t)1 << 31), CDML_E_UNSUPPORTED,
               "gemm_bf16x3_tnk: an operand exceeds the 2 GiB buffer-descriptor range");
  BArgs g{};
  g.A = reinterpret_cast<const bf16 *>(A) + (int64_t)a_col0 * 8; g.lda = (int64_t)ma * 8;
  g.B = reinterpret_cast<const bf16 *>(B) + (int64_t)b_col0 * 8; g.ldb = (int64_t)nb * 8;
  g.M = M; g.N = N;
  g.x3_tpp = K / 64; g.x3_plane_a = (int64_t)(K / 8) * ma * 8; g.x3_plane_b = (int64_t)(K / 8) * nb * 8;
  g.x3_products = 6;
  const int ktiles = 6 * g.x3_tpp;
  g.K = ktiles * 64;
  g.tiles_m = M / 256; g.tiles_n = N / 256;
  int splits = gemm_bf16_256_splits(M, N, g.K), per = 0;
  x3_split_geometry(ktiles, splits, 6, per, splits);
  const bool slabs = splits > 1;
  const size_t slab_bytes = slabs ? (size_t)splits * M * N * sizeof(float) : 0;
  const size_t cs_rows = (size_t)splits * g.tiles_m * 2;
  const size_t need = slab_bytes + (colsum ? cs_rows * N * sizeof(float) : 0);
  CDML_REQUIRE(need == 0 || (workspace && workspace_bytes >= need && aligned16(workspace)), CDML_E_BADARG,
               "gemm_bf16x3_tnk: workspace of %zu bytes required (cdml_gemm_bf16x3_workspace)", need);
  hipStream_t s = (hipStream_t)stream;
  g.k_per_split = per * 64;
  g.slab_stride = (int64_t)M * N;
  g.C = slabs ? workspace : static_cast<void *>(C);
  g.ldc = slabs ? N : ldc;
  g.colsum_partial = colsum ? reinterpret_cast<float *>(static_cast<char *>(workspace) + slab_bytes) : nullptr;
  int rc = launch_gemm_x3_tnk(g, splits, s);
  if (rc) return rc;
  if (slabs || colsum) {
    const int sb = slabs ? grid1d((int64_t)M * N / 4) : 0, cb = colsum ? (N + 15) / 16 : 0;
    hipLaunchKernelGGL(k_x3_sum_slabs, dim3(sb + cb), dim3(kThreads), 0, s, static_cast<const float *>(workspace), g.slab_stride,
                       splits, M, N, static_cast<const float *>(nullptr), 0.f, C, ldc, sb, g.colsum_partial, (int)cs_rows, colsum, 1.0f);
    rc = check_launch("gemm_bf16x3_tnk combine");
  }
  return rc;
}
