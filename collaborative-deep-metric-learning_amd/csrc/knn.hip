// Exact k-nearest-neighbour export (reference: faiss_knn.py:82-131 `calc_knn`,
// which builds a faiss HNSW index over the l2-normalised embeddings and returns
// squared-L2 distances D and neighbour ids I, nearest first, query included).
// Here it is brute force: the inner products of a query block with a block of
// the catalogue come from the fp32 MFMA GEMM (cdml_fc_bwd_data with no mask =
// C = A.B^T), and this file keeps, per query, the running list of the best
// candidates while the score blocks stream by.
//
//   k_row_sqnorm   |x_r|^2 per row (one wave per row)
//   k_knn_merge    one wave per query row: scan a [nq, nb] block of inner products,
//                  d = |q|^2 + |b|^2 - 2 q.b, keep everything that beats the current
//                  k-th key in a 256-entry LDS buffer, compact it with a bitonic sort
//                  whenever it fills, write the 128 best (sorted by (d, id)) back.
//                  HBM-bound: reads nq*nb*4 B of scores once.
//   k_knn_merge_list  (round 6) the same per-query list fed from a CANDIDATE list instead of a score row: the plane GEMM's
//                  kNN-filter epilogue (gemm_bf16_256.hip, BE_KNN_X3) appends every element within a query's current
//                  k-th best distance to that query's list; no score block is written at all.
#include "gemm_bf16.h"

namespace cdml {
namespace {

constexpr int kListCap = CDML_KNN_LIST;   // entries kept per query (k <= kListCap)
constexpr int kBuf = 2 * kListCap;        // LDS buffer per wave
constexpr int kRowsPerBlock = 4;          // one wave per query row
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_row_sqnorm(const float *x, int64_t ldx, int n, int D, float *out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float *p = x + (int64_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s = fmaf(p[c], p[c], s);
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

__device__ __forceinline__ bool key_less(float d1, int i1, float d2, int i2) {
  return d1 < d2 || (d1 == d2 && i1 < i2);
}

// In-place bitonic sort of the wave's 256 (d, id) keys in LDS, ascending.  All 64
// lanes of ONE wave take part; LDS operations of a wave retire in order, so the
// only fence needed between passes is against compiler reordering.
__device__ __forceinline__ void sort_buffer(float *kd, int *ki, int lane) {
#pragma unroll 1
  for (int size = 2; size <= kBuf; size <<= 1) {
#pragma unroll 1
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
      for (int t = 0; t < kBuf / 128; ++t) {
        const int p = lane + 64 * t;
        const int i = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
        const int j = i + stride;
        const bool up = (i & size) == 0;
        const float di = kd[i], dj = kd[j];
        const int ii = ki[i], ij = ki[j];
        if (key_less(dj, ij, di, ii) == up) {
          kd[i] = dj; ki[i] = ij;
          kd[j] = di; ki[j] = ii;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

__global__ void __launch_bounds__(256)
k_knn_merge(const float *scores, int64_t lds, int nq, int nb, int col0, int n_valid, const float *q_sq,
            const float *b_sq, int k, float *best_d, int *best_i, int first) {
  __shared__ float s_d[kRowsPerBlock][kBuf];
  __shared__ int s_i[kRowsPerBlock][kBuf];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowsPerBlock + w;
  if (row >= nq) return;                       // whole wave leaves; no block barrier below
  float *kd = s_d[w];
  int *ki = s_i[w];
  const float inf = __builtin_inff();
  float *bd = best_d + (int64_t)row * kListCap;
  int *bi = best_i + (int64_t)row * kListCap;
#pragma unroll
  for (int t = 0; t < kListCap / 64; ++t) {
    const int p = lane + 64 * t;
    kd[p] = first ? inf : bd[p];
    ki[p] = first ? 0x7fffffff : bi[p];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float tau_d = kd[k - 1];
  int tau_i = ki[k - 1];
  int n_buf = kListCap;                        // wave-uniform

  auto compact = [&]() {
#pragma unroll
    for (int t = 0; t < kBuf / 64; ++t) {
      const int p = lane + 64 * t;
      if (p >= n_buf) { kd[p] = inf; ki[p] = 0x7fffffff; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    sort_buffer(kd, ki, lane);
    tau_d = kd[k - 1];
    tau_i = ki[k - 1];
    n_buf = kListCap;
  };

  const float qs = q_sq[row];
  const float *srow = scores + (int64_t)row * lds;
  for (int c0 = 0; c0 < nb; c0 += 256) {
    const int c = c0 + lane * 4;
    f32x4 s = (f32x4)(0.f), bs = (f32x4)(0.f);
    if (c < nb) {                              // nb % 4 == 0 (checked on the host)
      s = *reinterpret_cast<const f32x4 *>(srow + c);
      bs = *reinterpret_cast<const f32x4 *>(b_sq + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int id = col0 + c + e;
      const float d = fmaxf((qs + bs[e]) - 2.f * s[e], 0.f);
      const bool pass = (c + e < nb) && (id < n_valid) && key_less(d, id, tau_d, tau_i);
      const unsigned long long m = __ballot(pass);
      if (m == 0ull) continue;
      const int cnt = __popcll(m);
      if (n_buf + cnt > kBuf) compact();       // frees kListCap >= 64 slots
      if (pass) {
        const int pos = n_buf + __popcll(m & ((1ull << lane) - 1ull));
        kd[pos] = d;
        ki[pos] = id;
      }
      n_buf += cnt;
    }
  }
  if (n_buf > kListCap || first) compact();
#pragma unroll
  for (int t = 0; t < kListCap / 64; ++t) {
    const int p = lane + 64 * t;
    bd[p] = kd[p];
    bi[p] = ki[p];
  }
}

// One wave per query row: the row's current list (kListCap sorted keys) + its n = min(cnt[row], cap) candidates -> the
// kListCap best, sorted by (d, id).  cnt[row] > cap raises *overflow (candidates were dropped: the caller redoes the search).
__global__ void __launch_bounds__(256)
k_knn_merge_list(const uint2 *__restrict__ cand, int32_t *__restrict__ cnt, int cap, int nq, int k,
                 float *__restrict__ best_d, int *__restrict__ best_i, int32_t *__restrict__ overflow) {
  __shared__ float s_d[kRowsPerBlock][kBuf];
  __shared__ int s_i[kRowsPerBlock][kBuf];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowsPerBlock + w;
  if (row >= nq) return;                       // whole wave leaves; no block barrier below
  float *kd = s_d[w];
  int *ki = s_i[w];
  const float inf = __builtin_inff();
  float *bd = best_d + (int64_t)row * kListCap;
  int *bi = best_i + (int64_t)row * kListCap;
#pragma unroll
  for (int t = 0; t < kListCap / 64; ++t) {
    const int p = lane + 64 * t;
    kd[p] = bd[p];
    ki[p] = bi[p];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float tau_d = kd[k - 1];
  int tau_i = ki[k - 1];
  int n_buf = kListCap;                        // wave-uniform
  auto compact = [&]() {
#pragma unroll
    for (int t = 0; t < kBuf / 64; ++t) {
      const int p = lane + 64 * t;
      if (p >= n_buf) { kd[p] = inf; ki[p] = 0x7fffffff; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    sort_buffer(kd, ki, lane);
    tau_d = kd[k - 1];
    tau_i = ki[k - 1];
    n_buf = kListCap;
  };
  const int n_all = cnt[row];
  if (n_all > cap && lane == 0) *overflow = 1;
  const int n = min(n_all, cap);
  const uint2 *src = cand + (int64_t)row * cap;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int c = c0 + lane;
    uint2 v = make_uint2(0u, 0u);
    if (c < n) v = src[c];
    const float d = __uint_as_float(v.x);
    const int id = (int)v.y;
    const bool pass = c < n && key_less(d, id, tau_d, tau_i);
    const unsigned long long m = __ballot(pass);
    if (m == 0ull) continue;
    const int pc = __popcll(m);
    if (n_buf + pc > kBuf) compact();          // frees kListCap >= 64 slots
    if (pass) {
      const int pos = n_buf + __popcll(m & ((1ull << lane) - 1ull));
      kd[pos] = d;
      ki[pos] = id;
    }
    n_buf += pc;
  }
  if (n_buf > kListCap) compact();
#pragma unroll
  for (int t = 0; t < kListCap / 64; ++t) {
    const int p = lane + 64 * t;
    bd[p] = kd[p];
    bi[p] = ki[p];
  }
  if (lane == 0) cnt[row] = 0;                 // clean for the next filter launch
}

}  // namespace
}  // namespace cdml

using namespace cdml;

extern "C" int cdml_knn_list_capacity(void) { return kListCap; }

// The kNN export without a score matrix (round 6; faiss_knn.py:82-131): the query x catalogue-block product on the plane
// kernels (six bf16 plane products per fp32 product) whose epilogue appends every element with d <= tau[query] to the
// query's candidate list.  Q, Bk: fp32 rows as three bf16 planes [rows][hi D | mid D | lo D] (plane strides plane_q /
// plane_b); n_cols (multiple of 256) catalogue rows starting at catalogue row col0, rows >= n_valid are padding.
static int knn_filter_impl(float out_scale, const uint16_t *Q, int64_t ldq, int64_t plane_q, const uint16_t *Bk, int64_t ldb,
                           int64_t plane_b, int nq, int n_cols, int D, const float *q_sq, const float *b_sq,
                           const float *tau, int col0, int n_valid, int32_t *cnt, void *cand, int cap,
                           cdml_stream_t stream) {
  const bool h2 = out_scale > 0.f;                           // two fp16 planes per row, three products (cdml_knn_filter_h2)
  const int np1 = h2 ? 1 : 2;
  CDML_REQUIRE(Q && Bk && q_sq && b_sq && tau && cnt && cand, CDML_E_BADARG, "knn_filter_x3: null pointer");
  CDML_REQUIRE(nq > 0 && n_cols > 0 && D > 0 && cap > 0 && col0 >= 0 && n_valid > 0, CDML_E_BADARG, "knn_filter_x3: bad size");
  CDML_REQUIRE(n_cols % 256 == 0 && D % 64 == 0, CDML_E_UNSUPPORTED,
               "knn_filter_x3: the catalogue block must be a multiple of 256 rows and D of 64, got %d, %d", n_cols, D);
  CDML_REQUIRE(aligned16(Q) && aligned16(Bk) && aligned16(b_sq) && !(ldq & 7) && !(ldb & 7) && !(plane_q & 7) && !(plane_b & 7) &&
                   plane_q >= D && plane_b >= D && ldq >= np1 * plane_q + D && ldb >= np1 * plane_b + D &&
                   (reinterpret_cast<uintptr_t>(cand) & 7) == 0,
               CDML_E_ALIGN, "knn_filter_x3: 16-B aligned operands, strides multiples of 8, ld >= 2 plane + D");
  CDML_REQUIRE(((int64_t)nq + 256) * ldq * 2 < ((int64_t)1 << 31) && (int64_t)n_cols * ldb * 2 < ((int64_t)1 << 31), CDML_E_UNSUPPORTED,
               "knn_filter_x3: an operand exceeds the 2 GiB buffer-descriptor range (split the launch)");
  CDML_REQUIRE((int64_t)((nq + 255) / 256) * (n_cols / 256) < ((int64_t)1 << 31), CDML_E_UNSUPPORTED, "knn_filter_x3: too many tiles");
  BArgs g{};
  g.A = reinterpret_cast<const bf16 *>(Q); g.lda = ldq;
  g.B = reinterpret_cast<const bf16 *>(Bk); g.ldb = ldb;
  g.M = nq; g.N = n_cols;
  const int prod = h2 ? 3 : 6;
  g.x3_tpp = D / 64; g.x3_plane_a = plane_q; g.x3_plane_b = plane_b; g.x3_products = prod;
  g.K = prod * g.x3_tpp * 64; g.k_per_split = g.K;
  g.out_scale = h2 ? out_scale : 1.0f; g.c_scale = 1.0f;
  g.tiles_m = (nq + 255) / 256; g.tiles_n = n_cols / 256;
  g.knn_qsq = q_sq; g.knn_bsq = b_sq; g.knn_tau = tau; g.knn_cnt = cnt; g.knn_cand = static_cast<uint2 *>(cand);
  g.knn_cap = cap; g.knn_col0 = col0; g.knn_n_valid = n_valid;
  return h2 ? launch_gemm_f16x2_knn(g, (hipStream_t)stream) : launch_gemm_x3_knn(g, (hipStream_t)stream);
}

extern "C" int cdml_knn_filter_x3(const uint16_t *Q, int64_t ldq, int64_t plane_q, const uint16_t *Bk, int64_t ldb,
                                  int64_t plane_b, int nq, int n_cols, int D, const float *q_sq, const float *b_sq,
                                  const float *tau, int col0, int n_valid, int32_t *cnt, void *cand, int cap,
                                  cdml_stream_t stream) {
  return knn_filter_impl(0.f, Q, ldq, plane_q, Bk, ldb, plane_b, nq, n_cols, D, q_sq, b_sq, tau, col0, n_valid, cnt, cand, cap, stream);
}

// The same filter on TWO fp16 planes per row (cdml_split_f32_f16x2 of the queries at scale sq, of the catalogue at sb):
// out_scale = 1 / (sq sb) multiplies the accumulated inner product; three plane products on the fp16 MFMA.
extern "C" int cdml_knn_filter_h2(const uint16_t *Q, int64_t ldq, int64_t plane_q, const uint16_t *Bk, int64_t ldb,
                                  int64_t plane_b, int nq, int n_cols, int D, float out_scale, const float *q_sq,
                                  const float *b_sq, const float *tau, int col0, int n_valid, int32_t *cnt, void *cand, int cap,
                                  cdml_stream_t stream) {
  CDML_REQUIRE(out_scale > 0.f, CDML_E_BADARG, "knn_filter_h2: a positive out_scale");
  return knn_filter_impl(out_scale, Q, ldq, plane_q, Bk, ldb, plane_b, nq, n_cols, D, q_sq, b_sq, tau, col0, n_valid, cnt, cand, cap, stream);
}

// Merge every query's candidate list (cdml_knn_filter_x3) into its running top-k list (best_d / best_i: [nq][list capacity],
// sorted; as cdml_knn_merge keeps them); cnt is reset to 0; *overflow = 1 if a query had more than `cap` candidates.
extern "C" int cdml_knn_merge_list(const void *cand, int32_t *cnt, int cap, int nq, int k, float *best_d, int32_t *best_i,
                                   int32_t *overflow, cdml_stream_t stream) {
  CDML_REQUIRE(cand && cnt && best_d && best_i && overflow && nq > 0 && cap > 0, CDML_E_BADARG, "knn_merge_list: bad argument");
  CDML_REQUIRE(k >= 1 && k <= kListCap, CDML_E_UNSUPPORTED, "knn_merge_list: k must be in [1, %d], got %d", kListCap, k);
  hipLaunchKernelGGL(k_knn_merge_list, dim3((nq + kRowsPerBlock - 1) / kRowsPerBlock), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const uint2 *>(cand), cnt, cap, nq, k, best_d, best_i, overflow);
  return check_launch("knn_merge_list");
}

extern "C" int cdml_row_sqnorm(const float *x, int64_t ldx, int n_rows, int D, float *out,
                               cdml_stream_t stream) {
  CDML_REQUIRE(x && out && n_rows > 0 && D > 0 && ldx >= D, CDML_E_BADARG, "row_sqnorm: bad argument");
  hipLaunchKernelGGL(k_row_sqnorm, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, n_rows,
                     D, out);
  return check_launch("row_sqnorm");
}

extern "C" int cdml_knn_merge(const float *scores, int64_t lds, int nq, int nb, int col0, int n_valid,
                              const float *q_sq, const float *b_sq, int k, float *best_d, int32_t *best_i,
                              int first, cdml_stream_t stream) {
  CDML_REQUIRE(scores && q_sq && b_sq && best_d && best_i, CDML_E_BADARG, "knn_merge: null pointer");
  CDML_REQUIRE(nq > 0 && nb > 0 && col0 >= 0 && n_valid > 0, CDML_E_BADARG, "knn_merge: bad size");
  CDML_REQUIRE(k >= 1 && k <= kListCap, CDML_E_UNSUPPORTED, "knn_merge: k must be in [1, %d], got %d",
               kListCap, k);
  CDML_REQUIRE((nb & 3) == 0 && (lds & 3) == 0 && lds >= nb && aligned16(scores) && aligned16(b_sq),
               CDML_E_ALIGN, "knn_merge: nb and the score stride must be multiples of 4, buffers 16-B aligned");
  hipLaunchKernelGGL(k_knn_merge, dim3((nq + kRowsPerBlock - 1) / kRowsPerBlock), dim3(256), 0,
                     (hipStream_t)stream, scores, lds, nq, nb, col0, n_valid, q_sq, b_sq, k, best_d, best_i,
                     first);
  return check_launch("knn_merge");
}
