// Trainable catalogue rows (BASELINE north_star: "the catalogue feature table and its Adam
// states shard row-wise"; the reference keeps the features frozen -- train.py:265 feeds them
// through a placeholder -- so this is build-defined, off by default, spec: oracle/table.py).
//
// Input of the update: G[r] = dLoss/d x_hat[r] for the R gathered rows (x_hat = the
// l2-normalised row, models.py:58), i.e. the data gradient of the first layer
// (cdml_fc_bwd_data(dz1, W1, NULL)).  Per touched catalogue row, once per step:
//     G    = sum over the batch rows r that gathered this catalogue row (ascending r)
//     dx   = inv * (G - x_hat * <x_hat, G>),  inv = rsqrt(max(|x|^2, 1e-12))   (l2norm backward)
//     lazy Adam on the row: m, v, x touched only here (tf.contrib.opt.LazyAdamOptimizer's
//     rule; the arithmetic is k_adam's).
// Duplicates are the rule (a video sits in several triplets), and the result must not depend
// on scheduling: k_rows_link threads the batch rows of a catalogue row into a list
// (atomicExch on head[row]); which batch row ends up as the head is arbitrary, but the SET is
// not, and the head's wave adds the members in ascending order.  The head resets head[row] to
// -1, so the scratch is clean for the next step without a memset.
// HBM-bound: R * F * 4 B of gradients in, 3 rows read + 3 rows written per touched row.
#include "common.h"

namespace cdml {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 8;   // rows up to 2048 floats

__global__ void __launch_bounds__(kThreads)
k_rows_link(const int32_t *__restrict__ idx, int n_idx, int64_t row0, int64_t n_rows,
            int32_t *__restrict__ head, int32_t *__restrict__ next) {
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r >= n_idx) return;
  const int64_t lr = (int64_t)idx[r] - row0;
  next[r] = (lr >= 0 && lr < n_rows) ? atomicExch(&head[lr], r) : -1;
}

template <int NCH>
__global__ void __launch_bounds__(kThreads)
k_rows_adam(float *__restrict__ table, int64_t row0, int64_t n_rows, int64_t row_stride, int F,
            const int32_t *__restrict__ idx, int n_idx, const float *__restrict__ G, int64_t ldg,
            float *__restrict__ m_tab, float *__restrict__ v_tab, int32_t *__restrict__ head,
            const int32_t *__restrict__ next, float grad_scale, float lr_imm,
            const float *__restrict__ lr_dev, float b1, float b2, float eps, int64_t t_imm,
            const uint64_t *__restrict__ t_dev) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_idx) return;
  const int64_t lr = (int64_t)idx[r] - row0;
  if (lr < 0 || lr >= n_rows) return;
  if (head[lr] != r) return;                        // one wave per touched row: the list head
  const int nq = (F + 3) >> 2;
  float4 g[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) g[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  // members in ascending order of r: repeatedly take the smallest one above the last taken
  int last = -1;
  while (true) {
    int best = 0x7fffffff;
    for (int cur = r; cur >= 0; cur = next[cur])
      if (cur > last && cur < best) best = cur;
    if (best == 0x7fffffff) break;
    const float4 *src = reinterpret_cast<const float4 *>(G + (int64_t)best * ldg);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + 64 * c;
      if (q < nq) {
        const float4 v = src[q];
        g[c].x += v.x; g[c].y += v.y; g[c].z += v.z; g[c].w += v.w;
      }
    }
    last = best;
  }
  if (grad_scale != 1.0f) {   // data-parallel runs: 1/world, the mean over the GLOBAL batch
#pragma unroll
    for (int c = 0; c < NCH; ++c) { g[c].x *= grad_scale; g[c].y *= grad_scale; g[c].z *= grad_scale; g[c].w *= grad_scale; }
  }
  // l2norm backward from the raw row
  float4 *xrow = reinterpret_cast<float4 *>(table + lr * row_stride);
  float4 x[NCH];
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int q = lane + 64 * c;
    x[c] = (q < nq) ? xrow[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int j = 4 * q;                            // pad columns of the row never take part
    if (j + 1 >= F) x[c].y = 0.f;
    if (j + 2 >= F) x[c].z = 0.f;
    if (j + 3 >= F) x[c].w = 0.f;
    if (j >= F) x[c].x = 0.f;
    ss += x[c].x * x[c].x + x[c].y * x[c].y + x[c].z * x[c].z + x[c].w * x[c].w;
    dot += x[c].x * g[c].x + x[c].y * g[c].y + x[c].z * g[c].z + x[c].w * g[c].w;
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
  const float s = dot * inv * inv;                  // <x_hat, G> * inv, as a factor of x
  const double t = (double)t_imm + (t_dev ? (double)(*t_dev) : 0.0);
  const double lrate = lr_dev ? (double)(*lr_dev) : (double)lr_imm;
  const float lr_t = (float)(lrate * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  float4 *mrow = reinterpret_cast<float4 *>(m_tab + lr * row_stride);
  float4 *vrow = reinterpret_cast<float4 *>(v_tab + lr * row_stride);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int q = lane + 64 * c;
    if (q >= nq) continue;
    float4 m4 = mrow[q], v4 = vrow[q], w4 = xrow[q];
    const int j = 4 * q;
#define CDML_ROW_ADAM1(comp, col)                                         \
  if ((col) < F) {                                                        \
    const float dx = inv * (g[c].comp - x[c].comp * s);                   \
    m4.comp += (dx - m4.comp) * omb1;                                     \
    v4.comp += (dx * dx - v4.comp) * omb2;                                \
    w4.comp -= (m4.comp * lr_t) / (sqrtf(v4.comp) + eps);                 \
  }
    CDML_ROW_ADAM1(x, j) CDML_ROW_ADAM1(y, j + 1) CDML_ROW_ADAM1(z, j + 2) CDML_ROW_ADAM1(w, j + 3)
#undef CDML_ROW_ADAM1
    mrow[q] = m4; vrow[q] = v4; xrow[q] = w4;
  }
  if (lane == 0) head[lr] = -1;
}

}  // namespace
}  // namespace cdml

using namespace cdml;

extern "C" int cdml_table_adam_rows(float *table, int64_t row0, int64_t n_rows, int64_t row_stride,
                                    int F, const int32_t *idx, int n_idx, const float *grad_xhat,
                                    int64_t ldg, float *m_table, float *v_table, int32_t *head,
                                    int32_t *next, float grad_scale, float lr, const float *lr_dev, float beta1,
                                    float beta2, float eps, int64_t t, const uint64_t *t_dev,
                                    cdml_stream_t stream) {
  CDML_REQUIRE(table && idx && grad_xhat && m_table && v_table && head && next, CDML_E_BADARG,
               "table_adam_rows: null pointer");
  CDML_REQUIRE(n_rows > 0 && n_idx > 0 && F > 0 && row0 >= 0, CDML_E_BADARG, "table_adam_rows: bad size");
  CDML_REQUIRE(t >= (t_dev ? 0 : 1), CDML_E_BADARG, "table_adam_rows: step t is 1-based");
  CDML_REQUIRE(row_stride >= F && (row_stride & 3) == 0 && (ldg & 3) == 0 && ldg >= ((F + 3) & ~3) &&
                   aligned16(table) && aligned16(grad_xhat) && aligned16(m_table) && aligned16(v_table),
               CDML_E_ALIGN, "table_adam_rows: 16-B aligned buffers, strides multiples of 4 and >= F");
  const int nch = ((F + 3) / 4 + 63) / 64;
  CDML_REQUIRE(nch <= kMaxChunks, CDML_E_UNSUPPORTED, "table_adam_rows: rows of at most %d floats",
               kMaxChunks * 256);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_rows_link, dim3((n_idx + kThreads - 1) / kThreads), dim3(kThreads), 0, s, idx, n_idx,
                     row0, n_rows, head, next);
  int rc = check_launch("table_adam_rows link");
  if (rc) return rc;
#define CDML_LAUNCH_RA(N)                                                                              \
  hipLaunchKernelGGL((k_rows_adam<N>), dim3((n_idx + 3) / 4), dim3(kThreads), 0, s, table, row0, n_rows, \
                     row_stride, F, idx, n_idx, grad_xhat, ldg, m_table, v_table, head, next, grad_scale, lr, lr_dev, \
                     beta1, beta2, eps, t, t_dev)
  if (nch <= 2) CDML_LAUNCH_RA(2);
  else if (nch <= 6) CDML_LAUNCH_RA(6);
  else CDML_LAUNCH_RA(8);
#undef CDML_LAUNCH_RA
  return check_launch("table_adam_rows");
}
