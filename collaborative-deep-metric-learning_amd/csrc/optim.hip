// Fused optimizer updates (train.py:108-125,146) + ABI base (version, errors).
//
// Adam: tf.train.AdamOptimizer (build_graph's default, train.py:82), the arithmetic
// of TF's ApplyAdam functor: m += (g-m)(1-b1); v += (g*g-v)(1-b2); w -= alpha*m/(sqrt(v)+eps).
// LARS: tf.contrib.opt.LARSOptimizer with contrib defaults (train.py:354).
// Roofline: HBM -- Adam streams 7 floats per parameter (r: w,g,m,v  w: w,m,v),
// LARS 5 (+2 for the norm pass).  16-B lane accesses, grid-stride.
#include "common.h"
#include <string.h>

namespace cdml {

char *err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

namespace {

constexpr int kThreads = 256;
constexpr int kLarsBlocks = 1024;

// One Adam update (TF's ApplyAdam functor), with the roundings spelled out -- m += (g-m)(1-b1);
// v += (g*g-v)(1-b2); w -= (m*lr_t)/(sqrt(v)+eps) -- so that every kernel form (float4 stream, scalar tail,
// matrix tiles, the bias vector riding along) gives the same bits whatever the compiler contracts.
__device__ __forceinline__ void adam1(float &w, const float g, float &m, float &v, const float omb1,
                                      const float omb2, const float lr_t, const float eps) {
  m = __fmaf_rn(g - m, omb1, m);
  v = __fmaf_rn(__fmaf_rn(g, g, -v), omb2, v);
  w = w - __fdiv_rn(m * lr_t, __fsqrt_rn(v) + eps);
}

// One LARS update (tf.contrib.opt.LARSOptimizer: compute_lr + apply_momentum) given the variable's scaled learning
// rate slr = lr * trust: g' = g + wd w; acc = momentum acc + slr g'; w -= acc.  Roundings spelled out for the same
// reason as adam1: the flat kernel and the matrix kernel (which also writes the GEMMs' operand copies) agree bit for bit.
__device__ __forceinline__ void lars1(float &w, const float g, float &a, const float slr, const float momentum,
                                      const float wd) {
  a = __fmaf_rn(momentum, a, slr * __fmaf_rn(wd, w, g));
  w = w - a;
}

// tf.train.MomentumOptimizer (TF's ApplyMomentum): accum = accum*momentum + g; w -= nesterov ? g*lr + accum*momentum*lr : accum*lr
__device__ __forceinline__ void momentum1(float &w, const float g, float &a, const float lr, const float momentum,
                                          const int nesterov) {
  a = __fmaf_rn(a, momentum, g);
  w = w - (nesterov ? __fmaf_rn(g, lr, a * momentum * lr) : a * lr);
}

__global__ void __launch_bounds__(kThreads)
k_adam(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
       float *__restrict__ v, int64_t n, float lr_imm, const float *__restrict__ lr_dev, float b1,
       float b2, float eps, int64_t t_imm, uint64_t *__restrict__ t_dev, int advance,
       uint32_t *__restrict__ tickets) {
  __shared__ float s_lr_t;
  if (threadIdx.x == 0) {
    const double t = (double)t_imm + (t_dev ? (double)(*t_dev) : 0.0);
    const double lr = lr_dev ? (double)(*lr_dev) : (double)lr_imm;
    s_lr_t = (float)(lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const float lr_t = s_lr_t;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 w4 = reinterpret_cast<float4 *>(w)[i];
    const float4 g4 = reinterpret_cast<const float4 *>(g)[i];
    float4 m4 = reinterpret_cast<float4 *>(m)[i];
    float4 v4 = reinterpret_cast<float4 *>(v)[i];
#define CDML_ADAM1(c) adam1(w4.c, g4.c, m4.c, v4.c, omb1, omb2, lr_t, eps);
    CDML_ADAM1(x) CDML_ADAM1(y) CDML_ADAM1(z) CDML_ADAM1(w)
#undef CDML_ADAM1
    reinterpret_cast<float4 *>(w)[i] = w4;
    reinterpret_cast<float4 *>(m)[i] = m4;
    reinterpret_cast<float4 *>(v)[i] = v4;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float wi = w[i], mi = m[i], vi = v[i];
    adam1(wi, g[i], mi, vi, omb1, omb2, lr_t, eps);
    m[i] = mi;
    v[i] = vi;
    w[i] = wi;
  }
  // global_step += 1 (train.py:146 apply_gradients(global_step=...)) by the last block to
  // finish: every block has read *t_dev by then, and no other kernel runs beside this one
  if (advance && grid_last_block(tickets) && threadIdx.x == 0) *t_dev += 1;
}

// Adam on a weight MATRIX W[K][N] (row-major, ld = N) that also writes the bf16 operand copies the
// config-4 GEMMs read: W^T as bf16 [N][K] (wt, nullable) and W as bf16 [K][N] (wc, nullable) -- what
// k_transpose_bf16 / k_cast_f32_bf16 would produce from the updated weights, without reading them
// again.  One 64 x 64 tile per block: 16-B accesses on every fp32 stream, the transpose through LDS,
// whole 128-B lines on the bf16 rows.
// Same arithmetic per element as k_adam (bit-equal).
// The layer's BIAS vector (bw/bg/bm/bv, bn elements, nullable) rides along in the first blocks -- a
// 5 000-element Adam launch of its own costs 5-7 us for 100 KB -- and on request the last block to
// finish advances the step counter (as k_adam does): the optimizer of the config-4 step is two launches.
// PLANES = 3 (precision "f32x3"): the copies are the three bf16 planes hi | mid | lo of the new weights (their
// sum is the fp32 value; plane p of W^T starts plane_t elements after plane p - 1 in every row, of W plane_c).
constexpr int kAT = 64;
// The bf16 operand copies of one updated 64 x 64 tile (res[p][q][u] = new W[k0 + 32 p + 2 tr + q][n0 + c4 + u]): wc = W as
// bf16 [K][N], wt = W^T as bf16 [N][K] through the LDS tile sT; PLANES = 3: the three planes hi | mid | lo of the value.
template <int PLANES>
__device__ __forceinline__ void tile_copies(float (&res)[2][2][4], uint32_t (&sT)[64][64 / 2 + 1], __bf16 *__restrict__ wt,
                                            int64_t ldt, __bf16 *__restrict__ wc, int64_t ldc, int64_t plane_t,
                                            int64_t plane_c, int k0, int n0, float h2_scale = 0.f) {
  using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
  const int tr = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  if constexpr (PLANES == 2) {
    // precision "f16x2": the copies are the two fp16 planes hi | lo of the new weights times h2_scale (saturating); the
    // same tile walk, the 16-bit words being fp16
    using half4v = __attribute__((ext_vector_type(4))) _Float16;
    using half2v = __attribute__((ext_vector_type(2))) _Float16;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) res[p][q][u] = __builtin_amdgcn_fmed3f(res[p][q][u] * h2_scale, -65504.f, 65504.f);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      if (pl && wt) __syncthreads();
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int r = p * 32 + 2 * tr;
        half4v o[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            o[q][u] = (_Float16)res[p][q][u];
            res[p][q][u] -= (float)o[q][u];
          }
          if (wc) *reinterpret_cast<half4v *>(wc + (int64_t)(k0 + r + q) * ldc + pl * plane_c + n0 + c4) = o[q];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const half2v pr = {o[0][u], o[1][u]};
          sT[c4 + u][r >> 1] = __builtin_bit_cast(uint32_t, pr);
        }
      }
      if (wt) {
        __syncthreads();
        const int sr = threadIdx.x >> 3, seg = (threadIdx.x & 7) * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int row = q * 32 + sr;
          uint4 o;
          o.x = sT[row][seg]; o.y = sT[row][seg + 1]; o.z = sT[row][seg + 2]; o.w = sT[row][seg + 3];
          *reinterpret_cast<uint4 *>(wt + (int64_t)(n0 + row) * ldt + pl * plane_t + k0 + seg * 2) = o;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int pl = 0; pl < PLANES; ++pl) {
    if (pl && wt) __syncthreads();                     // the previous plane's tile has been read out
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = p * 32 + 2 * tr;
      bf16x4 o[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          o[q][u] = (__bf16)res[p][q][u];
          if (PLANES > 1) res[p][q][u] -= (float)o[q][u];
        }
        if (wc) *reinterpret_cast<bf16x4 *>(wc + (int64_t)(k0 + r + q) * ldc + pl * plane_c + n0 + c4) = o[q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
        const bf16x2 pr = {o[0][u], o[1][u]};            // rows r (low half), r + 1 (high half)
        sT[c4 + u][r >> 1] = __builtin_bit_cast(uint32_t, pr);
      }
    }
    if (wt) {                                          // (uniform)
      __syncthreads();
      // transposed tile: row n of W^T holds 64 consecutive k = 128 B; 8 threads x 16 B per row
      const int sr = threadIdx.x >> 3, seg = (threadIdx.x & 7) * 4;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = q * 32 + sr;
        uint4 o;
        o.x = sT[row][seg]; o.y = sT[row][seg + 1]; o.z = sT[row][seg + 2]; o.w = sT[row][seg + 3];
        *reinterpret_cast<uint4 *>(wt + (int64_t)(n0 + row) * ldt + pl * plane_t + k0 + seg * 2) = o;
      }
    }
  }
}

template <int PLANES>
__global__ void __launch_bounds__(kThreads)
k_adam_matrix_bf16(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                   float *__restrict__ v, int K, int N, float lr_imm, const float *__restrict__ lr_dev,
                   float b1, float b2, float eps, int64_t t_imm, uint64_t *__restrict__ t_dev,
                   __bf16 *__restrict__ wt, int64_t ldt, __bf16 *__restrict__ wc, int64_t ldc,
                   float *__restrict__ bw, const float *__restrict__ bg, float *__restrict__ bm,
                   float *__restrict__ bv, int bn, int advance, uint32_t *__restrict__ tickets,
                   int64_t plane_t, int64_t plane_c, float h2_scale) {
  __shared__ float s_lr_t;
  __shared__ uint32_t sT[kAT][kAT / 2 + 1];           // [column n][row pair]: two bf16 of one column per word
  if (threadIdx.x == 0) {
    const double t = (double)t_imm + (t_dev ? (double)(*t_dev) : 0.0);
    const double lr = lr_dev ? (double)(*lr_dev) : (double)lr_imm;
    s_lr_t = (float)(lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const float lr_t = s_lr_t;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const int tiles_n = N / kAT;
  const int k0 = (blockIdx.x / tiles_n) * kAT, n0 = (blockIdx.x % tiles_n) * kAT;
  const int tr = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  // a thread takes two adjacent rows of four columns per pass: the transpose goes through LDS as
  // 32-bit words (the two rows' values of one column), 33-word rows -> at most 2-way bank conflicts
  float res[2][2][4];                                  // the new weights (PLANES > 1: what the planes so far leave)
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = p * 32 + 2 * tr;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t i = (int64_t)(k0 + r + q) * N + n0 + c4;
      float4 w4 = *reinterpret_cast<float4 *>(w + i);
      const float4 g4 = *reinterpret_cast<const float4 *>(g + i);
      float4 m4 = *reinterpret_cast<float4 *>(m + i);
      float4 v4 = *reinterpret_cast<float4 *>(v + i);
#define CDML_ADAM1(c) adam1(w4.c, g4.c, m4.c, v4.c, omb1, omb2, lr_t, eps);
      CDML_ADAM1(x) CDML_ADAM1(y) CDML_ADAM1(z) CDML_ADAM1(w)
#undef CDML_ADAM1
      *reinterpret_cast<float4 *>(w + i) = w4;
      *reinterpret_cast<float4 *>(m + i) = m4;
      *reinterpret_cast<float4 *>(v + i) = v4;
      res[p][q][0] = w4.x; res[p][q][1] = w4.y; res[p][q][2] = w4.z; res[p][q][3] = w4.w;
    }
  }
  tile_copies<PLANES>(res, sT, wt, ldt, wc, ldc, plane_t, plane_c, k0, n0, h2_scale);
  if (bw) {                                          // the bias vector: element i of the first ceil(bn / 256) blocks
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < bn; i += gridDim.x * kThreads) {
      float wi = bw[i], mi = bm[i], vi = bv[i];
      adam1(wi, bg[i], mi, vi, omb1, omb2, lr_t, eps);
      bm[i] = mi;
      bv[i] = vi;
      bw[i] = wi;
    }
  }
  if (advance && grid_last_block(tickets) && threadIdx.x == 0) *t_dev += 1;
}

// scratch layout: [0]=|w|^2, [1]=|g|^2, then kLarsBlocks x 2 block partials
__global__ void __launch_bounds__(kThreads)
k_lars_norm_partial(const float *__restrict__ w, const float *__restrict__ g, int64_t n,
                    float *__restrict__ scratch) {
  __shared__ double s[2][kThreads / kWave];
  float sw = 0.f, sg = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float a = w[i], b = g[i];
    sw += a * a;
    sg += b * b;
  }
  sw = wave_sum(sw);
  sg = wave_sum(sg);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  if (lane == 0) { s[0][wave] = sw; s[1][wave] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int k = 0; k < kThreads / kWave; ++k) { a += s[0][k]; b += s[1][k]; }
    scratch[2 + 2 * blockIdx.x] = (float)a;
    scratch[3 + 2 * blockIdx.x] = (float)b;
  }
}

__global__ void k_lars_norm_final(float *__restrict__ scratch, int blocks) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0, b = 0;
    for (int k = 0; k < blocks; ++k) { a += scratch[2 + 2 * k]; b += scratch[3 + 2 * k]; }
    scratch[0] = (float)a;
    scratch[1] = (float)b;
  }
}

__global__ void __launch_bounds__(kThreads)
k_lars_apply(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ acc, int64_t n,
             float lr_imm, const float *__restrict__ lr_dev, float momentum, float wd, float eeta,
             float eps, const float *__restrict__ scratch) {
  const float lr = lr_dev ? *lr_dev : lr_imm;
  const float wn = sqrtf(scratch[0]), gn = sqrtf(scratch[1]);
  const float trust = (wn > 0.f && gn > 0.f) ? eeta * wn / (gn + wd * wn + eps) : 1.0f;
  const float slr = lr * trust;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float wi = w[i], a = acc[i];
    lars1(wi, g[i], a, slr, momentum, wd);
    acc[i] = a;
    w[i] = wi;
  }
}

// ---- multi-tensor LARS: every variable of the flat parameter buffer in TWO launches ----------
// The reference's own optimizer (train.py:354): per variable trust ratio, so per variable two norms.
// cdml_lars_step is one variable at a time in three launches with 4-B lane accesses (13 launches a
// step with the step counter: 196 us at the production shape against Adam's 48).  Here the
// variables are SEGMENTS of one contiguous buffer (offsets and sizes multiples of 4 floats):
//   k_lars_multi_norms  -- each block sums |w|^2 and |g|^2 over a slice of ONE segment (16-B loads,
//                          blocks dealt to segments in proportion to their size), one partial pair per block;
//   k_lars_multi_apply  -- every block first reduces the partials of all segments itself, in the
//                          same fixed order (<= kLarsBlocks pairs out of L2: no third launch, no
//                          hand-over between blocks), then streams its share of the flat buffer
//                          with 16-B accesses; the last block advances the step counter on request.
// Bytes: 2 x 4 x n for the norms + 5 x 4 x n for the update.
constexpr int kMaxSeg = 8;
struct LarsSegs {
  int n_seg;
  int64_t off[kMaxSeg + 1];       // float offsets into the flat buffer; off[n_seg] = total
  int blk[kMaxSeg + 1];           // first norm block of each segment; blk[n_seg] = total norm blocks
};

__global__ void __launch_bounds__(kThreads)
k_lars_multi_norms(const float *__restrict__ w, const float *__restrict__ g, LarsSegs S,
                   float *__restrict__ scratch) {
  __shared__ double s[2][kThreads / kWave];
  int seg = 0;
#pragma unroll
  for (int k = 1; k < kMaxSeg; ++k) seg += (k < S.n_seg && (int)blockIdx.x >= S.blk[k]) ? 1 : 0;
  const int nb = S.blk[seg + 1] - S.blk[seg], b = blockIdx.x - S.blk[seg];
  const int64_t lo4 = S.off[seg] >> 2, hi4 = S.off[seg + 1] >> 2;
  float sw = 0.f, sg = 0.f;
  for (int64_t i = lo4 + (int64_t)b * kThreads + threadIdx.x; i < hi4; i += (int64_t)nb * kThreads) {
    const float4 a = reinterpret_cast<const float4 *>(w)[i];
    const float4 c = reinterpret_cast<const float4 *>(g)[i];
    sw += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    sg += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
  }
  sw = wave_sum(sw);
  sg = wave_sum(sg);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  if (lane == 0) { s[0][wave] = sw; s[1][wave] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, c = 0;
    for (int k = 0; k < kThreads / kWave; ++k) { a += s[0][k]; c += s[1][k]; }
    scratch[2 * blockIdx.x] = (float)a;
    scratch[2 * blockIdx.x + 1] = (float)c;
  }
}

__global__ void __launch_bounds__(kThreads)
k_lars_multi_apply(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ acc, LarsSegs S,
                   float lr_imm, const float *__restrict__ lr_dev, float momentum, float wd, float eeta,
                   float eps, const float *__restrict__ scratch, float *__restrict__ norms_out,
                   uint64_t *__restrict__ step_dev, uint32_t *__restrict__ tickets) {
  __shared__ double s_part[2][kThreads / kWave];
  __shared__ float s_slr[kMaxSeg];
  const float lr = lr_dev ? *lr_dev : lr_imm;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  for (int seg = 0; seg < S.n_seg; ++seg) {           // every block, the same order: the same trust ratios
    double a = 0, c = 0;
    for (int k = S.blk[seg] + threadIdx.x; k < S.blk[seg + 1]; k += kThreads) {
      a += (double)scratch[2 * k];
      c += (double)scratch[2 * k + 1];
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, kWave);
      c += __shfl_xor(c, o, kWave);
    }
    if (lane == 0) { s_part[0][wave] = a; s_part[1][wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double wn2 = 0, gn2 = 0;
      for (int k = 0; k < kThreads / kWave; ++k) { wn2 += s_part[0][k]; gn2 += s_part[1][k]; }
      const float wn = sqrtf((float)wn2), gn = sqrtf((float)gn2);
      const float trust = (wn > 0.f && gn > 0.f) ? eeta * wn / (gn + wd * wn + eps) : 1.0f;
      s_slr[seg] = lr * trust;
      if (norms_out && blockIdx.x == 0) { norms_out[2 * seg] = wn; norms_out[2 * seg + 1] = gn; }
    }
    __syncthreads();
  }
  const int64_t n4 = S.off[S.n_seg] >> 2;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    int seg = 0;
#pragma unroll
    for (int k = 1; k < kMaxSeg; ++k) seg += (k < S.n_seg && 4 * i >= S.off[k]) ? 1 : 0;
    const float slr = s_slr[seg];
    float4 w4 = reinterpret_cast<float4 *>(w)[i];
    const float4 g4 = reinterpret_cast<const float4 *>(g)[i];
    float4 a4 = reinterpret_cast<float4 *>(acc)[i];
#define CDML_LARS1(c) lars1(w4.c, g4.c, a4.c, slr, momentum, wd);
    CDML_LARS1(x) CDML_LARS1(y) CDML_LARS1(z) CDML_LARS1(w)
#undef CDML_LARS1
    reinterpret_cast<float4 *>(acc)[i] = a4;
    reinterpret_cast<float4 *>(w)[i] = w4;
  }
  if (step_dev && grid_last_block(tickets) && threadIdx.x == 0) *step_dev += 1;
}

// ---- LARS / momentum on a weight MATRIX, writing the GEMMs' operand copies with the update ----------------------
// What k_adam_matrix_bf16 is for Adam: the bf16 (config 4) or three-plane (precision "f32x3") copies of the new
// weights come out of the update's own registers, so the reference's actual recipe (LARS, train.py:354) needs no
// separate split / transpose / cast launches after the optimizer.  RULE 1 = LARS: the trust ratios of the matrix'
// segment and of the bias' segment are reduced from k_lars_multi_norms' partials by every block itself, in the fixed
// order k_lars_multi_apply uses (same bits); RULE 2 = momentum (Nesterov on request).  Same element arithmetic as the
// flat kernels (lars1 / momentum1).
struct MatRule {
  float lr_imm; const float *lr_dev;
  float momentum, wd, eeta, eps;
  int nesterov;
  const float *scratch;          // LARS: k_lars_multi_norms' partial pairs
  int blk_w0, blk_w1, blk_b0, blk_b1;   // partial-pair ranges of the matrix' and of the bias' segment
  float *norms_w, *norms_b;      // optional: (|w|, |g|) of the two variables (block 0 writes them)
  float h2_scale;                // PLANES = 2 (precision "f16x2"): the copies are the fp16 planes of the new weights times this
};

template <int PLANES, int RULE>
__global__ void __launch_bounds__(kThreads)
k_rule_matrix(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ acc, int K, int N, MatRule R,
              __bf16 *__restrict__ wt, int64_t ldt, __bf16 *__restrict__ wc, int64_t ldc, float *__restrict__ bw,
              const float *__restrict__ bg, float *__restrict__ bacc, int bn, uint64_t *__restrict__ step_dev,
              uint32_t *__restrict__ tickets, int64_t plane_t, int64_t plane_c) {
  __shared__ uint32_t sT[kAT][kAT / 2 + 1];
  __shared__ double s_part[2][kThreads / kWave];
  __shared__ float s_slr[2];
  const float lr = R.lr_dev ? *R.lr_dev : R.lr_imm;
  if (RULE == 1) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    for (int v = 0; v < 2; ++v) {                        // the matrix, then its bias vector
      const int b0 = v ? R.blk_b0 : R.blk_w0, b1 = v ? R.blk_b1 : R.blk_w1;
      double a = 0, c = 0;
      for (int k = b0 + threadIdx.x; k < b1; k += kThreads) {
        a += (double)R.scratch[2 * k];
        c += (double)R.scratch[2 * k + 1];
      }
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, kWave);
        c += __shfl_xor(c, o, kWave);
      }
      if (lane == 0) { s_part[0][wave] = a; s_part[1][wave] = c; }
      __syncthreads();
      if (threadIdx.x == 0) {
        double wn2 = 0, gn2 = 0;
        for (int k = 0; k < kThreads / kWave; ++k) { wn2 += s_part[0][k]; gn2 += s_part[1][k]; }
        const float wn = sqrtf((float)wn2), gn = sqrtf((float)gn2);
        const float trust = (wn > 0.f && gn > 0.f) ? R.eeta * wn / (gn + R.wd * wn + R.eps) : 1.0f;
        s_slr[v] = (b1 > b0) ? lr * trust : lr;
        float *no = v ? R.norms_b : R.norms_w;
        if (no && blockIdx.x == 0 && b1 > b0) { no[0] = wn; no[1] = gn; }
      }
      __syncthreads();
    }
  }
  const float slr_w = RULE == 1 ? s_slr[0] : lr, slr_b = RULE == 1 ? s_slr[1] : lr;
  const int tiles_n = N / kAT;
  const int k0 = (blockIdx.x / tiles_n) * kAT, n0 = (blockIdx.x % tiles_n) * kAT;
  const int tr = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  float res[2][2][4];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = p * 32 + 2 * tr;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t i = (int64_t)(k0 + r + q) * N + n0 + c4;
      float4 w4 = *reinterpret_cast<float4 *>(w + i);
      const float4 g4 = *reinterpret_cast<const float4 *>(g + i);
      float4 a4 = *reinterpret_cast<float4 *>(acc + i);
#define CDML_RULE1(c)                                                  \
  if (RULE == 1) lars1(w4.c, g4.c, a4.c, slr_w, R.momentum, R.wd);     \
  else momentum1(w4.c, g4.c, a4.c, lr, R.momentum, R.nesterov);
      CDML_RULE1(x) CDML_RULE1(y) CDML_RULE1(z) CDML_RULE1(w)
#undef CDML_RULE1
      *reinterpret_cast<float4 *>(w + i) = w4;
      *reinterpret_cast<float4 *>(acc + i) = a4;
      res[p][q][0] = w4.x; res[p][q][1] = w4.y; res[p][q][2] = w4.z; res[p][q][3] = w4.w;
    }
  }
  tile_copies<PLANES>(res, sT, wt, ldt, wc, ldc, plane_t, plane_c, k0, n0, R.h2_scale);
  if (bw) {
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < bn; i += gridDim.x * kThreads) {
      float wi = bw[i], ai = bacc[i];
      if (RULE == 1) lars1(wi, bg[i], ai, slr_b, R.momentum, R.wd);
      else momentum1(wi, bg[i], ai, lr, R.momentum, R.nesterov);
      bacc[i] = ai;
      bw[i] = wi;
    }
  }
  if (step_dev && grid_last_block(tickets) && threadIdx.x == 0) *step_dev += 1;
}

// ---- build_graph's gradient options (train.py:133-145), off in the reference's own run -------
// One variable at a time, in place, before the optimizer:
//   g <- g + l2_scale * w                 the slim l2_regularizer term of a weight matrix
//                                         (models.py:28: 1e-8 * |W|^2 / 2, times regularization_penalty)
//   g <- g * clip / max(|g|_2, clip)      tf.clip_by_norm per variable (train.py:47-64)
// Pass 1 (k_prep_partial + k_lars_norm_final): |g + l2_scale*w|^2 and |w|^2, two-stage, fixed order.
__global__ void __launch_bounds__(kThreads)
k_prep_partial(const float *__restrict__ w, const float *__restrict__ g, int64_t n, float l2_scale,
               float *__restrict__ scratch) {
  __shared__ double s[2][kThreads / kWave];
  float sw = 0.f, sg = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float a = w[i], b = g[i] + l2_scale * a;
    sw += a * a;
    sg += b * b;
  }
  sw = wave_sum(sw);
  sg = wave_sum(sg);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  if (lane == 0) { s[0][wave] = sw; s[1][wave] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int k = 0; k < kThreads / kWave; ++k) { a += s[0][k]; b += s[1][k]; }
    scratch[2 + 2 * blockIdx.x] = (float)a;
    scratch[3 + 2 * blockIdx.x] = (float)b;
  }
}

__global__ void __launch_bounds__(kThreads)
k_prep_apply(float *__restrict__ g, const float *__restrict__ w, int64_t n, float l2_scale, float clip,
             const float *__restrict__ scratch, float *__restrict__ norms_out) {
  const float gn = sqrtf(scratch[1]);
  const float scale = (clip > 0.f) ? clip / fmaxf(gn, clip) : 1.0f;
  if (norms_out && blockIdx.x == 0 && threadIdx.x == 0) { norms_out[0] = gn; norms_out[1] = scratch[0]; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    g[i] = (g[i] + l2_scale * w[i]) * scale;
}

// tf.train.MomentumOptimizer(lr, momentum, use_nesterov) (train.py:115-116), TF's ApplyMomentum:
// accum = accum*momentum + g;  w -= nesterov ? g*lr + accum*momentum*lr : accum*lr
__global__ void __launch_bounds__(kThreads)
k_momentum(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ acc, int64_t n,
           float lr_imm, const float *__restrict__ lr_dev, float momentum, int nesterov) {
  const float lr = lr_dev ? *lr_dev : lr_imm;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float wi = w[i], a = acc[i];
    momentum1(wi, g[i], a, lr, momentum, nesterov);
    acc[i] = a;
    w[i] = wi;
  }
}

int grid_elems(int64_t n, int per_thread) {
  int64_t b = (n / per_thread + kThreads - 1) / kThreads;
  if (b > kNumCU * 8) b = kNumCU * 8;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace
}  // namespace cdml

using namespace cdml;

extern "C" int cdml_version(void) { return 3000; }
extern "C" const char *cdml_last_error(void) { return err_buf(); }

extern "C" int cdml_adam_step(float *w, const float *g, float *m, float *v, int64_t n, float lr,
                              const float *lr_dev, float beta1, float beta2, float eps, int64_t t,
                              uint64_t *t_dev, int advance_step, uint32_t *tickets,
                              cdml_stream_t stream) {
  CDML_REQUIRE(w && g && m && v && n > 0, CDML_E_BADARG, "adam_step: bad argument");
  CDML_REQUIRE(!advance_step || (t_dev && tickets), CDML_E_BADARG,
               "adam_step: advance_step needs the device step counter and the ticket words");
  CDML_REQUIRE(t >= (t_dev ? 0 : 1), CDML_E_BADARG, "adam_step: step t is 1-based");
  CDML_REQUIRE(aligned16(w) && aligned16(g) && aligned16(m) && aligned16(v), CDML_E_ALIGN,
               "adam_step: buffers must be 16-B aligned");
  hipLaunchKernelGGL(k_adam, dim3(grid_elems(n, 4)), dim3(kThreads), 0, (hipStream_t)stream, w, g, m,
                     v, n, lr, lr_dev, beta1, beta2, eps, t, t_dev, advance_step, tickets);
  return check_launch("adam_step");
}

static int adam_matrix_impl(float *w, const float *g, float *m, float *v, int K, int N, float lr,
                            const float *lr_dev, float beta1, float beta2, float eps, int64_t t,
                            uint64_t *t_dev, uint16_t *wt_bf16, int64_t ldt, uint16_t *wc_bf16,
                            int64_t ldc, float *bias_w, const float *bias_g, float *bias_m,
                            float *bias_v, int bias_n, int advance_step, uint32_t *tickets,
                            int planes, int64_t plane_t, int64_t plane_c, cdml_stream_t stream, float h2_scale = 0.f) {
  CDML_REQUIRE(w && g && m && v && K > 0 && N > 0, CDML_E_BADARG, "adam_matrix_bf16: bad argument");
  CDML_REQUIRE(!bias_w || (bias_g && bias_m && bias_v && bias_n > 0), CDML_E_BADARG,
               "adam_matrix_bf16: the bias vector needs its gradient and both moments");
  CDML_REQUIRE(!advance_step || (t_dev && tickets), CDML_E_BADARG,
               "adam_matrix_bf16: advance_step needs the device step counter and the ticket words");
  CDML_REQUIRE(K % kAT == 0 && N % kAT == 0, CDML_E_UNSUPPORTED,
               "adam_matrix_bf16: K and N must be multiples of 64, got K=%d N=%d", K, N);
  CDML_REQUIRE(t >= (t_dev ? 0 : 1), CDML_E_BADARG, "adam_matrix_bf16: step t is 1-based");
  CDML_REQUIRE(aligned16(w) && aligned16(g) && aligned16(m) && aligned16(v) &&
                   (!wt_bf16 || (aligned16(wt_bf16) && (ldt & 7) == 0 && ldt >= K)) &&
                   (!wc_bf16 || (aligned16(wc_bf16) && (ldc & 3) == 0 && ldc >= N)),
               CDML_E_ALIGN, "adam_matrix_bf16: 16-B aligned buffers, ldt a multiple of 8 (>= K), ldc of 4 (>= N)");
  if (planes == 2) {
    CDML_REQUIRE(h2_scale > 0.f && (!wt_bf16 || (!(plane_t & 7) && plane_t >= K && ldt >= plane_t + K)) &&
                     (!wc_bf16 || (!(plane_c & 3) && plane_c >= N && ldc >= plane_c + N)),
                 CDML_E_ALIGN, "adam_matrix_h2: a positive scale, plane strides (W^T: multiple of 8, >= K; W: multiple of 4, >= N) and "
                 "leading dimensions >= plane + the matrix width");
    hipLaunchKernelGGL(k_adam_matrix_bf16<2>, dim3((K / kAT) * (N / kAT)), dim3(kThreads), 0, (hipStream_t)stream, w, g,
                       m, v, K, N, lr, lr_dev, beta1, beta2, eps, t, t_dev, reinterpret_cast<__bf16 *>(wt_bf16), ldt,
                       reinterpret_cast<__bf16 *>(wc_bf16), ldc, bias_w, bias_g, bias_m, bias_v, bias_w ? bias_n : 0,
                       advance_step, tickets, plane_t, plane_c, h2_scale);
  } else if (planes == 3) {
    CDML_REQUIRE((!wt_bf16 || (!(plane_t & 7) && plane_t >= K && ldt >= 2 * plane_t + K)) &&
                     (!wc_bf16 || (!(plane_c & 3) && plane_c >= N && ldc >= 2 * plane_c + N)),
                 CDML_E_ALIGN, "adam_matrix_planes: plane strides (W^T: multiple of 8, >= K; W: multiple of 4, >= N) and "
                 "leading dimensions >= 2 planes + the matrix width");
    hipLaunchKernelGGL(k_adam_matrix_bf16<3>, dim3((K / kAT) * (N / kAT)), dim3(kThreads), 0, (hipStream_t)stream, w, g,
                       m, v, K, N, lr, lr_dev, beta1, beta2, eps, t, t_dev, reinterpret_cast<__bf16 *>(wt_bf16), ldt,
                       reinterpret_cast<__bf16 *>(wc_bf16), ldc, bias_w, bias_g, bias_m, bias_v, bias_w ? bias_n : 0,
                       advance_step, tickets, plane_t, plane_c, 0.f);
  } else {
    hipLaunchKernelGGL(k_adam_matrix_bf16<1>, dim3((K / kAT) * (N / kAT)), dim3(kThreads), 0, (hipStream_t)stream, w, g,
                       m, v, K, N, lr, lr_dev, beta1, beta2, eps, t, t_dev, reinterpret_cast<__bf16 *>(wt_bf16), ldt,
                       reinterpret_cast<__bf16 *>(wc_bf16), ldc, bias_w, bias_g, bias_m, bias_v, bias_w ? bias_n : 0,
                       advance_step, tickets, (int64_t)0, (int64_t)0, 0.f);
  }
  return check_launch("adam_matrix_bf16");
}

extern "C" int cdml_adam_matrix_bf16(float *w, const float *g, float *m, float *v, int K, int N, float lr,
                                     const float *lr_dev, float beta1, float beta2, float eps, int64_t t,
                                     uint64_t *t_dev, uint16_t *wt_bf16, int64_t ldt, uint16_t *wc_bf16,
                                     int64_t ldc, float *bias_w, const float *bias_g, float *bias_m,
                                     float *bias_v, int bias_n, int advance_step, uint32_t *tickets,
                                     cdml_stream_t stream) {
  return adam_matrix_impl(w, g, m, v, K, N, lr, lr_dev, beta1, beta2, eps, t, t_dev, wt_bf16, ldt, wc_bf16, ldc, bias_w,
                          bias_g, bias_m, bias_v, bias_n, advance_step, tickets, 1, 0, 0, stream);
}

// The same update writing the copies as the three bf16 planes hi | mid | lo of the new weights (precision "f32x3":
// wt = planes of W^T, [N][hi K | mid K | lo K] with plane stride plane_t; wc = planes of W, [K][hi N | ...], plane_c).
extern "C" int cdml_adam_matrix_planes(float *w, const float *g, float *m, float *v, int K, int N, float lr,
                                       const float *lr_dev, float beta1, float beta2, float eps, int64_t t,
                                       uint64_t *t_dev, uint16_t *wt_planes, int64_t ldt, int64_t plane_t,
                                       uint16_t *wc_planes, int64_t ldc, int64_t plane_c, float *bias_w,
                                       const float *bias_g, float *bias_m, float *bias_v, int bias_n,
                                       int advance_step, uint32_t *tickets, cdml_stream_t stream) {
  return adam_matrix_impl(w, g, m, v, K, N, lr, lr_dev, beta1, beta2, eps, t, t_dev, wt_planes, ldt, wc_planes, ldc, bias_w,
                          bias_g, bias_m, bias_v, bias_n, advance_step, tickets, 3, plane_t, plane_c, stream);
}

// The same update writing the copies as the two fp16 planes hi | lo of the new weights times `scale` (precision "f16x2";
// a power of two, values beyond fp16's range saturate): wt = planes of W^T [N][hi K | lo K], wc = planes of W [K][hi N | lo N].
extern "C" int cdml_adam_matrix_h2(float *w, const float *g, float *m, float *v, int K, int N, float lr,
                                   const float *lr_dev, float beta1, float beta2, float eps, int64_t t,
                                   uint64_t *t_dev, uint16_t *wt_planes, int64_t ldt, int64_t plane_t,
                                   uint16_t *wc_planes, int64_t ldc, int64_t plane_c, float scale, float *bias_w,
                                   const float *bias_g, float *bias_m, float *bias_v, int bias_n,
                                   int advance_step, uint32_t *tickets, cdml_stream_t stream) {
  return adam_matrix_impl(w, g, m, v, K, N, lr, lr_dev, beta1, beta2, eps, t, t_dev, wt_planes, ldt, wc_planes, ldc, bias_w,
                          bias_g, bias_m, bias_v, bias_n, advance_step, tickets, 2, plane_t, plane_c, stream, scale);
}

extern "C" size_t cdml_lars_scratch_floats(void) { return 2 + 2 * (size_t)kLarsBlocks; }

extern "C" int cdml_lars_step(float *w, const float *g, float *acc, int64_t n, float lr,
                              const float *lr_dev, float momentum, float weight_decay, float eeta,
                              float eps, float *scratch, cdml_stream_t stream) {
  CDML_REQUIRE(w && g && acc && scratch && n > 0, CDML_E_BADARG, "lars_step: bad argument");
  int blocks = grid_elems(n, 8);
  if (blocks > kLarsBlocks) blocks = kLarsBlocks;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_lars_norm_partial, dim3(blocks), dim3(kThreads), 0, s, w, g, n, scratch);
  hipLaunchKernelGGL(k_lars_norm_final, dim3(1), dim3(64), 0, s, scratch, blocks);
  hipLaunchKernelGGL(k_lars_apply, dim3(grid_elems(n, 4)), dim3(kThreads), 0, s, w, g, acc, n, lr,
                     lr_dev, momentum, weight_decay, eeta, eps, scratch);
  return check_launch("lars_step");
}

// segments -> LarsSegs: offsets, and the norm blocks dealt in proportion to the segment sizes (at least one each,
// kLarsBlocks at most in all).  One function for every LARS entry point: the norms launch and the launches that
// reduce its partials must agree on the layout.
static int lars_layout(const int64_t *seg_offsets, const int64_t *seg_sizes, int n_seg, LarsSegs &S, int &nb, int64_t &total) {
  CDML_REQUIRE(seg_offsets && seg_sizes, CDML_E_BADARG, "lars: segment arrays required");
  CDML_REQUIRE(n_seg >= 1 && n_seg <= kMaxSeg, CDML_E_UNSUPPORTED, "lars: 1..%d segments, got %d", kMaxSeg, n_seg);
  S.n_seg = n_seg;
  total = 0;
  for (int k = 0; k < n_seg; ++k) {
    CDML_REQUIRE(seg_sizes[k] > 0 && (seg_sizes[k] & 3) == 0 && seg_offsets[k] == total, CDML_E_BADARG,
                 "lars: segments must tile the buffer contiguously in multiples of 4 floats (segment %d)", k);
    S.off[k] = total;
    total += seg_sizes[k];
  }
  S.off[n_seg] = total;
  const int budget = kLarsBlocks - n_seg;
  nb = 0;
  for (int k = 0; k < n_seg; ++k) {
    S.blk[k] = nb;
    int64_t want = (seg_sizes[k] / 4 + kThreads * 8 - 1) / (kThreads * 8);          // >= 8 float4 per thread
    const int64_t share = 1 + (int64_t)((double)budget * (double)seg_sizes[k] / (double)total);
    if (want > share) want = share;
    nb += (int)(want < 1 ? 1 : want);
  }
  S.blk[n_seg] = nb;
  for (int k = n_seg + 1; k <= kMaxSeg; ++k) { S.off[k] = total; S.blk[k] = nb; }
  return CDML_OK;
}

extern "C" size_t cdml_lars_multi_scratch_floats(void) { return 2 * (size_t)kLarsBlocks; }

extern "C" int cdml_lars_multi(float *w, const float *g, float *acc, const int64_t *seg_offsets,
                               const int64_t *seg_sizes, int n_seg, float lr, const float *lr_dev,
                               float momentum, float weight_decay, float eeta, float eps, float *scratch,
                               float *norms_out, uint64_t *step_dev_advance, uint32_t *tickets,
                               cdml_stream_t stream) {
  CDML_REQUIRE(w && g && acc && scratch && seg_offsets && seg_sizes, CDML_E_BADARG, "lars_multi: bad argument");
  CDML_REQUIRE(!step_dev_advance || tickets, CDML_E_BADARG, "lars_multi: advancing the step counter needs the ticket words");
  CDML_REQUIRE(aligned16(w) && aligned16(g) && aligned16(acc), CDML_E_ALIGN, "lars_multi: buffers must be 16-B aligned");
  LarsSegs S;
  int nb = 0;
  int64_t total = 0;
  if (int rc = lars_layout(seg_offsets, seg_sizes, n_seg, S, nb, total)) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_lars_multi_norms, dim3(nb), dim3(kThreads), 0, s, w, g, S, scratch);
  hipLaunchKernelGGL(k_lars_multi_apply, dim3(grid_elems(total, 4)), dim3(kThreads), 0, s, w, g, acc, S, lr, lr_dev,
                     momentum, weight_decay, eeta, eps, scratch, norms_out, step_dev_advance, tickets);
  return check_launch("lars_multi");
}

// LARS with the GEMMs' operand copies written by the update (config 4: bf16, planes = 1; precision "f32x3": three bf16
// planes, planes = 3).  cdml_lars_multi_norms first (one launch over every variable of the flat buffer), then one
// cdml_lars_matrix launch per weight matrix, each with its bias vector riding along: w / g / acc are the FLAT buffers,
// seg_matrix / seg_bias (-1: none) index the segment arrays the norms launch was given, K x N the matrix' shape.
extern "C" int cdml_lars_multi_norms(const float *w, const float *g, const int64_t *seg_offsets, const int64_t *seg_sizes,
                                     int n_seg, float *scratch, cdml_stream_t stream) {
  CDML_REQUIRE(w && g && scratch, CDML_E_BADARG, "lars_multi_norms: bad argument");
  CDML_REQUIRE(aligned16(w) && aligned16(g), CDML_E_ALIGN, "lars_multi_norms: buffers must be 16-B aligned");
  LarsSegs S;
  int nb = 0;
  int64_t total = 0;
  if (int rc = lars_layout(seg_offsets, seg_sizes, n_seg, S, nb, total)) return rc;
  hipLaunchKernelGGL(k_lars_multi_norms, dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, w, g, S, scratch);
  return check_launch("lars_multi_norms");
}

static int check_copies(const char *who, int K, int N, const uint16_t *wt, int64_t ldt, int64_t plane_t, const uint16_t *wc,
                        int64_t ldc, int64_t plane_c, int planes) {
  CDML_REQUIRE(planes == 1 || planes == 2 || planes == 3, CDML_E_BADARG, "%s: planes must be 1 (bf16 copies), 2 (fp16 hi | lo) or 3 (hi | mid | lo)", who);
  CDML_REQUIRE(K > 0 && N > 0 && K % kAT == 0 && N % kAT == 0, CDML_E_UNSUPPORTED,
               "%s: K and N must be multiples of 64, got K=%d N=%d", who, K, N);
  CDML_REQUIRE((!wt || (aligned16(wt) && (ldt & 7) == 0 && ldt >= K)) && (!wc || (aligned16(wc) && (ldc & 3) == 0 && ldc >= N)),
               CDML_E_ALIGN, "%s: 16-B aligned copies, ldt a multiple of 8 (>= K), ldc of 4 (>= N)", who);
  if (planes >= 2)
    CDML_REQUIRE((!wt || (!(plane_t & 7) && plane_t >= K && ldt >= (planes - 1) * plane_t + K)) &&
                     (!wc || (!(plane_c & 3) && plane_c >= N && ldc >= (planes - 1) * plane_c + N)),
                 CDML_E_ALIGN, "%s: plane strides (W^T: multiple of 8, >= K; W: multiple of 4, >= N) and leading dimensions "
                 ">= (planes - 1) planes + the matrix width", who);
  return CDML_OK;
}

static int lars_matrix_impl(float h2_scale, float *w, const float *g, float *acc, const int64_t *seg_offsets, const int64_t *seg_sizes,
                            int n_seg, int seg_matrix, int seg_bias, int K, int N, float lr, const float *lr_dev,
                            float momentum, float weight_decay, float eeta, float eps, const float *scratch,
                            float *norms_out, uint16_t *wt, int64_t ldt, int64_t plane_t, uint16_t *wc, int64_t ldc,
                            int64_t plane_c, int planes, uint64_t *step_dev_advance, uint32_t *tickets,
                            cdml_stream_t stream) {
  CDML_REQUIRE((planes == 2) == (h2_scale > 0.f), CDML_E_BADARG, "lars_matrix: fp16 planes (planes = 2) go with a positive scale");
  CDML_REQUIRE(w && g && acc && scratch, CDML_E_BADARG, "lars_matrix: bad argument");
  CDML_REQUIRE(!step_dev_advance || tickets, CDML_E_BADARG, "lars_matrix: advancing the step counter needs the ticket words");
  CDML_REQUIRE(aligned16(w) && aligned16(g) && aligned16(acc), CDML_E_ALIGN, "lars_matrix: buffers must be 16-B aligned");
  LarsSegs S;
  int nb = 0;
  int64_t total = 0;
  if (int rc = lars_layout(seg_offsets, seg_sizes, n_seg, S, nb, total)) return rc;
  CDML_REQUIRE(seg_matrix >= 0 && seg_matrix < n_seg && seg_bias < n_seg && seg_bias != seg_matrix, CDML_E_BADARG,
               "lars_matrix: segment indices out of range");
  CDML_REQUIRE(seg_sizes[seg_matrix] == (int64_t)K * N, CDML_E_BADARG, "lars_matrix: segment %d holds %lld floats, not %d x %d",
               seg_matrix, (long long)seg_sizes[seg_matrix], K, N);
  if (int rc = check_copies("lars_matrix", K, N, wt, ldt, plane_t, wc, ldc, plane_c, planes)) return rc;
  MatRule R{};
  R.lr_imm = lr; R.lr_dev = lr_dev; R.momentum = momentum; R.wd = weight_decay; R.eeta = eeta; R.eps = eps;
  R.scratch = scratch; R.h2_scale = h2_scale;
  R.blk_w0 = S.blk[seg_matrix]; R.blk_w1 = S.blk[seg_matrix + 1];
  R.norms_w = norms_out ? norms_out + 2 * seg_matrix : nullptr;
  float *bw = nullptr, *bacc = nullptr;
  const float *bg = nullptr;
  int bn = 0;
  if (seg_bias >= 0) {
    R.blk_b0 = S.blk[seg_bias]; R.blk_b1 = S.blk[seg_bias + 1];
    R.norms_b = norms_out ? norms_out + 2 * seg_bias : nullptr;
    bw = w + S.off[seg_bias]; bg = g + S.off[seg_bias]; bacc = acc + S.off[seg_bias];
    bn = (int)seg_sizes[seg_bias];
  }
  const int64_t o = S.off[seg_matrix];
  const dim3 grid((K / kAT) * (N / kAT));
  if (planes == 2)
    hipLaunchKernelGGL((k_rule_matrix<2, 1>), grid, dim3(kThreads), 0, (hipStream_t)stream, w + o, g + o, acc + o, K, N, R,
                       reinterpret_cast<__bf16 *>(wt), ldt, reinterpret_cast<__bf16 *>(wc), ldc, bw, bg, bacc, bn,
                       step_dev_advance, tickets, plane_t, plane_c);
  else if (planes == 3)
    hipLaunchKernelGGL((k_rule_matrix<3, 1>), grid, dim3(kThreads), 0, (hipStream_t)stream, w + o, g + o, acc + o, K, N, R,
                       reinterpret_cast<__bf16 *>(wt), ldt, reinterpret_cast<__bf16 *>(wc), ldc, bw, bg, bacc, bn,
                       step_dev_advance, tickets, plane_t, plane_c);
  else
    hipLaunchKernelGGL((k_rule_matrix<1, 1>), grid, dim3(kThreads), 0, (hipStream_t)stream, w + o, g + o, acc + o, K, N, R,
                       reinterpret_cast<__bf16 *>(wt), ldt, reinterpret_cast<__bf16 *>(wc), ldc, bw, bg, bacc, bn,
                       step_dev_advance, tickets, (int64_t)0, (int64_t)0);
  return check_launch("lars_matrix");
}

extern "C" int cdml_lars_matrix(float *w, const float *g, float *acc, const int64_t *seg_offsets, const int64_t *seg_sizes,
                                int n_seg, int seg_matrix, int seg_bias, int K, int N, float lr, const float *lr_dev,
                                float momentum, float weight_decay, float eeta, float eps, const float *scratch,
                                float *norms_out, uint16_t *wt, int64_t ldt, int64_t plane_t, uint16_t *wc, int64_t ldc,
                                int64_t plane_c, int planes, uint64_t *step_dev_advance, uint32_t *tickets,
                                cdml_stream_t stream) {
  CDML_REQUIRE(planes != 2, CDML_E_BADARG, "lars_matrix: fp16 planes take a scale: cdml_lars_matrix_h2");
  return lars_matrix_impl(0.f, w, g, acc, seg_offsets, seg_sizes, n_seg, seg_matrix, seg_bias, K, N, lr, lr_dev, momentum, weight_decay,
                          eeta, eps, scratch, norms_out, wt, ldt, plane_t, wc, ldc, plane_c, planes, step_dev_advance, tickets, stream);
}

// cdml_lars_matrix / cdml_momentum_matrix writing the copies as the two fp16 planes hi | lo of the new weights times `scale`
// (precision "f16x2"; as cdml_adam_matrix_h2)
extern "C" int cdml_lars_matrix_h2(float *w, const float *g, float *acc, const int64_t *seg_offsets, const int64_t *seg_sizes,
                                   int n_seg, int seg_matrix, int seg_bias, int K, int N, float lr, const float *lr_dev,
                                   float momentum, float weight_decay, float eeta, float eps, const float *scratch,
                                   float *norms_out, uint16_t *wt, int64_t ldt, int64_t plane_t, uint16_t *wc, int64_t ldc,
                                   int64_t plane_c, float scale, uint64_t *step_dev_advance, uint32_t *tickets,
                                   cdml_stream_t stream) {
  return lars_matrix_impl(scale, w, g, acc, seg_offsets, seg_sizes, n_seg, seg_matrix, seg_bias, K, N, lr, lr_dev, momentum, weight_decay,
                          eeta, eps, scratch, norms_out, wt, ldt, plane_t, wc, ldc, plane_c, 2, step_dev_advance, tickets, stream);
}

// tf.train.MomentumOptimizer on a weight matrix W[K][N] (+ its bias vector), writing the operand copies like
// cdml_lars_matrix; step_dev_advance (with tickets): also global_step += 1 by the last block.
static int momentum_matrix_impl(float h2_scale, float *w, const float *g, float *acc, int K, int N, float lr, const float *lr_dev,
                                float momentum, int use_nesterov, uint16_t *wt, int64_t ldt, int64_t plane_t,
                                uint16_t *wc, int64_t ldc, int64_t plane_c, int planes, float *bias_w,
                                const float *bias_g, float *bias_acc, int bias_n, uint64_t *step_dev_advance,
                                uint32_t *tickets, cdml_stream_t stream) {
  CDML_REQUIRE((planes == 2) == (h2_scale > 0.f), CDML_E_BADARG, "momentum_matrix: fp16 planes (planes = 2) go with a positive scale");
  CDML_REQUIRE(w && g && acc, CDML_E_BADARG, "momentum_matrix: bad argument");
  CDML_REQUIRE(!bias_w || (bias_g && bias_acc && bias_n > 0), CDML_E_BADARG,
               "momentum_matrix: the bias vector needs its gradient and accumulator");
  CDML_REQUIRE(!step_dev_advance || tickets, CDML_E_BADARG, "momentum_matrix: advancing the step counter needs the ticket words");
  CDML_REQUIRE(aligned16(w) && aligned16(g) && aligned16(acc), CDML_E_ALIGN, "momentum_matrix: buffers must be 16-B aligned");
  if (int rc = check_copies("momentum_matrix", K, N, wt, ldt, plane_t, wc, ldc, plane_c, planes)) return rc;
  MatRule R{};
  R.lr_imm = lr; R.lr_dev = lr_dev; R.momentum = momentum; R.nesterov = use_nesterov; R.h2_scale = h2_scale;
  const dim3 grid((K / kAT) * (N / kAT));
  if (planes == 2)
    hipLaunchKernelGGL((k_rule_matrix<2, 2>), grid, dim3(kThreads), 0, (hipStream_t)stream, w, g, acc, K, N, R,
                       reinterpret_cast<__bf16 *>(wt), ldt, reinterpret_cast<__bf16 *>(wc), ldc, bias_w, bias_g, bias_acc,
                       bias_w ? bias_n : 0, step_dev_advance, tickets, plane_t, plane_c);
  else if (planes == 3)
    hipLaunchKernelGGL((k_rule_matrix<3, 2>), grid, dim3(kThreads), 0, (hipStream_t)stream, w, g, acc, K, N, R,
                       reinterpret_cast<__bf16 *>(wt), ldt, reinterpret_cast<__bf16 *>(wc), ldc, bias_w, bias_g, bias_acc,
                       bias_w ? bias_n : 0, step_dev_advance, tickets, plane_t, plane_c);
  else
    hipLaunchKernelGGL((k_rule_matrix<1, 2>), grid, dim3(kThreads), 0, (hipStream_t)stream, w, g, acc, K, N, R,
                       reinterpret_cast<__bf16 *>(wt), ldt, reinterpret_cast<__bf16 *>(wc), ldc, bias_w, bias_g, bias_acc,
                       bias_w ? bias_n : 0, step_dev_advance, tickets, (int64_t)0, (int64_t)0);
  return check_launch("momentum_matrix");
}

extern "C" int cdml_momentum_matrix(float *w, const float *g, float *acc, int K, int N, float lr, const float *lr_dev,
                                    float momentum, int use_nesterov, uint16_t *wt, int64_t ldt, int64_t plane_t,
                                    uint16_t *wc, int64_t ldc, int64_t plane_c, int planes, float *bias_w,
                                    const float *bias_g, float *bias_acc, int bias_n, uint64_t *step_dev_advance,
                                    uint32_t *tickets, cdml_stream_t stream) {
  CDML_REQUIRE(planes != 2, CDML_E_BADARG, "momentum_matrix: fp16 planes take a scale: cdml_momentum_matrix_h2");
  return momentum_matrix_impl(0.f, w, g, acc, K, N, lr, lr_dev, momentum, use_nesterov, wt, ldt, plane_t, wc, ldc, plane_c, planes, bias_w,
                              bias_g, bias_acc, bias_n, step_dev_advance, tickets, stream);
}

extern "C" int cdml_momentum_matrix_h2(float *w, const float *g, float *acc, int K, int N, float lr, const float *lr_dev,
                                       float momentum, int use_nesterov, uint16_t *wt, int64_t ldt, int64_t plane_t,
                                       uint16_t *wc, int64_t ldc, int64_t plane_c, float scale, float *bias_w,
                                       const float *bias_g, float *bias_acc, int bias_n, uint64_t *step_dev_advance,
                                       uint32_t *tickets, cdml_stream_t stream) {
  return momentum_matrix_impl(scale, w, g, acc, K, N, lr, lr_dev, momentum, use_nesterov, wt, ldt, plane_t, wc, ldc, plane_c, 2, bias_w,
                              bias_g, bias_acc, bias_n, step_dev_advance, tickets, stream);
}

extern "C" int cdml_grad_prepare(float *g, const float *w, int64_t n, float l2_scale, float clip_norm,
                                 float *scratch, float *norms_out, cdml_stream_t stream) {
  CDML_REQUIRE(g && w && scratch && n > 0, CDML_E_BADARG, "grad_prepare: bad argument");
  int blocks = grid_elems(n, 8);
  if (blocks > kLarsBlocks) blocks = kLarsBlocks;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_prep_partial, dim3(blocks), dim3(kThreads), 0, s, w, g, n, l2_scale, scratch);
  hipLaunchKernelGGL(k_lars_norm_final, dim3(1), dim3(64), 0, s, scratch, blocks);
  hipLaunchKernelGGL(k_prep_apply, dim3(grid_elems(n, 4)), dim3(kThreads), 0, s, g, w, n, l2_scale, clip_norm,
                     scratch, norms_out);
  return check_launch("grad_prepare");
}

extern "C" int cdml_momentum_step(float *w, const float *g, float *acc, int64_t n, float lr,
                                  const float *lr_dev, float momentum, int use_nesterov,
                                  cdml_stream_t stream) {
  CDML_REQUIRE(w && g && acc && n > 0, CDML_E_BADARG, "momentum_step: bad argument");
  hipLaunchKernelGGL(k_momentum, dim3(grid_elems(n, 4)), dim3(kThreads), 0, (hipStream_t)stream, w, g, acc, n, lr,
                     lr_dev, momentum, use_nesterov);
  return check_launch("momentum_step");
}
