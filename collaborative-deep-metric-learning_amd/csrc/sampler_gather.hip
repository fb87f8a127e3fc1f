// Triplet sampler + feature-row gather (+ fused input l2-normalise) for gfx950.
//
// Replaces the reference's host input pipe: MPTripletPipe.subprocess
// (inputs.py:102-142: pair stream + negative draw), get_batch's numpy
// fancy-index gather FEATURES[idx] (inputs.py:158), the reshape + feed_dict H2D
// copy (train.py:313,318) and the tower's first op tf.nn.l2_normalize
// (models.py:58).  The catalogue lives in HBM; nothing touches the host.
//
// Roofline: HBM.  Algorithmic bytes per gathered row = F*4 read + F*4 written.
// One wave owns one row: 16 B/lane loads (1 KiB per wave-instruction), all of
// a row's loads issued before the first use, wave-shuffle reduction for the
// norm.  Rows start 128-B aligned when row_stride*4 is a multiple of 128.
#include "common.h"

namespace cdml {
namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / kWave;

#ifndef CDML_GATHER_NT_STORE
#define CDML_GATHER_NT_STORE 0   // 1: non-temporal stores of the gathered rows
#endif
#ifndef CDML_GATHER_NT
#define CDML_GATHER_NT 1   // non-temporal table loads: a row is read once, keep it out of L2 / Infinity Cache
                           // (A/B on MI355X, 32 768 rows per launch: 5.3 -> 6.2 TB/s; profiles/r02_gather_variants.txt)
#endif
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float4 load_row_chunk(const float4 *p) {
#if CDML_GATHER_NT
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}

// ---------------------------------------------------------------- table fill --
__global__ void __launch_bounds__(kThreads)
k_fill_table(float *__restrict__ table, int64_t row0, int64_t n_rows, int F,
             int64_t row_stride, uint64_t seed) {
  const int64_t q_per_row = row_stride >> 2;
  const int64_t total = n_rows * q_per_row;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lr = i / q_per_row;
    const int64_t q = i - lr * q_per_row;
    const uint64_t r = (uint64_t)(row0 + lr);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t j = q * 4;
    if (j < F) {
      u32x4 w = philox4x32_10(u32x4{(uint32_t)q, (uint32_t)r, (uint32_t)(r >> 32), kTableTag}, k0, k1);
      const float s = 5.9604644775390625e-08f;  // 2^-24
      v.x = (float)(w.x >> 8) * s;
      v.y = (j + 1 < F) ? (float)(w.y >> 8) * s : 0.f;
      v.z = (j + 2 < F) ? (float)(w.z >> 8) * s : 0.f;
      v.w = (j + 3 < F) ? (float)(w.w >> 8) * s : 0.f;
    }
    reinterpret_cast<float4 *>(table)[i] = v;
  }
}

// ------------------------------------------------------------------- sampler --
__global__ void __launch_bounds__(kThreads)
k_sample_uniform(const int32_t *__restrict__ pairs, int64_t n_pairs, uint32_t n_rows,
                 uint64_t seed, uint64_t step_imm, const uint64_t *__restrict__ step_dev,
                 int batch, int64_t slot0, int64_t batch_global, int32_t *__restrict__ idx_out) {
  const uint64_t step = step_imm + (step_dev ? *step_dev : 0);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  const uint64_t slot = (uint64_t)(slot0 + i);
  const uint64_t q = (step * (uint64_t)batch_global + slot) % (uint64_t)n_pairs;
  const int32_t a = pairs[2 * q], p = pairs[2 * q + 1];
  const int32_t n = sample_uniform_negative(seed, step, (uint32_t)slot, a, p, n_rows);
  idx_out[3 * i + 0] = a;
  idx_out[3 * i + 1] = p;
  idx_out[3 * i + 2] = n;
}

__global__ void __launch_bounds__(kThreads)
k_sample_inbatch(const int32_t *__restrict__ pairs, int64_t n_pairs, uint64_t seed,
                 uint64_t step_imm, const uint64_t *__restrict__ step_dev, int batch,
                 int64_t slot0, int64_t batch_global, int32_t *__restrict__ rows_out,
                 int32_t *__restrict__ shift_out) {
  const uint64_t step = step_imm + (step_dev ? *step_dev : 0);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *shift_out = sample_inbatch_shift(seed, step, batch);
  if (i >= batch) return;
  const uint64_t slot = (uint64_t)(slot0 + i);
  const uint64_t q = (step * (uint64_t)batch_global + slot) % (uint64_t)n_pairs;
  rows_out[2 * i + 0] = pairs[2 * q];
  rows_out[2 * i + 1] = pairs[2 * q + 1];
}

__global__ void k_step_advance(uint64_t *step_dev) { *step_dev += 1; }

// -------------------------------------------------------------------- gather --
// One wave gathers (and optionally l2-normalises) one row.  NCH = float4 chunks
// per lane held in registers (covers F <= 256*NCH).
template <int NCH>
__device__ __forceinline__ void gather_one_row(const float *__restrict__ table, int64_t local_row,
                                               int64_t row_stride, int F, int normalize,
                                               float *__restrict__ dst, int64_t out_stride,
                                               float *__restrict__ inv_out, int lane) {
  const float4 *src = reinterpret_cast<const float4 *>(table + local_row * row_stride);
  const int nq = (F + 3) >> 2;
  float4 v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int q = lane + kWave * c;
    v[c] = (q < nq) ? load_row_chunk(src + q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (F & 3) {  // mask the tail so the table's pad content never leaks in
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j = 4 * (lane + kWave * c);
      if (j + 1 >= F && j < F) v[c].y = 0.f;
      if (j + 2 >= F && j < F) v[c].z = 0.f;
      if (j + 3 >= F && j < F) v[c].w = 0.f;
    }
  }
  float inv = 1.f;
  if (normalize) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      ss += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
    ss = wave_sum(ss);
    inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
  }
  if (inv_out && lane == 0) *inv_out = inv;
  float4 *d = reinterpret_cast<float4 *>(dst);
  const int oq = (int)(out_stride >> 2);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int q = lane + kWave * c;
    if (q < oq) d[q] = make_float4(v[c].x * inv, v[c].y * inv, v[c].z * inv, v[c].w * inv);
  }
  for (int q = lane + kWave * NCH; q < oq; q += kWave) d[q] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ int64_t clamp_row(int32_t id, int64_t row0, int64_t n_rows,
                                             int32_t *oob_flag) {
  int64_t lr = (int64_t)id - row0;
  if (lr < 0 || lr >= n_rows) {
    if (oob_flag) atomicOr(oob_flag, 1);
    lr = lr < 0 ? 0 : n_rows - 1;
  }
  return lr;
}

template <int NCH>
__global__ void __launch_bounds__(kThreads)
k_gather_rows(const float *__restrict__ table, int64_t row0, int64_t n_rows, int64_t row_stride,
              const int32_t *__restrict__ idx, int n_idx, int F, int normalize,
              float *__restrict__ x_out, int64_t out_stride, float *__restrict__ inv_norm_out,
              int32_t *__restrict__ oob_flag) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < n_idx; r += gridDim.x * kWavesPerBlock) {
    const int32_t id = idx[r];
    if (id == -1) {
      // owner side of a fixed-capacity exchange: a padding slot, left untouched.  Requester side
      // (flag bit 1 set): a request that found no slot -- the row is filled with all-ones words
      // (NaN as fp32 and as bf16 pairs), so the same step's loss shows the overflow
      if (normalize & 2) {
        float4 *d = reinterpret_cast<float4 *>(x_out + (int64_t)r * out_stride);
        const float nanf_ = __uint_as_float(0xFFFFFFFFu);
        for (int q = lane; q < (int)(out_stride >> 2); q += kWave) d[q] = make_float4(nanf_, nanf_, nanf_, nanf_);
      }
      continue;
    }
    const int64_t lr = clamp_row(id, row0, n_rows, oob_flag);
    gather_one_row<NCH>(table, lr, row_stride, F, normalize & 1, x_out + (int64_t)r * out_stride,
                        out_stride, inv_norm_out ? inv_norm_out + r : nullptr, lane);
  }
}

// out planes[r] = the three bf16 planes hi | mid | lo of the fp32 row src[idx[r]] AS STORED (no normalisation): the last step
// of the row exchange on the split-fp32 path (round 5) -- the received rows go into request order and into the GEMMs' operand
// form in one pass on the prefetch stream, instead of fp32 rows there and a split launch on the compute stream every step.
// idx -1 (flag bit 1): a request that found no slot -> all-ones words (NaN in every plane), as k_gather_rows.
template <int NCH>
__global__ void __launch_bounds__(kThreads)
k_gather_rows_planes(const float *__restrict__ src, int64_t n_rows, int64_t row_stride, const int32_t *__restrict__ idx, int n_idx,
                     int F, int flags, __bf16 *__restrict__ out, int64_t out_stride, int32_t *__restrict__ oob_flag) {
  using bf16x4v = __attribute__((ext_vector_type(4))) __bf16;
  using u32x2v = __attribute__((ext_vector_type(2))) uint32_t;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t plane = out_stride / 3;
  const int oq = (int)(plane >> 2), nq = (F + 3) >> 2;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < n_idx; r += gridDim.x * kWavesPerBlock) {
    const int32_t id = idx[r];
    __bf16 *dst = out + (int64_t)r * out_stride;
    if (id == -1) {
      if (flags & 2)
        for (int q = lane; q < 3 * oq; q += kWave) *reinterpret_cast<u32x2v *>(dst + 4 * q) = u32x2v{0xFFFFFFFFu, 0xFFFFFFFFu};
      continue;
    }
    const int64_t lr = clamp_row(id, 0, n_rows, oob_flag);
    const float4 *s4 = reinterpret_cast<const float4 *>(src + lr * row_stride);
    float4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      v[c] = (q < nq) ? s4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      if (q < oq) {
        float x[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
        if (F & 3) {                                        // the pad of the source row never leaks in
#pragma unroll
          for (int u = 0; u < 4; ++u) if (4 * q + u >= F) x[u] = 0.f;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          bf16x4v o;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            o[u] = (__bf16)x[u];
            x[u] -= (float)o[u];
          }
          *reinterpret_cast<bf16x4v *>(dst + pl * plane + 4 * q) = o;
        }
      }
    }
    const bf16x4v z4 = {0, 0, 0, 0};
    for (int q = lane + kWave * NCH; q < oq; q += kWave)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x4v *>(dst + pl * plane + 4 * q) = z4;
  }
}

// Persistent fused sampler + gather + input l2-normalise, one launch for n_steps consecutive
// training steps (the sampler is counter-based, so the triplets of step t+1 are known at step
// t; two or more steps per launch amortise the launch ramp and the ids -> row latency chain of
// a kernel that is otherwise only ~16 us of HBM traffic long).
//   * the launch's rows (n_steps x RPT x batch; RPT = 3 uniform, 2 in-batch) are cut into
//     chunks of kChunkRows; a block walks chunks c, c + grid, ...;
//   * the ids of a chunk are produced once by the block's first lanes -- pair stream position,
//     pair load, the Philox negative where the row is one -- and STAGED IN LDS (also written to
//     idx_out); the ids of the NEXT chunk are computed while this chunk's row loads are in
//     flight (two LDS buffers);
//   * every wave then owns kRowsPerWave = 2 rows of the chunk: both rows' loads (NCH x 1 KiB
//     each, 16 B per lane = whole 128-B lines) are issued before the first use, then the
//     wave-shuffle norm and the stores.
// MODE 0 = uniform negatives, MODE 1 = in-batch negatives.
#ifndef CDML_GATHER_ROWS_PER_WAVE
#define CDML_GATHER_ROWS_PER_WAVE 2   // rows a wave keeps in flight (A/B'd: tools/gather_variants.sh)
#endif
constexpr int kRowsPerWave = CDML_GATHER_ROWS_PER_WAVE;
constexpr int kChunkRows = kRowsPerWave * kWavesPerBlock;

template <int NCH>
struct RowRegs { float4 v[NCH]; };

template <int NCH>
__device__ __forceinline__ void row_issue(RowRegs<NCH> &R, const float *__restrict__ table, int64_t local_row,
                                          int64_t row_stride, int nq, int lane) {
  const float4 *src = reinterpret_cast<const float4 *>(table + local_row * row_stride);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int q = lane + kWave * c;
    R.v[c] = (q < nq) ? load_row_chunk(src + q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// same arithmetic as gather_one_row (normalize = 1), on rows already in registers
template <int NCH>
__device__ __forceinline__ void row_finish(RowRegs<NCH> &R, int F, float *__restrict__ dst, int64_t out_stride,
                                           int lane) {
  if (F & 3) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int j = 4 * (lane + kWave * c);
      if (j + 1 >= F && j < F) R.v[c].y = 0.f;
      if (j + 2 >= F && j < F) R.v[c].z = 0.f;
      if (j + 3 >= F && j < F) R.v[c].w = 0.f;
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    ss += R.v[c].x * R.v[c].x + R.v[c].y * R.v[c].y + R.v[c].z * R.v[c].z + R.v[c].w * R.v[c].w;
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
  float4 *d = reinterpret_cast<float4 *>(dst);
  const int oq = (int)(out_stride >> 2);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int q = lane + kWave * c;
    if (q < oq) {
#if CDML_GATHER_NT_STORE
      const f32x4 o = f32x4{R.v[c].x * inv, R.v[c].y * inv, R.v[c].z * inv, R.v[c].w * inv};
      __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(d + q));
#else
      d[q] = make_float4(R.v[c].x * inv, R.v[c].y * inv, R.v[c].z * inv, R.v[c].w * inv);
#endif
    }
  }
  for (int q = lane + kWave * NCH; q < oq; q += kWave) d[q] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Row policies of the fused kernel: how a wave loads one catalogue row into registers and how it
// normalises and stores it.  RowF32: fp32 table -> fp32 rows (configs 1-3).  RowF16: fp16 table ->
// bf16 rows, fp32 arithmetic (config 4; same formulas as k_gather_rows_f16).
template <int NCH>
struct RowF32 {
  using In = float;
  using Out = float;
  using Regs = RowRegs<NCH>;
  static constexpr int kPerChunk = 4;          // elements per 16-B chunk
  static constexpr int kRows = kRowsPerWave;   // rows a wave keeps in flight (6 KB each)
  __device__ static __forceinline__ void issue(Regs &R, const In *table, int64_t lr, int64_t stride, int F, int lane) {
    row_issue<NCH>(R, table, lr, stride, (F + 3) >> 2, lane);
  }
  __device__ static __forceinline__ void finish(Regs &R, int F, Out *dst, int64_t out_stride, int lane) {
    row_finish<NCH>(R, F, dst, out_stride, lane);
  }
};

// RowF32X3: fp32 table -> the l2-normalised row as three bf16 planes hi | mid | lo (hi + mid + lo = the fp32 value,
// exactly; the split-fp32 GEMMs of gemm_bf16x3.hip read them) -- out_stride = 3 planes of out_stride / 3 columns.
template <int NCH>
struct RowF32X3 {
  using In = float;
  using Out = __bf16;
  using Regs = RowRegs<NCH>;
  static constexpr int kPerChunk = 4;
  static constexpr int kRows = kRowsPerWave;
  // The k8-interleaved copy (round 5; k_sample_gather's x_ki): after finish() the registers hold the row with its pad
  // masked; the planes of chunk c are recomputed from the SAME rounded products (same expression, no contraction) with the
  // norm factor finish() RETURNS -- a second evaluation of the sum of squares in the same kernel was contracted into fmas
  // differently from the first and moved a few rows' factor by one ulp (tests: the copy == the interleaved row-major planes).
  template <typename V4>
  __device__ static __forceinline__ void chunk_planes(const Regs &R, int c, float inv, V4 (&o)[3]) {
#pragma clang fp contract(off)
    float r[4] = {R.v[c].x * inv, R.v[c].y * inv, R.v[c].z * inv, R.v[c].w * inv};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o[pl][u] = (__bf16)r[u];
        r[u] -= (float)o[pl][u];
      }
  }
  __device__ static __forceinline__ void issue(Regs &R, const In *table, int64_t lr, int64_t stride, int F, int lane) {
    row_issue<NCH>(R, table, lr, stride, (F + 3) >> 2, lane);
  }
  __device__ static __forceinline__ float finish(Regs &R, int F, Out *dst, int64_t out_stride, int lane) {
    using bf16x4v = __attribute__((ext_vector_type(4))) __bf16;
    if (F & 3) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int j = 4 * (lane + kWave * c);
        if (j + 1 >= F && j < F) R.v[c].y = 0.f;
        if (j + 2 >= F && j < F) R.v[c].z = 0.f;
        if (j + 3 >= F && j < F) R.v[c].w = 0.f;
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      ss += R.v[c].x * R.v[c].x + R.v[c].y * R.v[c].y + R.v[c].z * R.v[c].z + R.v[c].w * R.v[c].w;
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    const int64_t plane = out_stride / 3;
    const int oq = (int)(plane >> 2);
    const bf16x4v z4 = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      if (q < oq) {
        // the planes are those of the ROUNDED product, the value the fp32 gather stores: no contraction of the
        // product with the residual subtraction below into an fma on the unrounded one
#pragma clang fp contract(off)
        float r[4] = {R.v[c].x * inv, R.v[c].y * inv, R.v[c].z * inv, R.v[c].w * inv};
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          bf16x4v o;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            o[u] = (__bf16)r[u];
            r[u] -= (float)o[u];
          }
#if CDML_GATHER_NT_STORE
          __builtin_nontemporal_store(o, reinterpret_cast<bf16x4v *>(dst + pl * plane + 4 * q));
#else
          *reinterpret_cast<bf16x4v *>(dst + pl * plane + 4 * q) = o;
#endif
        }
      }
    }
    for (int q = lane + kWave * NCH; q < oq; q += kWave)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x4v *>(dst + pl * plane + 4 * q) = z4;
    return inv;
  }
};

// RowF32H2: fp32 table -> the l2-normalised row as TWO fp16 planes hi | lo of x_hat * 2^14 (precision "f16x2":
// gemm_f16x2_256.hip reads them).  |x_hat| <= 1, so the scale is a constant of the format (cdml.h: CDML_F16X2_X_SCALE) and
// nothing saturates; out_stride = 2 planes of out_stride / 2 columns: 6 000 B written per 1500-d row where the three bf16
// planes are 9 000.
constexpr float kH2XScale = 16384.f;
template <int NCH>
struct RowF32H2 {
  using In = float;
  using Out = _Float16;
  using Regs = RowRegs<NCH>;
  static constexpr int kPerChunk = 4;
  static constexpr int kRows = kRowsPerWave;
  __device__ static __forceinline__ void issue(Regs &R, const In *table, int64_t lr, int64_t stride, int F, int lane) {
    row_issue<NCH>(R, table, lr, stride, (F + 3) >> 2, lane);
  }
  __device__ static __forceinline__ void finish(Regs &R, int F, Out *dst, int64_t out_stride, int lane) {
    using half4v = __attribute__((ext_vector_type(4))) _Float16;
    if (F & 3) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int j = 4 * (lane + kWave * c);
        if (j + 1 >= F && j < F) R.v[c].y = 0.f;
        if (j + 2 >= F && j < F) R.v[c].z = 0.f;
        if (j + 3 >= F && j < F) R.v[c].w = 0.f;
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      ss += R.v[c].x * R.v[c].x + R.v[c].y * R.v[c].y + R.v[c].z * R.v[c].z + R.v[c].w * R.v[c].w;
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    const int64_t plane = out_stride / 2;
    const int oq = (int)(plane >> 2);
    const half4v z4 = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      if (q < oq) {
        // the planes are those of the ROUNDED product (the value the fp32 gather stores) times 2^14, exactly
#pragma clang fp contract(off)
        float r[4] = {R.v[c].x * inv, R.v[c].y * inv, R.v[c].z * inv, R.v[c].w * inv};
        half4v hi, lo;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float sv = r[u] * kH2XScale;
          hi[u] = (_Float16)sv;
          lo[u] = (_Float16)(sv - (float)hi[u]);
        }
#if CDML_GATHER_NT_STORE
        __builtin_nontemporal_store(hi, reinterpret_cast<half4v *>(dst + 4 * q));
        __builtin_nontemporal_store(lo, reinterpret_cast<half4v *>(dst + plane + 4 * q));
#else
        *reinterpret_cast<half4v *>(dst + 4 * q) = hi;
        *reinterpret_cast<half4v *>(dst + plane + 4 * q) = lo;
#endif
      }
    }
    for (int q = lane + kWave * NCH; q < oq; q += kWave) {
      *reinterpret_cast<half4v *>(dst + 4 * q) = z4;
      *reinterpret_cast<half4v *>(dst + plane + 4 * q) = z4;
    }
  }
};

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8v = __attribute__((ext_vector_type(8))) __bf16;
template <int NCH>
struct RowF16 {
  using In = _Float16;
  using Out = __bf16;
  struct Regs { half8 v[NCH]; };
  static constexpr int kPerChunk = 8;
#ifndef CDML_GATHER_ROWS_PER_WAVE_F16
#define CDML_GATHER_ROWS_PER_WAVE_F16 4
#endif
  static constexpr int kRows = CDML_GATHER_ROWS_PER_WAVE_F16;   // 3-KB rows: four in flight per wave
  __device__ static __forceinline__ void issue(Regs &R, const In *table, int64_t lr, int64_t stride, int F, int lane) {
    const half8 *src = reinterpret_cast<const half8 *>(table + lr * stride);
    const int nq = (F + 7) >> 3;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      half8 x = {0, 0, 0, 0, 0, 0, 0, 0};
      if (q < nq) {
#if CDML_GATHER_NT
        x = __builtin_nontemporal_load(src + q);
#else
        x = src[q];
#endif
      }
      R.v[c] = x;
    }
  }
  __device__ static __forceinline__ void finish(Regs &R, int F, Out *dst, int64_t out_stride, int lane) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) ss = f16_chunk_sumsq(R.v[c], lane + kWave * c, F, ss);
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    bf16x8v *d = reinterpret_cast<bf16x8v *>(dst);
    const int oq = (int)(out_stride >> 3);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      if (q < oq) {
        bf16x8v o;
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = (__bf16)((float)R.v[c][u] * inv);
#if CDML_GATHER_NT_STORE
        __builtin_nontemporal_store(o, d + q);
#else
        d[q] = o;
#endif
      }
    }
    const bf16x8v z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = lane + kWave * NCH; q < oq; q += kWave) d[q] = z8;
  }
};

// KI (RowF32X3 only; round 5): the chunk's 8 rows = ONE row group of the k8-interleaved copy x_ki -- per step
// [3 planes][rows_per_step / 8][plane columns][8 rows] bf16, what the weight gradient's k-strided product reads with one
// 16-B LDS read per fragment (gemm_bf16_256.hip, KI).  After the row-major planes are stored the block walks the plane in
// slices of 256 columns: every wave drops its two rows' planes of the slice into 12 KiB of LDS as [plane][row][column]
// (8-B writes), then 192 threads each read one 4-column group of all 8 rows (8-B reads), transpose 8 x 4 in registers and
// store 4 x 16 B = 64 contiguous bytes of x_ki.
template <int MODE, typename ROW, bool KI = false>
__global__ void __launch_bounds__(kThreads)
k_sample_gather(const int32_t *__restrict__ pairs, int64_t n_pairs, uint64_t seed,
                uint64_t step_imm, const uint64_t *__restrict__ step_dev, int batch,
                int64_t slot0, int64_t batch_global, const typename ROW::In *__restrict__ table,
                int64_t n_rows, int64_t row_stride, int F, int32_t *__restrict__ idx_out,
                int32_t *__restrict__ shift_out, typename ROW::Out *__restrict__ x_out, int64_t out_stride,
                int n_steps, int64_t x_step_stride, int64_t idx_step_stride,
                int32_t *__restrict__ oob_flag, __bf16 *__restrict__ x_ki = nullptr, int64_t ki_step_stride = 0) {
  constexpr int RPT = (MODE == 0) ? 3 : 2;  // rows per triplet
  constexpr int kRPW = ROW::kRows, kCR = kRPW * kWavesPerBlock;   // rows per wave / per chunk
  __shared__ int32_t s_id[2][kCR];
  __shared__ __attribute__((aligned(16))) __bf16 s_il[KI ? 3 : 1][KI ? 8 : 1][KI ? 256 : 4];
  static_assert(!KI || kCR == 8, "the interleaved copy takes a chunk of 8 rows = one row group");
  const uint64_t step0 = step_imm + (step_dev ? *step_dev : 0);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t rows_per_step = (int64_t)RPT * batch;
  const int64_t total_rows = rows_per_step * n_steps;
  const int64_t n_chunks = (total_rows + kCR - 1) / kCR;

  // ids of chunk c -> LDS buffer b (threads 0..kCR-1), and to idx_out / shift_out
  auto stage_ids = [&](int64_t c, int b) {
    if (threadIdx.x < kCR) {
      const int64_t g = c * kCR + threadIdx.x;
      int32_t id = 0;
      if (g < total_rows) {
        const int s = (int)(g / rows_per_step);
        const int r = (int)(g - (int64_t)s * rows_per_step);
        const int i = r / RPT, k = r - i * RPT;
        const uint64_t step = step0 + (uint64_t)s;
        const uint64_t slot = (uint64_t)(slot0 + i);
        const uint64_t q = (step * (uint64_t)batch_global + slot) % (uint64_t)n_pairs;
        if (MODE == 0 && k == 2) {
          id = sample_uniform_negative(seed, step, (uint32_t)slot, pairs[2 * q], pairs[2 * q + 1], (uint32_t)n_rows);
        } else {
          id = pairs[2 * q + (k ? 1 : 0)];
          // a pair id outside the catalogue (the reference raises IndexError, inputs.py:158): the row
          // load below clamps it for memory safety, the flag tells the caller (off the load path)
          if (oob_flag && (id < 0 || id >= n_rows)) atomicOr(oob_flag, 1);
        }
        idx_out[(int64_t)s * idx_step_stride + r] = id;
        if (MODE == 1 && r == 0) shift_out[s] = sample_inbatch_shift(seed, step, batch);
      }
      s_id[b][threadIdx.x] = id;
    }
  };

  int64_t c = blockIdx.x;
  if (c >= n_chunks) return;
  stage_ids(c, 0);
  __syncthreads();
  int b = 0;
  for (; c < n_chunks; c += gridDim.x, b ^= 1) {
    const int64_t g0 = c * kCR + wave * kRPW;
    typename ROW::Regs R[kRPW];
#pragma unroll
    for (int u = 0; u < kRPW; ++u) {
      const int32_t id = __builtin_amdgcn_readfirstlane(s_id[b][wave * kRPW + u]);
      if (g0 + u < total_rows) ROW::issue(R[u], table, clamp_row(id, 0, n_rows, nullptr), row_stride, F, lane);
    }
    const int64_t cn = c + gridDim.x;
    if (cn < n_chunks) stage_ids(cn, b ^ 1);       // under the row loads in flight
    float inv[KI ? kRPW : 1];
#pragma unroll
    for (int u = 0; u < kRPW; ++u) {
      const int64_t g = g0 + u;
      if (g < total_rows) {
        const int64_t s = g / rows_per_step, r = g - s * rows_per_step;
        if constexpr (KI) inv[u] = ROW::finish(R[u], F, x_out + s * x_step_stride + r * out_stride, out_stride, lane);
        else ROW::finish(R[u], F, x_out + s * x_step_stride + r * out_stride, out_stride, lane);
      }
    }
    if constexpr (KI) {
      using bf16x4v = __attribute__((ext_vector_type(4))) __bf16;
      using u32x2v = __attribute__((ext_vector_type(2))) uint32_t;
      using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;
      // (host: rows_per_step % 8 == 0, so the chunk's 8 rows are whole, of one step, and one row group of it)
      const int64_t gc = c * kCR;
      const int64_t sidx = gc / rows_per_step;
      const int64_t kg = (gc - sidx * rows_per_step) >> 3, kgs = rows_per_step >> 3;
      const int plane = (int)(out_stride / 3);
      __bf16 *const kbase = x_ki + sidx * ki_step_stride;
      constexpr int kSlices = (int)(sizeof(R[0].v) / sizeof(R[0].v[0]));   // the row registers: NCH chunks of 256 columns
#pragma unroll
      for (int sl = 0; sl < kSlices; ++sl) {                            // chunk sl of the row registers = columns 256 sl .. + 255
        if (sl * 256 >= plane) break;                                  // (uniform)
#pragma unroll
        for (int u = 0; u < kRPW; ++u) {
          bf16x4v o[3];
          ROW::chunk_planes(R[u], sl, inv[u], o);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x4v *>(&s_il[pl][wave * kRPW + u][4 * lane]) = o[pl];
        }
        __syncthreads();
        if (threadIdx.x < 192) {
          const int pl = threadIdx.x >> 6, fg = threadIdx.x & 63;
          u32x2v in[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) in[j] = *reinterpret_cast<const u32x2v *>(&s_il[pl][j][4 * fg]);
          __bf16 *dstp = kbase + (((int64_t)pl * kgs + kg) * plane + sl * 256 + 4 * fg) * 8;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            u32x4v o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t a = in[2 * k][e >> 1], b = in[2 * k + 1][e >> 1];
              o[k] = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
            }
            *reinterpret_cast<u32x4v *>(dstp + e * 8) = o;
          }
        }
        __syncthreads();
      }
    }
    __syncthreads();                               // ids of the next chunk are staged; this buffer is free
  }
}

// ------------------------------------------------ row exchange: routing (multi-GPU) --
// Row-sharded catalogue, fixed-capacity all-to-all (cdml_amd/dist.py): every rank sends each
// peer exactly `cap` id slots and gets `cap` row slots back, so nothing on the step path depends
// on a host-side count (enqueue-only, hipGraph-capturable).  k_route_rows puts request r of this
// rank (global id ids[r], owner = id / rows_per_shard) into the owner's segment of the send
// buffer in ASCENDING r (deterministic: the owner of a trainable table sums duplicate rows in the
// order it was asked), pads the rest with -1 and records the slot for the way back.  One block:
// R is a few 10^4 ids and the kernel sits on the prefetch stream, off the critical path.
constexpr int kRouteThreads = 1024;
constexpr int kRouteMaxWorld = 64;

__global__ void __launch_bounds__(kRouteThreads)
k_route_rows(const int32_t *__restrict__ ids, int n, int64_t rows_per_shard, int world, int cap,
             int32_t *__restrict__ send_ids, int32_t *__restrict__ slot, int32_t *__restrict__ overflow) {
  __shared__ int s_cnt[kRouteMaxWorld];
  __shared__ int s_wave[kRouteThreads / kWave][kRouteMaxWorld];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  for (int i = tid; i < world * cap; i += kRouteThreads) send_ids[i] = -1;
  if (tid < world) s_cnt[tid] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += kRouteThreads) {
    const int r = base + tid;
    int id = 0, o = -1;
    if (r < n) {
      id = ids[r];
      const int64_t q = (int64_t)id / rows_per_shard;
      if (id < 0 || q >= world) { atomicOr(overflow, 2); o = -1; }     // not a catalogue row
      else o = (int)q;
    }
    int my_rank = 0;
    for (int ow = 0; ow < world; ++ow) {
      const unsigned long long m = __ballot(o == ow);
      if (o == ow) my_rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[wave][ow] = __popcll(m);
    }
    __syncthreads();
    if (o >= 0) {
      int pos = s_cnt[o] + my_rank;
      for (int w = 0; w < wave; ++w) pos += s_wave[w][o];
      if (pos < cap) {
        send_ids[o * cap + pos] = id;
        slot[r] = o * cap + pos;
      } else {
        slot[r] = -1;                                                  // the row is not fetched: fatal, flagged
        atomicOr(overflow, 1);
      }
    } else if (r < n) {
      slot[r] = -1;
    }
    __syncthreads();
    if (tid < world) {
      int c = s_cnt[tid];
      for (int w = 0; w < kRouteThreads / kWave; ++w) c += s_wave[w][tid];
      s_cnt[tid] = c;
    }
    __syncthreads();
  }
}

// dst[slot[r]] = src[r] for r < n (slot -1: dropped): the reverse trip of the exchange (row
// gradients to their owners).  One wave per row, 16-B lanes.
__global__ void __launch_bounds__(kThreads)
k_scatter_rows(const float *__restrict__ src, int64_t lds, const int32_t *__restrict__ slot, int n, int width,
               float *__restrict__ dst, int64_t ldd) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = width >> 2;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < n; r += gridDim.x * kWavesPerBlock) {
    const int t = slot[r];
    if (t < 0) continue;
    const float4 *a = reinterpret_cast<const float4 *>(src + (int64_t)r * lds);
    float4 *d = reinterpret_cast<float4 *>(dst + (int64_t)t * ldd);
    for (int q = lane; q < nq; q += kWave) d[q] = a[q];
  }
}

int grid_for(int64_t work_items, int per_block) {
  int64_t b = (work_items + per_block - 1) / per_block;
  const int64_t cap = (int64_t)kNumCU * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// (round 5: a grid with every block walking the same number of chunks -- 1 536 blocks x 3 chunks instead of 2 048 blocks x
// 2.25 -- was measured SLOWER for the fp16 gather, 0.600 against 0.675 of 8 TB/s: what counts is 8 resident blocks per CU,
// not an even chunk count)
}  // namespace
}  // namespace cdml

using namespace cdml;

extern "C" int cdml_fill_uniform_table(float *table, int64_t row0, int64_t n_rows, int feature_size,
                                       int64_t row_stride, uint64_t seed, cdml_stream_t stream) {
  CDML_REQUIRE(table && n_rows > 0 && feature_size > 0 && row0 >= 0, CDML_E_BADARG,
               "fill_uniform_table: bad argument");
  CDML_REQUIRE(row_stride >= feature_size && (row_stride & 3) == 0 && aligned16(table), CDML_E_ALIGN,
               "fill_uniform_table: row_stride must be >= F and a multiple of 4, base 16-B aligned");
  const int64_t total = n_rows * (row_stride >> 2);
  hipLaunchKernelGGL(k_fill_table, dim3(grid_for(total, kThreads)), dim3(kThreads), 0,
                     (hipStream_t)stream, table, row0, n_rows, feature_size, row_stride, seed);
  return check_launch("fill_uniform_table");
}

extern "C" int cdml_sample_uniform(const int32_t *pairs, int64_t n_pairs, int64_t n_rows,
                                   uint64_t seed, uint64_t step, const uint64_t *step_dev,
                                   int batch, int64_t slot0, int64_t batch_global,
                                   int32_t *idx_out, cdml_stream_t stream) {
  CDML_REQUIRE(pairs && idx_out && n_pairs > 0 && batch > 0 && slot0 >= 0, CDML_E_BADARG,
               "sample_uniform: bad argument");
  CDML_REQUIRE(n_rows >= 3 && n_rows <= 0x7FFFFFFFll, CDML_E_BADARG,
               "sample_uniform: n_rows must be in [3, 2^31)");
  CDML_REQUIRE(batch_global >= slot0 + batch, CDML_E_BADARG,
               "sample_uniform: batch_global < slot0 + batch");
  hipLaunchKernelGGL(k_sample_uniform, dim3((batch + kThreads - 1) / kThreads), dim3(kThreads), 0,
                     (hipStream_t)stream, pairs, n_pairs, (uint32_t)n_rows, seed, step, step_dev,
                     batch, slot0, batch_global, idx_out);
  return check_launch("sample_uniform");
}

extern "C" int cdml_sample_inbatch(const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                                   uint64_t step, const uint64_t *step_dev, int batch,
                                   int64_t slot0, int64_t batch_global, int32_t *rows_out,
                                   int32_t *shift_out, cdml_stream_t stream) {
  CDML_REQUIRE(pairs && rows_out && shift_out && n_pairs > 0 && slot0 >= 0, CDML_E_BADARG,
               "sample_inbatch: bad argument");
  CDML_REQUIRE(batch >= 2, CDML_E_BADARG, "sample_inbatch: in-batch negatives need batch >= 2");
  CDML_REQUIRE(batch_global >= slot0 + batch, CDML_E_BADARG,
               "sample_inbatch: batch_global < slot0 + batch");
  hipLaunchKernelGGL(k_sample_inbatch, dim3((batch + kThreads - 1) / kThreads), dim3(kThreads), 0,
                     (hipStream_t)stream, pairs, n_pairs, seed, step, step_dev, batch, slot0,
                     batch_global, rows_out, shift_out);
  return check_launch("sample_inbatch");
}

extern "C" int cdml_step_advance(uint64_t *step_dev, cdml_stream_t stream) {
  CDML_REQUIRE(step_dev, CDML_E_BADARG, "step_advance: null pointer");
  hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  return check_launch("step_advance");
}

static int check_gather_layout(const char *who, const float *table, int64_t row_stride, int F,
                               const float *x_out, int64_t out_stride) {
  CDML_REQUIRE(F > 0 && F <= 2048, CDML_E_UNSUPPORTED, "%s: feature size %d outside (0, 2048]", who, F);
  CDML_REQUIRE(row_stride >= F && (row_stride & 3) == 0 && out_stride >= F && (out_stride & 3) == 0,
               CDML_E_ALIGN, "%s: strides must be >= F and multiples of 4", who);
  CDML_REQUIRE(aligned16(table) && aligned16(x_out), CDML_E_ALIGN, "%s: bases must be 16-B aligned", who);
  return CDML_OK;
}

extern "C" int cdml_gather_rows(const float *table, int64_t row0, int64_t n_rows, int64_t row_stride,
                                const int32_t *idx, int n_idx, int F, int normalize, float *x_out,
                                int64_t out_stride, float *inv_norm_out, int32_t *oob_flag,
                                cdml_stream_t stream) {
  CDML_REQUIRE(table && idx && x_out && n_rows > 0 && n_idx > 0 && row0 >= 0, CDML_E_BADARG,
               "gather_rows: bad argument");
  int rc = check_gather_layout("gather_rows", table, row_stride, F, x_out, out_stride);
  if (rc) return rc;
  const int grid = grid_for(n_idx, kWavesPerBlock);
  const int nch = ((F + 3) / 4 + kWave - 1) / kWave;
#define CDML_LAUNCH_GATHER(N)                                                                   \
  hipLaunchKernelGGL(k_gather_rows<N>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, table, \
                     row0, n_rows, row_stride, idx, n_idx, F, normalize, x_out, out_stride,     \
                     inv_norm_out, oob_flag)
  if (nch <= 2) CDML_LAUNCH_GATHER(2);
  else if (nch <= 4) CDML_LAUNCH_GATHER(4);
  else if (nch <= 6) CDML_LAUNCH_GATHER(6);
  else CDML_LAUNCH_GATHER(8);
#undef CDML_LAUNCH_GATHER
  return check_launch("gather_rows");
}

// rows AS STORED (fp32, e.g. the receive buffer of the row exchange) -> request order AND the three bf16 planes of the split-fp32
// GEMMs: x_out_planes[r] = planes of src[idx[r]] (out_stride = 3 planes of out_stride / 3 >= F columns); flags bit 1: idx -1 =
// a request that found no slot -> NaN planes (else: left untouched).
extern "C" int cdml_gather_rows_x3(const float *src, int64_t n_rows, int64_t row_stride, const int32_t *idx, int n_idx, int F,
                                   int flags, uint16_t *x_out_planes, int64_t out_stride, int32_t *oob_flag,
                                   cdml_stream_t stream) {
  CDML_REQUIRE(src && idx && x_out_planes && n_rows > 0 && n_idx > 0, CDML_E_BADARG, "gather_rows_x3: bad argument");
  CDML_REQUIRE(F > 0 && F <= 2048, CDML_E_UNSUPPORTED, "gather_rows_x3: feature size %d outside (0, 2048]", F);
  CDML_REQUIRE(row_stride >= F && (row_stride & 3) == 0 && out_stride % 3 == 0 && out_stride / 3 >= F && ((out_stride / 3) & 3) == 0 &&
                   aligned16(src) && (reinterpret_cast<uintptr_t>(x_out_planes) & 7) == 0,
               CDML_E_ALIGN, "gather_rows_x3: out_stride must be 3 planes of >= F columns (multiples of 4), 16-B aligned rows");
  const int grid = grid_for(n_idx, kWavesPerBlock);
  const int nch = ((F + 3) / 4 + kWave - 1) / kWave;
#define CDML_LAUNCH_GRP(N)                                                                                          \
  hipLaunchKernelGGL(k_gather_rows_planes<N>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, src, n_rows, row_stride, \
                     idx, n_idx, F, flags, reinterpret_cast<__bf16 *>(x_out_planes), out_stride, oob_flag)
  if (nch <= 2) CDML_LAUNCH_GRP(2);
  else if (nch <= 4) CDML_LAUNCH_GRP(4);
  else if (nch <= 6) CDML_LAUNCH_GRP(6);
  else CDML_LAUNCH_GRP(8);
#undef CDML_LAUNCH_GRP
  return check_launch("gather_rows_x3");
}

extern "C" int cdml_sample_gather(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                                  uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                                  int64_t batch_global, const float *table, int64_t n_rows,
                                  int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                                  float *x_out, int64_t out_stride, int n_steps, int64_t x_step_stride,
                                  int64_t idx_step_stride, int32_t *oob_flag, cdml_stream_t stream) {
  CDML_REQUIRE(mode == 0 || mode == 1, CDML_E_BADARG, "sample_gather: mode must be 0 or 1");
  CDML_REQUIRE(n_steps >= 1 && n_steps <= 64, CDML_E_BADARG, "sample_gather: n_steps must be in [1, 64]");
  CDML_REQUIRE(n_steps == 1 || (x_step_stride >= (int64_t)batch * (mode == 0 ? 3 : 2) * out_stride &&
                                (x_step_stride & 3) == 0 && idx_step_stride >= (int64_t)batch * (mode == 0 ? 3 : 2)),
               CDML_E_BADARG, "sample_gather: per-step strides too small for the batch");
  CDML_REQUIRE(pairs && table && idx_out && x_out && n_pairs > 0 && slot0 >= 0, CDML_E_BADARG,
               "sample_gather: bad argument");
  CDML_REQUIRE(n_rows >= 3 && n_rows <= 0x7FFFFFFFll, CDML_E_BADARG,
               "sample_gather: n_rows must be in [3, 2^31)");
  CDML_REQUIRE(batch >= (mode == 1 ? 2 : 1), CDML_E_BADARG, "sample_gather: batch too small");
  CDML_REQUIRE(mode == 0 || shift_out, CDML_E_BADARG, "sample_gather: shift_out required in mode 1");
  CDML_REQUIRE(batch_global >= slot0 + batch, CDML_E_BADARG,
               "sample_gather: batch_global < slot0 + batch");
  int rc = check_gather_layout("sample_gather", table, row_stride, F, x_out, out_stride);
  if (rc) return rc;
  const int grid = grid_for((int64_t)batch * (mode == 0 ? 3 : 2) * n_steps, RowF32<6>::kRows * kWavesPerBlock);
  const int nch = ((F + 3) / 4 + kWave - 1) / kWave;
#define CDML_LAUNCH_SG(M, N)                                                                      \
  hipLaunchKernelGGL((k_sample_gather<M, RowF32<N>>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, \
                     pairs, n_pairs, seed, step, step_dev, batch, slot0, batch_global, table,     \
                     n_rows, row_stride, F, idx_out, shift_out, x_out, out_stride, n_steps,       \
                     x_step_stride, idx_step_stride, oob_flag)
  if (mode == 0) {
    if (nch <= 2) CDML_LAUNCH_SG(0, 2);
    else if (nch <= 6) CDML_LAUNCH_SG(0, 6);
    else CDML_LAUNCH_SG(0, 8);
  } else {
    if (nch <= 2) CDML_LAUNCH_SG(1, 2);
    else if (nch <= 6) CDML_LAUNCH_SG(1, 6);
    else CDML_LAUNCH_SG(1, 8);
  }
#undef CDML_LAUNCH_SG
  return check_launch("sample_gather");
}

// The fused sampler + gather writing each row as three bf16 planes (precision "f32x3"): x_out = bf16
// [rows][out_stride], out_stride = 3 planes of out_stride / 3 >= F columns each (a multiple of 4).
static int sample_gather_x3_impl(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                                 uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                                 int64_t batch_global, const float *table, int64_t n_rows,
                                 int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                                 uint16_t *x_out_planes, int64_t out_stride, int n_steps, int64_t x_step_stride,
                                 int64_t idx_step_stride, int32_t *oob_flag, uint16_t *x_ki, int64_t ki_step_stride,
                                 cdml_stream_t stream) {
  CDML_REQUIRE(mode == 0 || mode == 1, CDML_E_BADARG, "sample_gather_x3: mode must be 0 or 1");
  CDML_REQUIRE(n_steps >= 1 && n_steps <= 64, CDML_E_BADARG, "sample_gather_x3: n_steps must be in [1, 64]");
  const int rpt = mode == 0 ? 3 : 2;
  CDML_REQUIRE(n_steps == 1 || (x_step_stride >= (int64_t)batch * rpt * out_stride && (x_step_stride & 3) == 0 &&
                                idx_step_stride >= (int64_t)batch * rpt),
               CDML_E_BADARG, "sample_gather_x3: per-step strides too small for the batch");
  CDML_REQUIRE(pairs && table && idx_out && x_out_planes && n_pairs > 0 && slot0 >= 0, CDML_E_BADARG,
               "sample_gather_x3: bad argument");
  CDML_REQUIRE(n_rows >= 3 && n_rows <= 0x7FFFFFFFll, CDML_E_BADARG, "sample_gather_x3: n_rows must be in [3, 2^31)");
  CDML_REQUIRE(batch >= (mode == 1 ? 2 : 1), CDML_E_BADARG, "sample_gather_x3: batch too small");
  CDML_REQUIRE(mode == 0 || shift_out, CDML_E_BADARG, "sample_gather_x3: shift_out required in mode 1");
  CDML_REQUIRE(batch_global >= slot0 + batch, CDML_E_BADARG, "sample_gather_x3: batch_global < slot0 + batch");
  CDML_REQUIRE(F > 0 && F <= 2048, CDML_E_UNSUPPORTED, "sample_gather_x3: feature size %d outside (0, 2048]", F);
  CDML_REQUIRE(row_stride >= F && (row_stride & 3) == 0 && out_stride % 3 == 0 && out_stride / 3 >= F &&
                   ((out_stride / 3) & 3) == 0 && aligned16(table) && (reinterpret_cast<uintptr_t>(x_out_planes) & 7) == 0,
               CDML_E_ALIGN, "sample_gather_x3: out_stride must be 3 planes of >= F columns (multiples of 4)");
  const int grid = grid_for((int64_t)batch * rpt * n_steps, RowF32X3<6>::kRows * kWavesPerBlock);
  const int nch = ((F + 3) / 4 + kWave - 1) / kWave;
  const int plane = (int)(out_stride / 3);
  if (x_ki) {
    // the interleaved copy: [3][rows_per_step / 8][plane][8] per step; whole row groups per step, plane in 256-column slices
    // that the row registers cover (NCH x 256 columns)
    CDML_REQUIRE(((int64_t)batch * rpt) % 8 == 0 && plane % 256 == 0 && plane <= (nch <= 6 ? 6 : 8) * 256 && aligned16(x_ki) &&
                     (n_steps == 1 || ki_step_stride >= (int64_t)3 * batch * rpt * plane) && RowF32X3<6>::kRows * kWavesPerBlock == 8,
                 CDML_E_UNSUPPORTED, "sample_gather_x3k: needs 8 | rows per step, plane columns a multiple of 256 (<= %d), 16-B aligned x_ki",
                 (nch <= 6 ? 6 : 8) * 256);
  }
#define CDML_LAUNCH_SGX(M, N, KI)                                                                              \
  hipLaunchKernelGGL((k_sample_gather<M, RowF32X3<N>, KI>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, \
                     pairs, n_pairs, seed, step, step_dev, batch, slot0, batch_global, table,               \
                     n_rows, row_stride, F, idx_out, shift_out, reinterpret_cast<__bf16 *>(x_out_planes),   \
                     out_stride, n_steps, x_step_stride, idx_step_stride, oob_flag, reinterpret_cast<__bf16 *>(x_ki), ki_step_stride)
  if (x_ki) {
    if (mode == 0) {
      if (nch <= 6) CDML_LAUNCH_SGX(0, 6, true); else CDML_LAUNCH_SGX(0, 8, true);
    } else {
      if (nch <= 6) CDML_LAUNCH_SGX(1, 6, true); else CDML_LAUNCH_SGX(1, 8, true);
    }
  } else if (mode == 0) {
    if (nch <= 6) CDML_LAUNCH_SGX(0, 6, false); else CDML_LAUNCH_SGX(0, 8, false);
  } else {
    if (nch <= 6) CDML_LAUNCH_SGX(1, 6, false); else CDML_LAUNCH_SGX(1, 8, false);
  }
#undef CDML_LAUNCH_SGX
  return check_launch("sample_gather_x3");
}

extern "C" int cdml_sample_gather_x3(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                                     uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                                     int64_t batch_global, const float *table, int64_t n_rows,
                                     int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                                     uint16_t *x_out_planes, int64_t out_stride, int n_steps, int64_t x_step_stride,
                                     int64_t idx_step_stride, int32_t *oob_flag, cdml_stream_t stream) {
  return sample_gather_x3_impl(mode, pairs, n_pairs, seed, step, step_dev, batch, slot0, batch_global, table, n_rows, row_stride, F,
                               idx_out, shift_out, x_out_planes, out_stride, n_steps, x_step_stride, idx_step_stride, oob_flag,
                               nullptr, 0, stream);
}

// cdml_sample_gather_x3 that ALSO writes every step's rows k8-interleaved (x_ki: bf16 [3][rows per step / 8][out_stride / 3][8]
// per step, steps ki_step_stride elements apart): the operand layout of cdml_gemm_bf16x3_tnk
extern "C" int cdml_sample_gather_x3k(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                                      uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                                      int64_t batch_global, const float *table, int64_t n_rows,
                                      int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                                      uint16_t *x_out_planes, int64_t out_stride, int n_steps, int64_t x_step_stride,
                                      int64_t idx_step_stride, int32_t *oob_flag, uint16_t *x_ki, int64_t ki_step_stride,
                                      cdml_stream_t stream) {
  CDML_REQUIRE(x_ki, CDML_E_BADARG, "sample_gather_x3k: x_ki required (cdml_sample_gather_x3 without it)");
  return sample_gather_x3_impl(mode, pairs, n_pairs, seed, step, step_dev, batch, slot0, batch_global, table, n_rows, row_stride, F,
                               idx_out, shift_out, x_out_planes, out_stride, n_steps, x_step_stride, idx_step_stride, oob_flag,
                               x_ki, ki_step_stride, stream);
}

// The fused sampler + gather writing each row as the two fp16 planes of x_hat * 2^14 (precision "f16x2"): x_out = fp16
// [rows][out_stride], out_stride = 2 planes of out_stride / 2 >= F columns each (a multiple of 4).
extern "C" int cdml_sample_gather_h2(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                                     uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                                     int64_t batch_global, const float *table, int64_t n_rows,
                                     int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                                     uint16_t *x_out_planes, int64_t out_stride, int n_steps, int64_t x_step_stride,
                                     int64_t idx_step_stride, int32_t *oob_flag, cdml_stream_t stream) {
  CDML_REQUIRE(mode == 0 || mode == 1, CDML_E_BADARG, "sample_gather_h2: mode must be 0 or 1");
  CDML_REQUIRE(n_steps >= 1 && n_steps <= 64, CDML_E_BADARG, "sample_gather_h2: n_steps must be in [1, 64]");
  const int rpt = mode == 0 ? 3 : 2;
  CDML_REQUIRE(n_steps == 1 || (x_step_stride >= (int64_t)batch * rpt * out_stride && (x_step_stride & 3) == 0 &&
                                idx_step_stride >= (int64_t)batch * rpt),
               CDML_E_BADARG, "sample_gather_h2: per-step strides too small for the batch");
  CDML_REQUIRE(pairs && table && idx_out && x_out_planes && n_pairs > 0 && slot0 >= 0, CDML_E_BADARG,
               "sample_gather_h2: bad argument");
  CDML_REQUIRE(n_rows >= 3 && n_rows <= 0x7FFFFFFFll, CDML_E_BADARG, "sample_gather_h2: n_rows must be in [3, 2^31)");
  CDML_REQUIRE(batch >= (mode == 1 ? 2 : 1), CDML_E_BADARG, "sample_gather_h2: batch too small");
  CDML_REQUIRE(mode == 0 || shift_out, CDML_E_BADARG, "sample_gather_h2: shift_out required in mode 1");
  CDML_REQUIRE(batch_global >= slot0 + batch, CDML_E_BADARG, "sample_gather_h2: batch_global < slot0 + batch");
  CDML_REQUIRE(F > 0 && F <= 2048, CDML_E_UNSUPPORTED, "sample_gather_h2: feature size %d outside (0, 2048]", F);
  CDML_REQUIRE(row_stride >= F && (row_stride & 3) == 0 && out_stride % 2 == 0 && out_stride / 2 >= F &&
                   ((out_stride / 2) & 3) == 0 && aligned16(table) && (reinterpret_cast<uintptr_t>(x_out_planes) & 7) == 0,
               CDML_E_ALIGN, "sample_gather_h2: out_stride must be 2 planes of >= F columns (multiples of 4)");
  const int grid = grid_for((int64_t)batch * rpt * n_steps, RowF32H2<6>::kRows * kWavesPerBlock);
  const int nch = ((F + 3) / 4 + kWave - 1) / kWave;
#define CDML_LAUNCH_SGH2(M, N)                                                                             \
  hipLaunchKernelGGL((k_sample_gather<M, RowF32H2<N>>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, \
                     pairs, n_pairs, seed, step, step_dev, batch, slot0, batch_global, table,               \
                     n_rows, row_stride, F, idx_out, shift_out, reinterpret_cast<_Float16 *>(x_out_planes), \
                     out_stride, n_steps, x_step_stride, idx_step_stride, oob_flag)
  if (mode == 0) {
    if (nch <= 6) CDML_LAUNCH_SGH2(0, 6); else CDML_LAUNCH_SGH2(0, 8);
  } else {
    if (nch <= 6) CDML_LAUNCH_SGH2(1, 6); else CDML_LAUNCH_SGH2(1, 8);
  }
#undef CDML_LAUNCH_SGH2
  return check_launch("sample_gather_h2");
}

extern "C" int cdml_route_rows(const int32_t *ids, int n, int64_t rows_per_shard, int world, int capacity,
                               int32_t *send_ids, int32_t *slot_out, int32_t *overflow_flag,
                               cdml_stream_t stream) {
  CDML_REQUIRE(ids && send_ids && slot_out && overflow_flag && n > 0 && rows_per_shard > 0 && capacity > 0,
               CDML_E_BADARG, "route_rows: bad argument");
  CDML_REQUIRE(world >= 1 && world <= kRouteMaxWorld, CDML_E_UNSUPPORTED, "route_rows: world size %d outside [1, %d]",
               world, kRouteMaxWorld);
  hipLaunchKernelGGL(k_route_rows, dim3(1), dim3(kRouteThreads), 0, (hipStream_t)stream, ids, n, rows_per_shard,
                     world, capacity, send_ids, slot_out, overflow_flag);
  return check_launch("route_rows");
}

extern "C" int cdml_scatter_rows(const float *src, int64_t ld_src, const int32_t *slot, int n, int width,
                                 float *dst, int64_t ld_dst, cdml_stream_t stream) {
  CDML_REQUIRE(src && slot && dst && n > 0 && width > 0, CDML_E_BADARG, "scatter_rows: bad argument");
  CDML_REQUIRE((width & 3) == 0 && ld_src >= width && ld_dst >= width && (ld_src & 3) == 0 && (ld_dst & 3) == 0 &&
                   aligned16(src) && aligned16(dst),
               CDML_E_ALIGN, "scatter_rows: width and leading dimensions multiples of 4, 16-B aligned bases");
  hipLaunchKernelGGL(k_scatter_rows, dim3(grid_for(n, kWavesPerBlock)), dim3(kThreads), 0, (hipStream_t)stream, src,
                     ld_src, slot, n, width, dst, ld_dst);
  return check_launch("scatter_rows");
}

extern "C" int cdml_sample_gather_f16(int mode, const int32_t *pairs, int64_t n_pairs, uint64_t seed,
                                      uint64_t step, const uint64_t *step_dev, int batch, int64_t slot0,
                                      int64_t batch_global, const uint16_t *table, int64_t n_rows,
                                      int64_t row_stride, int F, int32_t *idx_out, int32_t *shift_out,
                                      uint16_t *x_out_bf16, int64_t out_stride, int n_steps,
                                      int64_t x_step_stride, int64_t idx_step_stride, int32_t *oob_flag,
                                      cdml_stream_t stream) {
  CDML_REQUIRE(mode == 0 || mode == 1, CDML_E_BADARG, "sample_gather_f16: mode must be 0 or 1");
  CDML_REQUIRE(pairs && table && idx_out && x_out_bf16 && n_pairs > 0 && slot0 >= 0, CDML_E_BADARG,
               "sample_gather_f16: bad argument");
  CDML_REQUIRE(n_rows >= 3 && n_rows <= 0x7FFFFFFFll, CDML_E_BADARG, "sample_gather_f16: n_rows must be in [3, 2^31)");
  CDML_REQUIRE(batch >= (mode == 1 ? 2 : 1), CDML_E_BADARG, "sample_gather_f16: batch too small");
  CDML_REQUIRE(mode == 0 || shift_out, CDML_E_BADARG, "sample_gather_f16: shift_out required in mode 1");
  CDML_REQUIRE(batch_global >= slot0 + batch, CDML_E_BADARG, "sample_gather_f16: batch_global < slot0 + batch");
  CDML_REQUIRE(n_steps >= 1 && n_steps <= 64, CDML_E_BADARG, "sample_gather_f16: n_steps must be in [1, 64]");
  const int rpt = mode == 0 ? 3 : 2;
  CDML_REQUIRE(n_steps == 1 || (x_step_stride >= (int64_t)batch * rpt * out_stride && (x_step_stride & 7) == 0 &&
                                idx_step_stride >= (int64_t)batch * rpt),
               CDML_E_BADARG, "sample_gather_f16: per-step strides too small for the batch");
  CDML_REQUIRE(F > 0 && F <= 4096, CDML_E_UNSUPPORTED, "sample_gather_f16: feature size outside (0, 4096]");
  CDML_REQUIRE(row_stride >= F && (row_stride & 7) == 0 && out_stride >= F && (out_stride & 7) == 0 &&
                   aligned16(table) && aligned16(x_out_bf16),
               CDML_E_ALIGN, "sample_gather_f16: strides must be >= F and multiples of 8, bases 16-B aligned");
  const int grid = grid_for((int64_t)batch * rpt * n_steps, RowF16<3>::kRows * kWavesPerBlock);
  const int nch = ((F + 7) / 8 + kWave - 1) / kWave;
#define CDML_LAUNCH_SGH(M, N)                                                                               \
  hipLaunchKernelGGL((k_sample_gather<M, RowF16<N>>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream,   \
                     pairs, n_pairs, seed, step, step_dev, batch, slot0, batch_global,                       \
                     reinterpret_cast<const _Float16 *>(table), n_rows, row_stride, F, idx_out, shift_out,   \
                     reinterpret_cast<__bf16 *>(x_out_bf16), out_stride, n_steps, x_step_stride, idx_step_stride, \
                     oob_flag)
  if (mode == 0) {
    if (nch <= 1) CDML_LAUNCH_SGH(0, 1); else if (nch <= 3) CDML_LAUNCH_SGH(0, 3); else CDML_LAUNCH_SGH(0, 8);
  } else {
    if (nch <= 1) CDML_LAUNCH_SGH(1, 1); else if (nch <= 3) CDML_LAUNCH_SGH(1, 3); else CDML_LAUNCH_SGH(1, 8);
  }
#undef CDML_LAUNCH_SGH
  return check_launch("sample_gather_f16");
}
