// Shared host/device helpers for libcdml_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cdml.h"

namespace cdml {

// ---- per-thread error message ------------------------------------------------
char *err_buf();
int fail(int code, const char *fmt, ...);

#define CDML_REQUIRE(cond, code, ...)                  \
  do {                                                 \
    if (!(cond)) return ::cdml::fail((code), __VA_ARGS__); \
  } while (0)

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(CDML_E_HIP, "%s: %s", what, hipGetErrorString(e));
  return CDML_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int kWave = 64;
constexpr int kNumCU = 256;

// ---- device helpers -----------------------------------------------------------
// Sum of squares of one 16-B chunk (8 halfs) of an fp16 catalogue row, elements at or past F zeroed IN the chunk (the pad is
// never trusted): shared by the fused sampler + gather and k_gather_rows_f16, which must agree bit for bit.  Round 5: four
// v_dot2_f32_f16 (exact fp16 products, fp32 accumulate) instead of eight conversions + eight fmas, and the pad test only in
// the chunks that reach past F -- a timing ablation without the norm ran the fp16 gather 11 % faster: it is bound by its
// per-row instructions as much as by HBM (profiles/r05_gather_f16_norm_ablation.txt).
using cdml_half8 = __attribute__((ext_vector_type(8))) _Float16;
using cdml_half2 = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ float f16_chunk_sumsq(cdml_half8 &x, int q, int F, float ss) {
  if (8 * q + 8 > F) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (8 * q + u >= F) x[u] = 0;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const cdml_half2 p = {x[2 * e], x[2 * e + 1]};
    ss = __builtin_amdgcn_fdot2(p, p, ss, false);
  }
  return ss;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- "am I the last block of this grid to get here" ------------------------------------
// For a kernel whose last finisher does a piece of bookkeeping that needs no data from the other
// blocks (Adam: global_step += 1 once every block has read the old value).  Two-level tickets
// (groups of 32 blocks, then one top word) keep every word at <= 64 arrivals.  tickets:
// uint32[kTicketWords], zero before the first launch; the last arrivers reset them, so one buffer
// serves every later launch of the same kernel on the same stream.  No fences: nothing is handed
// over through memory (an agent-scope release per block is an L2 write-back per block: it cost
// the 2048-block Adam launch 100 us).
constexpr int kTicketWords = 80;          // 1 top + up to 64 groups (grids <= 2048 blocks) + slack
__device__ __forceinline__ bool grid_last_block(uint32_t *tickets) {
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned nb = gridDim.x, g = blockIdx.x >> 5, ngroups = (nb + 31) >> 5;
    const unsigned gsize = min(32u, nb - (g << 5));
    int last = 0;
    if (__hip_atomic_fetch_add(&tickets[1 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
      __hip_atomic_store(&tickets[1 + g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(&tickets[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) {
        __hip_atomic_store(&tickets[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

// Philox4x32-10 (Salmon et al., Random123); spec + known answers: oracle/sampler.py
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += W0;
    k1 += W1;
  }
  return c;
}

__device__ __forceinline__ uint32_t pick(const u32x4 &v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

constexpr int kMaxWords = 64;             // oracle/sampler.py MAX_WORDS
constexpr uint32_t kPurposeUniformNeg = 0;
constexpr uint32_t kPurposeInbatchShift = 1;
constexpr uint32_t kTableTag = 0x7AB1E000u;

// Word stream of (seed, step, slot, purpose); bounded() = Lemire multiply-shift.
struct WordStream {
  uint32_t slot, step_lo, step_hi, purpose, k0, k1;
  int j;
  u32x4 blk;
  __device__ WordStream(uint64_t seed, uint64_t step, uint32_t slot_, uint32_t purpose_)
      : slot(slot_), step_lo((uint32_t)step), step_hi((uint32_t)(step >> 32)),
        purpose(purpose_), k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), j(0) {}
  __device__ bool exhausted() const { return j >= kMaxWords; }
  __device__ uint32_t next() {
    if ((j & 3) == 0)
      blk = philox4x32_10(u32x4{slot, step_lo, step_hi, (purpose << 24) | (uint32_t)(j >> 2)}, k0, k1);
    uint32_t w = pick(blk, j & 3);
    ++j;
    return w;
  }
  // returns -1 when the stream is exhausted
  __device__ int64_t bounded(uint32_t n) {
    uint32_t thresh = (0u - n) % n;
    while (!exhausted()) {
      uint64_t m = (uint64_t)next() * n;
      if ((uint32_t)m >= thresh) return (int64_t)(m >> 32);
    }
    return -1;
  }
};

__device__ __forceinline__ int32_t sample_uniform_negative(uint64_t seed, uint64_t step,
                                                           uint32_t slot, int32_t a, int32_t p,
                                                           uint32_t n_rows) {
  WordStream ws(seed, step, slot, kPurposeUniformNeg);
  for (;;) {
    int64_t n = ws.bounded(n_rows);
    if (n < 0) break;
    if (n != a && n != p) return (int32_t)n;
  }
  for (uint32_t n = 0; n < n_rows; ++n)  // deterministic fallback
    if ((int32_t)n != a && (int32_t)n != p) return (int32_t)n;
  return 0;
}

__device__ __forceinline__ int32_t sample_inbatch_shift(uint64_t seed, uint64_t step, int batch) {
  WordStream ws(seed, step, 0u, kPurposeInbatchShift);
  int64_t r = ws.bounded((uint32_t)(batch - 1));
  return 1 + (int32_t)(r < 0 ? 0 : r);
}

// XCD-aware bijective remap (blocks b and b+8 share an XCD) followed by a
// grouped raster: 8 M-tiles x all N-tiles per group.
__device__ __forceinline__ int logical_block(int bid, int nwg) {
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// Row-major raster after the same XCD remap: consecutive blocks of an XCD walk ALONG an output
// row, so what the chip writes (and reads beside it) at any moment is long contiguous runs.
// For products with a short contraction, where the output stream is the cost.
__device__ __forceinline__ void tile_of_block_rowmajor(int bid, int nwg, int tiles_n, int &tm, int &tn) {
  const int logical = logical_block(bid, nwg);
  tm = logical / tiles_n;
  tn = logical - tm * tiles_n;
}

__device__ __forceinline__ void tile_of_block(int bid, int nwg, int tiles_m, int tiles_n, int &tm,
                                              int &tn) {
  const int logical = logical_block(bid, nwg);
  constexpr int GROUP_M = 8;
  const int width = GROUP_M * tiles_n;
  const int group = logical / width;
  const int first_m = group * GROUP_M;
  const int gsize = min(tiles_m - first_m, GROUP_M);
  const int in_group = logical - group * width;
  tm = first_m + in_group % gsize;
  tn = in_group / gsize;
}


}  // namespace cdml
