// EXPERIMENT, NOT PART OF THE LIBRARY (round 2): the fp32 counterpart of csrc/gemm_bf16_k256.hip, parity-green
// (tests of that day: fp64, vs the tiled kernel, strided, ragged M), measured in the step at 0.198 ms (two
// accumulator chains) / 0.210 ms (explicit fragment double-buffer) against the tiled kernel's 0.194-0.198 ms:
// at fp32 the product is MFMA-bound, not output-bound, and both arrangements keep the pipe ~0.70 busy.
// To build it again: copy into csrc/, declare gemm_f32_k256_usable / launch_gemm_f32_k256 in gemm_f32.h and
// call them from cdml_fc_bwd_data.
// fp32 data gradient of the OUTPUT layer at its 256-deep contraction (train.py:141 through
// models.py:60-61): dz1[M][N] = (dz2[M][256] . W2[N][256]^T) * lrelu'(h1[M][N]).
//
// The tiled kernel (gemm_f32.hip) runs this product at 0.68 of the fp32 MFMA peak: eight K-tiles per
// output tile leave its prologue and epilogue exposed.  The contraction is short enough to keep one
// operand in registers instead (the same arrangement as the bf16 kernel of gemm_bf16_k256.hip):
//   - a block owns a strip of 128 output columns; each of its 4 waves holds the 32 x 256 slice of W2 it
//     needs in 128 VGPRs (loaded once) and the block sweeps down M;
//   - rows arrive 32 at a time (32 KiB of dz2 by LDS-DMA, double-buffered, one barrier per chunk); a wave
//     feeds 128 v_mfma_f32_32x32x2_f32 per chunk from 32 ds_read_b128 (XOR-swizzled 1-KiB rows);
//   - k is permuted so that one 16-B read serves four k-steps: lane half h multiplies k = 128 h + ks at
//     k-step ks (both operands alike, so every k is still counted once);
//   - the accumulator layout already has 32 consecutive columns per row across the lanes, so the mask is
//     read and the result stored straight from registers in whole 128-B lines (no LDS staging);
//   - 64 KiB of LDS and <= 256 VGPRs: two blocks per CU, 512 blocks in all, each with the same number of
//     (strip, 32-row chunk) units in strip-major order -- one load of its W2 slice per block.
// Summation order over k differs from the tiled kernel's (fixed, so still bit-reproducible).
#include "gemm_f32.h"

namespace cdml {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kT = 256;
constexpr int kK = 256;
constexpr int kStripN = 128;        // 32 columns per wave
constexpr int kChunkM = 32;
constexpr int kABytes = kChunkM * kK * 4;       // 32 KiB per A buffer
constexpr int kSmem = 2 * kABytes;              // 64 KiB

__device__ __forceinline__ uint32_t lds_off(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void dma(i32x4 srd, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(srd) : "memory", "m0");
}
__device__ __forceinline__ i32x4 make_srd(const void *base, int64_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));
  r.z = __builtin_amdgcn_readfirstlane((int)(bytes > 0 ? bytes : 0));
  r.w = 0x00020000;
  return r;
}

struct FK256Args {
  const float *A; int64_t lda;      // [M][256]
  const float *B; int64_t ldb;      // [N][256]
  float *C; int64_t ldc;            // [M][N]
  const float *aux; int64_t ldaux;  // [M][N] or null
  float alpha;
  int M, N;
  int n_chunks, total_units, units_per_block;   // unit = (strip, chunk), strip-major
};

__global__ void __launch_bounds__(kT, 2) k_gemm_f32_k256_mask(FK256Args g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  int u = blockIdx.x * g.units_per_block;
  const int u_end = min(g.total_units, u + g.units_per_block);
  if (u >= u_end) return;            // whole block

  // ---- A chunks by LDS-DMA: one 1-KiB row per piece; wave w issues rows 8w .. 8w+7 ----
  const i32x4 srd_a = make_srd(g.A, (int64_t)g.M * g.lda * 4);
  uint32_t va[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wave * 8 + i;
    const int q = lane ^ row;                           // logical 16-B chunk stored at this position (row < 32)
    va[i] = (uint32_t)(((int64_t)row * g.lda + q * 4) * 4);
  }
  const uint32_t lds_a = __builtin_amdgcn_readfirstlane(lds_off(smem) + wave * 8192);
  const uint32_t chunk_stride = (uint32_t)(kChunkM * g.lda * 4);
  auto stage_chunk = [&](int c) {                       // rows beyond M read as zeros (descriptor bound)
    const uint32_t coff = (uint32_t)c * chunk_stride;
    const uint32_t base = lds_a + (c & 1) * kABytes;
#pragma unroll
    for (int i = 0; i < 8; ++i) dma(srd_a, va[i] + coff, base + i * 1024);
  };
  // fragment read j (k-steps 4j .. 4j+3): logical chunk 32 h + j of row l31, at position (32 h + j) ^ l31
  const unsigned char *a_rd = smem + l31 * 1024;
  const int hx = (32 * h) | l31;
  // this lane's part of every mask / result address: row 4 h of the chunk, column l31 of the wave's 32
  const int off_aux = (int)(4 * h * g.ldaux) + l31, off_c = (int)(4 * h * g.ldc) + l31;
  const bool has_aux = g.aux != nullptr;

  while (u < u_end) {                                   // at most two runs: a block's units may straddle two strips
    const int strip = u / g.n_chunks;
    const int c_begin = u - strip * g.n_chunks;
    const int c_end = min(g.n_chunks, c_begin + (u_end - u));
    u += c_end - c_begin;
    const int col0 = __builtin_amdgcn_readfirstlane(strip * kStripN + wave * 32);   // the wave's first output column

    // ---- W2 slice in registers: lane (l31, h) holds B[col0 + l31][128 h + ks], ks = 0..127 ----
    f32x4 bq[32];
    {
      const f32x4 *bp = reinterpret_cast<const f32x4 *>(g.B + (int64_t)(col0 + l31) * g.ldb + 128 * h);
#pragma unroll
      for (int j = 0; j < 32; ++j) bq[j] = bp[j];
    }
    __syncthreads();                                    // nobody still reads the previous run's chunks
    stage_chunk(c_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int c = c_begin; c < c_end; ++c) {
      const int m0 = c * kChunkM;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");           // chunk c visible to all; the other buffer is free
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < c_end) stage_chunk(c + 1);
      const bool full = m0 + kChunkM <= g.M;            // uniform
      // uniform row bases (scalar arithmetic), one 32-bit lane offset: accumulator register r is row
      // (r & 3) + 8 (r >> 2) + 4 h of the chunk
      const float *auxb = g.aux + (int64_t)m0 * g.ldaux + col0;
      float *cb = g.C + (int64_t)m0 * g.ldc + col0;
      float mk[16];
      if (has_aux) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          mk[r] = (full || m0 + rr + 4 * h < g.M) ? (auxb + (int64_t)rr * g.ldaux)[off_aux] : 1.f;
        }
      }
#ifndef CDML_FK256_VARIANT
#define CDML_FK256_VARIANT 1
#endif
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      const unsigned char *ab = a_rd + (c & 1) * kABytes;
#if CDML_FK256_VARIANT == 0
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(ab + ((hx ^ j) << 4));
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[j].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[j].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[j].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[j].w, acc1, 0, 0, 0);
      }
#else
      // two fragment registers sets, the read of group j + 2 issued behind the first MFMA of group j
      f32x4 a0 = *reinterpret_cast<const f32x4 *>(ab + ((hx ^ 0) << 4));
      f32x4 a1 = *reinterpret_cast<const f32x4 *>(ab + ((hx ^ 1) << 4));
#pragma unroll
      for (int j = 0; j < 32; j += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bq[j].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bq[j].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bq[j].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bq[j].w, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 2 < 32) a0 = *reinterpret_cast<const f32x4 *>(ab + ((hx ^ (j + 2)) << 4));
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bq[j + 1].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bq[j + 1].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bq[j + 1].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bq[j + 1].w, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 3 < 32) a1 = *reinterpret_cast<const f32x4 *>(ab + ((hx ^ (j + 3)) << 4));
      }
#endif
      // everything older than this chunk's stores (its mask, the next chunk's DMA, the previous chunk's stores)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        float v = acc0[r] + acc1[r];
        if (has_aux) v *= (mk[r] > 0.f) ? 1.f : g.alpha;
        if (full || m0 + rr + 4 * h < g.M) __builtin_nontemporal_store(v, (cb + (int64_t)rr * g.ldc) + off_c);
      }
    }
  }
}

}  // namespace

bool gemm_f32_k256_usable(int M, int N, int K, int64_t lda, int64_t ldb) {
  if (K != kK || N % kStripN || M < 1 || (lda & 3) || (ldb & 3)) return false;
  const int64_t lim = (int64_t)1 << 31;
  return ((int64_t)M + kChunkM) * lda * 4 < lim;
}

int launch_gemm_f32_k256(const GemmArgs &b, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32_k256_mask),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_f32_k256: cannot reserve %d B of LDS: %s", kSmem,
                                     hipGetErrorString(e));
    configured = true;
  }
  FK256Args g{};
  g.A = b.A; g.lda = b.lda; g.B = b.B; g.ldb = b.ldb; g.C = b.C; g.ldc = b.ldc;
  g.aux = b.aux; g.ldaux = b.ldaux; g.alpha = b.alpha; g.M = b.M; g.N = b.N;
  g.n_chunks = (b.M + kChunkM - 1) / kChunkM;
  g.total_units = (b.N / kStripN) * g.n_chunks;
  // two blocks per CU, every block the same number of (strip, chunk) units in strip-major order: one
  // load of the W2 slice per block (two where its units straddle a strip boundary)
  int blocks = 2 * kNumCU;
  g.units_per_block = (g.total_units + blocks - 1) / blocks;
  if (g.units_per_block < 4) g.units_per_block = 4;
  blocks = (g.total_units + g.units_per_block - 1) / g.units_per_block;
  hipLaunchKernelGGL(k_gemm_f32_k256_mask, dim3(blocks), dim3(kT), kSmem, s, g);
  return check_launch("gemm_f32_k256");
}

}  // namespace cdml
