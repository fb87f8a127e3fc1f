// bf16 projection GEMM, one-wave-per-SIMD form with WHOLE-LINE operand rows:
// C[M][N] = epilogue(A[M][K] . B[N][K]^T), both operands k-contiguous bf16, fp32 accumulation
// (v_mfma_f32_32x32x16_bf16).  Same layers and epilogues as gemm_bf16_256.hip.
//
// Second iteration of tools/experiments/gemm_bf16_w4.hip (one MFMA-issuing wave per SIMD, 16
// accumulators, every LDS read and DMA issue in the issue slots between the MFMAs).  That kernel's
// 32-deep K-tiles made 64-B image rows -- each 128-B line of an operand row fetched in two halves, two
// K-tiles apart -- and its loop was bound by the global -> LDS stream (0.40 of peak, 0.62 without the
// DMA).  Here the K-tile is 64 deep: 128-B image rows (8 rows per 1-KiB DMA piece, whole lines), two
// LDS stages of 64 KiB, the 16 pieces of K-tile T+1 spread over three of the four k-steps of K-tile T
// (5 + 5 behind k-steps 0 and 1, 6 behind k-step 3 of the K-tile before), k-step 2 left free for the
// last pieces to land, one barrier per K-tile.
#include "gemm_bf16.h"

namespace cdml {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kT = 256;
constexpr int kTile = 256, kBK = 64, kStages = 2;
constexpr int IMG = kTile * kBK * 2;          // 32 KiB: one operand's K-tile
constexpr int STAGE = 2 * IMG;                // A | B
constexpr int SMEM = kStages * STAGE;         // 128 KiB

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

#define CDML_W4B_BARRIER()                      \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("s_barrier" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

template <int EPI>
__global__ void __launch_bounds__(kT, 1) k_gemm_bf16_w4b(BArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  int tm, tn;
  if (g.K <= 512) tile_of_block_rowmajor(blockIdx.x, gridDim.x, g.tiles_n, tm, tn);
  else tile_of_block(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, tm, tn);
  const int m0 = tm * kTile, n0 = tn * kTile;
  const int split = blockIdx.y;
  const int k_begin = split * g.k_per_split;
  const int k_end = min(g.K, k_begin + g.k_per_split);
  const int n_kt = k_end > k_begin ? (k_end - k_begin) / kBK : 0;

  const i32x4 srd_a = make_srd(g.A, (int64_t)g.M * g.lda * 2);
  const i32x4 srd_b = make_srd(g.B, (int64_t)g.N * g.ldb * 2);

  // ---- DMA lane constants: piece p = wave*8 + i covers image rows p*8 .. p*8+7 (128-B rows) ----
  uint32_t va[8], vb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (wave * 8 + i) * 8 + (lane >> 3);
    const int q = (lane & 7) ^ ((row >> 1) & 7);               // logical 16-B chunk held at this position
    va[i] = (uint32_t)(((int64_t)(m0 + row) * g.lda + k_begin + q * 8) * 2);
    vb[i] = (uint32_t)(((int64_t)(n0 + row) * g.ldb + k_begin + q * 8) * 2);
  }
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane(lds_off(smem) + wave * 8192);
  // one wave-instruction (1 KiB) of this wave's share of K-tile `tile`: j = 0..7 A pieces, 8..15 B pieces
  auto stage_piece = [&](int tile, int j) {
    const uint32_t kb = tile < n_kt ? (uint32_t)(tile * kBK * 2) : 0x80000000u;   // beyond the range: zeros
    const uint32_t base = lds_piece + (tile & 1) * STAGE;
    if (j < 8) dma(srd_a, va[j] + kb, base + j * 1024);
    else dma(srd_b, vb[j - 8] + kb, base + IMG + (j - 8) * 1024);
  };

  // ---- fragment reads: lane (l31, h) holds k = 16*ks + 8*h .. +7 of image row l31 ----
  const int x = (l31 >> 1) & 7;
  const unsigned char *a_rd = smem + (wr * 128 + l31) * 128;
  const unsigned char *b_rd = smem + IMG + (wc * 128 + l31) * 128;
  int sw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sw[ks] = ((2 * ks + h) ^ x) * 16;
  // fragment j of k-step ks of K-tile `tile`: j = 0..3 the wave's four 32-row groups of A, 4..7 of B
  auto read_frag = [&](int tile, int ks, int j) {
    const int so = (tile & 1) * STAGE + sw[ks];
    return *reinterpret_cast<const bf16x8 *>((j < 4 ? a_rd : b_rd) + so + (j & 3) * 4096);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: K-tile 0 and the first six pieces of K-tile 1 in flight, K-tile 0 landed and visible
#pragma unroll
  for (int j = 0; j < 16; ++j) stage_piece(0, j);
#pragma unroll
  for (int j = 0; j < 6; ++j) stage_piece(1, j);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  CDML_W4B_BARRIER();
  bf16x8 f0[8], f1[8];                           // [0..3] A fragments, [4..7] B fragments; two register sets
#pragma unroll
  for (int j = 0; j < 8; ++j) f0[j] = read_frag(0, 0, j);
  // One K-tile = four k-steps of 16 MFMAs; the wave's other work rides in the issue slots between
  // the MFMAs: behind MFMAs 0..7 of a k-step the 8 fragment reads of the next k-step (other register
  // set), behind MFMAs 8..15 the DMA pieces:
  //   k-step 0: pieces 6..10 of K-tile T+1      k-step 1: pieces 11..15 of K-tile T+1
  //   k-step 2: none (the last pieces land), then vmcnt(0) + own reads retired + barrier: K-tile T+1
  //             is complete and visible, and nobody reads K-tile T's stage any more
  //   k-step 3: pieces 0..5 of K-tile T+2 (into K-tile T's stage), fragment reads from K-tile T+1
#ifndef CDML_W4B_SPREAD
#define CDML_W4B_SPREAD 1
#endif
#ifndef CDML_W4B_ABLATE   // timing ablations (WRONG RESULTS): bit 1 = no DMA issue in the loop, bit 2 = no fragment reads
#define CDML_W4B_ABLATE 0
#endif
#if CDML_W4B_ABLATE & 2
#define CDML_W4B_DMA_AT(j, DT, D0, DN)
#elif CDML_W4B_SPREAD    // the DN pieces of a k-step evenly over its 16 issue slots
#define CDML_W4B_DMA_AT(j, DT, D0, DN)                                                                \
  _Pragma("unroll") for (int i = 0; i < DN; ++i)                                                      \
    if ((i * 16) / (DN > 0 ? DN : 1) + 1 == j) stage_piece(DT, D0 + i);
#else                    // behind MFMAs 8..15 only (after the fragment reads)
#define CDML_W4B_DMA_AT(j, DT, D0, DN) if (j >= 8 && j - 8 < DN) stage_piece(DT, D0 + j - 8);
#endif
#define CDML_W4B_STEP(FA, FB, RT, RKS, DT, D0, DN)                                                    \
  _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                    \
    const int mi = j >> 2, ni = j & 3;                                                                \
    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[mi], FA[4 + ni], acc[mi][ni], 0, 0, 0);  \
    if (j < 8 && !(CDML_W4B_ABLATE & 4)) FB[j] = read_frag(RT, RKS, j);                                                   \
    CDML_W4B_DMA_AT(j, DT, D0, DN)                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
  for (int tile = 0; tile < n_kt; ++tile) {
    CDML_W4B_STEP(f0, f1, tile, 1, tile + 1, 6, 5)
    CDML_W4B_STEP(f1, f0, tile, 2, tile + 1, 11, 5)
    CDML_W4B_STEP(f0, f1, tile, 3, tile + 1, 0, 0)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    CDML_W4B_BARRIER();
    CDML_W4B_STEP(f1, f0, tile + 1, 0, tile + 2, 0, 6)
  }
#undef CDML_W4B_STEP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the out-of-range tail DMAs still write zeros
  CDML_W4B_BARRIER();

  // ---- epilogue: per wave, 32 x 128 strips through its private 32 KiB of LDS ----
  float *sC = reinterpret_cast<float *>(smem + wave * 32768);
  const int c4 = lane & 31;
  const int gcol = n0 + wc * 128 + c4 * 4;
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (EPI == BE_BIAS_LRELU_BF16 || EPI == BE_BIAS_LRELU_F32) bias4 = *reinterpret_cast<const f32x4 *>(g.bias + gcol);
  const bool has_aux = (EPI == BE_MASK_BF16) && g.aux != nullptr;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    float *strip = sC + (mi & 1) * 4096;                 // alternate halves: no wait for the readers
    const int row_base = m0 + wr * 128 + mi * 32;
    bf16x4 mk[16];
    if (EPI == BE_MASK_BF16 && has_aux) {                // the strip's 16 mask loads go out together
#pragma unroll
      for (int p = 0; p < 16; ++p)
        mk[p] = *reinterpret_cast<const bf16x4 *>(g.aux + (int64_t)min(row_base + p * 2 + h, g.M - 1) * g.ldaux + gcol);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        strip[row * 128 + ni * 32 + l31] = acc[mi][ni][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int lr = p * 2 + h;
      const int row = row_base + lr;
      f32x4 v = *reinterpret_cast<const f32x4 *>(strip + lr * 128 + c4 * 4);
      if (EPI == BE_BIAS_LRELU_BF16 || EPI == BE_BIAS_LRELU_F32) {
        v += bias4;
        v.x = fmaxf(v.x, v.x * g.alpha); v.y = fmaxf(v.y, v.y * g.alpha);
        v.z = fmaxf(v.z, v.z * g.alpha); v.w = fmaxf(v.w, v.w * g.alpha);
      } else if (EPI == BE_MASK_BF16) {
        if (has_aux) {
          const bf16x4 m = mk[p];
          v.x *= ((float)m.x > 0.f) ? 1.f : g.alpha; v.y *= ((float)m.y > 0.f) ? 1.f : g.alpha;
          v.z *= ((float)m.z > 0.f) ? 1.f : g.alpha; v.w *= ((float)m.w > 0.f) ? 1.f : g.alpha;
        }
      }
      if (row >= g.M) continue;
      if (EPI == BE_BIAS_LRELU_BF16 || EPI == BE_MASK_BF16) {
        bf16x4 o;
        o.x = (bf16)v.x; o.y = (bf16)v.y; o.z = (bf16)v.z; o.w = (bf16)v.w;
        *reinterpret_cast<bf16x4 *>(static_cast<bf16 *>(g.C) + (int64_t)row * g.ldc + gcol) = o;
      } else {
        float *C = static_cast<float *>(g.C) + (EPI == BE_F32 ? (int64_t)split * g.slab_stride : 0);
        *reinterpret_cast<f32x4 *>(C + (int64_t)row * g.ldc + gcol) = v;
      }
    }
  }
}

template <int EPI>
int launch(const BArgs &g, int splits, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16_w4b<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16_w4b: cannot reserve %d B of LDS: %s", SMEM,
                                     hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((k_gemm_bf16_w4b<EPI>), dim3(g.tiles_m * g.tiles_n, splits), dim3(kT), SMEM, s, g);
  return check_launch("gemm_bf16_w4b");
}

}  // namespace

// same shape contract as the 256x256 ping-pong kernel (N % 256, K per split % 64, 2 GiB windows)
int launch_gemm_bf16_w4b(const BArgs &g, int epilogue, int splits, hipStream_t s) {
  switch (epilogue) {
    case BE_BIAS_LRELU_BF16: return launch<BE_BIAS_LRELU_BF16>(g, splits, s);
    case BE_BIAS_LRELU_F32: return launch<BE_BIAS_LRELU_F32>(g, splits, s);
    case BE_MASK_BF16: return launch<BE_MASK_BF16>(g, splits, s);
    default: return launch<BE_F32>(g, splits, s);
  }
}

}  // namespace cdml
