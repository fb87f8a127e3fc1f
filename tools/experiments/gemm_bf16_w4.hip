// EXPERIMENT, NOT PART OF THE LIBRARY (measured slower than the shipped ping-pong kernel, see the
// end of this header and DESIGN.md section 5; it was built into the library behind CDML_BF16_W4=1,
// parity-tested with tests/test_gpu_bf16.py::test_gemm_bf16_epilogues and timed with
// gemm_bf16_w4_bench.py next to this file).
//
// bf16 projection GEMM, one-wave-per-SIMD form: C[M][N] = epilogue(A[M][K] . B[N][K]^T),
// both operands k-contiguous bf16, fp32 accumulation (v_mfma_f32_32x32x16_bf16).
// Same layers and epilogues as gemm_bf16_256.hip (models.py:59-60, train.py:141 at BASELINE
// config 4's precision).
//
// Why a second structure: tools/micro/mfma_peak_bf16 (profiles/r02_mfma_peak_bf16.txt) shows
// that on live data the matrix pipe sustains 0.71-0.75 of the 2.5 PF peak with ONE MFMA-issuing
// wave per SIMD and 0.60-0.65 with two (the chip is power-limited and two waves arbitrating for
// one pipe lose issue slots on top).  The 256x256 ping-pong kernel keeps two waves per SIMD and
// four barriers per 64-deep K-tile; this kernel keeps one:
//   * block = 4 waves (one per SIMD), tile 256x256, wave tile 128x128 = 16 accumulators
//     (256 accumulator registers; 1 block per CU);
//   * K-tile 32 deep, FOUR LDS stages of 32 KiB (A 256x32 + B 256x32 bf16), filled by LDS-DMA
//     three K-tiles ahead (counted vmcnt, never drained in the loop), one barrier per K-tile;
//   * a wave reads its fragments (8 ds_read_b128 per 16-deep k-step) one k-step ahead into a
//     second register set, so the reads of step s+1 sit between the 16 MFMAs of step s and the
//     matrix pipe never waits for LDS;
//   * 64-B image rows, 16-B chunks XOR-swizzled by (row>>2)&3 on the DMA's source address and on
//     the read (conflict-free for ds_read_b128's 16-lane groups).
// Result (R = 24 576, MI355X): long-K product 0.40 of 2.5 PF against the ping-pong kernel's 0.51,
// FC1 0.33-0.37 against 0.38.  Timing ablations (CDML_W4_ABLATE): without the barrier 0.43, without
// the fragment reads 0.41, WITHOUT THE LDS-DMA 0.62 -- the loop is bound by the global -> LDS
// stream, and its 64-B image rows fetch every 128-B line of an operand row in two halves, two
// K-tiles apart (the ping-pong kernel's 128-B rows fetch whole lines).
#include "gemm_bf16.h"

// timing ablations (WRONG RESULTS; tools/gemm_bf16_w4_bench.py with CDML_LIB_PATH): bit 0 = no
// barrier / DMA wait in the loop, bit 1 = no DMA issue in the loop, bit 2 = no fragment reads
#ifndef CDML_W4_ABLATE
#define CDML_W4_ABLATE 0
#endif

namespace cdml {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kT = 256;
constexpr int kTile = 256, kBK = 32, kStages = 4;
constexpr int IMG = kTile * kBK * 2;          // 16 KiB: one operand's K-tile
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

#define CDML_W4_BARRIER()                      \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("s_barrier" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

template <int EPI>
__global__ void __launch_bounds__(kT, 1) k_gemm_bf16_w4(BArgs g) {
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

  // ---- DMA lane constants: piece p = wave*4 + i covers image rows p*16 .. p*16+15 (64-B rows) ----
  uint32_t va[4], vb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 16 + (lane >> 2);
    const int q = (lane & 3) ^ ((row >> 2) & 3);               // logical 16-B chunk held at this position
    va[i] = (uint32_t)(((int64_t)(m0 + row) * g.lda + k_begin + q * 8) * 2);
    vb[i] = (uint32_t)(((int64_t)(n0 + row) * g.ldb + k_begin + q * 8) * 2);
  }
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane(lds_off(smem) + wave * 4096);
  // one wave-instruction (1 KiB) of this wave's share of K-tile `tile`: j = 0..3 A pieces, 4..7 B pieces
  auto stage_piece = [&](int tile, int j) {
    const uint32_t kb = tile < n_kt ? (uint32_t)(tile * kBK * 2) : 0x80000000u;   // beyond the range: zeros
    const uint32_t base = lds_piece + (tile & (kStages - 1)) * STAGE;
    if (j < 4) dma(srd_a, va[j] + kb, base + j * 1024);
    else dma(srd_b, vb[j - 4] + kb, base + IMG + (j - 4) * 1024);
  };
  auto stage_tile = [&](int tile) {
#pragma unroll
    for (int j = 0; j < 8; ++j) stage_piece(tile, j);
  };

  // ---- fragment reads: lane (l31, h) holds k = 16*ks + 8*h .. +7 of image row l31 ----
  const int x = (l31 >> 2) & 3;
  const unsigned char *a_rd = smem + (wr * 128 + l31) * 64;
  const unsigned char *b_rd = smem + IMG + (wc * 128 + l31) * 64;
  const int sw0 = ((0 + h) ^ x) * 16, sw1 = ((2 + h) ^ x) * 16;
  // fragment j of k-step ks of K-tile `tile`: j = 0..3 the wave's four 32-row groups of A, 4..7 of B
  auto read_frag = [&](int tile, int ks, int j) {
    const int so = (tile & (kStages - 1)) * STAGE + (ks ? sw1 : sw0);
    return *reinterpret_cast<const bf16x8 *>((j < 4 ? a_rd : b_rd) + so + (j & 3) * 2048);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: three K-tiles in flight, the first one landed and visible
  stage_tile(0);
  stage_tile(1);
  stage_tile(2);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  CDML_W4_BARRIER();
  bf16x8 f0[8], f1[8];                           // [0..3] A fragments, [4..7] B fragments; two register sets
#pragma unroll
  for (int j = 0; j < 8; ++j) f0[j] = read_frag(0, 0, j);
  // One K-tile = two blocks of 16 MFMAs.  Everything else the wave has to issue rides in the
  // issue slots BETWEEN the MFMAs (a 32x32x16 MFMA holds the pipe for 32 cycles and the wave is
  // free to issue one LDS read or one DMA meanwhile): put after the MFMAs as one lump, the 8 reads
  // + 8 DMAs + barrier left the pipe idle a third of the time (0.37 of peak on the long-K shape).
  //   block 0 (k-step 0, set f0): the 8 fragment reads of k-step 1 -> f1 behind MFMAs 0..7, the 8
  //            DMA pieces of K-tile tile+3 behind MFMAs 8..15 (its stage was last read in K-tile
  //            tile-1; every wave retired those reads before the barrier inside that K-tile);
  //   then     this wave's pieces of K-tile tile+1 have landed (16 newer may fly), own reads retired,
  //            barrier: that now holds for every wave;
  //   block 1 (k-step 1, set f1): the 8 fragment reads of K-tile tile+1, k-step 0 -> f0.
  for (int tile = 0; tile < n_kt; ++tile) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int mi = j >> 2, ni = j & 3;
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0[mi], f0[4 + ni], acc[mi][ni], 0, 0, 0);
      if (j < 8) { if (!(CDML_W4_ABLATE & 4)) f1[j] = read_frag(tile, 1, j); }
      else if (!(CDML_W4_ABLATE & 2)) stage_piece(tile + 3, j - 8);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(CDML_W4_ABLATE & 1)) {
      asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
      CDML_W4_BARRIER();
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int mi = j >> 2, ni = j & 3;
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1[mi], f1[4 + ni], acc[mi][ni], 0, 0, 0);
      if (j < 8 && !(CDML_W4_ABLATE & 4)) f0[j] = read_frag(tile + 1, 0, j);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the out-of-range tail DMAs still write zeros
  CDML_W4_BARRIER();

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
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16_w4<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16_w4: cannot reserve %d B of LDS: %s", SMEM,
                                     hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((k_gemm_bf16_w4<EPI>), dim3(g.tiles_m * g.tiles_n, splits), dim3(kT), SMEM, s, g);
  return check_launch("gemm_bf16_w4");
}

}  // namespace

// same shape contract as the 256x256 ping-pong kernel (N % 256, K per split % 64, 2 GiB windows)
int launch_gemm_bf16_w4(const BArgs &g, int epilogue, int splits, hipStream_t s) {
  switch (epilogue) {
    case BE_BIAS_LRELU_BF16: return launch<BE_BIAS_LRELU_BF16>(g, splits, s);
    case BE_BIAS_LRELU_F32: return launch<BE_BIAS_LRELU_F32>(g, splits, s);
    case BE_MASK_BF16: return launch<BE_MASK_BF16>(g, splits, s);
    default: return launch<BE_F32>(g, splits, s);
  }
}

}  // namespace cdml
