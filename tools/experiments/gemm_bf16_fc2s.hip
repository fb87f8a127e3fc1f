// STATUS (round 6): measured and NOT shipped -- profiles/r06_fc2_row_streaming_ab.txt.  Parity-green (against fp64, against the
// K-split form, ragged rows, two-K-tile contraction, run-to-run bit-identical) and no faster: alone 81 us against 88 (first
// form: A pieces on waves 0-3, B on 4-7) / 84.5 against 83 (this form), in config 4's step 99-102 us against 95-99.  Its K-tile
// is a serial chain on every wave -- wait, barrier, 14 fragment reads, barrier, 5-6 LDS-DMA pieces at 60-185 cycles each, 24
// MFMAs -- ~2 000 cycles for 384 of MFMA, in lock step across the block: what the ping-pong kernel's two row groups one
// barrier apart exist to avoid.  Kept here for the record (declare it in gemm_bf16.h and dispatch it from cdml_gemm_bf16_nt,
// epilogue 1, N = 256, to build it).
// The narrow forward layer at BASELINE config 4's precision (models.py:60: z = lrelu(h1 . W2 + b2), bf16 operands, fp32
// out) as a ROW-STREAMING kernel: C[M][256] = lrelu(A[M][K] . B[256][K]^T + bias), K-contiguous bf16 operands.
//
// Why a kernel of its own (round 6; VERDICT r5 #1c).  With one tile column the 256 x 256 ping-pong kernel has M / 256
// tiles -- 96 at config 4's 24 576 rows -- so rounds 2-5 split K in two to put 192 blocks on the chip, wrote two fp32
// slabs (50 MB) and combined them with bias + leaky-relu in a second launch: 72 + 13 us for a product whose floor is
// reading h1 once (252 MB: 40 us at the 6.3 TB/s the part's copy kernels reach; PMC: 4.5 TB/s, MFMA pipe 0.36 busy).
// It is not MFMA work (2 * 256 flop per 2 bytes of h1 = 256 flop/B against a machine ridge of ~310): it is a stream.
// Here:
//   * a block owns 96 rows (24 576 = 256 x 96: ONE round of the chip, every CU streaming) and all 256 columns, and walks
//     the whole contraction itself: no slabs, no combine launch, bias + leaky-relu in its epilogue;
//   * 8 waves = 2 row groups of 48 rows x 4 column strips of 64; per 64-deep K-tile a wave reads 6 A + 8 B fragments from
//     LDS (the 128-B-row images and XOR swizzle of gemm_bf16_256.hip) for 24 v_mfma_f32_16x16x32_bf16;
//   * operands by LDS-DMA with the two streams on DIFFERENT waves' counters: waves 0-1 issue only A pieces (h1: HBM,
//     12 KB per K-tile, SIX stages = five K-tiles in flight: what a 2-us HBM round trip needs at 22 GB/s per CU), waves 2-7
//     only B pieces (W2^T: 2.6 MB, L2-resident, two stages).  vmcnt retires in order per wave: with both streams on one
//     counter, waiting for the next B image would drain every older A load -- the deep A pipeline only exists because
//     the B loads are counted elsewhere;
//   * MFMA operands swapped (mfma(b, a): gemm_bf16_256.hip, round 6) so that a lane holds four consecutive columns of a
//     row: 16-B fp32 stores straight from the accumulators (the output is 25 MB against 252 MB read: its store pattern is
//     not what this kernel waits for);
//   * 160 KiB of LDS: 6 x 16 KiB A slots (96 of 128 rows used) + 2 x 32 KiB B slots.
// One pass over K in a fixed order: a row's result does not depend on M or on which block computed it.
#include "gemm_bf16.h"
#include <stdlib.h>

namespace cdml {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kT = 512;
constexpr int kRows = 96, kCols = 256, kTileK = 64;
constexpr int kASlot = 16384, kAStages = 6, kAhead = kAStages - 1;      // K-tiles of A in flight
constexpr int kBSlot = 32768, kBStages = 2;
constexpr int kSmem = kAStages * kASlot + kBStages * kBSlot;            // 160 KiB
// 44 pieces of 1 KiB per K-tile: A (12) on waves 0-1, six each; B (32) on waves 2-7, piece p on loader p % 6 (six or five
// each) -- an LDS-DMA piece costs its issuing wave 60-185 cycles, so the pieces per wave set the K-tile's length
constexpr int kAWaves = 2, kAPieces = 6, kBWaves = 6, kBPiecesMax = 6;

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
#define CDML_FC2S_BARRIER()                    \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("s_barrier" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

struct Fc2sArgs {
  const bf16 *A; int64_t lda;
  const bf16 *B; int64_t ldb;
  float *C; int64_t ldc;
  const float *bias;
  float alpha;
  int M, K;
};

__global__ void __launch_bounds__(kT, 1) k_fc2_stream_bf16(Fc2sArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, q16 = lane >> 4;
  const int m0 = blockIdx.x * kRows;
  const int n_kt = g.K / kTileK;
  const bool a_loader = wave < kAWaves;                 // waves 0-1: the A stream; waves 2-7: the B stream
  const int lw = a_loader ? wave : wave - kAWaves;      // index among the loaders of a stream
  const int n_pieces = a_loader ? kAPieces : (lw < 2 ? 6 : 5);     // B: 32 pieces dealt round-robin to six waves

  const i32x4 srd_a = make_srd(g.A, (int64_t)g.M * g.lda * 2);
  const i32x4 srd_b = make_srd(g.B, (int64_t)kCols * g.ldb * 2);
  // piece p = image rows 8 p .. 8 p + 7 (128 B each); lane l: row 8 p + (l >> 3), 16-B chunk (l & 7), source chunk swizzled
  uint32_t vo[kBPiecesMax];
#pragma unroll
  for (int i = 0; i < kBPiecesMax; ++i) {
    const int p = a_loader ? lw * kAPieces + i : lw + kBWaves * i;      // (B: p < 32 for i < n_pieces)
    const int r = p * 8 + (lane >> 3);
    const int sc = (lane & 7) ^ ((r >> 1) & 7);
    vo[i] = a_loader ? (uint32_t)(((int64_t)(m0 + r) * g.lda + sc * 8) * 2)     // rows past M: zeros (descriptor range)
                     : (uint32_t)(((int64_t)r * g.ldb + sc * 8) * 2);
  }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
  auto issue = [&](int kt) {                           // this wave's pieces of K-tile kt (kt < n_kt)
    const uint32_t kb = (uint32_t)kt * (kTileK * 2);
    if (a_loader) {
      const uint32_t base = lds0 + (uint32_t)(kt % kAStages) * kASlot + (uint32_t)(lw * kAPieces) * 1024;
#pragma unroll
      for (int i = 0; i < kAPieces; ++i) dma(srd_a, vo[i] + kb, base + i * 1024);
    } else {
      const uint32_t base = lds0 + kAStages * kASlot + (uint32_t)(kt % kBStages) * kBSlot + (uint32_t)lw * 1024;
#pragma unroll
      for (int i = 0; i < 5; ++i) dma(srd_b, vo[i] + kb, base + i * (kBWaves * 1024));
      if (n_pieces == 6) dma(srd_b, vo[5] + kb, base + 5 * (kBWaves * 1024));
    }
  };
  // this wave's pieces of K-tile kt have landed when at most `ahead` newer K-tiles of its stream are outstanding
  auto wait_own = [&](int ahead) {
    if (a_loader) {
      switch (ahead) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
      }
    } else {
      if (ahead == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (n_pieces == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    }
  };

  // fragment reads: lane (l15, q) holds k = 32 ks2 + 8 q .. + 7 of image row l15 (+ 16-row block)
  const int sw0 = ((q16) ^ ((l15 >> 1) & 7)) * 16, sw1 = ((4 + q16) ^ ((l15 >> 1) & 7)) * 16;
  const unsigned char *a_rd = smem + (grp * 48 + l15) * 128;
  const unsigned char *b_rd = smem + kAStages * kASlot + (wc * 64 + l15) * 128;

  f32x4 acc[3][4];
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: the A stream kAhead K-tiles ahead, the B stream one
  if (a_loader) {
    for (int s = 0; s < kAhead && s < n_kt; ++s) issue(s);
  } else {
    issue(0);
    if (n_kt > 1) issue(1);
  }
  for (int kt = 0; kt < n_kt; ++kt) {
    // K-tiles of this wave's stream issued beyond kt
    const int issued_a = min(kt + kAhead, n_kt) - 1 - kt;       // A: kt + 1 .. min(kt + kAhead, n_kt) - 1
    const int issued_b = min(kt + kBStages, n_kt) - 1 - kt;
    wait_own(a_loader ? issued_a : issued_b);
    CDML_FC2S_BARRIER();                                        // every wave's pieces of K-tile kt are in LDS
    bf16x8 fa[2][3], fb[2][4];
    const unsigned char *ai = a_rd + (kt % kAStages) * kASlot;
    const unsigned char *bi = b_rd + (kt % kBStages) * kBSlot;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      fb[0][cb] = *reinterpret_cast<const bf16x8 *>(bi + cb * 2048 + sw0);
      fb[1][cb] = *reinterpret_cast<const bf16x8 *>(bi + cb * 2048 + sw1);
    }
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) {
      fa[0][rb] = *reinterpret_cast<const bf16x8 *>(ai + rb * 2048 + sw0);
      fa[1][rb] = *reinterpret_cast<const bf16x8 *>(ai + rb * 2048 + sw1);
    }
    // the fragments are in registers BEFORE the barrier that frees their slots
    asm volatile("" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]));
    asm volatile("" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]), "+v"(fb[1][0]), "+v"(fb[1][1]),
                      "+v"(fb[1][2]), "+v"(fb[1][3]));
    CDML_FC2S_BARRIER();                                        // slots of K-tile kt (A: kt % 6, B: kt % 2) are free
    if (a_loader) {
      if (kt + kAhead < n_kt) issue(kt + kAhead);               // -> slot (kt - 1) % 6: read in the previous iteration
    } else {
      if (kt + kBStages < n_kt) issue(kt + kBStages);           // -> the slot just read
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 3; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)      // operands swapped: the lane holds ROW l15, columns 4 q .. 4 q + 3 of block (rb, cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks2][cb], fa[ks2][rb], acc[rb][cb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }

  // epilogue: bias + leaky-relu, 16-B fp32 stores
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int el15 = lane_e & 15, eq = lane_e >> 4;
  const int col = wc * 64 + 4 * eq;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(g.bias + col + cb * 16);
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) {
      const int row = m0 + grp * 48 + rb * 16 + el15;
      f32x4 v = acc[rb][cb] + b4;
      v.x = fmaxf(v.x, v.x * g.alpha); v.y = fmaxf(v.y, v.y * g.alpha);
      v.z = fmaxf(v.z, v.z * g.alpha); v.w = fmaxf(v.w, v.w * g.alpha);
      if (row < g.M) *reinterpret_cast<f32x4 *>(g.C + (int64_t)row * g.ldc + col + cb * 16) = v;
    }
  }
}

}  // namespace

// CDML_BF16_FC2_STREAM=0 (A/B timing, read per call): the K-split 256 x 256 form of rounds 2-5
bool gemm_bf16_fc2_stream_enabled() {
  const char *e = getenv("CDML_BF16_FC2_STREAM");
  return !e || atoi(e) != 0;
}

bool gemm_bf16_fc2_stream_usable(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc) {
  if (N != kCols || K % kTileK || K < 2 * kTileK || M < 1) return false;
  if ((lda & 7) || (ldb & 7) || (ldc & 3) || lda < K || ldb < K || ldc < N) return false;
  const int64_t lim = (int64_t)1 << 31;
  return ((int64_t)M + kRows) * lda * 2 < lim && (int64_t)N * ldb * 2 < lim;
}

int launch_gemm_bf16_fc2_stream(const bf16 *A, int64_t lda, const bf16 *B, int64_t ldb, int M, int K, float *C, int64_t ldc,
                                const float *bias, float alpha, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fc2_stream_bf16),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16 fc2 stream: cannot reserve %d B of LDS: %s", kSmem, hipGetErrorString(e));
    configured = true;
  }
  Fc2sArgs g{A, lda, B, ldb, C, ldc, bias, alpha, M, K};
  hipLaunchKernelGGL(k_fc2_stream_bf16, dim3((M + kRows - 1) / kRows), dim3(kT), kSmem, s, g);
  return check_launch("gemm_bf16 fc2 stream");
}

}  // namespace cdml
