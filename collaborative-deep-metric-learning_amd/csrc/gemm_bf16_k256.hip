// bf16 data gradient of the OUTPUT layer at its K = 256 contraction (train.py:141 through
// models.py:60-61): dz1[M][N] = (dz2[M][256] . W2[N][256]^T) * lrelu'(h1[M][N]), bf16 in, bf16 out.
//
// At K = 256 this product is not MFMA work: per output element it does 512 flops and moves 4 bytes
// (2 of mask in, 2 out) -- 124 flop/B against a machine ridge of ~310 -- so the bound is HBM
// (algorithmic bytes = 2 * M * N * 2 + M * 512 + N * 512).  The tiled GEMM kernels run it at 3 TB/s
// because a block loads, multiplies and stores one tile after the other with the whole LDS to itself.
// This kernel streams instead:
//   - a block owns a strip of 256 output columns; each of its 4 waves keeps the 64 x 256 slice of W2
//     it needs IN REGISTERS (32 fragments = 128 VGPRs, loaded once) and the block sweeps down M;
//   - rows arrive 32 at a time (16 KiB of dz2, LDS-DMA, double-buffered, one barrier per chunk); a wave
//     reads each A fragment once from LDS (XOR-swizzled 512-B rows, conflict-free) for two MFMAs;
//   - the 32 x 64 result goes through 8 KiB of wave-private LDS to get whole 128-B row segments per
//     store; the mask segment for chunk c is requested before chunk c's MFMAs;
//   - 64 KiB of LDS and <= 256 VGPRs: two blocks per CU, so one block's loads/stores run under the
//     other's MFMAs;
//   - blocks of one row segment are placed on one XCD (blockIdx % 8), so every dz2 row is fetched into
//     one L2 only.
// Results are bit-equal to the tiled kernels' mask epilogue (same fp32 accumulation order over k).
#include "gemm_bf16.h"

namespace cdml {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kT = 256;            // 4 waves
constexpr int kK = 256;            // the contraction this kernel is built for
constexpr int kKS = kK / 16;       // MFMA k-steps
constexpr int kStripN = 256;       // columns per block (64 per wave)
constexpr int kChunkM = 32;        // rows per chunk
constexpr int kABytes = kChunkM * kK * 2;                 // 16 KiB per A buffer
constexpr int kStageBytes = kChunkM * 64 * 4;             // 8 KiB per wave
constexpr int kSmem = 2 * kABytes + 4 * kStageBytes;      // 64 KiB
constexpr int kTargetBlocks = 512;                        // two per CU

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

struct K256Args {
  const bf16 *A; int64_t lda;      // [M][256]
  const bf16 *B; int64_t ldb;      // [N][256]
  bf16 *C; int64_t ldc;            // [M][N]
  const bf16 *aux; int64_t ldaux;  // [M][N] or null
  int aux_bits;                    // aux is the sign bitmask uint8 [M][ldaux bytes] (bit j of byte b = column 8b+j)
  float alpha;
  int M, N;
  int strips, segments, chunks_per_segment, n_chunks;
};

// BITS: aux is the sign bitmask (epilogue 5), a compile-time choice -- as a run-time branch beside the value mask it
// cost the kernel 25 % (registers for both forms, a select per element in the store loop).
template <bool BITS>
__global__ void __launch_bounds__(kT, 2) k_gemm_bf16_k256_mask(K256Args g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // block -> (column strip, row segment): all strips of a segment on one XCD when that divides evenly
  int strip, seg;
  if ((g.segments & 7) == 0) {
    const int i = blockIdx.x >> 3;
    strip = i % g.strips;
    seg = (blockIdx.x & 7) + 8 * (i / g.strips);
  } else {
    strip = blockIdx.x % g.strips;
    seg = blockIdx.x / g.strips;
  }
  const int c_begin = seg * g.chunks_per_segment;
  const int c_end = min(g.n_chunks, c_begin + g.chunks_per_segment);
  if (c_begin >= c_end) return;                        // whole block: no barrier is left behind
  const int ncol0 = strip * kStripN + wave * 64;       // this wave's first output column

  // ---- W2 slice in registers: fragment (ks, ni): lane (l31, h) holds k = 16 ks + 8 h .. +7 of column ni*32 + l31 ----
  bf16x8 bfrag[kKS][2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const bf16 *bp = g.B + (int64_t)(ncol0 + ni * 32 + l31) * g.ldb + h * 8;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) bfrag[ks][ni] = *reinterpret_cast<const bf16x8 *>(bp + ks * 16);
  }

  // ---- A chunks by LDS-DMA: piece p = 2 rows of 512 B; wave w issues pieces 4w .. 4w+3 ----
  const i32x4 srd_a = make_srd(g.A, (int64_t)g.M * g.lda * 2);
  uint32_t va[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 2 * (wave * 4 + i) + (lane >> 5);
    const int q = (lane & 31) ^ row;                   // logical 16-B chunk stored at this position (row < 32)
    va[i] = (uint32_t)(((int64_t)row * g.lda + q * 8) * 2);
  }
  const uint32_t lds_a = __builtin_amdgcn_readfirstlane(lds_off(smem) + wave * 4096);
  const uint32_t chunk_stride = (uint32_t)(kChunkM * g.lda * 2);
  auto stage_chunk = [&](int c) {                      // rows beyond M read as zeros (descriptor bound)
    const uint32_t coff = (uint32_t)c * chunk_stride;          // in the VGPR offset: the range check sees it
    const uint32_t base = lds_a + (c & 1) * kABytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(srd_a, va[i] + coff, base + i * 1024);
  };
  const unsigned char *a_rd = smem + l31 * 512;
  int a_sw[kKS];
#pragma unroll
  for (int ks = 0; ks < kKS; ++ks) a_sw[ks] = ((2 * ks + h) ^ l31) * 16;

  float *stage = reinterpret_cast<float *>(smem + 2 * kABytes + wave * kStageBytes);
  const int er = lane >> 3, ec = (lane & 7) * 8;       // epilogue: 8 rows x 64 columns per instruction
  const bool has_aux = g.aux != nullptr;

  stage_chunk(c_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int c = c_begin; c < c_end; ++c) {
    const int m0 = c * kChunkM;
    // this wave's pieces of chunk c have landed (waited for below, before the previous chunk's stores
    // went out) -> after the barrier the chunk is visible to all and nobody reads the other buffer
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < c_end) stage_chunk(c + 1);
    bf16x8 mk[BITS ? 1 : 4];
    unsigned mkb[BITS ? 4 : 1];
    if constexpr (!BITS) {
      if (has_aux) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          mk[p] = *reinterpret_cast<const bf16x8 *>(g.aux + (int64_t)min(m0 + p * 8 + er, g.M - 1) * g.ldaux + ncol0 + ec);
      }
    } else {                            // one byte = this lane's 8 columns
      const uint8_t *mb = reinterpret_cast<const uint8_t *>(g.aux);
#pragma unroll
      for (int p = 0; p < 4; ++p) mkb[p] = mb[(int64_t)min(m0 + p * 8 + er, g.M - 1) * g.ldaux + ((ncol0 + ec) >> 3)];
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const unsigned char *ab = a_rd + (c & 1) * kABytes;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8 *>(ab + a_sw[ks]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[ks][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[ks][1], acc1, 0, 0, 0);
    }
    // 32 x 64 fp32 through the wave's own 8 KiB
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      stage[row * 64 + l31] = acc0[r];
      stage[row * 64 + 32 + l31] = acc1[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // everything older than this chunk's stores: the mask (needed now), the next chunk's DMA (issued a
    // chunk of MFMAs ago) and the previous chunk's stores -- so the loop never waits on fresh stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = m0 + p * 8 + er;
      const f32x4 v0 = *reinterpret_cast<const f32x4 *>(stage + (p * 8 + er) * 64 + ec);
      const f32x4 v1 = *reinterpret_cast<const f32x4 *>(stage + (p * 8 + er) * 64 + ec + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (BITS) v[j] *= ((mkb[p] >> j) & 1u) ? 1.f : g.alpha;
        else if (has_aux) v[j] *= ((float)mk[p][j] > 0.f) ? 1.f : g.alpha;
        o[j] = (bf16)v[j];
      }
      // non-temporal: the result is read next by a different kernel on other CUs (5.2 vs 4.6 TB/s)
      if (row < g.M) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8 *>(g.C + (int64_t)row * g.ldc + ncol0 + ec));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// the mask / plain-bf16 epilogue (BE_MASK_BF16) at K == 256, N % 256 == 0, 16-B aligned rows
bool gemm_bf16_k256_usable(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, bool has_aux) {
  if (K != kK || N % kStripN || M < 1) return false;
  if ((lda & 7) || (ldb & 7) || (ldc & 7) || (has_aux && (ldaux & 7))) return false;   // (bitmask rows: ldaux in bytes)
  const int64_t lim = (int64_t)1 << 31;
  return ((int64_t)M + kChunkM) * lda * 2 < lim;
}

int launch_gemm_bf16_k256(const BArgs &b, hipStream_t s) {
  const bool bits = b.aux_bits && b.aux;
  static bool configured[2] = {false, false};
  if (!configured[bits]) {
    const void *fn = bits ? reinterpret_cast<const void *>(&k_gemm_bf16_k256_mask<true>)
                          : reinterpret_cast<const void *>(&k_gemm_bf16_k256_mask<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16_k256: cannot reserve %d B of LDS: %s", kSmem,
                                     hipGetErrorString(e));
    configured[bits] = true;
  }
  K256Args g{};
  g.A = b.A; g.lda = b.lda; g.B = b.B; g.ldb = b.ldb;
  g.C = static_cast<bf16 *>(b.C); g.ldc = b.ldc; g.aux = b.aux; g.ldaux = b.ldaux; g.alpha = b.alpha;
  g.aux_bits = b.aux_bits;
  g.M = b.M; g.N = b.N;
  g.strips = b.N / kStripN;
  g.n_chunks = (b.M + kChunkM - 1) / kChunkM;
  int segs = kTargetBlocks / g.strips;
  if (segs >= 8) segs &= ~7;                      // whole XCD groups
  if (segs < 1) segs = 1;
  if (segs > g.n_chunks) segs = g.n_chunks;
  g.chunks_per_segment = (g.n_chunks + segs - 1) / segs;
  g.segments = segs;
  if (bits) hipLaunchKernelGGL(k_gemm_bf16_k256_mask<true>, dim3(g.strips * g.segments), dim3(kT), kSmem, s, g);
  else hipLaunchKernelGGL(k_gemm_bf16_k256_mask<false>, dim3(g.strips * g.segments), dim3(kT), kSmem, s, g);
  return check_launch("gemm_bf16_k256");
}

}  // namespace cdml
