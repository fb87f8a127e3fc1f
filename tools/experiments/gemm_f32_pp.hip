// fp32 projection GEMMs, ping-pong form (models.py:59-60 forward, train.py:141
// backward): the same exact-fp32 arithmetic as gemm_f32.hip (v_mfma_f32_32x32x2_f32,
// fmaf chains) on a 128x256x32 tile with one block of 8 waves per CU.
//
// STATUS: opt-in experiment (CDML_F32_TILE=256), parity-tested, NOT the default: measured on
// MI355X it ties gemm_f32.hip on the forward product and loses on the weight gradient
// (DESIGN.md section 5).  Kept because the structure is what pays on the bf16 path
// (gemm_bf16_256.hip) and the measurement is worth having.
//
// Why a second structure: in gemm_f32.hip two independent blocks share a CU and overlap
// their LDS reads / barriers / DMA waits with each other's MFMAs only statistically
// (MFMA pipe busy 87 % in the step).  Here the overlap is arranged: the block's two row
// groups (4 waves each, one of each group per SIMD) run one barrier apart, a phase is
//     [fragment reads + DMA issue + counted wait]  barrier  [32 MFMAs]  barrier
// and while one group's waves are in their MFMA part the other group's are in their read
// part.  All operands of a phase are in registers before its MFMAs start, so the MFMA
// part is 32 back-to-back matrix instructions (2048 cycles) with nothing to wait for.
//
// Tile: 128 rows (row group = 64) x 256 columns (two halves of 128; strip wc owns columns
// wc*32..+31 of each half), wave tile 64x64 = 2x2 accumulators.  A K-tile (32) is two
// phases: phase A multiplies by the first column half (reads the A image and B-h0),
// phase B by the second (reads B-h1, A stays in registers).  Three 16-KiB images per
// K-tile, two buffers = 96 KiB LDS, filled by LDS-DMA (buffer_load_dwordx4 ... lds; rows
// outside the descriptor read zeros: ragged M, ragged contraction, and the dummy tiles
// past the end of the split).
//   k-contiguous operand image: [128 rows][32 k] = 128-B rows, chunks swizzled by
//     (row>>1)&7 on the DMA source, read with ds_read_b128 (4 k per lane feed 4 MFMAs,
//     the same k permutation on both operands);
//   k-strided operand image:    [32 k][128 columns] = 512-B rows, read with ds_read_b32.
// DMA schedule (tile T in buffer T&1), never drained in the loop:
//   phase A(T) issues B-h1(T+1)            (other buffer; last read in phase B(T-1))
//   phase B(T) issues A(T+2), B-h0(T+2)    (this buffer; read in phase A(T), and fragment
//                                           reads are retired before every barrier)
//   each phase then waits vmcnt(6): phase A for B-h1(T) (read in phase B(T)), phase B
//   for A(T+1), B-h0(T+1) (read in phase A(T+1)); the read comes two barriers after the
//   wait for either group.
#include "gemm_f32.h"

namespace cdml {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kT = 512;
constexpr int kBM = 128, kBN = 256, kBK = 32;
constexpr int IMG = 16384;
constexpr int BUF = 3 * IMG;     // A | B-h0 | B-h1
constexpr int SMEM = 2 * BUF;    // 96 KiB

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

#ifndef CDML_PP_SETPRIO
#define CDML_PP_SETPRIO 1        // priority of a wave while it issues its MFMA part
#endif
#define PP_BARRIER()                           \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("s_barrier" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

// FORM 0 = NN, 1 = NT, 2 = TN (see gemm_f32.h)
template <int FORM, int EPI>
__global__ void __launch_bounds__(kT, 1) k_gemm_f32_pp(GemmArgs g) {
  constexpr bool AKC = FORM != 2;   // A rows are k-contiguous
  constexpr bool BKC = FORM == 1;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, h = lane >> 5;

  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, tm, tn);
  const int m0 = tm * kBM, n0 = tn * kBN;
  const int split = blockIdx.y;
  const int k_begin = split * g.k_per_split;
  const int k_end = min(g.K, k_begin + g.k_per_split);
  const int n_ktiles = k_end > k_begin ? (((k_end - k_begin + kBK - 1) / kBK + 1) & ~1) : 0;   // even

  const i32x4 srd_a = make_srd(g.A, (int64_t)(AKC ? g.M : k_end) * g.lda * 4);
  const i32x4 srd_b = make_srd(g.B, (int64_t)(BKC ? g.N : k_end) * g.ldb * 4);

  // ---- DMA lane constants (two 1-KiB pieces per wave per image) ----
  uint32_t va[2], vb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = wave * 2 + i;
    if constexpr (AKC) {
      const int r = pc * 8 + (lane >> 3);
      const int sc = (lane & 7) ^ ((r >> 1) & 7);
      va[i] = (uint32_t)(((int64_t)(m0 + r) * g.lda + sc * 4) * 4);
    } else {
      const int r = pc * 2 + (lane >> 5);
      va[i] = (uint32_t)(((int64_t)r * g.lda + m0 + (lane & 31) * 4) * 4);
    }
    if constexpr (BKC) {
      const int r = pc * 8 + (lane >> 3);
      const int sc = (lane & 7) ^ ((r >> 1) & 7);
      vb[i] = (uint32_t)(((int64_t)(n0 + r) * g.ldb + sc * 4) * 4);
    } else {
      const int r = pc * 2 + (lane >> 5);
      vb[i] = (uint32_t)(((int64_t)r * g.ldb + n0 + (lane & 31) * 4) * 4);
    }
  }
  const uint32_t d_b = BKC ? (uint32_t)(128 * g.ldb * 4) : 512u;      // second column half
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane(lds_off(smem) + wave * 2048);

  // img 0 = A, 1 = B-h0, 2 = B-h1
  auto stage = [&](int img, int tile, int buf) {
    const int64_t kk = (int64_t)k_begin + (int64_t)tile * kBK;
    const int64_t k_elems = img == 0 ? (AKC ? kk : kk * g.lda) : (BKC ? kk : kk * g.ldb);
    const uint32_t kb = tile < n_ktiles ? (uint32_t)(k_elems * 4) : 0x80000000u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t voff = (img == 0 ? va[i] : vb[i] + (img == 2 ? d_b : 0u)) + kb;
      dma(img == 0 ? srd_a : srd_b, voff, lds_piece + i * 1024 + buf * BUF + img * IMG);
    }
  };

  // ---- fragment reads: lane (l31, h), step (j, e) multiplies k = 8j + 4h + e ----
  const int x = (l31 >> 1) & 7;
  int sw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) sw[j] = ((2 * j + h) ^ x) * 16;
  const unsigned char *a_kc = smem + (grp * 64 + l31) * 128;                 // + mi*4096 + sw[j]
  const unsigned char *a_ks = smem + h * 2048 + (grp * 64 + l31) * 4;        // + (8j+e)*512 + mi*128
  const unsigned char *b_kc = smem + IMG + (wc * 32 + l31) * 128;            // + ni*IMG + sw[j]
  const unsigned char *b_ks = smem + IMG + h * 2048 + (wc * 32 + l31) * 4;   // + ni*IMG + (8j+e)*512
  auto read_a = [&](int buf, int mi, int j) {
    if constexpr (AKC) {
      return *reinterpret_cast<const f32x4 *>(a_kc + buf * BUF + mi * 4096 + sw[j]);
    } else {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = *reinterpret_cast<const float *>(a_ks + buf * BUF + (8 * j + e) * 512 + mi * 128);
      return v;
    }
  };
  auto read_b = [&](int buf, int ni, int j) {
    if constexpr (BKC) {
      return *reinterpret_cast<const f32x4 *>(b_kc + buf * BUF + ni * IMG + sw[j]);
    } else {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = *reinterpret_cast<const float *>(b_ks + buf * BUF + ni * IMG + (8 * j + e) * 512);
      return v;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 fa[2][4], fb[4];

  // bias gradient (weight-gradient form): column sums of B, the K-tiles of a B tile shared
  // between the tiles_m blocks x 2 row groups that read it (tile t -> owner t % (2*tiles_m))
  float cs[2] = {0.f, 0.f};
  const bool cs_on = (EPI == EPI_SLAB_COLSUM) && g.colsum != nullptr;
  const int cs_owner = 2 * tm + grp, cs_period = 2 * g.tiles_m;

  auto mfma_phase = [&](int ni) {
    __builtin_amdgcn_s_setprio(CDML_PP_SETPRIO);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][j][e], fb[j][e], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto colsum_phase = [&](int ni, int tile) {
    if (EPI == EPI_SLAB_COLSUM && cs_on && (tile % cs_period) == cs_owner) {
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[ni] += (fb[j][0] + fb[j][1]) + (fb[j][2] + fb[j][3]);
    }
  };

  // The fragments as in/out operands of an empty asm: hipcc must have every one of them in its
  // registers here, i.e. it waits for the LDS reads BEFORE this point and cannot sink them
  // (or their waits) below the barrier that follows.  The hazard analysis above relies on it:
  // an image may be refilled one barrier after it was read.
  auto pin_a = [&]() {
    asm volatile("" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]),
                      "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]));
  };
  auto pin_b = [&]() { asm volatile("" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3])); };

  auto do_tile = [&](const int buf, const int tile) {
    // ---- phase A: first column half ----
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = read_b(buf, 0, j);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[mi][j] = read_a(buf, mi, j);
    stage(2, tile + 1, buf ^ 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    pin_b();
    pin_a();
    colsum_phase(0, tile);
    PP_BARRIER();
    mfma_phase(0);
    PP_BARRIER();
    // ---- phase B: second column half (A stays in registers) ----
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = read_b(buf, 1, j);
    stage(0, tile + 2, buf);
    stage(1, tile + 2, buf);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    pin_b();
    colsum_phase(1, tile);
    PP_BARRIER();
    mfma_phase(1);
    PP_BARRIER();
  };

  // prologue: the steady state at phase A of tile 0
  stage(0, 0, 0); stage(1, 0, 0);
  stage(2, 0, 0);
  stage(0, 1, 1); stage(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  PP_BARRIER();
  if (grp == 1) PP_BARRIER();                            // group 1 runs one barrier behind
  for (int tile = 0; tile < n_ktiles; tile += 2) {
    do_tile(0, tile);
    do_tile(1, tile + 1);
  }
  if (grp == 0) PP_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the dummy tail DMAs still write zeros
  PP_BARRIER();

  if (EPI == EPI_SLAB_COLSUM && cs_on) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const float v = cs[ni] + __shfl_xor(cs[ni], 32, 64);
      if (h == 0)
        g.colsum[(int64_t)((split * g.tiles_m + tm) * 2 + grp) * g.N + n0 + ni * 128 + wc * 32 + l31] = v;
    }
  }

  // ---- epilogue: 32x64 strips through the wave's private 12 KiB of LDS, 128-B row segments out ----
  float *strip = reinterpret_cast<float *>(smem + wave * 12288);
  const int c4 = lane & 15;
  const int gcol = n0 + (c4 >> 3) * 128 + wc * 32 + (c4 & 7) * 4;
  auto out_row = [&](int mi, int p) { return m0 + grp * 64 + mi * 32 + p * 4 + (lane >> 4); };
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (EPI == EPI_BIAS_LRELU) bias4 = *reinterpret_cast<const f32x4 *>(g.bias + gcol);
  const bool has_aux = (EPI == EPI_LRELU_MASK) && g.aux != nullptr;
  f32x4 mk[2][8];
  if (EPI == EPI_LRELU_MASK && has_aux) {                // all mask loads in flight together
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int p = 0; p < 8; ++p)
        mk[mi][p] = *reinterpret_cast<const f32x4 *>(g.aux + (int64_t)min(out_row(mi, p), g.M - 1) * g.ldaux + gcol);
  }
  float *C = g.C + (EPI == EPI_SLAB_COLSUM ? (int64_t)split * g.slab_stride : 0);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        strip[row * 64 + ni * 32 + l31] = acc[mi][ni][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int lr = p * 4 + (lane >> 4);
      const int row = out_row(mi, p);
      f32x4 v = *reinterpret_cast<const f32x4 *>(strip + lr * 64 + c4 * 4);
      if (EPI == EPI_BIAS_LRELU) {
        v += bias4;
        v.x = fmaxf(v.x, v.x * g.alpha); v.y = fmaxf(v.y, v.y * g.alpha);
        v.z = fmaxf(v.z, v.z * g.alpha); v.w = fmaxf(v.w, v.w * g.alpha);
      } else if (EPI == EPI_LRELU_MASK) {
        if (has_aux) {
          const f32x4 m = mk[mi][p];
          v.x *= (m.x > 0.f) ? 1.f : g.alpha; v.y *= (m.y > 0.f) ? 1.f : g.alpha;
          v.z *= (m.z > 0.f) ? 1.f : g.alpha; v.w *= (m.w > 0.f) ? 1.f : g.alpha;
        }
      }
      if (row < g.M) *reinterpret_cast<f32x4 *>(C + (int64_t)row * g.ldc + gcol) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int FORM, int EPI>
int launch(const GemmArgs &g, int splits, hipStream_t s) {
  static bool configured = false;   // raising the dynamic-LDS limit is idempotent
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_f32_pp<FORM, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess)
      return fail(CDML_E_HIP, "gemm_f32_pp: cannot reserve %d B of LDS: %s", SMEM, hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((k_gemm_f32_pp<FORM, EPI>), dim3(g.tiles_m * g.tiles_n, splits), dim3(kT), SMEM, s, g);
  return check_launch("gemm_f32_pp");
}

}  // namespace

bool gemm_f32_pp_usable(int form, int M, int N, int K, int64_t lda, int64_t ldb) {
  if (M < 1 || N % kBN || K < 1) return false;
  const int64_t lim = (int64_t)1 << 31;
  if (form == 2) {                                       // A[K][M], B[K][N]: any K (rows read as zeros past it)
    if (M % kBM) return false;
    return ((int64_t)K + 2 * kBK) * lda * 4 < lim && ((int64_t)K + 2 * kBK) * ldb * 4 < lim;
  }
  if (K % (2 * kBK)) return false;                       // k-contiguous rows cannot be zero-filled past K
  if (((int64_t)M + kBM) * lda * 4 >= lim) return false;
  return form == 1 ? (int64_t)N * ldb * 4 < lim : (int64_t)K * ldb * 4 < lim;
}

int gemm_f32_pp_splits(int M, int N, int K) {
  const int64_t tiles = (int64_t)(M / kBM) * (N / kBN);
  const int max_by_k = K / 512 > 0 ? K / 512 : 1;
  int best = 1;
  double best_eff = 0.0;
  for (int s = 1; s <= 16 && s <= max_by_k; ++s) {
    const int64_t blocks = tiles * s;
    const double eff = (double)blocks / (double)(((blocks + kNumCU - 1) / kNumCU) * kNumCU);
    if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
  }
  return best;
}

int gemm_f32_pp_colsum_chunks(int M, int splits) { return splits * (M / kBM) * 2; }

int launch_gemm_f32_pp(int form, GemmArgs g, int splits, hipStream_t s) {
  g.tiles_m = (g.M + kBM - 1) / kBM;
  g.tiles_n = g.N / kBN;
  switch (form) {
    case 0: return launch<0, EPI_BIAS_LRELU>(g, splits, s);
    case 1: return launch<1, EPI_LRELU_MASK>(g, splits, s);
    default: return launch<2, EPI_SLAB_COLSUM>(g, splits, s);
  }
}

}  // namespace cdml
