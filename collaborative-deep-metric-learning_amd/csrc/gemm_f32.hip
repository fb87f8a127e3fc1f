// fp32 MFMA GEMMs of the embedding tower (gfx950, v_mfma_f32_32x32x2_f32).
//
// Reference: the two slim.fully_connected layers of VNet (models.py:59-60 via
// :19-30) and their autodiff (train.py:141): per row 2*F*H + 2*H*D flop forward
// and 2*F*H + 4*H*D backward (no dX: the input is a placeholder, train.py:265).
//
// Roofline: MFMA.  fp32-in/fp32-acc MFMA is an exact k-ordered fmaf chain, so
// the 1e-5 parity bound against the fp32 reference path holds; peak 157.3 TF.
//
// One kernel template covers the three operand layouts:
//   fwd         y  = lrelu(x  @ W + b)      A k-contiguous, B k-strided
//   bwd data    dx = (dy @ W^T) * lrelu'    A k-contiguous, B k-contiguous
//   bwd weight  dW = x^T @ dy (+ colsum dy) A k-strided,    B k-strided, split-K
// Block = 4 waves (2x2), wave tile = (32*TM) x (32*TN) via TM*TN 32x32x2 MFMAs
// per k-pair.  Two LDS stages of one K-tile (BKT = 32 or 16 deep) each, filled
// by LDS-DMA (no VGPR staging, no ds_write), one barrier per K-tile.
//   * k-contiguous operands: lane-linear rows of BKT floats whose 16-B chunks are
//     XOR-swizzled by the row (on the DMA's per-lane source address and on the
//     ds_read_b128) -> conflict-free; lane-half h takes k = 8g+4h+t for MFMA t
//     and the other operand uses the same k permutation;
//   * k-strided operands: [k][cols], read with ds_read_b32, fetched through a
//     buffer descriptor that ends at the split's last row, so the ragged last
//     K-tile of the weight gradient reads zeros with no predication.
// Workgroups are dealt to XCDs in contiguous chunks of a grouped raster so the
// tiles sharing an A/B panel hit the same L2.
#include "gemm_f32.h"

namespace cdml {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// native vector: HIP's float4 is a struct, and struct copies through pointers become
// memcpy-to-alloca (scratch) when the copy is conditional
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kThreads = 256;

// tuning switches (A/B'd on MI355X with tools/gemm_variants.sh)
#ifndef CDML_GEMM_FRAG_PREFETCH
#define CDML_GEMM_FRAG_PREFETCH 2   // read k-group g+1's fragments while group g's MFMAs issue (2: bwd-weight only)
#endif
#ifndef CDML_GEMM_BK
#define CDML_GEMM_BK 32             // K-tile depth of the big-tile kernels (32 or 16)
#endif
#ifndef CDML_GEMM_STAGES
#define CDML_GEMM_STAGES 2          // LDS stages of the big-tile kernels: 3 keeps the DMA two K-tiles ahead
#endif
#ifndef CDML_GEMM_BLOCKS_PER_CU
#define CDML_GEMM_BLOCKS_PER_CU 2   // launch-bounds occupancy request
#endif

// ---- LDS-DMA primitives.  Issued through inline asm on purpose: with the
// builtins hipcc cannot prove that the DMA into LDS stage b^1 does not alias
// the ds_reads of stage b and drains the DMA (s_waitcnt vmcnt(0)) before the
// first fragment read of every K-tile, i.e. in front of the MFMAs.  The asm
// loads are invisible to its wait-count pass; the kernel waits for them itself
// (dma_wait_all) right before the barrier that publishes the tile.
__device__ __forceinline__ uint32_t lds_offset(const float *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float *)p;
}
// 64 lanes x 16 B: global (per-lane address) -> LDS (m0 = wave-uniform base, + lane*16)
__device__ __forceinline__ void dma_global_to_lds(const float *gptr, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(lds_base), "v"(gptr) : "memory", "m0");
}
// same through a buffer descriptor: lanes whose offset is outside the range read zeros
__device__ __forceinline__ void dma_buffer_to_lds(i32x4 srd, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(srd) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// A 16-B load the compiler's wait-count pass does not see either (it would answer a visible load issued between
// the DMA pieces with vmcnt(0)): the destination is valid only after a wait_loads<N>() that names it.
__device__ __forceinline__ void load16_async(f32x4 &dst, const float *gptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(gptr) : "memory");
}
// Wait until at most N of this wave's loads are in flight.  Returns 0.0f produced BEHIND the wait: what reads
// an asynchronously loaded register compares against it, so that the compiler cannot schedule the read in
// front of the wait (and a and b as plain inputs are not copied on the way in, which a tied operand would be
// -- a copy of a register still in flight).
template <int N>
__device__ __forceinline__ float wait_loads(const f32x4 &a, const f32x4 &b) {
  float zero;
  asm volatile("s_waitcnt vmcnt(%1)\n\tv_mov_b32 %0, 0" : "=v"(zero) : "n"(N), "v"(a), "v"(b) : "memory");
  return zero;
}
__device__ __forceinline__ uint32_t sign_bits(f32x4 m, float zero) {
  return (m.x > zero ? 1u : 0u) | (m.y > zero ? 2u : 0u) | (m.z > zero ? 4u : 0u) | (m.w > zero ? 8u : 0u);
}

__device__ __forceinline__ i32x4 make_srd(const float *base, int64_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));  // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)(bytes > 0 ? bytes : 0));
  r.w = 0x00020000;
  return r;
}

// AKC/BKC: operand is k-contiguous in memory.  TM/TN: MFMA tiles per wave.
// BKT: K-tile depth.  LEPI: stage the C tile through LDS for coalesced 16-B row
// stores (pays when K is short); NBLK: blocks per CU asked of the register allocator.
template <bool AKC, bool BKC, int TM, int TN, int EPI, int BKT, bool LEPI, int NBLK, int NSTG>
__global__ void __launch_bounds__(kThreads, NBLK) k_gemm_f32(GemmArgs g) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int A_TILE = BM * BKT, B_TILE = BKT * BN;       // floats; no padding (lane-linear DMA image)
  constexpr int STAGE = A_TILE + B_TILE;
  constexpr int SMEM = (LEPI && BM * BN > NSTG * STAGE) ? BM * BN : NSTG * STAGE;
  static_assert(NSTG == 2 || NSTG == 3, "LDS stages");
  constexpr int CPR = BKT / 4;                               // 16-B chunks per k-contiguous row
  constexpr int SWZ_SHIFT = (BKT == 32) ? 1 : 2;             // rows per 256-B bank row: 2 or 4
  constexpr int RPP = 256 / BKT;                             // k-contiguous rows per 1-KiB DMA piece
  constexpr bool KPRED = !AKC && !BKC;                       // only the bwd-weight GEMM has a ragged K
  static_assert(BKT == 32 || BKT == 16, "K-tile depth");
  __shared__ __attribute__((aligned(16))) float smem[SMEM];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, g.tiles_m, g.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int split = blockIdx.y;
  const int k_begin = split * g.k_per_split;
  const int k_end = min(g.K, k_begin + g.k_per_split);
  const int n_ktiles = (k_end - k_begin + BKT - 1) / BKT;

  f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = (EPI == EPI_SLAB_COLSUM) && g.colsum && tm == 0;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  const i32x4 srd_a = (!AKC) ? make_srd(g.A + (int64_t)k_begin * g.lda, (int64_t)(k_end - k_begin) * g.lda * 4)
                             : i32x4{0, 0, 0, 0};
  const i32x4 srd_b = (!BKC) ? make_srd(g.B + (int64_t)k_begin * g.ldb, (int64_t)(k_end - k_begin) * g.ldb * 4)
                             : i32x4{0, 0, 0, 0};

  // Each wave-instruction moves 64 x 16 B = 1 KiB ("piece") of tile kt into stage buf.
  auto issue_tile = [&](int buf, int kt) {
    const float *sA = smem + buf * STAGE;
    const float *sB = sA + A_TILE;
    const uint32_t la = __builtin_amdgcn_readfirstlane(lds_offset(sA) + wave * 1024);
    const uint32_t lb = __builtin_amdgcn_readfirstlane(lds_offset(sB) + wave * 1024);
    const int k0 = k_begin + kt * BKT;
    constexpr int PA = A_TILE / 256 / 4, PB = B_TILE / 256 / 4;   // pieces per wave
    if (AKC) {
#pragma unroll
      for (int j = 0; j < PA; ++j) {                 // piece = wave + 4j: RPP rows of BKT floats
        const int row = (wave + 4 * j) * RPP + lane / CPR;
        const int c = (lane % CPR) ^ ((row >> SWZ_SHIFT) % CPR);
        dma_global_to_lds(g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + k0 + 4 * c, la + j * 4096);
      }
    } else {
      constexpr int KPP = 256 / BM;                  // k rows per piece
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const int f = lane * 4;
        const int k = kt * BKT + (wave + 4 * j) * KPP + f / BM;
        dma_buffer_to_lds(srd_a, (uint32_t)(((int64_t)k * g.lda + m0 + f % BM) * 4), la + j * 4096);
      }
    }
    if (BKC) {
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const int row = (wave + 4 * j) * RPP + lane / CPR;
        const int c = (lane % CPR) ^ ((row >> SWZ_SHIFT) % CPR);
        dma_global_to_lds(g.B + (int64_t)(n0 + row) * g.ldb + k0 + 4 * c, lb + j * 4096);
      }
    } else {
      constexpr int KPP = 256 / BN;
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const int f = lane * 4;
        const int k = kt * BKT + (wave + 4 * j) * KPP + f / BN;
        dma_buffer_to_lds(srd_b, (uint32_t)(((int64_t)k * g.ldb + n0 + f % BN) * 4), lb + j * 4096);
      }
    }
  };

  // LDS -> MFMA operand fragments of k-group grp (8 k values; lane-half h owns 4).
  // Returned by value: arrays passed by reference into a lambda end up in scratch.
  struct Frag { float a[TM][4]; float b[TN][4]; };
  auto load_frags = [&](const float *sA, const float *sB, int grp) {
    Frag f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int row = wm * 32 * TM + mi * 32 + l31;
      if (AKC) {
        const int off = row * BKT + (((2 * grp + h) ^ ((row >> SWZ_SHIFT) % CPR)) << 2);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(sA + off);
        f.a[mi][0] = v.x; f.a[mi][1] = v.y; f.a[mi][2] = v.z; f.a[mi][3] = v.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) f.a[mi][u] = sA[(8 * grp + 4 * h + u) * BM + row];
      }
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int col = wn * 32 * TN + ni * 32 + l31;
      if (BKC) {
        const int off = col * BKT + (((2 * grp + h) ^ ((col >> SWZ_SHIFT) % CPR)) << 2);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(sB + off);
        f.b[ni][0] = v.x; f.b[ni][1] = v.y; f.b[ni][2] = v.z; f.b[ni][3] = v.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) f.b[ni][u] = sB[(8 * grp + 4 * h + u) * BN + col];
      }
    }
    return f;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // MFMAs of one K-tile held in LDS stage `buf`
  auto compute_tile = [&](int buf) {
    const float *sA = smem + buf * STAGE;
    const float *sB = sA + A_TILE;
#define CDML_MFMA_GROUP(F)                                                                   \
  _Pragma("unroll") for (int u = 0; u < 4; ++u)                                              \
  _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                          \
  _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                          \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32((F).a[mi][u], (F).b[ni][u], acc[mi][ni], 0, 0, 0)
    // CDML_GEMM_FRAG_PREFETCH: 0 off, 1 every kernel, 2 only the bwd-weight GEMM (both
    // operands k-strided: 2x the LDS read instructions; measured +5 % there, -2 % on fwd)
    constexpr bool PF = (BKT == 32) &&
                        ((CDML_GEMM_FRAG_PREFETCH == 1) || (CDML_GEMM_FRAG_PREFETCH == 2 && KPRED));
    if constexpr (PF) {
      // fragments double-buffered in registers: group g+1 is read from LDS while the
      // 4*TM*TN MFMAs of group g issue.  The sched_barrier keeps the reads ahead (hipcc
      // otherwise sinks each read to just before its use and stalls on lgkmcnt(0)).
      Frag f0 = load_frags(sA, sB, 0);
      Frag f1 = load_frags(sA, sB, 1);
      __builtin_amdgcn_sched_barrier(0);
      CDML_MFMA_GROUP(f0);
      f0 = load_frags(sA, sB, 2);
      __builtin_amdgcn_sched_barrier(0);
      CDML_MFMA_GROUP(f1);
      f1 = load_frags(sA, sB, 3);
      __builtin_amdgcn_sched_barrier(0);
      CDML_MFMA_GROUP(f0);
      CDML_MFMA_GROUP(f1);
    } else {
#pragma unroll
      for (int grp = 0; grp < BKT / 8; ++grp) {
        const Frag f = load_frags(sA, sB, grp);
        CDML_MFMA_GROUP(f);
      }
    }
#undef CDML_MFMA_GROUP
  };

  // bias gradient of the bwd-weight GEMM: column sums of the dy tile now in LDS
  auto colsum_tile = [&](int buf) {
    const float *sB = smem + buf * STAGE + A_TILE;
    constexpr int CB = BN / 4, KRB = kThreads / CB;
    if constexpr (BKT >= KRB) {
#pragma unroll
      for (int j = 0; j < BKT / KRB; ++j)
        bsum += *reinterpret_cast<const f32x4 *>(sB + (j * KRB + t / CB) * BN + (t % CB) * 4);
    } else {
      if (t / CB < BKT) bsum += *reinterpret_cast<const f32x4 *>(sB + (t / CB) * BN + (t % CB) * 4);
    }
  };

  // EPI_LRELU_MASK through LDS (the data gradient, short contraction): the tile's 64 KB of mask operand, loaded
  // in the epilogue, were 40 us of the 197-us dH1 (tools/f32_nt_probe.py) -- HBM latency with nothing to hide
  // it.  Instead each of the first eight K-tiles requests two of the epilogue's sixteen 16-B pieces per thread
  // right after its operand DMA; the counted wait at the end of the K-tile leaves them in flight, the next K-tile
  // collects them as 8 sign bits.  (All sixteen requested at once after the first K-tile only moved the stall
  // to that K-tile's vmcnt(0): 197 -> 194 us.)
  constexpr bool MPF = (EPI == EPI_LRELU_MASK) && LEPI && NSTG == 2 && BM == 128 && BN == 128;
  constexpr int MPF_KT = 8, P_TILE = (A_TILE + B_TILE) / 256 / 4;      // DMA pieces per wave and K-tile
  const bool mask_ahead = MPF && g.aux != nullptr && m0 + BM <= g.M && n_ktiles == MPF_KT;   // uniform
  uint64_t mbits = 0;

  if constexpr (NSTG == 2) {
    if (n_ktiles > 0) issue_tile(0, 0);
    dma_wait_all();
    __syncthreads();
    if (MPF && mask_ahead) {
      // a loop of its own: the plain loop below pays ~4 % for carrying these branches
      f32x4 mq0 = zero4, mq1 = zero4;
      const float *mptr = g.aux + (int64_t)(m0 + t / 32) * g.ldaux + n0 + (t % 32) * 4;
      for (int kt = 0; kt < MPF_KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < MPF_KT) issue_tile(buf ^ 1, kt + 1);
        if (kt >= 1) {                                       // the pieces requested one K-tile ago
          const float z = (kt + 1 < MPF_KT) ? wait_loads<P_TILE>(mq0, mq1) : wait_loads<0>(mq0, mq1);
          mbits |= (uint64_t)(sign_bits(mq0, z) | (sign_bits(mq1, z) << 4)) << (8 * (kt - 1));
        }
        load16_async(mq0, mptr);                             // rows 16 kt + t/32 and + 8 of the tile
        load16_async(mq1, mptr + 8 * g.ldaux);
        mptr += 16 * g.ldaux;
        compute_tile(buf);
        // this wave's pieces of tile kt+1 are in LDS (the two mask pieces requested after them may still fly) ...
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();   // ... and so are everybody else's; stage `buf` is free again
      }
      const float z = wait_loads<0>(mq0, mq1);               // the last two pieces
      mbits |= (uint64_t)(sign_bits(mq0, z) | (sign_bits(mq1, z) << 4)) << (8 * (MPF_KT - 1));
    } else {
      for (int kt = 0; kt < n_ktiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < n_ktiles) issue_tile(buf ^ 1, kt + 1);  // lands in the other stage under the MFMAs
        if (KPRED && EPI == EPI_SLAB_COLSUM && do_colsum) colsum_tile(buf);
        compute_tile(buf);
        dma_wait_all();    // this wave's pieces of tile kt+1 are in LDS ...
        __syncthreads();   // ... and so are everybody else's; stage `buf` is free again
      }
    }
  } else {
    // three stages: the DMA runs TWO K-tiles ahead, so a tile has two compute phases to
    // arrive; the wait before the barrier leaves the newest tile's pieces in flight
    // (counted vmcnt -- the kernel counts its own asm loads: P pieces per wave per tile)
    constexpr int P = (A_TILE + B_TILE) / 256 / 4;
    if (n_ktiles > 0) issue_tile(0, 0);
    if (n_ktiles > 1) {
      issue_tile(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");
    } else {
      dma_wait_all();
    }
    __syncthreads();
    int buf = 0;
    for (int kt = 0; kt < n_ktiles; ++kt) {
      const int nxt2 = buf >= 1 ? buf - 1 : 2;             // (kt + 2) % 3
      if (kt + 2 < n_ktiles) issue_tile(nxt2, kt + 2);
      if (KPRED && EPI == EPI_SLAB_COLSUM && do_colsum) colsum_tile(buf);
      compute_tile(buf);
      if (kt + 2 < n_ktiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");
      else dma_wait_all();
      __syncthreads();
      buf = buf == 2 ? 0 : buf + 1;
    }
  }

  float *C = g.C + (EPI == EPI_SLAB_COLSUM ? (int64_t)split * g.slab_stride : 0);
  const bool has_aux = (EPI == EPI_LRELU_MASK) && g.aux != nullptr;  // uniform
  if constexpr (!LEPI) {
    // direct epilogue (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)):
    // one dword per lane per store, two 128-B row segments per instruction
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = n0 + wn * 32 * TN + ni * 32 + l31;
        float bias = 0.f;
        if (EPI == EPI_BIAS_LRELU) bias = g.bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < g.M) {
            float v = acc[mi][ni][r];
            if (EPI == EPI_BIAS_LRELU) {
              v += bias;
              v = fmaxf(v, v * g.alpha);
            } else if (EPI == EPI_LRELU_MASK) {
              if (has_aux) v *= (g.aux[(int64_t)row * g.ldaux + col] > 0.f) ? 1.f : g.alpha;
            }
            C[(int64_t)row * g.ldc + col] = v;
          }
        }
      }
  } else {
    // accumulators -> LDS C tile, then whole 16-B row segments per lane so the global
    // stores (and the aux / bias loads) are coalesced
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          smem[row * BN + wn * 32 * TN + ni * 32 + l31] = acc[mi][ni][r];
        }
    __syncthreads();

    constexpr int C4 = BN / 4, ROWS_PER_PASS = kThreads / C4, PASSES = BM / ROWS_PER_PASS;
    const int c4 = t % C4;
    const int col = n0 + c4 * 4;
    const int lr0 = t / C4;
    f32x4 bias4 = zero4;
    if (EPI == EPI_BIAS_LRELU) bias4 = *reinterpret_cast<const f32x4 *>(g.bias + col);

    auto finish = [&](f32x4 v, f32x4 m) {
      if (EPI == EPI_BIAS_LRELU) {
        v += bias4;
        v.x = fmaxf(v.x, v.x * g.alpha); v.y = fmaxf(v.y, v.y * g.alpha);
        v.z = fmaxf(v.z, v.z * g.alpha); v.w = fmaxf(v.w, v.w * g.alpha);
      } else if (EPI == EPI_LRELU_MASK) {
        if (has_aux) {
          v.x *= (m.x > 0.f) ? 1.f : g.alpha; v.y *= (m.y > 0.f) ? 1.f : g.alpha;
          v.z *= (m.z > 0.f) ? 1.f : g.alpha; v.w *= (m.w > 0.f) ? 1.f : g.alpha;
        }
      }
      return v;
    };

    if (m0 + BM <= g.M) {
      // full tile (wave-uniform test): no per-lane branches, so every aux load and
      // every store is in flight at once instead of one round trip per pass
      f32x4 m[PASSES];
      if (EPI == EPI_LRELU_MASK && has_aux) {
        if (MPF && mask_ahead) {
          static_assert(!MPF || (PASSES == 2 * MPF_KT && ROWS_PER_PASS == 8 && C4 == 32), "mask pieces per K-tile");
#pragma unroll
          for (int p = 0; p < PASSES; ++p) {               // finish() only looks at the signs
            const uint32_t b = (uint32_t)(mbits >> (4 * p));
            m[p] = f32x4{(b & 1u) ? 1.f : 0.f, (b & 2u) ? 1.f : 0.f, (b & 4u) ? 1.f : 0.f, (b & 8u) ? 1.f : 0.f};
          }
        } else {
#pragma unroll
          for (int p = 0; p < PASSES; ++p)
            m[p] = *reinterpret_cast<const f32x4 *>(g.aux + (int64_t)(m0 + p * ROWS_PER_PASS + lr0) * g.ldaux + col);
        }
      }
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int lr = p * ROWS_PER_PASS + lr0;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + lr * BN + c4 * 4);
        *reinterpret_cast<f32x4 *>(C + (int64_t)(m0 + lr) * g.ldc + col) =
            finish(v, (EPI == EPI_LRELU_MASK && has_aux) ? m[p] : zero4);
      }
    } else {
      for (int p = 0; p < PASSES; ++p) {
        const int lr = p * ROWS_PER_PASS + lr0;
        const int row = m0 + lr;
        if (row < g.M) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + lr * BN + c4 * 4);
          f32x4 mm = zero4;
          if (EPI == EPI_LRELU_MASK && has_aux)
            mm = *reinterpret_cast<const f32x4 *>(g.aux + (int64_t)row * g.ldaux + col);
          *reinterpret_cast<f32x4 *>(C + (int64_t)row * g.ldc + col) = finish(v, mm);
        }
      }
    }
  }

  if (EPI == EPI_SLAB_COLSUM && do_colsum) {
    // threads with equal (t % C4) hold partial sums of the same 4 columns
    constexpr int C4 = BN / 4;
    constexpr int KR = kThreads / C4;
    __syncthreads();
    f32x4 *red = reinterpret_cast<f32x4 *>(smem);
    red[t] = bsum;
    __syncthreads();
    if (t < C4) {
      f32x4 s = red[t];
      for (int j = 1; j < KR; ++j) s += red[t + j * C4];
      *reinterpret_cast<f32x4 *>(g.colsum + (int64_t)split * g.N + n0 + t * 4) = s;
    }
  }
}

// out[i] = sum_z slab[z][i] in a fixed order (deterministic split-K combine).
__global__ void __launch_bounds__(kThreads)
k_sum_slabs(const float *__restrict__ slabs, int64_t slab_stride, int splits, int64_t n4,
            float *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4 *>(slabs)[i];
    for (int z = 1; z < splits; ++z) s += reinterpret_cast<const f32x4 *>(slabs + (int64_t)z * slab_stride)[i];
    reinterpret_cast<f32x4 *>(out)[i] = s;
  }
}

// dW[K][lddw] <- slab[K][N] rows (when lddw != N the combine is row-wise)
__global__ void __launch_bounds__(kThreads)
k_sum_slabs_2d(const float *__restrict__ slabs, int64_t slab_stride, int splits, int rows, int N,
               float *__restrict__ out, int64_t ldo) {
  const int n4 = N >> 2;
  const int64_t total = (int64_t)rows * n4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / n4;
    const int c = (int)(i - r * n4);
    f32x4 s = reinterpret_cast<const f32x4 *>(slabs)[i];
    for (int z = 1; z < splits; ++z) s += reinterpret_cast<const f32x4 *>(slabs + (int64_t)z * slab_stride)[i];
    reinterpret_cast<f32x4 *>(out + r * ldo)[c] = s;
  }
}

// ------------------------------------------------------ stream-K weight gradients ----
// Both weight gradients of the tower in ONE launch, data-parallel + stream-K.
// At the step's shapes dW1 is 12 x 40 = 480 tiles of 128 x 128 and dW2 40 x 2 = 80, every tile
// R/32 K-tiles deep: as two launches on 512 block slots (256 CUs x 2) that is one round at
// 480/512 occupancy and then a split-K round for dW2 -- 1.19 tile-times.  Here every one of
// the 512 resident blocks takes one whole tile (512 of the 560), and the K-tiles of the 48
// tiles left over are dealt evenly to all blocks (24 each at R = 8192): 1.09 tile-times, no
// idle CU.  A block's share of the left-over tiles is a run of consecutive iterations in
// tile-major order, so it covers at most the tail of one tile and the head of the next; whole
// tiles are written in place (with their bias-gradient column sums), parts go to the block's
// two slab slots -- [0] a part that starts inside a tile, [1] a part that starts a tile but
// does not finish it -- and k_gemm_f32_sk_fixup adds a split tile's parts in block order
// (deterministic, no atomics).  Kernel body = k_gemm_f32<false,false,2,2,EPI_SLAB_COLSUM,32,...>
// (k-strided operands through buffer descriptors, LDS-DMA, register-prefetched fragments).
struct SkProblem {
  const float *A; int64_t lda;     // x     [K][M]  (k-strided)
  const float *B; int64_t ldb;     // dy    [K][N]
  float *C; int64_t ldc;           // dW    [M][N]
  float *colsum;                   // db    [N] or null
  int tiles_m, tiles_n, tile0;
};
struct SkArgs {
  SkProblem p[2];
  int n_problems, K, n_kt, total_tiles;
  int whole_rounds;                // every block first takes this many WHOLE tiles (lb, lb+grid, ...)
  int ipb;                         // then this many iterations of the remaining tiles' joint space
  float *slabs;                    // [grid][2][128*128]
  float *cs_slabs;                 // [grid][2][128]
};

__device__ __forceinline__ void sk_tile(const SkArgs &g, int T, int &pi, int &tm, int &tn) {
  pi = (g.n_problems > 1 && T >= g.p[1].tile0) ? 1 : 0;
  const int tiles_m = pi ? g.p[1].tiles_m : g.p[0].tiles_m, tiles_n = pi ? g.p[1].tiles_n : g.p[0].tiles_n;
  const int lt = T - (pi ? g.p[1].tile0 : g.p[0].tile0);
  constexpr int GROUP_M = 8;
  const int width = GROUP_M * tiles_n;
  const int group = lt / width;
  const int first_m = group * GROUP_M;
  const int gsize = min(tiles_m - first_m, GROUP_M);
  const int in_group = lt - group * width;
  tm = first_m + in_group % gsize;
  tn = in_group / gsize;
}

__global__ void __launch_bounds__(kThreads, 2) k_gemm_f32_sk(SkArgs g) {
  constexpr int TM = 2, TN = 2, BKT = 32, BM = 128, BN = 128;
  constexpr int A_TILE = BM * BKT, B_TILE = BKT * BN, STAGE = A_TILE + B_TILE;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int lb = logical_block(blockIdx.x, gridDim.x);
  // Whole tiles first: all blocks then sweep k = 0 .. n_kt-1 together, so the blocks of an XCD that
  // share an operand panel read the same K-slice at the same time and the slice comes out of
  // their L2 once (a pure stream-K split starts every block at another k: each then pulls its own
  // copy of both panels -- 4.5 GB instead of 0.9 GB of L2 fills per launch, measured).  Only the
  // tiles left over after the whole rounds are split.
  const int first_split = g.whole_rounds * (int)gridDim.x;
  const int64_t split_iters = (int64_t)(g.total_tiles - first_split) * g.n_kt;
  int64_t it = (int64_t)lb * g.ipb;
  const int64_t it_end = min(it + (int64_t)g.ipb, split_iters);

  for (int seg = 0;; ++seg) {
    int T, kt0, kt1;
    if (seg < g.whole_rounds) {
      T = lb + seg * (int)gridDim.x;
      kt0 = 0;
      kt1 = g.n_kt;
    } else {
      if (it >= it_end) break;
      const int ts = (int)(it / g.n_kt);
      T = first_split + ts;
      kt0 = (int)(it - (int64_t)ts * g.n_kt);
      kt1 = (int)min((int64_t)g.n_kt, kt0 + (it_end - it));
      it += kt1 - kt0;
    }
    int pi, tm, tn;
    sk_tile(g, T, pi, tm, tn);
    const float *pA = pi ? g.p[1].A : g.p[0].A, *pB = pi ? g.p[1].B : g.p[0].B;
    const int64_t lda = pi ? g.p[1].lda : g.p[0].lda, ldb = pi ? g.p[1].ldb : g.p[0].ldb;
    float *pcs = pi ? g.p[1].colsum : g.p[0].colsum;
    const int m0 = tm * BM, n0 = tn * BN;
    const int k_begin = kt0 * BKT, k_end = min(g.K, kt1 * BKT);
    const int n_ktiles = kt1 - kt0;
    const bool whole = (kt0 == 0) && (kt1 == g.n_kt);
    const bool do_colsum = pcs != nullptr && tm == 0;

    const i32x4 srd_a = make_srd(pA + (int64_t)k_begin * lda, (int64_t)(k_end - k_begin) * lda * 4);
    const i32x4 srd_b = make_srd(pB + (int64_t)k_begin * ldb, (int64_t)(k_end - k_begin) * ldb * 4);

    auto issue_tile = [&](int buf, int kt) {
      const float *sA = smem + buf * STAGE;
      const float *sB = sA + A_TILE;
      const uint32_t la = __builtin_amdgcn_readfirstlane(lds_offset(sA) + wave * 1024);
      const uint32_t lbo = __builtin_amdgcn_readfirstlane(lds_offset(sB) + wave * 1024);
      constexpr int PA = A_TILE / 256 / 4, PB = B_TILE / 256 / 4, KPP = 256 / BM;
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const int f = lane * 4;
        const int k = kt * BKT + (wave + 4 * j) * KPP + f / BM;
        dma_buffer_to_lds(srd_a, (uint32_t)(((int64_t)k * lda + m0 + f % BM) * 4), la + j * 4096);
      }
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const int f = lane * 4;
        const int k = kt * BKT + (wave + 4 * j) * KPP + f / BN;
        dma_buffer_to_lds(srd_b, (uint32_t)(((int64_t)k * ldb + n0 + f % BN) * 4), lbo + j * 4096);
      }
    };

    struct Frag { float a[TM][4]; float b[TN][4]; };
    auto load_frags = [&](const float *sA, const float *sB, int grp) {
      Frag f;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int row = wm * 32 * TM + mi * 32 + l31;
#pragma unroll
        for (int u = 0; u < 4; ++u) f.a[mi][u] = sA[(8 * grp + 4 * h + u) * BM + row];
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = wn * 32 * TN + ni * 32 + l31;
#pragma unroll
        for (int u = 0; u < 4; ++u) f.b[ni][u] = sB[(8 * grp + 4 * h + u) * BN + col];
      }
      return f;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};

#define CDML_SK_MFMA(F)                                                                      \
  _Pragma("unroll") for (int u = 0; u < 4; ++u)                                              \
  _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                          \
  _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                          \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32((F).a[mi][u], (F).b[ni][u], acc[mi][ni], 0, 0, 0)
    if (n_ktiles > 0) issue_tile(0, 0);
    dma_wait_all();
    __syncthreads();
    for (int kt = 0; kt < n_ktiles; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < n_ktiles) issue_tile(buf ^ 1, kt + 1);
      const float *sA = smem + buf * STAGE;
      const float *sB = sA + A_TILE;
      if (do_colsum) {
        constexpr int CB = BN / 4, KRB = kThreads / CB;          // 32 column quads x 8 k-rows per pass
#pragma unroll
        for (int j = 0; j < BKT / KRB; ++j)
          bsum += *reinterpret_cast<const f32x4 *>(sB + (j * KRB + t / CB) * BN + (t % CB) * 4);
      }
      Frag f0 = load_frags(sA, sB, 0);
      Frag f1 = load_frags(sA, sB, 1);
      __builtin_amdgcn_sched_barrier(0);
      CDML_SK_MFMA(f0);
      f0 = load_frags(sA, sB, 2);
      __builtin_amdgcn_sched_barrier(0);
      CDML_SK_MFMA(f1);
      f1 = load_frags(sA, sB, 3);
      __builtin_amdgcn_sched_barrier(0);
      CDML_SK_MFMA(f0);
      CDML_SK_MFMA(f1);
      dma_wait_all();
      __syncthreads();
    }
#undef CDML_SK_MFMA

    // epilogue: a whole tile goes to dW, a part to the block's slab slot
    const int which = (kt0 == 0) ? 1 : 0;
    float *C = whole ? (pi ? g.p[1].C : g.p[0].C) + (int64_t)m0 * (pi ? g.p[1].ldc : g.p[0].ldc) + n0
                     : g.slabs + ((int64_t)lb * 2 + which) * (BM * BN);
    const int64_t ldc = whole ? (pi ? g.p[1].ldc : g.p[0].ldc) : BN;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = wn * 32 * TN + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 32 * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          C[(int64_t)row * ldc + col] = acc[mi][ni][r];
        }
      }
    if (do_colsum) {       // threads with equal (t % 32) hold partial sums of the same 4 columns
      constexpr int C4 = BN / 4, KR = kThreads / C4;
      f32x4 *red = reinterpret_cast<f32x4 *>(smem);
      red[t] = bsum;
      __syncthreads();
      if (t < C4) {
        f32x4 sum = red[t];
        for (int j = 1; j < KR; ++j) sum += red[t + j * C4];
        float *dst = whole ? pcs + n0 : g.cs_slabs + ((int64_t)lb * 2 + which) * BN;
        *reinterpret_cast<f32x4 *>(dst + t * 4) = sum;
      }
      __syncthreads();     // the staging buffers are reused by the next segment
    }
  }
}

// A tile that was computed in parts is the sum of its parts, added in block order (block b0
// holds the part that starts the tile in its slot 1, every later block its part in slot 0).
// 16 blocks per split tile (8 rows each: one float4 per thread and part, all parts' loads in
// flight together) -- one block per tile left a 11-part tile to 704 KB of serial reads.
constexpr int kFixChunks = 16;
__global__ void __launch_bounds__(kThreads) k_gemm_f32_sk_fixup(SkArgs g, int first_split) {
  const int ts = blockIdx.x / kFixChunks, chunk = blockIdx.x % kFixChunks;
  const int T = first_split + ts, t = threadIdx.x;
  const int64_t first = (int64_t)ts * g.n_kt;
  const int b0 = (int)(first / g.ipb), b1 = (int)((first + g.n_kt - 1) / g.ipb);
  if (b0 == b1) return;                                          // written in place by one block
  int pi, tm, tn;
  sk_tile(g, T, pi, tm, tn);
  float *C = (pi ? g.p[1].C : g.p[0].C);
  const int64_t ldc = pi ? g.p[1].ldc : g.p[0].ldc;
  float *pcs = pi ? g.p[1].colsum : g.p[0].colsum;
  const int m0 = tm * 128, n0 = tn * 128;
  const int row = chunk * 8 + (t >> 5), c4 = t & 31;
  const int64_t off = (int64_t)row * 128 + c4 * 4;
  f32x4 sum = *reinterpret_cast<const f32x4 *>(g.slabs + ((int64_t)b0 * 2 + 1) * 16384 + off);
  for (int b = b0 + 1; b <= b1; ++b)
    sum += *reinterpret_cast<const f32x4 *>(g.slabs + ((int64_t)b * 2) * 16384 + off);
  *reinterpret_cast<f32x4 *>(C + (int64_t)(m0 + row) * ldc + n0 + c4 * 4) = sum;
  if (pcs && tm == 0 && chunk == 0 && t < 32) {
    f32x4 cs = *reinterpret_cast<const f32x4 *>(g.cs_slabs + ((int64_t)b0 * 2 + 1) * 128 + t * 4);
    for (int b = b0 + 1; b <= b1; ++b)
      cs += *reinterpret_cast<const f32x4 *>(g.cs_slabs + ((int64_t)b * 2) * 128 + t * 4);
    *reinterpret_cast<f32x4 *>(pcs + n0 + t * 4) = cs;
  }
}

constexpr int kSkGrid = 2 * kNumCU;     // every block resident at once: 2 per CU

// LEPI per layout: the C tile goes through LDS only for the data-gradient GEMM,
// whose contraction is short (K = 256: 8 K-tiles per output tile) so the epilogue
// is a visible share of the tile; the long-K kernels keep the small LDS footprint
// that lets more blocks share a CU.
template <bool AKC, bool BKC, int EPI>
int launch_gemm(GemmArgs g, int tm_sel, int tn_sel, int splits, hipStream_t s) {
  const dim3 block(kThreads);
  constexpr bool NT = AKC && BKC;
  // (the forward kernel also gains ~2 % from it while its two stages are 64 KB anyway)
  constexpr bool LEPI = NT || (AKC && CDML_GEMM_BK == 32);
  constexpr int BKT = NT ? 32 : CDML_GEMM_BK, NB = NT ? 2 : CDML_GEMM_BLOCKS_PER_CU;
  constexpr int NS = NT ? 2 : CDML_GEMM_STAGES;
  if (tm_sel == 2 && tn_sel == 2) {
    g.tiles_m = (g.M + 127) / 128; g.tiles_n = g.N / 128;
    hipLaunchKernelGGL((k_gemm_f32<AKC, BKC, 2, 2, EPI, BKT, LEPI, NB, NS>), dim3(g.tiles_m * g.tiles_n, splits),
                       block, 0, s, g);
  } else if (tm_sel == 1 && tn_sel == 2) {
    g.tiles_m = (g.M + 63) / 64; g.tiles_n = g.N / 128;
    hipLaunchKernelGGL((k_gemm_f32<AKC, BKC, 1, 2, EPI, 32, AKC, 2, 2>), dim3(g.tiles_m * g.tiles_n, splits),
                       block, 0, s, g);
  } else {
    g.tiles_m = (g.M + 63) / 64; g.tiles_n = g.N / 64;
    hipLaunchKernelGGL((k_gemm_f32<AKC, BKC, 1, 1, EPI, 32, AKC, 2, 2>), dim3(g.tiles_m * g.tiles_n, splits),
                       block, 0, s, g);
  }
  return check_launch("gemm_f32");
}

// Pick the largest tile that still gives the chip enough workgroups
// (256 CUs x 2 resident blocks).
void pick_tile(int M, int N, int &tm, int &tn) {
  const int64_t want = 384;
  auto tiles = [&](int bm, int bn) { return (int64_t)((M + bm - 1) / bm) * (N / bn); };
  if (N % 128 == 0 && tiles(128, 128) >= want) { tm = 2; tn = 2; return; }
  if (N % 128 == 0 && tiles(64, 128) >= want) { tm = 1; tn = 2; return; }
  if (tiles(64, 64) >= want || N % 128 != 0) { tm = 1; tn = 1; return; }
  if (tiles(128, 128) >= 192) { tm = 2; tn = 2; return; }
  tm = 1; tn = (tiles(64, 128) >= 128) ? 2 : 1;
}

int check_mat(const char *who, const void *p, int64_t ld, int64_t min_ld) {
  CDML_REQUIRE(p, CDML_E_BADARG, "%s: null pointer", who);
  CDML_REQUIRE(aligned16(p) && (ld & 3) == 0 && ld >= min_ld, CDML_E_ALIGN,
               "%s: base must be 16-B aligned, leading dimension a multiple of 4 and >= %lld", who,
               (long long)min_ld);
  return CDML_OK;
}

// Split-K supplies the parallelism for the weight gradient, so take the
// largest tile that divides the output.
void pick_tile_bwd_weight(int K, int N, int &tm, int &tn) {
  if (K % 128 == 0 && N % 128 == 0) { tm = 2; tn = 2; }
  else if (N % 128 == 0) { tm = 1; tn = 2; }
  else { tm = 1; tn = 1; }
}

int bwd_weight_splits(int M, int K, int N) {
  int tm, tn;
  pick_tile_bwd_weight(K, N, tm, tn);
  const int64_t tiles = (int64_t)(K / (64 * tm)) * (N / (64 * tn));
  int64_t splits = (480 + tiles - 1) / tiles;
  const int64_t max_by_k = (M + 255) / 256;  // at least 256 contraction rows per split
  if (splits > max_by_k) splits = max_by_k;
  if (splits > 32) splits = 32;
  if (splits < 1) splits = 1;
  return (int)splits;
}

}  // namespace
}  // namespace cdml

using namespace cdml;

extern "C" int cdml_fc_lrelu_fwd(const float *x, int64_t ldx, const float *W, int64_t ldw,
                                 const float *b, float alpha, int M, int K, int N, float *y,
                                 int64_t ldy, cdml_stream_t stream) {
  CDML_REQUIRE(M > 0 && K > 0 && N > 0 && b, CDML_E_BADARG, "fc_lrelu_fwd: bad argument");
  CDML_REQUIRE(K % 32 == 0 && N % 64 == 0, CDML_E_UNSUPPORTED,
               "fc_lrelu_fwd: K must be a multiple of 32 and N of 64 (pad with zeros), got K=%d N=%d", K, N);
  int rc;
  if ((rc = check_mat("fc_lrelu_fwd x", x, ldx, K))) return rc;
  if ((rc = check_mat("fc_lrelu_fwd W", W, ldw, N))) return rc;
  if ((rc = check_mat("fc_lrelu_fwd y", y, ldy, N))) return rc;
  CDML_REQUIRE((int64_t)K * ldw * 4 < (1ll << 31), CDML_E_UNSUPPORTED,
               "fc_lrelu_fwd: weight matrix exceeds the 2 GiB buffer-descriptor range");
  GemmArgs g{};
  g.A = x; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = y; g.ldc = ldy;
  g.bias = b; g.alpha = alpha; g.M = M; g.N = N; g.K = K; g.k_per_split = K;
  int tm, tn;
  pick_tile(M, N, tm, tn);
  return launch_gemm<true, false, EPI_BIAS_LRELU>(g, tm, tn, 1, (hipStream_t)stream);
}

extern "C" int cdml_fc_bwd_data(const float *dy, int64_t lddy, const float *W, int64_t ldw,
                                const float *x_post, int64_t ldxp, float alpha, int M, int K, int N,
                                float *dx, int64_t lddx, cdml_stream_t stream) {
  CDML_REQUIRE(M > 0 && K > 0 && N > 0, CDML_E_BADARG, "fc_bwd_data: bad argument");
  CDML_REQUIRE(N % 32 == 0 && K % 64 == 0, CDML_E_UNSUPPORTED,
               "fc_bwd_data: N must be a multiple of 32 and K of 64, got K=%d N=%d", K, N);
  int rc;
  if ((rc = check_mat("fc_bwd_data dy", dy, lddy, N))) return rc;
  if ((rc = check_mat("fc_bwd_data W", W, ldw, N))) return rc;
  if ((rc = check_mat("fc_bwd_data dx", dx, lddx, K))) return rc;
  if (x_post && (rc = check_mat("fc_bwd_data x_post", x_post, ldxp, K))) return rc;
  GemmArgs g{};  // dx[M][K] = dy[M][N] @ W[K][N]^T : output cols = K, contraction = N
  g.A = dy; g.lda = lddy; g.B = W; g.ldb = ldw; g.C = dx; g.ldc = lddx;
  g.aux = x_post; g.ldaux = ldxp; g.alpha = alpha; g.M = M; g.N = K; g.K = N; g.k_per_split = N;
  int tm, tn;
  pick_tile(M, K, tm, tn);
  return launch_gemm<true, true, EPI_LRELU_MASK>(g, tm, tn, 1, (hipStream_t)stream);
}

extern "C" size_t cdml_fc_bwd_weight_workspace(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0 || K % 64 || N % 64) return 0;
  const int splits = bwd_weight_splits(M, K, N);   // enough for whichever kernel is dispatched
  const size_t slabs = splits > 1 ? (size_t)splits * K * N : 0, chunks = (size_t)splits;
  return (slabs + chunks * N) * sizeof(float);
}

extern "C" int cdml_fc_bwd_weight(const float *x, int64_t ldx, const float *dy, int64_t lddy, int M,
                                  int K, int N, float *dW, int64_t lddw, float *db, void *workspace,
                                  size_t workspace_bytes, cdml_stream_t stream) {
  CDML_REQUIRE(M > 0 && K > 0 && N > 0, CDML_E_BADARG, "fc_bwd_weight: bad argument");
  CDML_REQUIRE(K % 64 == 0 && N % 64 == 0, CDML_E_UNSUPPORTED,
               "fc_bwd_weight: K and N must be multiples of 64, got K=%d N=%d", K, N);
  int rc;
  if ((rc = check_mat("fc_bwd_weight x", x, ldx, K))) return rc;
  if ((rc = check_mat("fc_bwd_weight dy", dy, lddy, N))) return rc;
  if ((rc = check_mat("fc_bwd_weight dW", dW, lddw, N))) return rc;
  CDML_REQUIRE((int64_t)M * ldx * 4 < (1ll << 31) && (int64_t)M * lddy * 4 < (1ll << 31), CDML_E_UNSUPPORTED,
               "fc_bwd_weight: an operand exceeds the 2 GiB buffer-descriptor range (M=%d)", M);
  const size_t need = cdml_fc_bwd_weight_workspace(M, K, N);
  CDML_REQUIRE(workspace && workspace_bytes >= need && aligned16(workspace), CDML_E_BADARG,
               "fc_bwd_weight: workspace of %zu bytes (16-B aligned) required", need);
  const int splits = bwd_weight_splits(M, K, N);
  float *slabs = static_cast<float *>(workspace);
  float *colsum = slabs + (splits > 1 ? (size_t)splits * K * N : (size_t)0);
  // no combine pass for one split: the GEMM writes dW (and db) itself
  const bool direct = (splits == 1);
  GemmArgs g{};  // dW[K][N] = x[M][K]^T @ dy[M][N] : output rows = K, contraction = M
  g.A = x; g.lda = ldx; g.B = dy; g.ldb = lddy;
  g.C = direct ? dW : slabs; g.ldc = direct ? lddw : N;
  g.colsum = db ? (direct ? db : colsum) : nullptr; g.M = K; g.N = N; g.K = M;
  int kps = (M + splits - 1) / splits;
  kps = (kps + 63) / 64 * 64;
  g.k_per_split = kps;
  g.slab_stride = (int64_t)K * N;
  int tm, tn;
  pick_tile_bwd_weight(K, N, tm, tn);
  if ((rc = launch_gemm<false, false, EPI_SLAB_COLSUM>(g, tm, tn, splits, (hipStream_t)stream))) return rc;
  if (direct) return rc;
  const int64_t total4 = (int64_t)K * N / 4;
  int grid = (int)((total4 + kThreads - 1) / kThreads);
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(k_sum_slabs_2d, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, slabs,
                     g.slab_stride, splits, K, N, dW, lddw);
  if ((rc = check_launch("fc_bwd_weight combine"))) return rc;
  if (db) {
    hipLaunchKernelGGL(k_sum_slabs, dim3((N / 4 + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, colsum, (int64_t)N, splits, (int64_t)(N / 4), db);
    rc = check_launch("fc_bwd_weight bias combine");
  }
  return rc;
}

// ---- both weight gradients in one stream-K launch -------------------------------------------
static bool sk_usable(int M, int K1, int N1, int64_t ldx1, int64_t lddy1, int K2, int N2, int64_t ldx2, int64_t lddy2) {
  if (M < 256 || K1 % 128 || N1 % 128 || K2 % 128 || N2 % 128) return false;
  const int64_t lim = (int64_t)1 << 31;
  if ((int64_t)M * ldx1 * 4 >= lim || (int64_t)M * lddy1 * 4 >= lim || (int64_t)M * ldx2 * 4 >= lim ||
      (int64_t)M * lddy2 * 4 >= lim) return false;
  const int64_t tiles = (int64_t)(K1 / 128) * (N1 / 128) + (int64_t)(K2 / 128) * (N2 / 128);
  return tiles >= kSkGrid / 2;            // fewer tiles: the split-K kernels are the better fit
}

extern "C" size_t cdml_fc_bwd_weight2_workspace(int M, int K1, int N1, int K2, int N2) {
  if (!sk_usable(M, K1, N1, K1, N1, K2, N2, K2, N2)) return 0;
  return (size_t)kSkGrid * 2 * (128 * 128 + 128) * sizeof(float);
}

extern "C" int cdml_fc_bwd_weight2(const float *x1, int64_t ldx1, const float *dy1, int64_t lddy1, int K1, int N1,
                                   float *dW1, int64_t lddw1, float *db1, const float *x2, int64_t ldx2,
                                   const float *dy2, int64_t lddy2, int K2, int N2, float *dW2, int64_t lddw2,
                                   float *db2, int M, void *workspace, size_t workspace_bytes,
                                   cdml_stream_t stream) {
  CDML_REQUIRE(M > 0 && K1 > 0 && N1 > 0 && K2 > 0 && N2 > 0, CDML_E_BADARG, "fc_bwd_weight2: bad argument");
  int rc;
  if ((rc = check_mat("fc_bwd_weight2 x1", x1, ldx1, K1))) return rc;
  if ((rc = check_mat("fc_bwd_weight2 dy1", dy1, lddy1, N1))) return rc;
  if ((rc = check_mat("fc_bwd_weight2 dW1", dW1, lddw1, N1))) return rc;
  if ((rc = check_mat("fc_bwd_weight2 x2", x2, ldx2, K2))) return rc;
  if ((rc = check_mat("fc_bwd_weight2 dy2", dy2, lddy2, N2))) return rc;
  if ((rc = check_mat("fc_bwd_weight2 dW2", dW2, lddw2, N2))) return rc;
  CDML_REQUIRE(sk_usable(M, K1, N1, ldx1, lddy1, K2, N2, ldx2, lddy2), CDML_E_UNSUPPORTED,
               "fc_bwd_weight2: needs K and N multiples of 128, M >= 256, operands below 2 GiB and >= %d tiles "
               "(use cdml_fc_bwd_weight per layer)", kSkGrid / 2);
  const size_t need = cdml_fc_bwd_weight2_workspace(M, K1, N1, K2, N2);
  CDML_REQUIRE(workspace && workspace_bytes >= need && aligned16(workspace), CDML_E_BADARG,
               "fc_bwd_weight2: workspace of %zu bytes (16-B aligned) required", need);
  SkArgs g{};
  g.p[0] = SkProblem{x1, ldx1, dy1, lddy1, dW1, lddw1, db1, K1 / 128, N1 / 128, 0};
  g.p[1] = SkProblem{x2, ldx2, dy2, lddy2, dW2, lddw2, db2, K2 / 128, N2 / 128, (K1 / 128) * (N1 / 128)};
  g.n_problems = 2;
  g.K = M;
  g.n_kt = (M + 31) / 32;
  g.total_tiles = g.p[1].tile0 + (K2 / 128) * (N2 / 128);
  g.whole_rounds = g.total_tiles / kSkGrid;
  const int n_split = g.total_tiles - g.whole_rounds * kSkGrid;
  const int64_t total = (int64_t)n_split * g.n_kt;
  g.ipb = (int)((total + kSkGrid - 1) / kSkGrid);
  g.slabs = static_cast<float *>(workspace);
  g.cs_slabs = g.slabs + (size_t)kSkGrid * 2 * 128 * 128;
  hipLaunchKernelGGL(k_gemm_f32_sk, dim3(kSkGrid), dim3(kThreads), 0, (hipStream_t)stream, g);
  if ((rc = check_launch("fc_bwd_weight2"))) return rc;
  if (n_split > 0) {
    hipLaunchKernelGGL(k_gemm_f32_sk_fixup, dim3(n_split * kFixChunks), dim3(kThreads), 0, (hipStream_t)stream, g,
                       g.whole_rounds * kSkGrid);
    rc = check_launch("fc_bwd_weight2 fix-up");
  }
  return rc;
}
