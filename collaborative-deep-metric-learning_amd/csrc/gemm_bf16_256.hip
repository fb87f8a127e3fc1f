// bf16 projection GEMM, large-tile form: C[M][N] = epilogue(A[M][K] . B[N][K]^T),
// both operands k-contiguous bf16, fp32 accumulation (v_mfma_f32_32x32x16_bf16).
// Same layers and epilogues as k_gemm_bf16_nt (models.py:59-60, train.py:141 at
// BASELINE config 4's precision); this kernel is the one the big products run on.
//
// Structure (one block per CU, 512 threads, 128 KiB LDS):
//   * tile 256x256x64; 8 waves = 2 row groups x 4 column strips, wave tile 128x64
//     (8 accumulators of 32x32);
//   * operands reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds) in full 128-B
//     rows; a K-tile is four 16-KiB "half images" -- A-h0 / A-h1 hold the first /
//     second 64 rows of each row group, B-h0 / B-h1 the first / second 32 columns of
//     each strip -- so that an image is read in exactly ONE phase of the K-tile and
//     can be refilled for tile t+2 while tile t is still being multiplied;
//   * a K-tile is two phases of 16 MFMAs per wave: phase A multiplies the wave's first 64
//     rows by both column halves (reads A-h0, B-h0, B-h1: 16 ds_read_b128), phase B its second
//     64 rows (reads A-h1: 8; the B fragments stay in registers).  Each phase =
//     [fragment reads + two half-image DMAs + counted wait]  barrier  [16 MFMAs]  barrier,
//     and every fragment of a phase is in registers before its first barrier;
//   * the two row groups run one barrier apart (ping-pong): while one group's waves
//     issue MFMAs the other group's waves, on the same SIMDs, issue their LDS reads
//     and DMA, so the matrix pipe always has a wave feeding it;
//   * the DMA is never drained inside the loop (counted s_waitcnt vmcnt(8) / vmcnt(6): three
//     to four half images stay in flight across every barrier).
//   (A four-phase, quadrant-per-phase schedule with five half images in flight was measured at
//   the same throughput and removed.)
//
// DMA schedule and hazards (tile T lives in LDS buffer T&1; group 1 runs one barrier behind
// group 0; every phase has two barriers; LDS reads are retired BEFORE a phase's first barrier):
//   phase A(T) issues A-h0, A-h1 of tile T+1 -> other buffer; their last readers were phases
//              A(T-1) / B(T-1): at least one barrier back for the lagging group, and reads are
//              retired before a barrier is entered;
//   phase B(T) issues B-h0, B-h1 of tile T+2 -> this buffer; read in phase A(T), whose reads
//              the lagging group retired before ITS first barrier, one barrier back;
//   phase A(T) waits vmcnt(8): everything but the four newest half images has landed, i.e.
//              A-h1(T), read in phase B(T);
//   phase B(T) waits vmcnt(6): B-h0, B-h1, A-h0 of T+1 have landed, read in phase A(T+1).
//   A wave waits for its OWN pieces before a barrier; the read of the image comes two barriers
//   later, after every wave of both groups has passed its wait.
#include "gemm_bf16.h"
#include <type_traits>
#include <stdlib.h>

#ifndef CDML_BF16_MFMA_DEFAULT
#define CDML_BF16_MFMA_DEFAULT 16   // measured: profiles/r03_bf16_mfma_shape.txt
#endif

namespace cdml {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;

constexpr int kT = 512;
constexpr int kTileM = 256, kTileN = 256, kTileK = 64;
constexpr int IMG = 16384;       // one half image: 128 rows x 128 B
constexpr int BUF = 4 * IMG;     // one K-tile: A-h0, A-h1, B-h0, B-h1
// LDS by OPERAND: [A: buf0 h0 | buf0 h1 | buf1 h0 | buf1 h1][B: likewise] (round 4; rounds 1-3 laid it out by buffer).  Every
// fragment read of an operand is then within 64 KiB of ONE lane base, i.e. inside the 16-bit offset immediate of a ds_read:
// the second buffer costs no v_add per read (k-strided form: 24 fewer VALU per K-tile, -2 % measured) and no second set
// of base registers (k-contiguous forms: 229-240 -> 205-226 VGPRs).
constexpr int SMEM = 2 * BUF;    // 128 KiB
constexpr int SMEM_R6 = 10 * IMG;  // 160 KiB (the whole LDS of a CU): the resident-plane walk's 3 A slots + 2 B slots

__device__ __forceinline__ uint32_t lds_off(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// 64 lanes x 16 B through a buffer descriptor into LDS at m0 + lane*16; lanes whose
// offset is outside the descriptor's range deliver zeros.  Inline asm: invisible to
// hipcc's wait-count pass, the kernel counts these loads itself.
__device__ __forceinline__ void dma(i32x4 srd, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(srd) : "memory", "m0");
}
// the same with the wave-uniform part of the source offset in an SGPR (the instruction's soffset field): the per-lane
// offset register is then loop-invariant -- no v_add per piece, and nothing for the compiler to hoist into extra VGPRs
// when a loop is unrolled over many (plane, half, K-tile) combinations
__device__ __forceinline__ void dma_s(i32x4 srd, uint32_t voff, uint32_t soff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(srd), "s"(soff) : "memory", "m0");
}
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
// CDML_F16X2 (gemm_f16x2_256.hip compiles THIS file with it): the 16-bit operands are fp16 -- v_mfma_f32_16x16x32_f16 -- and
// the plane-output epilogues write TWO fp16 planes hi | lo of (value * BArgs::c_scale) instead of three bf16 planes.  The
// tile, images, DMA schedule, fragment layouts and phases are those of the bf16 form (both types are 16 bits wide; a
// fragment is eight of them in four registers either way).  Only the split-fp32 (X3) launchers are exported from that build.
#ifdef CDML_F16X2
constexpr bool kF16 = true;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
#define CDML_MFMA16(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0)
// two fp32 -> one dword of two fp16 (round to nearest even; element 0 in the low half), and the halves back as fp32
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  uint32_t w = __builtin_bit_cast(uint32_t, half2v{(_Float16)a, (_Float16)b});
  asm("" : "+v"(w));
  return w;
}
__device__ __forceinline__ float lo_of(uint32_t w) { return (float)__builtin_bit_cast(half2v, w)[0]; }
__device__ __forceinline__ float hi_of(uint32_t w) { return (float)__builtin_bit_cast(half2v, w)[1]; }
#else
constexpr bool kF16 = false;
#define CDML_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
// two fp32 -> one dword of two bf16 (round to nearest even; element 0 in the low half), and the halves back as fp32
// (the dword is made opaque: hipcc otherwise sees through `pack << 16` and converts the low element a second time on its own)
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  uint32_t w = __builtin_bit_cast(uint32_t, bf16x2{(__bf16)a, (__bf16)b});
  asm("" : "+v"(w));
  return w;
}
__device__ __forceinline__ float lo_of(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float hi_of(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
#endif
constexpr int kPlanesOut = kF16 ? 2 : 3;             // planes a plane-output epilogue writes
__device__ __forceinline__ i32x4 make_srd(const void *base, int64_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffff));  // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)(bytes > 0 ? bytes : 0));
  r.w = 0x00020000;
  return r;
}

// Every half image is waited for one phase before the phase that reads it, which
// always leaves the five newest images (10 wave-instructions) in flight.
#define CDML_BARRIER()                         \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("s_barrier" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

// TN = false: A[M][K], B[N][K] (k-contiguous rows; fragments by ds_read_b128).
// TN = true : A[K][M], B[K][N] (the weight gradients x^T.dy: the contraction runs over
//   the batch rows, so operand rows are k).  Half images are [64 k][128 columns] with
//   256-B rows, A-h0/A-h1 = columns 0-127 / 128-255 of the tile (a row group owns 64 of
//   each), B likewise (a strip owns 32 of each); chunk swizzle
//   ch ^ (((row&3)<<2) | ((row>>2)&3)) on the DMA source; fragments come out of LDS
//   already transposed through ds_read_b64_tr_b16 (two per 32x16 fragment), so no
//   transposed copy of the activations is ever made.
// S16: the same tile, images, DMA schedule and phases on v_mfma_f32_16x16x32_bf16 (32 MFMAs of 16 cycles
// per phase instead of 16 of 32: equal cycles per flop; the chip holds a higher clock on this shape under
// load, MI355X_MICROARCH.md 'DVFS give-back' item 7).  A fragment = 16 rows x 32 k (lane l: row l & 15,
// k = 8 (l >> 4) .. +7): one ds_read_b128 of the same 128-B-row images; accumulators 8 x 4 blocks of 16 x 16.
// One output tile over one K range: operands of `g` at tile origin (m0, n0), K-tiles k_begin/64 ..
// + n_ktiles (even; 0 = nothing is multiplied, zeros come out).  The result goes to c_base as
// c_base[(c_row0 + r) * c_ld + c_col0 + c] for tile-local (r, c) -- the caller's C (c_row0 = m0, c_col0 =
// n0) or a tile-local partial slab (0, 0).  cs_row (TN, nullable): the column sums of B over the K-tiles
// this (tile, row group) OWNS -- absolute K-tile index kt belongs to (tm, grp) = ((kt % (2 tiles_m)) >> 1,
// kt & 1), so across the tiles_m tiles that share B every K-tile is counted once -- written to
// cs_row[grp * cs_grp_stride + tile-local column].  Called by every wave of the block with the same arguments.
// X3: the operands are three bf16 planes each (BArgs::x3_*): the K-tiles walk the six plane products.
// F6 (X3 only): the walk is K-major over six products in whole six-step periods (the host guarantees it): the loop is
// the unrolled period with its compile-time DMA skipping, and the general loop is not compiled in.
// NTCS (X3, k-contiguous form): also sum B over k per column (its own instantiation: compiled into the plain kernels the
// four sums and their branch cost FC1 25 % -- 514 -> 642 us, measured).
// KI (k-strided form of the resident-plane walk only): the operands are stored k8-INTERLEAVED -- [plane][k / 8][column][8 k]
// bf16, lda / ldb = elements per k-group (8 x the operand's columns), x3_plane_* = elements per plane -- so that a fragment (8
// consecutive k of one column) is ONE aligned 16-B LDS read instead of two transposed 8-B reads, and a half image
// [8 k-groups][128 columns][16 B] is filled by 1-KiB pieces that are contiguous in memory.  Same images, slots, DMA schedule,
// accumulation order and results as the k-strided form (test_gemm_x3_tnk_equals_tn: bit for bit).
template <bool TN, int EPI, bool S16, bool X3 = false, bool F6 = false, bool NTCS = false, bool R6 = false, bool NARROW = false,
          bool KI = false>
__device__ __forceinline__ void run_tile(const BArgs &g, const int tm, const int m0, const int n0, const int k_begin,
                                         const int n_ktiles, void *c_base, const int64_t c_ld, const int c_row0,
                                         const int c_col0, float *cs_row, const int64_t cs_grp_stride,
                                         unsigned char *smem) {
  static_assert(!TN || EPI == BE_F32, "the k-strided form only serves the weight gradients");
  static_assert(!NARROW || (X3 && S16 && R6 && !TN && (EPI == BE_BIAS_LRELU_X3 || EPI == BE_MASK_X3 || EPI == BE_ROWBIAS_LRELU_X3 || EPI == BE_F32)),
                "the 128 x 256 half tile exists for the plane-output products of the resident-plane walk and for the fp32 slabs of the narrow layer");
  static_assert(X3 || (EPI != BE_BIAS_LRELU_X3 && EPI != BE_MASK_X3 && EPI != BE_ROWBIAS_LRELU_X3),
                "plane outputs belong to the split-fp32 form");
  static_assert((EPI != BE_MINE_X3 && EPI != BE_KNN_X3) || (X3 && S16 && R6 && !TN && !NARROW),
                "the mining / kNN-filter epilogues ride on the resident-plane walk");
  static_assert(!KI || (TN && X3 && S16 && R6), "the k8-interleaved operands exist for the k-strided resident-plane walk");
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, h = lane >> 5;
  // Round 6: the MFMA's two operands SWAPPED -- a fragment of the row operand and one of the column operand have the same
  // register layout (16 rows x 32 k, lane -> (row l & 15, k-chunk l >> 4)), so mfma(b, a) instead of mfma(a, b) costs
  // nothing and leaves the TRANSPOSED 16 x 16 block in the accumulator: lane (l15, q) then holds output ROW l15 and the
  // four consecutive COLUMNS 4q .. 4q + 3 of the block instead of column l15 and four rows.  An epilogue that works along
  // rows (the miner: the best column per row) then needs no trip through LDS: a lane reduces its own 16 columns of a row,
  // the four lanes that share the row meet in two cross-lane steps.
  // (The plane-output epilogues were built in this layout too -- one 8-B store per plane straight from the accumulators, no
  // strip, no barrier -- and measured SLOWER: FC1 + 4 %, the data gradient + 30 %, profiles/r06_swapped_plane_epilogue_ab.txt.
  // A store instruction then writes 16 rows x 32 B and a 128-B line is assembled from four of them; the LDS transpose of
  // tail16 below is what makes every store a whole line.  The swapped layout stays where nothing of the tile is stored.)
  constexpr bool kSwap = EPI == BE_MINE_X3 || EPI == BE_KNN_X3;

  const int k_rows = X3 ? g.x3_tpp * kTileK : g.K;              // k-strided form: rows of the operands in memory
  const i32x4 srd_a = make_srd(g.A, KI ? 3 * g.x3_plane_a * 2 : (int64_t)(TN ? k_rows : g.M) * g.lda * 2);
  const i32x4 srd_b = make_srd(g.B, KI ? 3 * g.x3_plane_b * 2 : (int64_t)(TN ? k_rows : g.N) * g.ldb * 2);

  // ---- DMA lane constants.  NT: piece pc = wave*2+i covers image rows pc*8 .. pc*8+7
  //      (128-B rows); TN: k-rows pc*4 .. pc*4+3 (256-B rows) ----
  uint32_t va[2], vb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if constexpr (!TN) {
      const int r = (wave * 2 + i) * 8 + (lane >> 3);
      const int sc = (lane & 7) ^ ((r >> 1) & 7);            // swizzle on the SOURCE chunk
      const int row_a = NARROW ? r : (r >> 6) * 128 + (r & 63);   // A-h0 (A-h1: + 64 rows); half tile: 64 rows per row group
      const int col_b = (r >> 5) * 64 + (r & 31);            // B-h0 (B-h1: + 32 columns)
      va[i] = (uint32_t)(((int64_t)(m0 + row_a) * g.lda + sc * 8) * 2);
      vb[i] = (uint32_t)(((int64_t)(n0 + col_b) * g.ldb + sc * 8) * 2);
    } else if constexpr (KI) {
      // piece pc = k-group pc >> 1, columns (pc & 1) * 64 + lane of the half image: 64 x 16 B contiguous in memory
      const int pc = wave * 2 + i;
      va[i] = (uint32_t)(((int64_t)(pc >> 1) * g.lda + (int64_t)(m0 + (pc & 1) * 64 + lane) * 8) * 2);   // A-h1: + 128 columns
      vb[i] = (uint32_t)(((int64_t)(pc >> 1) * g.ldb + (int64_t)(n0 + (pc & 1) * 64 + lane) * 8) * 2);
    } else {
      const int r = (wave * 2 + i) * 4 + (lane >> 4);
      const int sc = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
      va[i] = (uint32_t)(((int64_t)r * g.lda + m0 + sc * 8) * 2);   // A-h1: + 128 columns
      vb[i] = (uint32_t)(((int64_t)r * g.ldb + n0 + sc * 8) * 2);
    }
  }
  const uint32_t d_a = TN ? 256u : (uint32_t)(64 * g.lda * 2), d_b = TN ? 256u : (uint32_t)(32 * g.ldb * 2);
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane(lds_off(smem) + wave * 2048);

  // img 0 = A, 1 = B; hh = half; tile beyond the split's range -> every lane out of range (zeros)
  // X3: K-tile v of the whole walk = K-tile v % tpp of plane product v / tpp
  // (K-major, x3_products > 0: K-tile v = product v % P of K-tile v / P -- the three reads of an operand's hi
  // plane K-tile, two of its mid plane, come within six steps of each other, out of L2 instead of HBM)
  const bool x3_kmajor = X3 && g.x3_products > 0;
  // (through readfirstlane: a uniform float computed on the VALU otherwise lives in a VGPR, and the k-strided kernel
  // has none to spare -- it was keeping 12 B of stack)
  const float x3_inv = X3 ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(
                                int, 1.0f / (float)(x3_kmajor ? g.x3_products : g.x3_tpp)))) : 0.f;
  const int x3_t0 = k_begin / kTileK;
  auto x3_div = [&](int v) { return __builtin_amdgcn_readfirstlane((int)(((float)v + 0.5f) * x3_inv)); };
  // product (0..5) and K-tile within the plane of walk position v
  // (selects, not assignments under a branch: with those hipcc kept the two results on the stack and read the product
  // back behind an s_waitcnt vmcnt(0) -- draining the DMA pipeline wherever the k-strided form asked for it)
  auto x3_where = [&](int v, int &sgm, int &w) {
    const int qd = x3_div(v);
    const int rem = v - qd * (x3_kmajor ? g.x3_products : g.x3_tpp);
    sgm = x3_kmajor ? rem : qd;
    w = x3_kmajor ? qd : rem;
  };
  auto x3_segment = [&](int v) { int sgm, w; x3_where(v, sgm, w); return sgm; };
  auto stage = [&](int img, int hh, int tile, int buf) {
    int64_t k_elems;
    if constexpr (X3) {
      int sgm, w;
      x3_where(x3_t0 + tile, sgm, w);
      const int plane = ((img == 0 ? 0x120100 : 0x102010) >> (4 * sgm)) & 3;
      k_elems = (int64_t)w * kTileK * (TN ? (img == 0 ? g.lda : g.ldb) : 1) + plane * (img == 0 ? g.x3_plane_a : g.x3_plane_b);
    } else {
      k_elems = (int64_t)(k_begin + tile * kTileK) * (TN ? (img == 0 ? g.lda : g.ldb) : 1);
    }
    const uint32_t kb = tile < n_ktiles ? (uint32_t)(k_elems * 2) : 0x80000000u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t voff = (img == 0 ? va[i] + hh * d_a : vb[i] + hh * d_b) + kb;
      dma(img == 0 ? srd_a : srd_b, voff, lds_piece + i * 1024 + img * (4 * IMG) + buf * (2 * IMG) + hh * IMG);
    }
  };

  // X3, general loop, K-major: walk positions packed as 8 w + product -- of the step being multiplied (c), of step + 1
  // (a: its A images are staged) and of step + 2 (b: its B images) -- advanced once per step by VALUE (no references:
  // hipcc puts referenced loop-carried scalars on the stack, and this loop cannot afford a stack): no division per DMA
  const bool x3_run = X3 && !F6 && x3_kmajor;
  auto x3_pack = [&](int v) { const int qd = x3_div(v); return qd * 8 + (v - qd * g.x3_products); };
  auto x3_step = [&](int pos) { return ((pos & 7) + 1 == g.x3_products) ? (pos & ~7) + 8 : pos + 1; };
  int xp_c = 0, xp_a = 0, xp_b = 0;
  if (x3_run) { xp_c = x3_pack(x3_t0); xp_a = x3_pack(x3_t0 + 1); xp_b = x3_pack(x3_t0 + 2); }
  // X3 fast walk: the image of plane `plane` (compile time at the call sites), K-tile w of the plane
  auto stage_pw = [&](int img, int hh, int plane, int w, int tile, int buf) {
    const int64_t k_elems = (int64_t)w * kTileK * (TN ? (img == 0 ? g.lda : g.ldb) : 1) +
                            plane * (img == 0 ? g.x3_plane_a : g.x3_plane_b);
    const uint32_t kb = tile < n_ktiles ? (uint32_t)(k_elems * 2) : 0x80000000u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t voff = (img == 0 ? va[i] + hh * d_a : vb[i] + hh * d_b) + kb;
      dma(img == 0 ? srd_a : srd_b, voff, lds_piece + i * 1024 + img * (4 * IMG) + buf * (2 * IMG) + hh * IMG);
    }
  };

  // ---- fragment reads: lane (l31, h) holds k = 16*ks + 8*h .. +7 of image row l31 ----
  const int x = (l31 >> 1) & 7;
  const unsigned char *a_rd = smem + (grp * 64 + l31) * 128;
  const unsigned char *b_rd = smem + 4 * IMG + (wc * 32 + l31) * 128;
  int sw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sw[ks] = ((2 * ks + h) ^ x) * 16;
  // TN: lane 4q+p of a 16-lane group addresses row r0+q, columns 4p..4p+3 of a 4x16 block and
  // receives column (lane & 15) of its 4 rows; a fragment = rows 8h..8h+3 and 8h+4..8h+7 of
  // k-step ks for the 32 columns lane&31 (cdna guide T10, image (b))
  const int tq = (lane >> 2) & 3, tp = lane & 3, tmb = (lane >> 4) & 1;
  auto tr_off = [&](int c0, int sh) {
    const int row = 8 * h + 4 * sh + tq;
    const int swz = (tq << 2) | ((2 * h + sh) & 3);
    return 256 * row + 16 * ((c0 + tmb * 2 + (tp >> 1)) ^ swz) + 8 * (tp & 1);
  };
  int ta[2][2], tb[2];
#pragma unroll
  for (int sh = 0; sh < 2; ++sh) {
    ta[0][sh] = tr_off(grp * 8, sh);
    ta[1][sh] = tr_off(grp * 8 + 4, sh);
    tb[sh] = tr_off(wc * 4, sh);
  }
  auto tr_read = [&](const unsigned char *img, int off0, int off1) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4 *)(img + off0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4 *)(img + off1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_a = [&](int buf, int hh, int mi, int ks) {
    if constexpr (!TN)
      return *reinterpret_cast<const bf16x8 *>(a_rd + buf * (2 * IMG) + hh * IMG + mi * 4096 + sw[ks]);
    else
      return tr_read(smem + buf * (2 * IMG) + hh * IMG + ks * 4096, ta[mi][0], ta[mi][1]);
  };
  auto read_b = [&](int buf, int hh, int ks) {
    if constexpr (!TN)
      return *reinterpret_cast<const bf16x8 *>(b_rd + buf * (2 * IMG) + hh * IMG + sw[ks]);
    else
      return tr_read(smem + 4 * IMG + buf * (2 * IMG) + hh * IMG + ks * 4096, tb[0], tb[1]);
  };

  // S16 fragment reads: lane (l15, q) holds k = 32*ks2 + 8*q .. +7 of image row l15 (+ 16-row block)
  const int l15 = lane & 15, q16 = lane >> 4;
  const unsigned char *a16_rd = smem + (grp * 64 + l15) * 128;
  const unsigned char *b16_rd = smem + 4 * IMG + (wc * 32 + l15) * 128;
  const int sw16[2] = {((q16) ^ ((l15 >> 1) & 7)) * 16, ((4 + q16) ^ ((l15 >> 1) & 7)) * 16};
  // TN: a 16x16x32 fragment = k-rows 8q .. 8q+7 of the 32-deep k-step for the 16 columns l15 of a column
  // block: the 16-lane group q reads rows 8q+4sh .. +3 (sh = 0, 1) x 16 columns through ds_read_b64_tr_b16;
  // a 32-lane half takes two blocks 8 rows apart in the same columns (conflict-free on this image, T10)
  auto tr_off16 = [&](int c0, int sh) {
    const int row = 8 * q16 + 4 * sh + tq;
    const int swz = (tq << 2) | ((2 * q16 + sh) & 3);
    return 256 * row + 16 * ((c0 + (tp >> 1)) ^ swz) + 8 * (tp & 1);
  };
  int ta16[4][2], tb16[2][2];
#pragma unroll
  for (int sh = 0; sh < 2; ++sh) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) ta16[rb][sh] = tr_off16(grp * 8 + 2 * rb, sh);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) tb16[cb][sh] = tr_off16(wc * 4 + 2 * cb, sh);
  }
  auto read_a16 = [&](int buf, int hh, int rb, int ks2) {
    if constexpr (!TN)
      return *reinterpret_cast<const bf16x8 *>(a16_rd + buf * (2 * IMG) + hh * IMG + rb * 2048 + sw16[ks2]);
    else
      return tr_read(smem + buf * (2 * IMG) + hh * IMG + ks2 * 8192, ta16[rb][0], ta16[rb][1]);
  };
  auto read_b16 = [&](int buf, int hh, int cb, int ks2) {
    if constexpr (!TN)
      return *reinterpret_cast<const bf16x8 *>(b16_rd + buf * (2 * IMG) + hh * IMG + cb * 2048 + sw16[ks2]);
    else
      return tr_read(smem + 4 * IMG + buf * (2 * IMG) + hh * IMG + ks2 * 8192, tb16[cb][0], tb16[cb][1]);
  };

  f32x16 acc[S16 ? 1 : 4][S16 ? 1 : 2];
  f32x4 acc16[S16 ? 8 : 1][S16 ? 4 : 1];
#pragma unroll
  for (int i = 0; i < (S16 ? 1 : 4); ++i)
#pragma unroll
    for (int j = 0; j < (S16 ? 1 : 2); ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (S16 ? 8 : 1); ++i)
#pragma unroll
    for (int j = 0; j < (S16 ? 4 : 1); ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][4], fb0[4], fb1[4];     // 32x32x16: [mi][ks], [ks];  16x16x32: fa[ks2][rb], fb0 = [cb of half 0..1][ks2] ...

  // Bias gradient riding along (k-strided form): db[n] = sum_k B[k][n].  The tiles_m blocks and
  // two row groups that share a B tile split its K-tiles between them (tile t belongs to
  // (tm, grp) = (t % (2*tiles_m)) >> 1, & 1); the owner adds its fragments' 8 k-values per lane
  // in the read part of the phase, where the wave only waits anyway.
  float cs[2] = {0.f, 0.f};
  float cs16[4] = {0.f, 0.f, 0.f, 0.f};      // S16: column blocks 0, 1 of B-h0 and of B-h1
  // (X3, k-contiguous form too: B[N][K] there, and a B fragment has the register layout the transposed read gives
  // the k-strided form, so the same per-lane sums serve: colsum[n] = sum_k B[n][k])
  const bool cs_on = (TN || (NTCS && X3 && S16)) && cs_row != nullptr;
  const int cs_period = 2 * g.tiles_m;
  const int cs_owner = (2 * tm + grp + cs_period - (k_begin / kTileK) % cs_period) % cs_period;   // in K-tiles from k_begin
  // acc + the 8 values of a fragment: four v_dot2c_f32_bf16 against (1, 1) -- no conversions, no temporaries (the
  // shift / mask / add form of rounds 1-3 took 17 VALU instructions and 16 VGPRs of temporaries per owned K-tile pair)
  auto dot_sum = [&](const bf16x8 &f, float acc) {
#ifdef CDML_F16X2
    const half8 hf = __builtin_bit_cast(half8, f);
    const half2v one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const half2v pr = {hf[2 * e], hf[2 * e + 1]};
      acc = __builtin_amdgcn_fdot2(pr, one, acc, false);
    }
#else
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    const bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bf16x2 pr = {f[2 * e], f[2 * e + 1]};
      acc = __builtin_amdgcn_fdot2_f32_bf16(pr, one, acc, false);
    }
#endif
    return acc;
  };
  // K-tiles until the next one this (tile, row group) owns (0x40000000: none -- no bias gradient asked for)
  int cs_left = __builtin_amdgcn_readfirstlane(cs_on ? cs_owner : 0x40000000);
  auto frag_sum = [&](const bf16x8 &f) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)f[e];
    return s;
  };

  // The K-tile in two phases: phase A = Q00 + Q01 (16 fragment
  // reads, 16 MFMAs), phase B = Q11 + Q10 (8 reads, 16 MFMAs): half the barriers, and the
  // read parts have a 512-cycle MFMA part of the other group to hide under.  Fragment reads
  // are retired (lgkmcnt(0)) BEFORE the phase's first barrier, so an image is dead one
  // barrier after the lagging group read it:
  //   phase A of tile T issues A-h0, A-h1 of tile T+1 (other buffer; last read in tile T-1);
  //   phase B of tile T issues B-h0, B-h1 of tile T+2 (this buffer; read in phase A of T);
  //   phase B waits vmcnt(6): B-h0, B-h1, A-h0 of T+1 landed (read in phase A of T+1);
  //   phase A waits vmcnt(8): A-h1 of T landed (read in phase B of T).
  // Fragments as in/out operands of an empty asm: hipcc has to have them in registers here,
  // so the LDS reads are waited for BEFORE the barrier that follows (it would otherwise sink
  // the waits into the MFMA part, and the images are refilled one barrier after their reads).
  auto pin_a = [&]() {
    asm volatile("" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]),
                      "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]));
  };
  auto pin_b = [&]() {
    asm volatile("" : "+v"(fb0[0]), "+v"(fb0[1]), "+v"(fb0[2]), "+v"(fb0[3]),
                      "+v"(fb1[0]), "+v"(fb1[1]), "+v"(fb1[2]), "+v"(fb1[3]));
  };
  // 16x16x32 form of the K-tile: fb0[2*cb + ks2] = column block cb (0, 1) of B-h0, fb1 likewise of B-h1,
  // fa[ks2][rb] = row block rb (0..3) of the phase's A half image
  auto do_tile2_s16 = [&](const int buf, const int tile) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        fb0[2 * cb + ks2] = read_b16(buf, 0, cb, ks2);
        fb1[2 * cb + ks2] = read_b16(buf, 1, cb, ks2);
      }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) fa[ks2][rb] = read_a16(buf, 0, rb, ks2);
    if (X3 && x3_run) {
      const int pl = (0x120100 >> (4 * (xp_a & 7))) & 3;
      stage_pw(0, 0, pl, xp_a >> 3, tile + 1, buf ^ 1);
      stage_pw(0, 1, pl, xp_a >> 3, tile + 1, buf ^ 1);
    } else {
      stage(0, 0, tile + 1, buf ^ 1);
      stage(0, 1, tile + 1, buf ^ 1);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    pin_b();
    pin_a();
    // (the owner test as a countdown in an SGPR: `tile % cs_period` was a dozen scalar instructions per K-tile)
    const bool cs_mine = cs_left == 0;
    cs_left = cs_mine ? cs_period - 1 : cs_left - 1;
    if (cs_mine && (!X3 || ((0xB >> (x3_run ? (xp_c & 7) : x3_segment(x3_t0 + tile))) & 1))) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          cs16[cb] = dot_sum(fb0[2 * cb + ks2], cs16[cb]);
          cs16[2 + cb] = dot_sum(fb1[2 * cb + ks2], cs16[2 + cb]);
        }
    }
    CDML_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          acc16[rb][cb] = CDML_MFMA16(fa[ks2][rb], fb0[2 * cb + ks2], acc16[rb][cb]);
          acc16[rb][2 + cb] = CDML_MFMA16(fa[ks2][rb], fb1[2 * cb + ks2], acc16[rb][2 + cb]);
        }
    __builtin_amdgcn_s_setprio(0);
    CDML_BARRIER();
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) fa[ks2][rb] = read_a16(buf, 1, rb, ks2);
    if (X3 && x3_run) {
      const int pl = (0x102010 >> (4 * (xp_b & 7))) & 3;
      stage_pw(1, 0, pl, xp_b >> 3, tile + 2, buf);
      stage_pw(1, 1, pl, xp_b >> 3, tile + 2, buf);
    } else {
      stage(1, 0, tile + 2, buf);
      stage(1, 1, tile + 2, buf);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    pin_a();
    CDML_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          acc16[4 + rb][cb] = CDML_MFMA16(fa[ks2][rb], fb0[2 * cb + ks2], acc16[4 + rb][cb]);
          acc16[4 + rb][2 + cb] = CDML_MFMA16(fa[ks2][rb], fb1[2 * cb + ks2], acc16[4 + rb][2 + cb]);
        }
    __builtin_amdgcn_s_setprio(0);
    CDML_BARRIER();
    if (X3 && x3_run) { xp_c = x3_step(xp_c); xp_a = x3_step(xp_a); xp_b = x3_step(xp_b); }
  };
  // X3, K-major, six products: step S (compile time) of the six-step period of K-tile w.  Same phases as
  // do_tile2_s16; what is known at compile time here -- which planes a step multiplies and stages for -- lets the
  // loop SKIP the DMA of an image its LDS buffer already holds (B = hi on steps 0, 2, 4: the B image staged for
  // step 0 serves 2 and 4; A = hi on steps 1 and 3) with the counted waits adjusted as immediates: 9 image loads
  // per K-tile instead of 12, no division per DMA.  (Decided at run time the same skipping cost 19 %.)
  auto step6 = [&](auto Sc, const int tile, const int w) {
    constexpr int S = decltype(Sc)::value;
    constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};      // (A, B) planes of the six products
    constexpr int buf = S & 1;
    constexpr bool skip_a = (S >= 1 && S <= 4) && PA[(S + 1) % 6] == PA[(S + 5) % 6];   // A of step + 1 == A of step - 1
    constexpr bool skip_b = (S <= 3) && PB[(S + 2) % 6] == PB[S];                       // B of step + 2 == B of this step
    constexpr int SP = (S + 5) % 6;                                                      // the previous step
    constexpr bool b_next_issued = !((SP <= 3) && PB[(SP + 2) % 6] == PB[SP]);           // B of step + 1: staged by it?
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        fb0[2 * cb + ks2] = read_b16(buf, 0, cb, ks2);
        fb1[2 * cb + ks2] = read_b16(buf, 1, cb, ks2);
      }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) fa[ks2][rb] = read_a16(buf, 0, rb, ks2);
    if constexpr (!skip_a) {
      stage_pw(0, 0, PA[(S + 1) % 6], w + (S == 5 ? 1 : 0), tile + 1, buf ^ 1);
      stage_pw(0, 1, PA[(S + 1) % 6], w + (S == 5 ? 1 : 0), tile + 1, buf ^ 1);
    }
    // in flight behind A-h1 of this step (read in phase B): B of step + 1 and A of step + 1, where issued
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (int)b_next_issued + 4 * (int)!skip_a) : "memory");
    pin_b();
    pin_a();
    if (cs_on && ((0xB >> S) & 1) && (tile % cs_period) == cs_owner) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          cs16[cb] += frag_sum(fb0[2 * cb + ks2]);
          cs16[2 + cb] += frag_sum(fb1[2 * cb + ks2]);
        }
    }
    CDML_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          acc16[rb][cb] = CDML_MFMA16(fa[ks2][rb], fb0[2 * cb + ks2], acc16[rb][cb]);
          acc16[rb][2 + cb] = CDML_MFMA16(fa[ks2][rb], fb1[2 * cb + ks2], acc16[rb][2 + cb]);
        }
    __builtin_amdgcn_s_setprio(0);
    CDML_BARRIER();
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) fa[ks2][rb] = read_a16(buf, 1, rb, ks2);
    if constexpr (!skip_b) {
      stage_pw(1, 0, PB[(S + 2) % 6], w + (S >= 4 ? 1 : 0), tile + 2, buf);
      stage_pw(1, 1, PB[(S + 2) % 6], w + (S >= 4 ? 1 : 0), tile + 2, buf);
    }
    // in flight behind A-h0 and B of step + 1 (read in its phase A): A-h1 of step + 1 and B of step + 2, where issued
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (int)!skip_a + 4 * (int)!skip_b) : "memory");
    pin_a();
    CDML_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          acc16[4 + rb][cb] = CDML_MFMA16(fa[ks2][rb], fb0[2 * cb + ks2], acc16[4 + rb][cb]);
          acc16[4 + rb][2 + cb] = CDML_MFMA16(fa[ks2][rb], fb1[2 * cb + ks2], acc16[4 + rb][2 + cb]);
        }
    __builtin_amdgcn_s_setprio(0);
    CDML_BARRIER();
  };
  auto do_tile2 = [&](const int buf, const int tile) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb0[ks] = read_b(buf, 0, ks);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb1[ks] = read_b(buf, 1, ks);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[mi][ks] = read_a(buf, 0, mi, ks);
    stage(0, 0, tile + 1, buf ^ 1);
    stage(0, 1, tile + 1, buf ^ 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    pin_b();
    pin_a();
    if (TN && cs_on && (tile % cs_period) == cs_owner && (!X3 || ((0xB >> x3_segment(x3_t0 + tile)) & 1))) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        cs[0] += frag_sum(fb0[ks]);
        cs[1] += frag_sum(fb1[ks]);
      }
    }
    CDML_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][ks], fb0[ks], acc[mi][0], 0, 0, 0);
        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][ks], fb1[ks], acc[mi][1], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
    CDML_BARRIER();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[mi][ks] = read_a(buf, 1, mi, ks);
    stage(1, 0, tile + 2, buf);
    stage(1, 1, tile + 2, buf);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    pin_a();
    CDML_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc[2 + mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][ks], fb0[ks], acc[2 + mi][0], 0, 0, 0);
        acc[2 + mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi][ks], fb1[ks], acc[2 + mi][1], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
    CDML_BARRIER();
  };

  // ---- X3, six products, whole periods: the RESIDENT-PLANE walk (R6; round 4) ------------------------------------------
  // What bounds the ping-pong loop is the length of a phase's READ part -- an in-order wave issues its fragment reads,
  // its LDS-DMA pieces (60-185 cycles each to the issuing wave), their address arithmetic and the walk's scalar
  // bookkeeping one after the other, and that sum, not the other group's 512-cycle MFMA part, sets the barrier interval
  // (profiles/r04_tn_ablations.txt: without the DMA issues the k-strided product runs 21 % faster, with half the A
  // reads 6 %, without the bias-gradient sums 5 %; profiles/r04_tn_lds_counters.csv: the LDS array itself is 20-28 % busy,
  // no bank conflicts).  The six plane products of a K-tile use A0 three times, A1 twice, A2 once (B alike), and the
  // K-major walks of round 3 staged 12 (general loop) or 9 (F6) images per K-tile for them.  Here every plane image of a
  // K-tile is staged ONCE -- 6 images, the minimum -- by keeping the planes RESIDENT: three A slots and two B slots fill
  // the CU's whole 160 KiB of LDS, the products are ordered B-constant
  //     (A0,B0) (A1,B0) (A2,B0) (A0,B1) (A1,B1) (A0,B2)          [hh, mh, lh, hm, mm, hl: the same six products]
  // so that the B FRAGMENTS are read from LDS on three of the six steps only and stay in registers in between (their
  // slot is free again right after that read), and every one of the period's 12 phases issues exactly ONE half image
  // (2 pieces per wave) with its source offset in an SGPR: half the DMA pieces of the general loop, two thirds of F6's,
  // 120 fragment reads per period for 144, no division, no per-piece VALU.
  //   phase p = 2 S + h (step S, row half h) reads A(S) half h [+ the B fragments at p = 0, 6, 10] and issues:
  //     p:     0      1      2      3      4      5       6       7       8       9       10      11
  //     load:  A2.h1  B1.h0  B1.h1  B2.h0  B2.h1  A0'.h0  A0'.h1  B0'.h0  B0'.h1  A1'.h0  A1'.h1  A2'.h0     (' = next K-tile)
  //     need:  5      6      6      10     10     12      13      12      12      14      15      16         (first read, phase)
  //     vmcnt: 8      8      8      8      8      6       8       10      12      10      12      6          (what may stay in flight)
  //   Slots (parity = K-tile & 1 of the block's walk): A1 always slot 1; A0 / A2 slots 0 / 2, swapped on odd K-tiles
  //   (A0' is loaded into the slot A2 left at phase 5, A2' into the one A0 leaves at phase 11); B0, B1, B2, B0', ...
  //   alternate between the two B slots.  Every overwrite is issued at least one phase after the last read of what it
  //   replaces (reads are retired before a phase's first barrier, the lagging group is one barrier behind: the rule of
  //   the schedule at the top of this file), every image is waited for one phase before its first read.
  constexpr bool kR6 = X3 && S16 && R6;
  if constexpr (kR6) {
    constexpr int kPer = kF16 ? 3 : 6;                     // steps per K-tile: the plane products
    const int n_per = n_ktiles / kPer;                     // whole periods (host)
    if (n_per > 0) {                                       // (an empty split writes zeros)
      const int w_first = x3_t0 / kPer, w_last = w_first + n_per - 1;
      const uint32_t ws_a = KI ? (uint32_t)(8 * g.lda * 2) : TN ? (uint32_t)(kTileK * g.lda * 2) : (uint32_t)(kTileK * 2);   // bytes per K-tile of a plane
      const uint32_t ws_b = KI ? (uint32_t)(8 * g.ldb * 2) : TN ? (uint32_t)(kTileK * g.ldb * 2) : (uint32_t)(kTileK * 2);
      const uint32_t ps_a = (uint32_t)(g.x3_plane_a * 2), ps_b = (uint32_t)(g.x3_plane_b * 2);
      const unsigned char *b16r = smem + 6 * IMG + (wc * 32 + l15) * 128;
      // k-strided form: the transposed reads address LDS as (lane offset register) + (16-bit immediate).  160 KiB need
      // three 64-KiB windows: A slots 0, 1 from the lane offsets as they are, A slot 2 and the B slots from copies moved
      // by 64 / 96 KiB -- made opaque to the optimizer, which otherwise forms one hoisted register per DISTINCT large
      // constant (dozens over the 24 unrolled phases: 61 spilled VGPRs, measured)
      int ta_w2[4][2], tb_w[2][2];
      if constexpr (TN && !KI) {
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            ta_w2[rb][sh] = ta16[rb][sh] + 4 * IMG;
            asm volatile("" : "+v"(ta_w2[rb][sh]));
          }
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            tb_w[cb][sh] = tb16[cb][sh] + 6 * IMG;
            asm volatile("" : "+v"(tb_w[cb][sh]));
          }
        }
      }
      // k8-interleaved images [8 k-groups][128 columns][16 B]: lane (l15, q) reads k-group 4 ks2 + q of its column
      const unsigned char *ka_rd = smem + (q16 * 128 + grp * 64 + l15) * 16;
      const unsigned char *kb_rd = smem + 6 * IMG + (q16 * 128 + wc * 32 + l15) * 16;
      auto rd_a = [&](int slot, int hh, int rb, int ks2) {
        if constexpr (!TN)
          return *reinterpret_cast<const bf16x8 *>(a16_rd + slot * 2 * IMG + hh * IMG + rb * 2048 + sw16[ks2]);
        else if constexpr (KI)
          return *reinterpret_cast<const bf16x8 *>(ka_rd + slot * 2 * IMG + hh * IMG + ks2 * 8192 + rb * 256);
        else if (slot < 2)
          return tr_read(smem + slot * 2 * IMG + hh * IMG + ks2 * 8192, ta16[rb][0], ta16[rb][1]);
        else
          return tr_read(smem + hh * IMG + ks2 * 8192, ta_w2[rb][0], ta_w2[rb][1]);
      };
      auto rd_b = [&](int slot, int hh, int cb, int ks2) {
        if constexpr (!TN)
          return *reinterpret_cast<const bf16x8 *>(b16r + slot * 2 * IMG + hh * IMG + cb * 2048 + sw16[ks2]);
        else if constexpr (KI)
          return *reinterpret_cast<const bf16x8 *>(kb_rd + slot * 2 * IMG + hh * IMG + ks2 * 8192 + cb * 256);
        else
          return tr_read(smem + slot * 2 * IMG + hh * IMG + ks2 * 8192, tb_w[cb][0], tb_w[cb][1]);
      };
      // one half image: plane pl, half hh of operand img at K-tile offset kw (= w * ws_x) -> slot.  The SGPR part of the
      // source offset carries only what moves ALONG a row (plane, K-tile; k-strided form: the column half too); the
      // k-contiguous form's second half is 64 / 32 ROWS further, and rows past the operand's last one are zero-filled
      // by the descriptor's range check on the per-lane offset -- so there the half lives in a second lane register
      // (whether the check also sees the scalar offset is not something to depend on)
      uint32_t va_h1[2] = {va[0] + d_a, va[1] + d_a}, vb_h1[2] = {vb[0] + d_b, vb[1] + d_b};
      auto issue = [&](int img, int pl, int hh, int slot, uint32_t kw) {
        const uint32_t so = (img == 0 ? pl * ps_a : pl * ps_b) + kw + (KI ? hh * 2048u : TN ? hh * 256u : 0u);
        const uint32_t dst = lds_piece + (img == 0 ? 0 : 6 * IMG) + slot * 2 * IMG + hh * IMG;
        const bool h1 = !TN && hh == 1;
        dma_s(img == 0 ? srd_a : srd_b, img == 0 ? (h1 ? va_h1[0] : va[0]) : (h1 ? vb_h1[0] : vb[0]), so, dst);
        dma_s(img == 0 ? srd_a : srd_b, img == 0 ? (h1 ? va_h1[1] : va[1]) : (h1 ? vb_h1[1] : vb[1]), so, dst + 1024);
      };
      if constexpr (kF16) {
      // ---- TWO planes, THREE products: the resident-plane walk of the fp16 form (R3; round 6) ----------------------------
      // The same idea on the period (A0,B0) (A1,B0) (A0,B1) [hi.hi, lo.hi, hi.lo]: every plane image of a K-tile is staged
      // ONCE -- 4 images = 8 half images in the period's 6 phases, where the general loop stages 12 -- into the same five
      // slots: A0 alternates between A slots 0 and 2 with the K-tile's parity (it is read on steps 0 AND 2, so the next
      // K-tile's A0 needs a slot of its own), A1 lives in A slot 1, B0 / B1 in B slots 0 / 1; the B fragments are read on
      // steps 0 and 2 only and stay in registers in between.
      //   phase p = 2 S + h reads A(S) half h [+ the B fragments at p = 0, 4] and issues:
      //     p:      0             1        2        3        4               5
      //     load:   B1.h0 B1.h1   A0'.h0   A0'.h1   B0'.h0   B0'.h1 A1'.h0   A1'.h1          (' = next K-tile)
      //     need:   4     4       6        7        6        6      8        9               (first read, phase)
      //     vmcnt:  8             8        8        6        10              4               (what may stay in flight)
      //   Every overwrite is issued at least one phase after the last read of what it replaces (B1 after phase 4 of the K-tile
      //   before; A0' into the slot A0 of the K-tile BEFORE left at phases 4 / 5; B0' after phase 0; A1' halves after phases
      //   2 / 3), every image is waited for one phase before its first read (replayed from this source by
      //   tests/test_r6_schedule.py).
      static_assert(!NARROW && !KI, "the fp16 form has the full tile on row-major operands only");
      auto phase3 = [&](auto phc, auto parc, const uint32_t kwb_c, const uint32_t kwa_n, const uint32_t kwb_n, const bool own) {
        constexpr int PH = decltype(phc)::value, PAR = decltype(parc)::value;
        constexpr int S = PH >> 1, HALF = PH & 1;
        constexpr int PA3[3] = {0, 1, 0}, PB3[3] = {0, 0, 1};
        constexpr int SLOT_A3[2][2] = {{0, 1}, {2, 1}};          // [parity][plane]
        constexpr int sa = SLOT_A3[PAR][PA3[S]];
        constexpr bool rdB = HALF == 0 && (S == 0 || S == 2);
        if constexpr (rdB) {
          constexpr int sb = PB3[S];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
              fb0[2 * cb + ks2] = rd_b(sb, 0, cb, ks2);
              fb1[2 * cb + ks2] = rd_b(sb, 1, cb, ks2);
            }
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) fa[ks2][rb] = rd_a(sa, HALF, rb, ks2);
        // Two placements of the period's eight loads (SCH).  0: the table above.  1: one load in each of the fragment-heavy
        // phases 0 / 4 (they also read the B fragments) and two in phases 1 / 3 -- B1.h0 | B1.h1 A0'.h0 | A0'.h1 | B0'.h0 B0'.h1 |
        // A1'.h0 | A1'.h1, waits 6 8 8 8 10 4: FC1 554.7 against 560.7 us, the other k-contiguous products even
        // (gpurun call r06z, alternating processes).  The k-contiguous kernels run 1; the k-strided kernel keeps 0 (with 1 it
        // needs 20 B of scratch: 256 VGPRs).
        constexpr int SCH = TN ? 0 : 1;
        if constexpr (SCH == 0 && PH == 0) { issue(1, 1, 0, 1, kwb_c); issue(1, 1, 1, 1, kwb_c); }
        if constexpr (SCH == 0 && PH == 1) { issue(0, 0, 0, SLOT_A3[PAR ^ 1][0], kwa_n); }
        if constexpr (SCH == 0 && PH == 2) { issue(0, 0, 1, SLOT_A3[PAR ^ 1][0], kwa_n); }
        if constexpr (SCH == 0 && PH == 3) { issue(1, 0, 0, 0, kwb_n); }
        if constexpr (SCH == 0 && PH == 4) { issue(1, 0, 1, 0, kwb_n); issue(0, 1, 0, 1, kwa_n); }
        if constexpr (SCH == 0 && PH == 5) { issue(0, 1, 1, 1, kwa_n); }
        if constexpr (SCH == 1 && PH == 0) { issue(1, 1, 0, 1, kwb_c); }
        if constexpr (SCH == 1 && PH == 1) { issue(1, 1, 1, 1, kwb_c); issue(0, 0, 0, SLOT_A3[PAR ^ 1][0], kwa_n); }
        if constexpr (SCH == 1 && PH == 2) { issue(0, 0, 1, SLOT_A3[PAR ^ 1][0], kwa_n); }
        if constexpr (SCH == 1 && PH == 3) { issue(1, 0, 0, 0, kwb_n); issue(1, 0, 1, 0, kwb_n); }
        if constexpr (SCH == 1 && PH == 4) { issue(0, 1, 0, 1, kwa_n); }
        if constexpr (SCH == 1 && PH == 5) { issue(0, 1, 1, 1, kwa_n); }
        constexpr int VM3[2][6] = {{8, 8, 8, 6, 10, 4}, {6, 8, 8, 8, 10, 4}};
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM3[SCH][PH]) : "memory");
        if constexpr (rdB) pin_b();
        pin_a();
        if constexpr (rdB) {
          if (own) {                                       // bias gradient: this (tile, row group) owns the K-tile
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
              for (int ks2 = 0; ks2 < 2; ++ks2) {
                cs16[cb] = dot_sum(fb0[2 * cb + ks2], cs16[cb]);
                cs16[2 + cb] = dot_sum(fb1[2 * cb + ks2], cs16[2 + cb]);
              }
          }
        }
        CDML_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              if constexpr (kSwap) {                       // (the miner / kNN filter: the transposed block, a row's columns per lane)
                acc16[4 * HALF + rb][cb] = CDML_MFMA16(fb0[2 * cb + ks2], fa[ks2][rb], acc16[4 * HALF + rb][cb]);
                acc16[4 * HALF + rb][2 + cb] = CDML_MFMA16(fb1[2 * cb + ks2], fa[ks2][rb], acc16[4 * HALF + rb][2 + cb]);
              } else {
                acc16[4 * HALF + rb][cb] = CDML_MFMA16(fa[ks2][rb], fb0[2 * cb + ks2], acc16[4 * HALF + rb][cb]);
                acc16[4 * HALF + rb][2 + cb] = CDML_MFMA16(fa[ks2][rb], fb1[2 * cb + ks2], acc16[4 * HALF + rb][2 + cb]);
              }
            }
        __builtin_amdgcn_s_setprio(0);
        CDML_BARRIER();
      };
      auto period3 = [&](auto parc, const int w, const bool own) {
        const int wn = min(w + 1, w_last);                 // past the end: a valid K-tile again (its images are never read)
        const uint32_t kwb_c = (uint32_t)w * ws_b, kwa_n = (uint32_t)wn * ws_a, kwb_n = (uint32_t)wn * ws_b;
        phase3(std::integral_constant<int, 0>{}, parc, kwb_c, kwa_n, kwb_n, own);
        phase3(std::integral_constant<int, 1>{}, parc, kwb_c, kwa_n, kwb_n, own);
        phase3(std::integral_constant<int, 2>{}, parc, kwb_c, kwa_n, kwb_n, own);
        phase3(std::integral_constant<int, 3>{}, parc, kwb_c, kwa_n, kwb_n, own);
        phase3(std::integral_constant<int, 4>{}, parc, kwb_c, kwa_n, kwb_n, own);
        phase3(std::integral_constant<int, 5>{}, parc, kwb_c, kwa_n, kwb_n, own);
      };
      // R3 prologue: the steady state at phase 0 of the first K-tile (parity 0): what phases 1 .. 5 of a previous period
      // would have issued, in their order, then the wait of its phase 5
      {
        const uint32_t ka3 = (uint32_t)w_first * ws_a, kb3 = (uint32_t)w_first * ws_b;
        issue(0, 0, 0, 0, ka3); issue(0, 0, 1, 0, ka3);
        issue(1, 0, 0, 0, kb3); issue(1, 0, 1, 0, kb3);
        issue(0, 1, 0, 1, ka3); issue(0, 1, 1, 1, ka3);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      CDML_BARRIER();
      if (grp == 1) CDML_BARRIER();                        // group 1 runs one barrier behind
      int cs_wait3 = __builtin_amdgcn_readfirstlane(cs_on ? (2 * tm + grp + cs_period - w_first % cs_period) % cs_period : 0x40000000);
      for (int w = w_first;;) {
        bool own = cs_wait3 == 0;
        cs_wait3 = own ? cs_period - 1 : cs_wait3 - 1;
        period3(std::integral_constant<int, 0>{}, w, own);
        if (++w > w_last) break;
        own = cs_wait3 == 0;
        cs_wait3 = own ? cs_period - 1 : cs_wait3 - 1;
        period3(std::integral_constant<int, 1>{}, w, own);
        if (++w > w_last) break;
      }
      } else if constexpr (!NARROW) {
      // the phase: PH = 2 S + half (compile time), PAR = parity of the K-tile; kw_* = byte offsets of this / the next K-tile
      auto phase = [&](auto phc, auto parc, const uint32_t kwa_c, const uint32_t kwb_c, const uint32_t kwa_n,
                       const uint32_t kwb_n, const bool own) {
        constexpr int PH = decltype(phc)::value, PAR = decltype(parc)::value;
        constexpr int S = PH >> 1, HALF = PH & 1;
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {0, 0, 0, 1, 1, 2};
        constexpr int SLOT_A[2][3] = {{0, 1, 2}, {2, 1, 0}};     // [parity][plane]
        constexpr int sa = SLOT_A[PAR][PA[S]];
        constexpr bool rdB = HALF == 0 && (S == 0 || S == 3 || S == 5);
        if constexpr (rdB) {
          constexpr int sb = (PB[S] & 1) ^ PAR;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
              fb0[2 * cb + ks2] = rd_b(sb, 0, cb, ks2);
              fb1[2 * cb + ks2] = rd_b(sb, 1, cb, ks2);
            }
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) fa[ks2][rb] = rd_a(sa, HALF, rb, ks2);
        if constexpr (PH == 0) issue(0, 2, 1, SLOT_A[PAR][2], kwa_c);
        if constexpr (PH == 1) issue(1, 1, 0, 1 ^ PAR, kwb_c);
        if constexpr (PH == 2) issue(1, 1, 1, 1 ^ PAR, kwb_c);
        if constexpr (PH == 3) issue(1, 2, 0, PAR, kwb_c);
        if constexpr (PH == 4) issue(1, 2, 1, PAR, kwb_c);
        if constexpr (PH == 5) issue(0, 0, 0, SLOT_A[PAR ^ 1][0], kwa_n);
        if constexpr (PH == 6) issue(0, 0, 1, SLOT_A[PAR ^ 1][0], kwa_n);
        if constexpr (PH == 7) issue(1, 0, 0, PAR ^ 1, kwb_n);
        if constexpr (PH == 8) issue(1, 0, 1, PAR ^ 1, kwb_n);
        if constexpr (PH == 9) issue(0, 1, 0, 1, kwa_n);
        if constexpr (PH == 10) issue(0, 1, 1, 1, kwa_n);
        if constexpr (PH == 11) issue(0, 2, 0, SLOT_A[PAR ^ 1][2], kwa_n);
        constexpr int VM[12] = {8, 8, 8, 8, 8, 6, 8, 10, 12, 10, 12, 6};
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM[PH]) : "memory");
        if constexpr (rdB) pin_b();
        pin_a();
        if constexpr (rdB) {
          if (own) {                                       // bias gradient: this (tile, row group) owns the K-tile
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
              for (int ks2 = 0; ks2 < 2; ++ks2) {
                cs16[cb] = dot_sum(fb0[2 * cb + ks2], cs16[cb]);
                cs16[2 + cb] = dot_sum(fb1[2 * cb + ks2], cs16[2 + cb]);
              }
          }
        }
        CDML_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              if constexpr (kSwap) {
                acc16[4 * HALF + rb][cb] =
                    CDML_MFMA16(fb0[2 * cb + ks2], fa[ks2][rb], acc16[4 * HALF + rb][cb]);
                acc16[4 * HALF + rb][2 + cb] =
                    CDML_MFMA16(fb1[2 * cb + ks2], fa[ks2][rb], acc16[4 * HALF + rb][2 + cb]);
              } else {
              acc16[4 * HALF + rb][cb] =
                  CDML_MFMA16(fa[ks2][rb], fb0[2 * cb + ks2], acc16[4 * HALF + rb][cb]);
              acc16[4 * HALF + rb][2 + cb] =
                  CDML_MFMA16(fa[ks2][rb], fb1[2 * cb + ks2], acc16[4 * HALF + rb][2 + cb]);
              }
            }
        __builtin_amdgcn_s_setprio(0);
        CDML_BARRIER();
      };
      // (round 6 pointed the last period's run-ahead loads -- seven half images "for the next K-tile" that nobody reads -- at the
      // NEXT tile's first K-tile, one resident block per CU walking its tiles: the next tile then starts without a prologue.
      // Measured on the K = 256 miner, with and without its epilogue: nothing, profiles/r06_miner_runahead_and_lds_free_
      // epilogue_ab.txt -- a prologue out of L2 is not what a short tile waits for.  Removed.)
      auto period = [&](auto parc, const int w, const bool own) {
        const int wn = min(w + 1, w_last);                 // past the end: a valid K-tile again (its images are never read)
        const uint32_t kwa_c = (uint32_t)w * ws_a, kwb_c = (uint32_t)w * ws_b;
        const uint32_t kwa_n = (uint32_t)wn * ws_a, kwb_n = (uint32_t)wn * ws_b;
        phase(std::integral_constant<int, 0>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 1>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 2>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 3>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 4>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 5>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 6>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 7>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 8>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 9>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 10>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
        phase(std::integral_constant<int, 11>{}, parc, kwa_c, kwb_c, kwa_n, kwb_n, own);
      };
      // prologue: the steady state at phase 0 of the first K-tile (parity 0): what phases 5 .. 11 of a previous period
      // would have issued, in their order, then the wait of its phase 11
      {
        const uint32_t ka = (uint32_t)w_first * ws_a, kb_ = (uint32_t)w_first * ws_b;
        issue(0, 0, 0, 0, ka); issue(0, 0, 1, 0, ka);
        issue(1, 0, 0, 0, kb_); issue(1, 0, 1, 0, kb_);
        issue(0, 1, 0, 1, ka); issue(0, 1, 1, 1, ka);
        issue(0, 2, 0, 2, ka);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
      CDML_BARRIER();
      if (grp == 1) CDML_BARRIER();                        // group 1 runs one barrier behind
      // the bias gradient's owner of K-tile w: (w mod 2 tiles_m) == 2 tm + grp -- every K-tile of a B column tile counted once
      // (through readfirstlane: the integer remainder runs on the VALU and would leave the countdown in a VGPR)
      int cs_wait = __builtin_amdgcn_readfirstlane(cs_on ? (2 * tm + grp + cs_period - w_first % cs_period) % cs_period : 0x40000000);
      for (int w = w_first;;) {
        bool own = cs_wait == 0;
        cs_wait = own ? cs_period - 1 : cs_wait - 1;
        period(std::integral_constant<int, 0>{}, w, own);
        if (++w > w_last) break;
        own = cs_wait == 0;
        cs_wait = own ? cs_period - 1 : cs_wait - 1;
        period(std::integral_constant<int, 1>{}, w, own);
        if (++w > w_last) break;
      }
      } else {
      // ---- the HALF TILE (NARROW): 128 rows x 256 columns, row groups of 64 rows -----------------------------------
      // For the last, partly filled round of a launch (FC1 at config 1: 640 tiles = 2.5 rounds of 256 CUs): its tiles are
      // computed as two half tiles each, on twice the CUs.  Halving the ROWS keeps every phase at 32 MFMAs per wave (a
      // phase is bound by its read part: halving the columns would halve the MFMAs of a phase and not its length); a K-tile
      // is the six steps themselves, one phase each.  Per K-tile 3 A half images (the only half there is) + 6 B half
      // images = 9 in 6 phases.  LDS: four A slots of 16 KiB (A0 alternates between slots 0 and 3, A1 / A2 swap slots 1 / 2
      // on odd K-tiles) + three B slots of 32 KiB (plane p in slot p) = the same 160 KiB.
      //   step S:   0            1            2        3     4            5
      //   reads:    A0 B0        A1           A2       A0 B1 A1           A0 B2
      //   issues:   B2.h0 B2.h1  B0'.h0 A0'   B0'.h1   A1'   B1'.h0 B1'.h1  A2'           (' = next K-tile)
      //   vmcnt:    10           8            10       12    12           8
      //   Every overwrite comes at least one step after the last read of what it replaces (B2 after step 5 of the K-tile
      //   before, B0' after step 0, A1' -> A2's slot after step 2, B1' after step 3, A2' -> A1's slot after step 4, A0' ->
      //   the other A0 slot), every image is waited for one step before its first read (replayed from this source by
      //   tests/test_r6_schedule.py).
      auto rd_an = [&](int slot, int rb, int ks2) {
        return *reinterpret_cast<const bf16x8 *>(a16_rd + slot * IMG + rb * 2048 + sw16[ks2]);
      };
      const unsigned char *b16n = smem + 4 * IMG + (wc * 32 + l15) * 128;
      auto rd_bn = [&](int slot, int hh, int cb, int ks2) {
        return *reinterpret_cast<const bf16x8 *>(b16n + slot * 2 * IMG + hh * IMG + cb * 2048 + sw16[ks2]);
      };
      auto issue_n = [&](int img, int pl, int hh, int slot, uint32_t kw) {
        const uint32_t so = (img == 0 ? pl * ps_a : pl * ps_b) + kw;
        const uint32_t dst = lds_piece + (img == 0 ? slot * IMG : 4 * IMG + slot * 2 * IMG + hh * IMG);
        const bool h1 = hh == 1;
        dma_s(img == 0 ? srd_a : srd_b, img == 0 ? va[0] : (h1 ? vb_h1[0] : vb[0]), so, dst);
        dma_s(img == 0 ? srd_a : srd_b, img == 0 ? va[1] : (h1 ? vb_h1[1] : vb[1]), so, dst + 1024);
      };
      auto phase_n = [&](auto sc, auto parc, const uint32_t kwb_c, const uint32_t kwa_n, const uint32_t kwb_n) {
        constexpr int S = decltype(sc)::value, PAR = decltype(parc)::value;
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {0, 0, 0, 1, 1, 2};
        constexpr int NSLOT_A[2][3] = {{0, 1, 2}, {3, 2, 1}};    // [parity][plane]
        constexpr bool rdB = S == 0 || S == 3 || S == 5;
        if constexpr (rdB) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
              fb0[2 * cb + ks2] = rd_bn(PB[S], 0, cb, ks2);
              fb1[2 * cb + ks2] = rd_bn(PB[S], 1, cb, ks2);
            }
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) fa[ks2][rb] = rd_an(NSLOT_A[PAR][PA[S]], rb, ks2);
        if constexpr (S == 0) { issue_n(1, 2, 0, 2, kwb_c); issue_n(1, 2, 1, 2, kwb_c); }
        if constexpr (S == 1) { issue_n(1, 0, 0, 0, kwb_n); issue_n(0, 0, 0, NSLOT_A[PAR ^ 1][0], kwa_n); }
        if constexpr (S == 2) { issue_n(1, 0, 1, 0, kwb_n); }
        if constexpr (S == 3) { issue_n(0, 1, 0, NSLOT_A[PAR ^ 1][1], kwa_n); }
        if constexpr (S == 4) { issue_n(1, 1, 0, 1, kwb_n); issue_n(1, 1, 1, 1, kwb_n); }
        if constexpr (S == 5) { issue_n(0, 2, 0, NSLOT_A[PAR ^ 1][2], kwa_n); }
        constexpr int NVM[6] = {10, 8, 10, 12, 12, 8};
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NVM[S]) : "memory");
        if constexpr (rdB) pin_b();
        pin_a();
        CDML_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              acc16[rb][cb] = CDML_MFMA16(fa[ks2][rb], fb0[2 * cb + ks2], acc16[rb][cb]);
              acc16[rb][2 + cb] = CDML_MFMA16(fa[ks2][rb], fb1[2 * cb + ks2], acc16[rb][2 + cb]);
            }
        __builtin_amdgcn_s_setprio(0);
        CDML_BARRIER();
      };
      auto period_n = [&](auto parc, const int w) {
        const int wn = min(w + 1, w_last);                 // past the end: a valid K-tile again (its images are never read)
        const uint32_t kwb_c = (uint32_t)w * ws_b, kwa_n = (uint32_t)wn * ws_a, kwb_n = (uint32_t)wn * ws_b;
        phase_n(std::integral_constant<int, 0>{}, parc, kwb_c, kwa_n, kwb_n);
        phase_n(std::integral_constant<int, 1>{}, parc, kwb_c, kwa_n, kwb_n);
        phase_n(std::integral_constant<int, 2>{}, parc, kwb_c, kwa_n, kwb_n);
        phase_n(std::integral_constant<int, 3>{}, parc, kwb_c, kwa_n, kwb_n);
        phase_n(std::integral_constant<int, 4>{}, parc, kwb_c, kwa_n, kwb_n);
        phase_n(std::integral_constant<int, 5>{}, parc, kwb_c, kwa_n, kwb_n);
      };
      // narrow prologue: the steady state at step 0 of the first K-tile (parity 0): what steps 1 .. 5 of a previous
      // period would have issued, in their order, then the wait of its step 5
      {
        const uint32_t ka = (uint32_t)w_first * ws_a, kb_ = (uint32_t)w_first * ws_b;
        issue_n(1, 0, 0, 0, kb_); issue_n(0, 0, 0, 0, ka);
        issue_n(1, 0, 1, 0, kb_);
        issue_n(0, 1, 0, 1, ka);
        issue_n(1, 1, 0, 1, kb_); issue_n(1, 1, 1, 1, kb_);
        issue_n(0, 2, 0, 2, ka);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      CDML_BARRIER();
      if (grp == 1) CDML_BARRIER();                        // group 1 runs one barrier behind
      for (int w = w_first;;) {
        period_n(std::integral_constant<int, 0>{}, w);
        if (++w > w_last) break;
        period_n(std::integral_constant<int, 1>{}, w);
        if (++w > w_last) break;
      }
      }
      if (grp == 0) CDML_BARRIER();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead loads of the last K-tile
      CDML_BARRIER();
    }
  } else {
  // prologue: the steady state at the first phase of tile 0
  stage(1, 0, 0, 0); stage(1, 1, 0, 0); stage(0, 0, 0, 0); stage(0, 1, 0, 0);
  stage(1, 0, 1, 1); stage(1, 1, 1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  CDML_BARRIER();
  if (grp == 1) CDML_BARRIER();                          // group 1 runs one barrier behind
  constexpr bool kFast6 = X3 && S16 && F6;
  if constexpr (kFast6) {
    {
      int w = x3_t0 / 6;
      for (int tile = 0; tile < n_ktiles; tile += 6, ++w) {
        step6(std::integral_constant<int, 0>{}, tile, w);
        step6(std::integral_constant<int, 1>{}, tile + 1, w);
        step6(std::integral_constant<int, 2>{}, tile + 2, w);
        step6(std::integral_constant<int, 3>{}, tile + 3, w);
        step6(std::integral_constant<int, 4>{}, tile + 4, w);
        step6(std::integral_constant<int, 5>{}, tile + 5, w);
      }
    }
  } else {
    for (int tile = 0; tile < n_ktiles; tile += 2) {
      if constexpr (S16) {
        do_tile2_s16(0, tile);
        do_tile2_s16(1, tile + 1);
      } else {
        do_tile2(0, tile);
        do_tile2(1, tile + 1);
      }
    }
  }

  if (grp == 0) CDML_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the out-of-range tail DMAs still write zeros
  CDML_BARRIER();
  }   // (!kR6)

  if (TN && cs_on && !S16) {   // lanes l31 / l31+32 hold the two k-halves of column l31
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float v = cs[ct] + __shfl_xor(cs[ct], 32, 64);
      if (h == 0) cs_row[grp * cs_grp_stride + ct * 128 + wc * 32 + l31] = v;
    }
  }
  if (!TN && cs_on && S16) {   // k-contiguous form: B-h0 / B-h1 = the first / second 32 columns of the strip's 64
#pragma unroll
    for (int cbt = 0; cbt < 4; ++cbt) {
      float v = cs16[cbt] + __shfl_xor(cs16[cbt], 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q16 == 0) cs_row[grp * cs_grp_stride + wc * 64 + (cbt >> 1) * 32 + (cbt & 1) * 16 + l15] = v;
    }
  }
  if (TN && cs_on && S16) {    // the four 16-lane groups hold the four k-quarters of column l15
#pragma unroll
    for (int cbt = 0; cbt < 4; ++cbt) {
      float v = cs16[cbt] + __shfl_xor(cs16[cbt], 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (q16 == 0) cs_row[grp * cs_grp_stride + (cbt >> 1) * 128 + wc * 32 + (cbt & 1) * 16 + l15] = v;
    }
  }

  // ---- semi-hard mining as the epilogue of the score product (BASELINE config 2; spec: oracle/tower.py semihard_select,
  // the rule of k_semihard_select in loss.hip) -------------------------------------------------------------------------
  // The tile holds S[i][c] = <anchor i, row c> for 256 anchors x 256 embedded rows and writes none of it.  Lane (l15, q)
  // holds, per 16 x 16 block (rbb, cb), column cb*16 + l15 and rows rbb*16 + 4q .. +3.  Per element: d = |a|^2 + |c|^2 - 2 S,
  // eligible = the row's video is neither the anchor's nor its positive's; the lane keeps, per ROW, its best "outside"
  // candidate (closest d > d_p) and its farthest eligible one over its four columns.  The 16 lanes that share a row then
  // meet through the wave's private 16 KiB of LDS (64 rows at a time, slots XOR-swizzled by the row: conflict-free both
  // ways): lane L merges the 16 slots of row L and stores ONE 16-B record -- rows contiguous, 1 KiB per wave store --
  // into the strip's plane of `mine_out`.  k_semihard_finish merges an anchor's 4 tiles_n strips.  (d, c) orders are total
  // (ties -> smaller column), so no merge order can change the result.
  if constexpr (EPI == BE_MINE_X3) {
    // Round 6: no LDS.  With the operands swapped (kSwap) lane (l15, q) holds, per 16 x 16 block (rbb, cb), ROW rbb*16 + l15
    // and the four consecutive columns cb*16 + 4q .. + 3: a lane's 16 columns of a row (four blocks x four) are reduced in
    // registers, the four lanes that share the row (q = 0 .. 3: lanes l15, l15 + 16, + 32, + 48) meet in two cross-lane
    // steps, lane q = 0 stores the row's 16-B record -- 16 rows x 16 B contiguous per store.  (Rounds 5's form: column per
    // lane, a 16-B record per (row, lane) written to the wave's 16 KiB of LDS, read back transposed, merged 16-way: 2 246
    // vector + 1 868 scalar instructions per wave and tile, ~12 us of a 49-us tile.)
    // Everything below is computed from a lane id the optimizer cannot see through: derived from the function's own
    // `lane`, the epilogue's addresses were hoisted ABOVE the K loop and kept live across it (256 VGPRs and 17 of them
    // spilled to scratch inside the loop).  A scratch store is a vector-memory operation: it is counted by the loop's
    // hand-counted s_waitcnt vmcnt(N), which would then let fragments be read before their DMA has landed --
    // __graft_entry__.build() refuses a build in which a GEMM kernel has a scratch frame.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15 = lane_e & 15, q16 = lane_e >> 4;
    const float inf = __builtin_huge_valf();
    // the lane's 16 columns: c(cb, r) = cbase + cb*16 + r -- their |c|^2 and video ids as four 16-B loads each
    const int cbase = n0 + wc * 64 + 4 * q16;                 // < N (N is a multiple of 256)
    f32x4 ncv[4];
    i32x4 idv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      ncv[cb] = *reinterpret_cast<const f32x4 *>(g.mine_sqn + cbase + cb * 16);
      idv[cb] = *reinterpret_cast<const i32x4 *>(g.mine_ids + cbase + cb * 16);
    }
    // the lane's 8 rows: |a|^2, d_p and the anchor's / the positive's video id, all loads in flight together
    const int row_base = m0 + grp * 128 + l15;
    float sa8[8], dp8[8];
    int va8[8], vp8[8];
#pragma unroll
    for (int rbb = 0; rbb < 8; ++rbb) {
      const int i = min(row_base + rbb * 16, g.M - 1);
      sa8[rbb] = g.mine_sqn[2 * i];
      dp8[rbb] = g.mine_dp[i];
      va8[rbb] = g.mine_ids[2 * i];
      vp8[rbb] = g.mine_ids[2 * i + 1];
    }
    const float two_s = kF16 ? 2.0f * g.out_scale : 2.0f;     // (fp16 planes: the accumulator holds 2^(sa + sb) <a, c>)
    auto closer = [](float d, int c, float bd, int bc) { return d < bd || (d == bd && c < bc); };
    auto farther = [](float d, int c, float bd, int bc) { return d > bd || (d == bd && c < bc); };
    MineCand *dst = g.mine_out + (int64_t)((n0 / kTileN) * 4 + wc) * g.mine_ld;
    // The "farthest eligible" candidate is the rule's FALLBACK -- taken only when an anchor has no outside candidate in any
    // strip.  A strip that has found an outside candidate for a row therefore never needs to report a farthest one for it
    // (the anchor has an outside candidate, full stop); a strip that has not reports its farthest as before -- if NO strip
    // has an outside candidate, every strip reported its farthest, and k_semihard_finish's merge is what it was.  So the
    // second scan runs only for the 16-row blocks in which some row's strip came up empty (wave-uniform test; rare: 64
    // columns, about half of them outside): the epilogue is bound by exactly these compare-and-select chains
    // (profiles/r06_miner_runahead_and_lds_free_epilogue_ab.txt), and this removes a third of them.
#pragma unroll
    for (int rbb = 0; rbb < 8; ++rbb) {
      const float sa = sa8[rbb], dpv = dp8[rbb];
      const int va = va8[rbb], vp = vp8[rbb];
      float od = inf, id_ = -inf;
      int oc = 0x7fffffff, ic = 0x7fffffff;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                        // columns ascending: a tie keeps the smaller column without a test
          const int c = cbase + cb * 16 + r;
          const float d = (sa + ncv[cb][r]) - two_s * acc16[rbb][cb][r];
          const int vid = idv[cb][r];
          if (vid != va && vid != vp && d > dpv && d < od) { od = d; oc = c; }
        }
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {              // the four lanes of a row: (d, c) orders are total, any merge order
        const float od2 = __shfl_xor(od, off, 64);
        const int oc2 = __shfl_xor(oc, off, 64);
        if (closer(od2, oc2, od, oc)) { od = od2; oc = oc2; }
      }
      if (__ballot(oc == 0x7fffffff) != 0ull) {              // some row of this block has no outside candidate in this strip
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = cbase + cb * 16 + r;
            const float d = (sa + ncv[cb][r]) - two_s * acc16[rbb][cb][r];
            const int vid = idv[cb][r];
            if (vid != va && vid != vp && d > id_) { id_ = d; ic = c; }
          }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
          const float id2 = __shfl_xor(id_, off, 64);
          const int ic2 = __shfl_xor(ic, off, 64);
          if (farther(id2, ic2, id_, ic)) { id_ = id2; ic = ic2; }
        }
        if (oc != 0x7fffffff) { id_ = -inf; ic = 0x7fffffff; }   // (a row that HAS an outside candidate reports none: one rule per row)
      }
      const int i = row_base + rbb * 16;
      if (q16 == 0 && i < g.M) dst[i] = MineCand{od, oc, id_, ic};
    }
    return;
  }

  // ---- the kNN export's threshold filter as the epilogue of the query x catalogue score product (faiss_knn.py:82-131; exact
  // search) ---------------------------------------------------------------------------------------------------------------
  // The tile holds <q_i, b_c> for 256 queries x 256 catalogue rows and writes none of it (rounds 1-5 wrote 128-MiB score
  // blocks and merged them out of the Infinity Cache).  Every query already has a k-th best distance tau from a first block
  // of the catalogue; an element within it -- a fraction k / (columns seen) of them: a few per tile -- is appended to its
  // query's candidate list (atomic slot counter; the lists are merged into the exact top-k by k_knn_merge_list).  Same
  // operand-swapped register layout as the miner: lane (l15, q) holds row rbb*16 + l15, columns cb*16 + 4q .. + 3.
  if constexpr (EPI == BE_KNN_X3) {
    int lane_e = lane;                                       // (opaque: nothing of this is hoisted above the K loop)
    asm volatile("" : "+v"(lane_e));
    const int l15 = lane_e & 15, q16 = lane_e >> 4;
    const int cbase = n0 + wc * 64 + 4 * q16;                // launch-local column of the lane's first element
    f32x4 bsv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bsv[cb] = *reinterpret_cast<const f32x4 *>(g.knn_bsq + cbase + cb * 16);
    const int row_base = m0 + grp * 128 + l15;
    float qs8[8], tau8[8];
#pragma unroll
    for (int rbb = 0; rbb < 8; ++rbb) {
      const int i = min(row_base + rbb * 16, g.M - 1);
      qs8[rbb] = g.knn_qsq[i];
      tau8[rbb] = g.knn_tau[i];
    }
#pragma unroll
    for (int rbb = 0; rbb < 8; ++rbb) {
      const int i = row_base + rbb * 16;
      const float qs = qs8[rbb], tau = tau8[rbb];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int id = g.knn_col0 + cbase + cb * 16 + r;
          const float d = fmaxf((qs + bsv[cb][r]) - (kF16 ? 2.0f * g.out_scale : 2.0f) * acc16[rbb][cb][r], 0.f);
          if (d <= tau && id < g.knn_n_valid && i < g.M) {   // rare
            const int pos = atomicAdd(g.knn_cnt + i, 1);
            if (pos < g.knn_cap) g.knn_cand[(int64_t)i * g.knn_cap + pos] = make_uint2(__float_as_uint(d), (uint32_t)id);
          }
        }
    }
    return;
  }

  // ---- epilogue: per wave, 32x64 strips through its private 16 KiB of LDS ----
  float *sC = reinterpret_cast<float *>(smem + wave * 16384);
  const int c4 = lane & 15;
  // NT: the strip's 64 columns are contiguous; TN: 32 of each column half
  const int lcol = TN ? (c4 >> 3) * 128 + wc * 32 + (c4 & 7) * 4 : wc * 64 + c4 * 4;     // tile-local column
  const int gcol = n0 + lcol;
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (EPI == BE_BIAS_LRELU_BF16 || EPI == BE_BIAS_LRELU_F32 || EPI == BE_BIAS_LRELU_X3)
    bias4 = *reinterpret_cast<const f32x4 *>(g.bias + gcol);
  const bool has_aux = (EPI == BE_MASK_BF16 || EPI == BE_MASK_X3) && g.aux != nullptr;
  constexpr int GRg = NARROW ? 64 : 128, RTg = GRg / 32;     // rows per row group (half tile: 64), its 32-row strips
  auto out_row = [&](int rt, int p) {
    const int lr = p * 4 + (lane >> 4);
    return TN ? m0 + (rt >> 1) * 128 + grp * 64 + (rt & 1) * 32 + lr : m0 + grp * GRg + rt * 32 + lr;
  };
  constexpr bool kRowBias = EPI == BE_ROWBIAS_LRELU_X3;
  constexpr bool kBiasEpi = EPI == BE_BIAS_LRELU_BF16 || EPI == BE_BIAS_LRELU_X3 || kRowBias;
  constexpr bool kMaskEpi = EPI == BE_MASK_BF16 || EPI == BE_MASK_X3;
  constexpr bool kPlanes = EPI == BE_BIAS_LRELU_X3 || EPI == BE_MASK_X3 || kRowBias;
  if constexpr (!TN && (kBiasEpi || kMaskEpi)) {
    // bf16 outputs of the k-contiguous form: 16 B per lane and store (8 rows x 128 B per instruction) instead of
    // 8 B -- half the store instructions of the tile's tail (cdna guide T21: such a tail is issue-bound)
    const int c8 = lane & 7, lcol8 = wc * 64 + c8 * 8;
    f32x2 bb2[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    if (kBiasEpi && !kRowBias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4 *>(g.bias + n0 + lcol8);
      const f32x4 b1 = *reinterpret_cast<const f32x4 *>(g.bias + n0 + lcol8 + 4);
      bb2[0] = f32x2{b0.x, b0.y}; bb2[1] = f32x2{b0.z, b0.w}; bb2[2] = f32x2{b1.x, b1.y}; bb2[3] = f32x2{b1.z, b1.w};
    }
    // Round 4 (late): this tail was 2700 vector + 1500 scalar instructions per wave in the data gradient -- a branch per
    // ELEMENT on the runtime-uniform "which mask form" switches, 64-bit address arithmetic per store.  Now every such switch
    // is resolved ONCE per tile (MM: 0 = no mask operand, 1 = bf16 values, 2 = one bit per element; WB = also write the
    // sign bitmask), a store's address is a wave-uniform base (SGPRs, scalar adds) plus one 32-bit lane offset that never
    // changes, the bias / leaky-relu / residual arithmetic is written on float pairs (v_pk_*), a sign bit costs a compare
    // and an add-with-carry, a bit-masked element a bit-field extract and a bit-field insert.
    // KO (plane outputs only; round 5): the planes are written k8-INTERLEAVED -- [plane][row / 8][column][8 rows], c_ld =
    // columns per row group -- for a consumer that contracts over the rows (the weight gradient reads the data gradient this
    // way: one 16-B LDS read per fragment).  The masked / activated values go back into the strip they came from, and a
    // second pass reads it by COLUMNS: lane = column, eight rows per 16-B store, 64 lanes x 16 B contiguous per plane.
    auto tail16 = [&](auto mm_c, auto wb_c, auto ko_c) {
      constexpr int MM = decltype(mm_c)::value;
      constexpr bool WB = decltype(wb_c)::value;
      constexpr bool KO = decltype(ko_c)::value;
      static_assert(!KO || (kPlanes && S16 && !WB), "interleaved plane output: a plane epilogue of the 16x16x32 form");
      constexpr int GR = NARROW ? 64 : 128, RT = GR / 32;     // rows per row group (half tile: 64), 32-row strips of it
      const int lrow = lane >> 3;                                  // the lane's row inside an 8-row store
      const int rows_left = g.M - m0 - grp * GR - lrow;          // rows rt*32 + p*8 + lrow < ... are inside the matrix
      const uint32_t lane_c = (uint32_t)(((int64_t)lrow * c_ld + lcol8) * 2);
      const uint32_t lane_m = (uint32_t)((int64_t)lrow * g.ldmask + (lcol8 >> 3));
      const uint32_t plane_bytes = (uint32_t)(g.x3_plane_c * 2);
      bf16x8 mk8[MM == 1 ? 4 : 1][MM == 1 ? 4 : 1];
      int mb8[MM == 2 ? 4 : 1][MM == 2 ? 4 : 1];
      // the leaky-relu' mask operand: bf16 values (sign of the hi plane) or, split-fp32 form with g.aux_bits, ONE BIT per
      // element written by the forward layer's epilogue below (this lane's 8 columns = one byte; ldaux in bytes) -- the
      // value form reads 84 MB of h1's hi plane from HBM inside the store-bound epilogue (30 us of the data gradient's 151).
      // All 16 mask loads of the wave go out together (rows clamped, not branched around)
      if constexpr (MM != 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int64_t rowc = m0 + grp * GR + min(rt * 32 + p * 8 + lrow, g.M - 1 - m0 - grp * GR);
            if constexpr (MM == 2) mb8[rt][p] = reinterpret_cast<const uint8_t *>(g.aux)[rowc * g.ldaux + ((n0 + lcol8) >> 3)];
            else mk8[rt][p] = *reinterpret_cast<const bf16x8 *>(g.aux + rowc * g.ldaux + n0 + lcol8);
          }
      }
      const f32x2 alpha2 = f32x2{g.alpha, g.alpha};
      char *ub_c = static_cast<char *>(c_base) + ((int64_t)(grp * GR + c_row0) * c_ld + c_col0) * 2;   // 8 rows further per store group
      uint8_t *ub_m = WB ? g.mask_out + (int64_t)(m0 + grp * GR) * g.ldmask + (n0 >> 3) : nullptr;
      const int64_t row8_c = c_ld * 16, row8_m = g.ldmask * 8;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float *strip = sC + (rt & 1) * 2048;                 // alternate halves: no wait for the readers
        if constexpr (S16) {
#pragma unroll
          for (int rbb = 0; rbb < 2; ++rbb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                strip[(rbb * 16 + q16 * 4 + r) * 64 + ((cb * 16 + l15) ^ ((q16 & 1) << 4))] = acc16[2 * rt + rbb][cb][r];
        } else {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
              strip[row * 64 + ct * 32 + l31] = acc[rt][ct][r];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int lr = p * 8 + lrow;
          const int sw = S16 ? (((lr >> 2) & 1) << 4) : 0;
          const f32x4 v0 = *reinterpret_cast<const f32x4 *>(strip + lr * 64 + ((c8 * 8) ^ sw));
          const f32x4 v1 = *reinterpret_cast<const f32x4 *>(strip + lr * 64 + ((c8 * 8 + 4) ^ sw));
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if constexpr (kF16) {                               // the accumulator holds 2^(sa + sb) times the product
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= g.out_scale;
          }
          if constexpr (kBiasEpi) {
            f32x2 bq[4] = {bb2[0], bb2[1], bb2[2], bb2[3]};
            if constexpr (kRowBias) {
              const float br = g.bias[m0 + grp * GR + min(rt * 32 + lr, g.M - 1 - m0 - grp * GR)];
#pragma unroll
              for (int k = 0; k < 4; ++k) bq[k] = f32x2{br, br};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f32x2 x = f32x2{v[2 * k], v[2 * k + 1]} + bq[k];
              const f32x2 t = x * alpha2;
              v[2 * k] = fmaxf(x.x, t.x);
              v[2 * k + 1] = fmaxf(x.y, t.y);
            }
          } else if constexpr (MM == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const f32x2 t = f32x2{v[2 * k], v[2 * k + 1]} * alpha2;   // bit set: the value; clear: alpha times it
              const float te[2] = {t.x, t.y};
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                int keep;                                    // 0 or all ones (in asm: hipcc turns the C form back into and / cmp / cndmask)
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep) : "v"(mb8[rt][p]), "n"(2 * k + e));
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(v[2 * k + e]) : "v"(keep), "v"(v[2 * k + e]), "v"(te[e]));
              }
            }
          } else if constexpr (MM == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= ((float)mk8[rt][p][j] > 0.f) ? 1.f : g.alpha;
          }
          if constexpr (kPlanes) {                            // the planes are those of the ROUNDED value
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v[j]));
          }
          if constexpr (kF16 && kPlanes) {                    // fp16 planes of value * c_scale (a power of two), inside fp16's range
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j] * g.c_scale, -65504.f, 65504.f);
          }
          if constexpr (KO) {                                 // back where they came from; the column pass below stores them
            *reinterpret_cast<f32x4 *>(strip + lr * 64 + ((c8 * 8) ^ sw)) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4 *>(strip + lr * 64 + ((c8 * 8 + 4) ^ sw)) = f32x4{v[4], v[5], v[6], v[7]};
            continue;
          }
          // a pair of values -> one dword of two bf16 (v_cvt_pk_bf16_f32), and back by a shift / a mask (written out: hipcc
          // converted every element a second time on its own to get its rounded value back -- 256 extra conversions per wave)
          u32x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = pack2(v[2 * k], v[2 * k + 1]);
          unsigned bits = 0;
          if constexpr (WB) {                                 // bit j = (value j > 0), most significant first:
#pragma unroll
            for (int j = 7; j >= 0; --j) {                    // bits = 2 bits + carry, the carry being the compare
              const float f = X3 ? v[j] : ((j & 1) ? hi_of(o[j >> 1]) : lo_of(o[j >> 1]));   // X3: sign of the fp32 activation itself; else of the stored bf16
              asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(f) : "vcc");
            }
          }
          char *const ub = ub_c;                               // (running uniform bases: two scalar adds per store group)
          uint8_t *const um = ub_m;
          ub_c += row8_c;
          if constexpr (WB) ub_m += row8_m;
          if (rt * 32 + p * 8 >= rows_left) continue;         // stores only below this line
          if constexpr (WB) um[lane_m] = (uint8_t)bits;       // this lane's 8 columns = one byte of the sign bitmask
          *reinterpret_cast<u32x4 *>(ub + lane_c) = o;
          if constexpr (kPlanes) {
            // three roundings to nearest hold the 24 significant bits: hi + mid + lo == v exactly
            // (fp16 form: hi + lo holds 22 of them -- what the three-product sum keeps anyway)
#pragma unroll
            for (int pl = 1; pl < kPlanesOut; ++pl) {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const f32x2 r = f32x2{v[2 * k], v[2 * k + 1]} - f32x2{lo_of(o[k]), hi_of(o[k])};
                v[2 * k] = r.x;
                v[2 * k + 1] = r.y;
                o[k] = pack2(r.x, r.y);
              }
              *reinterpret_cast<u32x4 *>(ub + (size_t)pl * plane_bytes + lane_c) = o;
            }
          }
        }
        if constexpr (KO) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // rows (c_row0 + grp GR + rt 32 + rg 8 .. + 7) of column c_col0 + wc 64 + lane: element (r, c, plane) of the output at
          // plane * x3_plane_c + ((r / 8) * c_ld + c) * 8 + r % 8
          char *const kb = static_cast<char *>(c_base) + ((int64_t)((c_row0 + grp * GR + rt * 32) >> 3) * c_ld + c_col0 + wc * 64 + lane) * 16;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = strip[(rg * 8 + j) * 64 + (lane ^ ((j >> 2) << 4))];
            if (m0 + grp * GR + rt * 32 + rg * 8 >= g.M) continue;          // (M is a multiple of 8: a row group is inside or outside)
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack2(v[2 * k], v[2 * k + 1]);
            char *const dstp = kb + (int64_t)rg * c_ld * 16;
            *reinterpret_cast<u32x4 *>(dstp) = o;
#pragma unroll
            for (int pl = 1; pl < 3; ++pl) {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const f32x2 r = f32x2{v[2 * k], v[2 * k + 1]} - f32x2{lo_of(o[k]), hi_of(o[k])};
                v[2 * k] = r.x;
                v[2 * k + 1] = r.y;
                o[k] = pack2(r.x, r.y);
              }
              *reinterpret_cast<u32x4 *>(dstp + (size_t)pl * plane_bytes) = o;
            }
          }
        }
      }
    };
    using std::integral_constant;
    using no_t = integral_constant<bool, false>;
    if constexpr (kMaskEpi) {
      if constexpr (X3 && S16 && EPI == BE_MASK_X3) {
        if (g.c_kint) {                                       // (host: the bitmask form or no mask)
          if (has_aux) tail16(integral_constant<int, 2>{}, no_t{}, integral_constant<bool, true>{});
          else tail16(integral_constant<int, 0>{}, no_t{}, integral_constant<bool, true>{});
          return;
        }
      }
      if (!has_aux) tail16(integral_constant<int, 0>{}, no_t{}, no_t{});
      else if (X3 && g.aux_bits) tail16(integral_constant<int, X3 ? 2 : 1>{}, no_t{}, no_t{});
      else tail16(integral_constant<int, 1>{}, no_t{}, no_t{});
    } else {
      if (g.mask_out) tail16(integral_constant<int, 0>{}, integral_constant<bool, true>{}, no_t{});
      else tail16(integral_constant<int, 0>{}, no_t{}, no_t{});
    }
    return;
  }
  // all 32 mask loads of the wave go out together (rows clamped, not branched around: a load
  // under a branch is waited for on the spot, 32 dependent round trips per tile)
  bf16x4 mk[4][8];
  if ((EPI == BE_MASK_BF16 || EPI == BE_MASK_X3) && has_aux) {
#pragma unroll
    for (int rt = 0; rt < RTg; ++rt)
#pragma unroll
      for (int p = 0; p < 8; ++p)
        mk[rt][p] = *reinterpret_cast<const bf16x4 *>(g.aux + (int64_t)min(out_row(rt, p), g.M - 1) * g.ldaux + gcol);
  }
#pragma unroll
  for (int rt = 0; rt < RTg; ++rt) {
    float *strip = sC + (rt & 1) * 2048;                 // alternate halves: no wait for the readers
    if constexpr (S16) {
      // 16x16 blocks: lane (l15, q) holds column l15, rows 4q .. 4q+3.  Rows 4 apart share their banks
      // (64-float rows): the column block is XOR-ed with 16 on odd q, so a store is 2-way (free), not 4-way
#pragma unroll
      for (int rbb = 0; rbb < 2; ++rbb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            strip[(rbb * 16 + q16 * 4 + r) * 64 + ((cb * 16 + l15) ^ ((q16 & 1) << 4))] = acc16[2 * rt + rbb][cb][r];
    } else {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          strip[row * 64 + ct * 32 + l31] = acc[rt][ct][r];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int lr = p * 4 + (lane >> 4);
      const int row = out_row(rt, p);
      f32x4 v = *reinterpret_cast<const f32x4 *>(strip + lr * 64 + ((c4 * 4) ^ (S16 ? (((lr >> 2) & 1) << 4) : 0)));
      if constexpr (kF16) v *= g.out_scale;                  // the accumulator holds 2^(sa + sb) times the product
      if (EPI == BE_BIAS_LRELU_BF16 || EPI == BE_BIAS_LRELU_F32 || EPI == BE_BIAS_LRELU_X3) {
        v += bias4;
        v.x = fmaxf(v.x, v.x * g.alpha); v.y = fmaxf(v.y, v.y * g.alpha);
        v.z = fmaxf(v.z, v.z * g.alpha); v.w = fmaxf(v.w, v.w * g.alpha);
      } else if (EPI == BE_MASK_BF16 || EPI == BE_MASK_X3) {
        if (has_aux) {
          const bf16x4 m = mk[rt][p];
          v.x *= ((float)m.x > 0.f) ? 1.f : g.alpha; v.y *= ((float)m.y > 0.f) ? 1.f : g.alpha;
          v.z *= ((float)m.z > 0.f) ? 1.f : g.alpha; v.w *= ((float)m.w > 0.f) ? 1.f : g.alpha;
        }
      }
      bf16x4 o;
      o.x = (bf16)v.x; o.y = (bf16)v.y; o.z = (bf16)v.z; o.w = (bf16)v.w;
      unsigned nib = 0;
      if (EPI == BE_BIAS_LRELU_BF16 && g.mask_out) {       // sign bits of the 4 stored values; lanes c4, c4^1
        nib = ((float)o.x > 0.f ? 1u : 0u) | ((float)o.y > 0.f ? 2u : 0u) | ((float)o.z > 0.f ? 4u : 0u) |
              ((float)o.w > 0.f ? 8u : 0u);                 // share a byte (8 columns): the even lane stores it
        const unsigned other = __shfl_xor(nib, 1, 64);
        nib |= other << 4;
      }
      if (row >= g.M) continue;                            // stores only below this line
      if constexpr (EPI == BE_BIAS_LRELU_X3 || EPI == BE_MASK_X3) {
        // three roundings to nearest hold the 24 significant bits: hi + mid + lo == v exactly (v pinned as
        // computed: the products above must not be contracted with the residual subtractions into fmas)
        bf16 *dst = static_cast<bf16 *>(c_base) + (int64_t)(row - m0 + c_row0) * c_ld + c_col0 + lcol;
        asm volatile("" : "+v"(v));
        f32x4 r = v;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          bf16x4 q;
          q.x = (bf16)r.x; q.y = (bf16)r.y; q.z = (bf16)r.z; q.w = (bf16)r.w;
          *reinterpret_cast<bf16x4 *>(dst + pl * g.x3_plane_c) = q;
          r.x -= (float)q.x; r.y -= (float)q.y; r.z -= (float)q.z; r.w -= (float)q.w;
        }
        continue;
      }
      if (EPI == BE_BIAS_LRELU_BF16 || EPI == BE_MASK_BF16) {
        if (EPI == BE_BIAS_LRELU_BF16 && g.mask_out && !(c4 & 1))
          g.mask_out[(int64_t)row * g.ldmask + (gcol >> 3)] = (uint8_t)nib;
        *reinterpret_cast<bf16x4 *>(static_cast<bf16 *>(c_base) + (int64_t)(row - m0 + c_row0) * c_ld + c_col0 + lcol) = o;
      } else {
        *reinterpret_cast<f32x4 *>(static_cast<float *>(c_base) + (int64_t)(row - m0 + c_row0) * c_ld + c_col0 + lcol) = v;
      }
    }
  }
}

// one block of a launch: block index `bid` of this launch -> (half) tile -> run_tile
template <bool TN, int EPI, bool S16, bool X3, bool F6, bool NTCS, bool R6, bool NARROW, bool KI = false>
__device__ __forceinline__ void block_of_launch(const BArgs &g, int bid, unsigned char *smem) {
  int tm, tn;
  // the block -> tile map of the WHOLE tile grid, of which this launch may cover the first blocks only (grid_tiles) or,
  // NARROW, the rest as two half tiles each: block (xcd, local) -> half local / per of the tile of block (narrow_first / 8
  // + local % per, xcd) -- both halves of a tile and its neighbours on the XCD whose L2 holds their operand panels
  const int nwg = g.grid_tiles > 0 ? g.grid_tiles : (int)gridDim.x;
  int half = 0;
  if constexpr (NARROW) {
    if (((nwg - g.narrow_first) & 7) == 0) {
      const int per = (nwg - g.narrow_first) >> 3, xcd = bid & 7, loc = bid >> 3;
      half = loc / per;
      bid = (((g.narrow_first >> 3) + (loc - half * per)) << 3) | xcd;
    } else {                      // a tile count that is no multiple of 8 (the narrow layer at 3 072 rows: 12 tiles): halves side by side
      half = bid & 1;
      bid = g.narrow_first + (bid >> 1);
    }
  }
  if (EPI != BE_MINE_X3 && EPI != BE_KNN_X3 && g.K <= (X3 ? 3072 : 512)) tile_of_block_rowmajor(bid, nwg, g.tiles_n, tm, tn);   // output-bound
  else tile_of_block(bid, nwg, g.tiles_m, g.tiles_n, tm, tn);
  const int m0 = tm * kTileM + half * (kTileM / 2), n0 = tn * kTileN;
  const int split = blockIdx.y;
  const int k_begin = split * g.k_per_split;
  const int k_end = min(g.K, k_begin + g.k_per_split);
  const int n_ktiles = k_end > k_begin ? (k_end - k_begin) / kTileK : 0;   // even (host)
  void *c_base = EPI == BE_F32 ? static_cast<void *>(static_cast<float *>(g.C) + (int64_t)split * g.slab_stride) : g.C;
  float *cs_row = ((TN || NTCS) && g.colsum_partial) ? g.colsum_partial + (int64_t)((split * g.tiles_m + tm) * 2) * g.N + n0 : nullptr;
  run_tile<TN, EPI, S16, X3, F6, NTCS, R6, NARROW, KI>(g, tm, m0, n0, k_begin, n_ktiles, c_base, g.ldc, m0, n0, cs_row, g.N, smem);
}

template <bool TN, int EPI, bool S16, bool X3 = false, bool F6 = false, bool NTCS = false, bool R6 = false, bool NARROW = false,
          bool KI = false>
__global__ void __launch_bounds__(kT, 1) k_gemm_bf16_256(BArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // (round 5: ONE resident block per CU walking the launch's tiles -- no block dispatch between a CU's tiles -- was built
  // and measured: miner 0.437-0.442 against 0.429-0.430 ms, headline step 2.629-2.633 against 2.621-2.623 ms,
  // profiles/r05_persistent_tiles_ab.txt; the loop around run_tile also cost the k-strided kernel its last free VGPRs --
  // 72 B of scratch.  Removed.)
  block_of_launch<TN, EPI, S16, X3, F6, NTCS, R6, NARROW, KI>(g, blockIdx.x, smem);
}

// full tiles and the last round's half tiles in ONE launch (blocks [0, narrow_first): full tiles; the rest: half tiles):
// a half tile starts on whichever CU finishes its last full tile first, not after the whole launch of full tiles has drained
template <int EPI>
__global__ void __launch_bounds__(kT, 1) k_gemm_x3_rounds(BArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // dispatch order (blocks start in index order): [stagger_lead half-tile blocks][the full tiles][the other half-tile blocks]
  const int b = blockIdx.x, lead = g.stagger_lead;
  if (b >= lead && b < lead + g.narrow_first) block_of_launch<false, EPI, true, true, false, false, true, false>(g, b - lead, smem);
  else block_of_launch<false, EPI, true, true, false, false, true, true>(g, b < lead ? b : b - g.narrow_first, smem);
}

// ---- both weight gradients of the tower in ONE launch (k-strided form; dW1 = x_hat^T dz1, dW2 = h1^T dz2) ----
// The two products contract over the same K (the batch rows).  Their output tiles, each cut into UNITS of two
// K-tiles, form one line of work: tiles of problem 0 then problem 1, a tile's units in k order, tiles ordered
// tm fastest (neighbouring blocks share the B panel).  The launch has one block per CU and block b takes the
// units [b U, (b+1) U): at most a tile's tail and the next tile's head (U < units per tile at the shapes this is
// used for; any number of pieces is handled).  Every piece -- a SEGMENT, the line cut at block AND tile
// boundaries -- is written as a tile-local fp32 partial (256 x 256, contiguous) to slot number
// u/U + u/upt - u/lcm(U, upt) (u = its first unit), with its owned column sums of B beside it; k_sk_fixup_tn
// adds a tile's partials in k order (fixed order: bit-reproducible) and finishes the bias gradients.
// What it buys over one split-K launch per product: the second layer's product (N = 256: 20 tiles, bound by
// streaming h1 from HBM) no longer has the chip to itself -- its blocks run beside the first layer's MFMA-bound
// ones -- and every CU gets the same number of K-tiles.
struct SKArgs {
  BArgs p[2];
  int tiles0;              // tiles of problem 0
  int upt;                 // units per tile = K / 128
  int total_units, units_per_block, lcm_units;
  float *partials;         // [slots][256 * 256]
  float *cs_partials;      // [slots][2][256] (column sums of B per row group) or null
};

__device__ __forceinline__ int sk_slot(const SKArgs &a, int u) {
  return u / a.units_per_block + u / a.upt - u / a.lcm_units;
}

template <bool S16>
__global__ void __launch_bounds__(kT, 1) k_gemm_bf16_sk(SKArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  int u = blockIdx.x * a.units_per_block;
  const int end = min(u + a.units_per_block, a.total_units);
  bool first = true;
  while (u < end) {
    const int tg = u / a.upt, ku = u - tg * a.upt;
    const int len = min(a.upt - ku, end - u);
    const int pi = tg >= a.tiles0 ? 1 : 0;
    const BArgs &g = a.p[pi];
    const int tl = tg - (pi ? a.tiles0 : 0);
    const int tn = tl / g.tiles_m, tm = tl - tn * g.tiles_m;
    const int slot = sk_slot(a, u);
    if (!first) CDML_BARRIER();                      // the previous segment's epilogue strips are LDS images again
    first = false;
    run_tile<true, BE_F32, S16>(g, tm, tm * kTileM, tn * kTileN, ku * 2 * kTileK, len * 2,
                                a.partials + (int64_t)slot * (kTileM * kTileN), kTileN, 0, 0,
                                a.cs_partials ? a.cs_partials + (int64_t)slot * (2 * kTileN) : nullptr, kTileN, smem);
    u += len;
  }
}

// Sum the partials of every tile (in k order) into the two outputs and finish the column sums.
// Blocks [0, 16 n_tiles): 16 rows x 256 columns of one tile each; then one block per (problem, column tile).
__global__ void __launch_bounds__(256) k_sk_fixup_tn(SKArgs a, float *db0, float *db1) {
  const int n_tiles = a.total_units / a.upt;
  const int U = a.units_per_block;
  if ((int)blockIdx.x < 16 * n_tiles) {
    const int tg = blockIdx.x >> 4, part = blockIdx.x & 15;
    const int pi = tg >= a.tiles0 ? 1 : 0;
    const BArgs &g = a.p[pi];
    const int tl = tg - (pi ? a.tiles0 : 0);
    const int tn = tl / g.tiles_m, tm = tl - tn * g.tiles_m;
    const int g0 = tg * a.upt, g1 = g0 + a.upt;
    const int r = part * 16 + (threadIdx.x >> 4), c = (threadIdx.x & 15) * 16;      // 16 floats per thread
    f32x4 s[4];
    bool have = false;
    for (int st = g0; st < g1; st = (st / U + 1) * U) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(a.partials + (int64_t)sk_slot(a, st) * (kTileM * kTileN) + r * kTileN + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = have ? s[j] + src[j] : src[j];
      have = true;
    }
    float *dst = static_cast<float *>(g.C) + (int64_t)(tm * kTileM + r) * g.ldc + tn * kTileN + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) reinterpret_cast<f32x4 *>(dst)[j] = s[j];
    return;
  }
  if (!a.cs_partials) return;
  int ct = blockIdx.x - 16 * n_tiles;                 // (problem, column tile)
  const int pi = ct >= a.p[0].tiles_n ? 1 : 0;
  const BArgs &g = a.p[pi];
  float *db = pi ? db1 : db0;
  if (!db) return;
  const int tn = ct - (pi ? a.p[0].tiles_n : 0);
  float acc = 0.f;
  for (int tm = 0; tm < g.tiles_m; ++tm) {
    const int tg = (pi ? a.tiles0 : 0) + tn * g.tiles_m + tm;
    const int g0 = tg * a.upt, g1 = g0 + a.upt;
    for (int st = g0; st < g1; st = (st / U + 1) * U) {
      const float *src = a.cs_partials + (int64_t)sk_slot(a, st) * (2 * kTileN) + threadIdx.x;
      acc += src[0];
      acc += src[kTileN];
    }
  }
  db[tn * kTileN + threadIdx.x] = acc;
}

template <bool TN, int EPI, bool S16>
int launch1(const BArgs &g, int splits, hipStream_t s) {
  static bool configured = false;   // raising the dynamic-LDS limit is idempotent; a race only repeats it
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16_256<TN, EPI, S16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16_256: cannot reserve %d B of LDS: %s", SMEM,
                                     hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((k_gemm_bf16_256<TN, EPI, S16>), dim3(g.tiles_m * g.tiles_n, splits), dim3(kT), SMEM, s, g);
  return check_launch("gemm_bf16_256");
}

// MFMA shape: CDML_BF16_MFMA = "32" (v_mfma_f32_32x32x16_bf16) or "16"
// (v_mfma_f32_16x16x32_bf16); read per call so that one process can time both on one device
static bool shape16() {
  const char *e = getenv("CDML_BF16_MFMA");
  return e ? atoi(e) == 16 : CDML_BF16_MFMA_DEFAULT == 16;
}

template <bool TN, int EPI>
int launch(const BArgs &g, int splits, hipStream_t s) {
  if (shape16()) return launch1<TN, EPI, true>(g, splits, s);
  return launch1<TN, EPI, false>(g, splits, s);
}

}  // namespace

#ifndef CDML_F16X2
bool gemm_bf16_256_usable(int M, int N, int K, int64_t lda, int64_t ldb) {
  if (N % kTileN || K % (2 * kTileK) || M < 1) return false;
  const int64_t lim = (int64_t)1 << 31;
  return ((int64_t)M + kTileM) * lda * 2 < lim && (int64_t)N * ldb * 2 < lim;
}

// split-K for the skinny weight gradients: fill one round of 256 CUs as evenly as
// possible with the fewest slabs
int gemm_bf16_256_splits(int M, int N, int K) {
  const int64_t tiles = (int64_t)((M + kTileM - 1) / kTileM) * (N / kTileN);
  const int max_by_k = K / 1024 > 0 ? K / 1024 : 1;
  int best = 1;
  double best_eff = 0.0;
  for (int s = 1; s <= 16 && s <= max_by_k; ++s) {
    const int64_t blocks = tiles * s;
    const double eff = (double)blocks / (double)(((blocks + kNumCU - 1) / kNumCU) * kNumCU);
    if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
  }
  return best;
}

int launch_gemm_bf16_256(const BArgs &g, int epilogue, int splits, hipStream_t s) {
  switch (epilogue) {
    case BE_BIAS_LRELU_BF16: return launch<false, BE_BIAS_LRELU_BF16>(g, splits, s);
    case BE_BIAS_LRELU_F32: return launch<false, BE_BIAS_LRELU_F32>(g, splits, s);
    case BE_MASK_BF16: return launch<false, BE_MASK_BF16>(g, splits, s);
    default: return launch<false, BE_F32>(g, splits, s);
  }
}
#endif

namespace {
template <bool TN, int EPI, bool F6, bool NTCS = false, bool R6 = false, bool NARROW = false>
int launch_x3_1(const BArgs &g, int blocks, int splits, hipStream_t s) {
  static bool configured = false;
  constexpr int smem = R6 ? SMEM_R6 : SMEM;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16_256<TN, EPI, true, true, F6, NTCS, R6, NARROW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16x3: cannot reserve %d B of LDS: %s", smem, hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((k_gemm_bf16_256<TN, EPI, true, true, F6, NTCS, R6, NARROW>), dim3(blocks, splits), dim3(kT), smem, s, g);
  return check_launch("gemm_bf16x3");
}
// The last round of a plane-output product in HALF TILES: a launch of T tiles runs T / 256 full rounds of the 256 CUs and
// then a round with T % 256 of them busy (FC1 and the data gradient at config 1: 640 tiles, the third round on half the
// chip).  When that remainder fits the chip twice, its tiles are computed by a second launch as two 128 x 256 halves
// each -- same operands, same K order per output element (bit-identical results), twice the CUs at a little over half a
// tile's time each.  CDML_X3_HALFTILES=0 turns it off (A/B timing); read per call.
bool x3_half_tiles() {
  const char *e = getenv("CDML_X3_HALFTILES");
  return !e || atoi(e) != 0;
}
// CDML_X3_HALFTILES=2: the half tiles as a launch of their own behind the full tiles' (A/B timing)
bool x3_one_launch() {
  const char *e = getenv("CDML_X3_HALFTILES");
  return !e || atoi(e) != 2;
}
// which K loop walks the six plane products: CDML_X3_WALK = "r6" (default: the resident-plane walk, both forms),
// "f6" (round 3's unrolled six-step period with DMA skipping; k-contiguous form only), "general" (round 3's general loop);
// read per call so that one process can time them against each other
int x3_walk() {
  const char *e = getenv("CDML_X3_WALK");
  return !e ? 2 : (e[0] == 'g' ? 0 : (e[0] == 'f' ? 1 : 2));
}
// the unrolled walks need every block's K range to be whole periods of the K-major six-product walk
template <bool TN, int EPI>
int launch_x3(const BArgs &g, int splits, hipStream_t s) {
#ifdef CDML_F16X2
  // three products: the resident-plane walk of the three-step period (R3) wherever every block's K range is whole periods;
  // CDML_X3_WALK=general: the general K-major loop (A/B runs; read per call)
  const int kt3 = g.K / kTileK, per3 = g.k_per_split / kTileK;
  const bool whole3 = g.x3_products == 3 && kt3 % 3 == 0 && per3 % 3 == 0 && per3 > 0;
  if (whole3 && x3_walk() == 2) return launch_x3_1<TN, EPI, false, false, true>(g, g.tiles_m * g.tiles_n, splits, s);
  return launch_x3_1<TN, EPI, false>(g, g.tiles_m * g.tiles_n, splits, s);
#else
  const int kt = g.K / kTileK, per = g.k_per_split / kTileK;
  const bool whole = g.x3_products == 6 && kt % 6 == 0 && per % 6 == 0 && per > 0;
  const int walk = whole ? x3_walk() : 0;
  if constexpr (!TN && EPI == BE_F32) {   // the weight gradients of the transposed activation layout: with the column sums
    // (the general loop: round 3's unrolled period plus the sums was 14 VGPRs over the budget)
    if (g.colsum_partial) return launch_x3_1<TN, EPI, false, true>(g, g.tiles_m * g.tiles_n, splits, s);
  }
  const int tiles = g.tiles_m * g.tiles_n;
  // (round 6: the unsplit fp32-output product too -- the trainable table's row gradient dz1 . W1^T is 384 tiles at 16 384 rows:
  // a full round and half a round)
  if constexpr (!TN && (EPI == BE_BIAS_LRELU_X3 || EPI == BE_MASK_X3 || EPI == BE_ROWBIAS_LRELU_X3 || EPI == BE_F32)) {
    const int full = tiles / kNumCU * kNumCU, rem = tiles - full;
    if (walk == 2 && splits == 1 && rem > 0 && 2 * rem <= kNumCU && rem % 8 == 0 && full % 8 == 0 && x3_half_tiles()) {
      BArgs h = g;
      h.grid_tiles = tiles;
      h.narrow_first = full;
      if (full > 0 && x3_one_launch()) {
        // CDML_X3_STAGGER=1 (A/B, read per call): half of the half-tile blocks go first
        const char *st = getenv("CDML_X3_STAGGER");
        if (st && atoi(st) != 0 && full + rem >= kNumCU && rem % 8 == 0) h.stagger_lead = rem;
        static bool configured = false;
        if (!configured) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_x3_rounds<EPI>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_R6);
          if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16x3: cannot reserve %d B of LDS: %s", SMEM_R6, hipGetErrorString(e));
          configured = true;
        }
        hipLaunchKernelGGL((k_gemm_x3_rounds<EPI>), dim3(full + 2 * rem), dim3(kT), SMEM_R6, s, h);
        return check_launch("gemm_bf16x3");
      }
      if (full > 0) {
        const int rc = launch_x3_1<TN, EPI, false, false, true>(h, full, 1, s);
        if (rc) return rc;
      }
      return launch_x3_1<TN, EPI, false, false, true, true>(h, 2 * rem, 1, s);
    }
  }
  if constexpr (!TN && EPI == BE_F32) {
    // The narrow layer's fp32 slabs (N = one tile column, K split into slabs): when the full tiles leave a third of the
    // chip or more idle, the same slabs are computed as 128 x 256 HALF tiles on twice the blocks -- same operands, same K
    // order per output element: bit-identical (BASELINE config 1: 32 row tiles x 4 slabs = 128 blocks -> 256)
    if (walk == 2 && splits > 1 && !g.colsum_partial && 3 * tiles * splits <= 2 * kNumCU && x3_half_tiles()) {
      BArgs h = g;
      h.grid_tiles = tiles;
      h.narrow_first = 0;
      return launch_x3_1<TN, EPI, false, false, true, true>(h, 2 * tiles, splits, s);
    }
  }
  if (walk == 2) return launch_x3_1<TN, EPI, false, false, true>(g, tiles, splits, s);
  if constexpr (!TN) {
    // (round 3: the k-strided form did not fit the F6 period into 256 VGPRs -- 58 spilled, 2.4 x slower)
    if (walk == 1) return launch_x3_1<TN, EPI, true>(g, tiles, splits, s);
  }
  return launch_x3_1<TN, EPI, false>(g, tiles, splits, s);
#endif
}
}  // namespace

#ifdef CDML_F16X2
// the score products whose epilogue is the selection (semi-hard mining) / the threshold filter (kNN export), on fp16 planes:
// three products, the three-step resident-plane walk (K = D: whole periods always)
int launch_gemm_f16x2_mine(const BArgs &g, hipStream_t s) {
  return launch_x3_1<false, BE_MINE_X3, false, false, true>(g, g.tiles_m * g.tiles_n, 1, s);
}
int launch_gemm_f16x2_knn(const BArgs &g, hipStream_t s) {
  return launch_x3_1<false, BE_KNN_X3, false, false, true>(g, g.tiles_m * g.tiles_n, 1, s);
}
int launch_gemm_f16x2_256(const BArgs &g, bool tn, int epilogue, int splits, hipStream_t s) {
  if (tn) return launch_x3<true, BE_F32>(g, splits, s);
  switch (epilogue) {
    case BE_BIAS_LRELU_F32: return launch_x3<false, BE_BIAS_LRELU_F32>(g, splits, s);
    case BE_BIAS_LRELU_X3: return launch_x3<false, BE_BIAS_LRELU_X3>(g, splits, s);
    case BE_MASK_X3: return launch_x3<false, BE_MASK_X3>(g, splits, s);
    default: return launch_x3<false, BE_F32>(g, splits, s);
  }
}
#else
// the k-strided product on k8-interleaved operands (resident-plane walk, six products, whole periods per split)
int launch_gemm_x3_tnk(const BArgs &g, int splits, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_bf16_256<true, BE_F32, true, true, false, false, true, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_R6);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16x3_tnk: cannot reserve %d B of LDS: %s", SMEM_R6, hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((k_gemm_bf16_256<true, BE_F32, true, true, false, false, true, false, true>), dim3(g.tiles_m * g.tiles_n, splits),
                     dim3(kT), SMEM_R6, s, g);
  return check_launch("gemm_bf16x3_tnk");
}

int launch_gemm_x3_mine(const BArgs &g, hipStream_t s) {
  return launch_x3_1<false, BE_MINE_X3, false, false, true>(g, g.tiles_m * g.tiles_n, 1, s);
}

int launch_gemm_x3_knn(const BArgs &g, hipStream_t s) {
  return launch_x3_1<false, BE_KNN_X3, false, false, true>(g, g.tiles_m * g.tiles_n, 1, s);
}

int launch_gemm_bf16_256_x3(const BArgs &g, bool tn, int epilogue, int splits, hipStream_t s) {
  if (tn) return launch_x3<true, BE_F32>(g, splits, s);
  switch (epilogue) {
    case BE_BIAS_LRELU_F32: return launch_x3<false, BE_BIAS_LRELU_F32>(g, splits, s);
    case BE_BIAS_LRELU_X3: return launch_x3<false, BE_BIAS_LRELU_X3>(g, splits, s);
    case BE_MASK_X3: return launch_x3<false, BE_MASK_X3>(g, splits, s);
    case BE_ROWBIAS_LRELU_X3: return launch_x3<false, BE_ROWBIAS_LRELU_X3>(g, splits, s);
    default: return launch_x3<false, BE_F32>(g, splits, s);
  }
}

bool gemm_bf16_tn_usable(int M, int N, int K, int64_t lda, int64_t ldb) {
  if (M % kTileM || N % kTileN || K % (2 * kTileK) || K < 2 * kTileK) return false;
  const int64_t lim = (int64_t)1 << 31;
  return (int64_t)K * lda * 2 < lim && (int64_t)K * ldb * 2 < lim;
}

int launch_gemm_bf16_tn(const BArgs &g, int splits, hipStream_t s) { return launch<true, BE_F32>(g, splits, s); }

// ---- the joint launch: geometry shared by the workspace query and the launch ----
namespace {
struct SKGeom { int tiles0, tiles1, upt, total, U, lcm, slots; };
int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
bool sk_geometry(int M1, int N1, int M2, int N2, int K, SKGeom &q) {
  if (M1 % kTileM || N1 % kTileN || M2 % kTileM || N2 % kTileN || K % (2 * kTileK) || K < 2 * kTileK) return false;
  q.tiles0 = (M1 / kTileM) * (N1 / kTileN);
  q.tiles1 = (M2 / kTileM) * (N2 / kTileN);
  q.upt = K / (2 * kTileK);
  const int64_t total = (int64_t)(q.tiles0 + q.tiles1) * q.upt;
  if (total > (1 << 28)) return false;
  q.total = (int)total;
  q.U = (q.total + kNumCU - 1) / kNumCU;
  const int64_t l = (int64_t)q.U / gcd_i(q.U, q.upt) * q.upt;
  q.lcm = l > (int64_t)q.total + 1 ? q.total + 1 : (int)l;           // beyond the line: never reached
  q.slots = q.total / q.U + q.total / q.upt - q.total / q.lcm + 1;
  return true;
}
}  // namespace

size_t gemm_bf16_tn2_workspace(int M1, int N1, int M2, int N2, int K) {
  SKGeom q;
  if (!sk_geometry(M1, N1, M2, N2, K, q)) return 0;
  return (size_t)q.slots * (kTileM * kTileN + 2 * kTileN) * sizeof(float);
}

int launch_gemm_bf16_tn2(const BArgs &g1, const BArgs &g2, float *db1, float *db2, void *workspace, size_t workspace_bytes,
                         hipStream_t s) {
  SKGeom q;
  if (!sk_geometry(g1.M, g1.N, g2.M, g2.N, g1.K, q) || g1.K != g2.K)
    return fail(CDML_E_UNSUPPORTED, "gemm_bf16_tn2: shapes must be multiples of 256 (M, N) and 128 (one K for both)");
  const size_t need = (size_t)q.slots * (kTileM * kTileN + 2 * kTileN) * sizeof(float);
  if (!workspace || workspace_bytes < need) return fail(CDML_E_BADARG, "gemm_bf16_tn2: workspace of %zu bytes required", need);
  SKArgs a{};
  a.p[0] = g1; a.p[1] = g2;
  a.p[0].tiles_m = g1.M / kTileM; a.p[0].tiles_n = g1.N / kTileN;
  a.p[1].tiles_m = g2.M / kTileM; a.p[1].tiles_n = g2.N / kTileN;
  a.tiles0 = q.tiles0; a.upt = q.upt; a.total_units = q.total; a.units_per_block = q.U; a.lcm_units = q.lcm;
  a.partials = static_cast<float *>(workspace);
  a.cs_partials = (db1 || db2) ? a.partials + (size_t)q.slots * (kTileM * kTileN) : nullptr;
  const bool s16 = shape16();
  static bool configured[2] = {false, false};
  if (!configured[s16]) {
    const void *fn = s16 ? reinterpret_cast<const void *>(&k_gemm_bf16_sk<true>) : reinterpret_cast<const void *>(&k_gemm_bf16_sk<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return fail(CDML_E_HIP, "gemm_bf16_tn2: cannot reserve %d B of LDS: %s", SMEM, hipGetErrorString(e));
    configured[s16] = true;
  }
  const int blocks = (q.total + q.U - 1) / q.U;
  if (s16) hipLaunchKernelGGL((k_gemm_bf16_sk<true>), dim3(blocks), dim3(kT), SMEM, s, a);
  else hipLaunchKernelGGL((k_gemm_bf16_sk<false>), dim3(blocks), dim3(kT), SMEM, s, a);
  int rc = check_launch("gemm_bf16_tn2");
  if (rc) return rc;
  const int n_tiles = q.tiles0 + q.tiles1;
  const int cs_blocks = a.cs_partials ? a.p[0].tiles_n + a.p[1].tiles_n : 0;
  hipLaunchKernelGGL(k_sk_fixup_tn, dim3(16 * n_tiles + cs_blocks), dim3(256), 0, s, a, db1, db2);
  return check_launch("gemm_bf16_tn2 fix-up");
}
#endif  // !CDML_F16X2

}  // namespace cdml
