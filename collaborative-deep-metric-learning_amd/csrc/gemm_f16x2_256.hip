// The 256 x 256 ping-pong GEMM of gemm_bf16_256.hip compiled for TWO fp16 planes per fp32 operand (precision "f16x2"):
//   x 2^s = hi + lo,  hi = fp16(x 2^s),  lo = fp16(x 2^s - hi)      (s: a per-tensor power of two, exact)
//   a b   = 2^-(sa + sb) (ah bh + ah bl + al bh)   [+ al bl, below 2^-22 |a b|: dropped]
// three v_mfma_f32_16x16x32_f16 products into one fp32 accumulator instead of the six bf16 ones of "f32x3".  What the
// two planes hold is 22 significant bits of the value where three bf16 planes hold all 24 -- the error against fp64 of
// every product of the tower stays at or below the fp32-MFMA kernel's own (profiles/r06_f16x2_probe.txt: the probe and
// its gate; tests/test_gpu_f16x2.py: the same bound on these kernels) -- and fp16's exponent range, which
// is why every plane tensor carries a scale (engine_f16x2.py).  Same reference lines as gemm_bf16x3.hip (models.py:59-60,
// train.py:141).  Everything but the 16-bit type, the plane count of the epilogues and the two scales of BArgs is the
// code of gemm_bf16_256.hip: the tile, the LDS images, the DMA schedule and its hand-counted waits do not know the type.
#define CDML_F16X2 1
#define k_gemm_bf16_256 k_gemm_f16x2_256          // (the kernels' own names in profiles and code objects)
#define k_gemm_x3_rounds k_gemm_f16x2_rounds
#define k_gemm_bf16_sk k_gemm_f16x2_sk_unused
#define k_sk_fixup_tn k_sk_fixup_tn_f16x2_unused
#include "gemm_bf16_256.hip"
