# STATUS: ADOPTED in round 4 (csrc/gemm_bf16_256.hip now has this layout; the recipe applies to the round-3 source only)
# LDS layout by operand instead of by buffer: [A: buf0 h0 | buf0 h1 | buf1 h0 | buf1 h1][B: likewise] -- every fragment read of
# an operand is then within 64 KiB of one lane base (16-bit ds offset immediates), no per-read v_add for the second buffer
EDITS = [
    ("gemm_bf16_256.hip", "buf * BUF + (hh * 2 + img) * IMG", "img * (4 * IMG) + buf * (2 * IMG) + hh * IMG", 2),
    ("gemm_bf16_256.hip", "smem + IMG + (wc * 32 + l31) * 128", "smem + 4 * IMG + (wc * 32 + l31) * 128"),
    ("gemm_bf16_256.hip", "smem + IMG + (wc * 32 + l15) * 128", "smem + 4 * IMG + (wc * 32 + l15) * 128"),
    ("gemm_bf16_256.hip", "buf * BUF + IMG + hh * 2 * IMG", "4 * IMG + buf * (2 * IMG) + hh * IMG", 2),
    ("gemm_bf16_256.hip", "buf * BUF + hh * 2 * IMG", "buf * (2 * IMG) + hh * IMG", 6),
]
