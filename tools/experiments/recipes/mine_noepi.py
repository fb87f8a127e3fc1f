# timing ablation: the mining launch WITHOUT its epilogue (results are garbage): what the K = 256 loop + prologue cost alone
EDITS = [("gemm_bf16_256.hip", "  if constexpr (EPI == BE_MINE_X3) {\n    const float inf = __builtin_huge_valf();",
          "  if constexpr (EPI == BE_MINE_X3) {\n    if (g.M > 0) return;\n    const float inf = __builtin_huge_valf();")]
