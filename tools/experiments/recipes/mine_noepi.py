# timing ablation: the mining launch WITHOUT its epilogue (results are garbage): what the K = 256 loop + prologue cost alone
# (round 6: the operand-swapped, LDS-free epilogue; `g.M > 0` keeps the accumulators -- and with them the loop -- alive)
EDITS = [("gemm_bf16_256.hip", "    int lane_e = lane;\n    asm volatile(\"\" : \"+v\"(lane_e));\n    const int l15 = lane_e & 15, q16 = lane_e >> 4;\n    const float inf = __builtin_huge_valf();\n    // the lane's 16 columns",
          "    if (g.M > 0) return;\n    int lane_e = lane;\n    asm volatile(\"\" : \"+v\"(lane_e));\n    const int l15 = lane_e & 15, q16 = lane_e >> 4;\n    const float inf = __builtin_huge_valf();\n    // the lane's 16 columns")]
