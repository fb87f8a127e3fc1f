# timing experiment: the plane stores of the 16-B epilogue (FC1's h1 planes, the data gradient's dz1 planes) as
# non-temporal stores (lines marked for early eviction from the XCD's L2, which also holds the operand panels)
EDITS = [("gemm_bf16_256.hip",
          "          *reinterpret_cast<u32x4 *>(ub + lane_c) = o;\n",
          "          __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(ub + lane_c));\n"),
         ("gemm_bf16_256.hip",
          "              *reinterpret_cast<u32x4 *>(ub + (size_t)pl * plane_bytes + lane_c) = o;\n",
          "              __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(ub + (size_t)pl * plane_bytes + lane_c));\n")]
