# timing experiment: non-temporal plane stores in the MASK epilogue only (the data gradient's dz1 planes)
EDITS = [("gemm_bf16_256.hip",
          "          *reinterpret_cast<u32x4 *>(ub + lane_c) = o;\n",
          "          if constexpr (kMaskEpi) __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(ub + lane_c));\n          else *reinterpret_cast<u32x4 *>(ub + lane_c) = o;\n"),
         ("gemm_bf16_256.hip",
          "              *reinterpret_cast<u32x4 *>(ub + (size_t)pl * plane_bytes + lane_c) = o;\n",
          "              if constexpr (kMaskEpi) __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(ub + (size_t)pl * plane_bytes + lane_c));\n              else *reinterpret_cast<u32x4 *>(ub + (size_t)pl * plane_bytes + lane_c) = o;\n")]
