# timing experiment: the plane-output epilogue computes everything and stores nothing (every store behind a never-true
# uniform condition, its operand pinned by an empty asm).  Results are wrong.  What do the epilogue's instructions cost
# without their stores?
EDITS = [("gemm_bf16_256.hip",
          "          *reinterpret_cast<bf16x8 *>(dst) = o;\n",
          "          asm volatile(\"\" :: \"v\"(o));\n          if (g.alpha == 12345.f) *reinterpret_cast<bf16x8 *>(dst) = o;\n"),
         ("gemm_bf16_256.hip",
          "            *reinterpret_cast<bf16x8 *>(dst + pl * g.x3_plane_c) = o;\n",
          "            asm volatile(\"\" :: \"v\"(o));\n            if (g.alpha == 12345.f) *reinterpret_cast<bf16x8 *>(dst + pl * g.x3_plane_c) = o;\n"),
         ("gemm_bf16_256.hip",
          "          if (kBiasEpi && g.mask_out)                        // this lane's 8 columns = one byte of the sign bitmask\n",
          "          asm volatile(\"\" :: \"v\"(bits));\n          if (kBiasEpi && g.mask_out && g.alpha == 12345.f)\n")]
