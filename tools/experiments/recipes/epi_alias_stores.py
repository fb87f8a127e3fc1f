# timing experiment: every tile of the plane-output epilogue stores into the region of tile (0, 0) (same instructions, the
# 393 KB per tile stay in L2: no fabric / HBM writes).  Results are wrong.  Is the epilogue bound by where its bytes go?
EDITS = [("gemm_bf16_256.hip",
          "          bf16 *dst = static_cast<bf16 *>(c_base) + (int64_t)(row - m0 + c_row0) * c_ld + c_col0 + lcol8;\n",
          "          bf16 *dst = static_cast<bf16 *>(c_base) + (int64_t)(row - m0) * c_ld + lcol8;\n"),
         ("gemm_bf16_256.hip",
          "            g.mask_out[(int64_t)row * g.ldmask + ((n0 + lcol8) >> 3)] = (uint8_t)bits;\n          // three roundings",
          "            g.mask_out[(int64_t)(row - m0) * g.ldmask + ((lcol8) >> 3)] = (uint8_t)bits;\n          // three roundings")]
