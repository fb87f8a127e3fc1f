# timing experiment: every tile of the 256x256 ping-pong kernel loads the operand panels of tile (0, 0) (same instruction
# stream and MFMA work on live data, one pair of panels serving the whole launch out of L2: fabric traffic gone); results
# are wrong.  What would a perfect L2 raster buy?  (The fp32 kernels: nothing, profiles/r03_f32_experiments.txt.)
EDITS = [("gemm_bf16_256.hip",
          "  run_tile<TN, EPI, S16, X3, F6, NTCS, R6>(g, tm, m0, n0, k_begin, n_ktiles, c_base, g.ldc, m0, n0, cs_row, g.N, smem);",
          "  run_tile<TN, EPI, S16, X3, F6, NTCS, R6>(g, tm, 0, 0, k_begin, n_ktiles, c_base, g.ldc, m0, n0, cs_row, g.N, smem);")]
