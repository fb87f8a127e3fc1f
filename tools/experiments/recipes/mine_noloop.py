# timing ablation: the mining launch WITHOUT its K loop (zeros are mined): what the epilogue costs alone
EDITS = [("gemm_bf16_256.hip", "    const int n_per = n_ktiles / 6;                        // whole periods (host)",
          "    const int n_per = EPI == BE_MINE_X3 ? 0 : n_ktiles / 6;")]
