# k-strided form without the bias-gradient column sums riding along (db comes out wrong: timing only)
EDITS = [("gemm_bf16_256.hip", "const bool cs_on = (TN || (NTCS && X3 && S16)) && cs_row != nullptr;",
          "const bool cs_on = (NTCS && X3 && S16) && cs_row != nullptr;")]
