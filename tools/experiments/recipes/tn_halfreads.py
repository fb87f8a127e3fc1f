# timing ablation: phase B of the 16x16x32 K-tile reuses phase A's A fragments (half the A fragment reads; wrong numbers)
EDITS = [("gemm_bf16_256.hip", """    CDML_BARRIER();
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) fa[ks2][rb] = read_a16(buf, 1, rb, ks2);
    if (X3 && x3_run) {""", """    CDML_BARRIER();
    if (X3 && x3_run) {""")]
