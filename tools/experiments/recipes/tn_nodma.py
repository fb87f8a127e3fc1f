# timing ablation: no LDS-DMA issued inside the K loop of the 16x16x32 forms (operands stay those of the prologue)
EDITS = [("gemm_bf16_256.hip", """    if (X3 && x3_run) {
      const int pl = (0x120100 >> (4 * (xp_a & 7))) & 3;
      stage_pw(0, 0, pl, xp_a >> 3, tile + 1, buf ^ 1);
      stage_pw(0, 1, pl, xp_a >> 3, tile + 1, buf ^ 1);
    } else {
      stage(0, 0, tile + 1, buf ^ 1);
      stage(0, 1, tile + 1, buf ^ 1);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");""", """    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");"""),
         ("gemm_bf16_256.hip", """    if (X3 && x3_run) {
      const int pl = (0x102010 >> (4 * (xp_b & 7))) & 3;
      stage_pw(1, 0, pl, xp_b >> 3, tile + 2, buf);
      stage_pw(1, 1, pl, xp_b >> 3, tile + 2, buf);
    } else {
      stage(1, 0, tile + 2, buf);
      stage(1, 1, tile + 2, buf);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");""", """    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");""")]
