# timing experiment: the half tile's loop without its LDS-DMA issues (results wrong): how much of the half-tile round is the
# DMA pieces in the phases' read parts?
EDITS = [("gemm_bf16_256.hip",
          "        dma_s(img == 0 ? srd_a : srd_b, img == 0 ? va[0] : (h1 ? vb_h1[0] : vb[0]), so, dst);\n        dma_s(img == 0 ? srd_a : srd_b, img == 0 ? va[1] : (h1 ? vb_h1[1] : vb[1]), so, dst + 1024);\n",
          "        asm volatile(\"\" :: \"s\"(so), \"s\"(dst), \"v\"(h1 ? vb_h1[0] : vb[0]));\n")]
