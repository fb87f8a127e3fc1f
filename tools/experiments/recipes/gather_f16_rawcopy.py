# timing ablation: the fp16 fused gather as a RAW COPY of the row (no conversion, no norm): the rate of random 3-KB rows read + written
EDITS = [("sampler_gather.hip", "    ss = wave_sum(ss);\n    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));\n    bf16x8v *d = reinterpret_cast<bf16x8v *>(dst);",
          "    const float inv = 1.0f;\n    bf16x8v *d = reinterpret_cast<bf16x8v *>(dst);"),
         ("sampler_gather.hip", "        for (int u = 0; u < 8; ++u) o[u] = (__bf16)((float)R.v[c][u] * inv);",
          "        o = __builtin_bit_cast(bf16x8v, R.v[c]);"),
         ("sampler_gather.hip", "      for (int u = 0; u < 8; ++u) {\n        if (8 * q + u >= F) R.v[c][u] = 0;       // never trust the pad\n        const float f = (float)R.v[c][u];\n        ss += f * f;\n      }",
          "      (void)q;")]
