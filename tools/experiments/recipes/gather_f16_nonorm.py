# timing ablation: the fp16 fused gather WITHOUT its norm (no wave reduction, scale 1): is the kernel bound by its per-row instructions?
EDITS = [("sampler_gather.hip", "    ss = wave_sum(ss);\n    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));\n    bf16x8v *d = reinterpret_cast<bf16x8v *>(dst);",
          "    const float inv = 1.0f + 0.f * ss;\n    bf16x8v *d = reinterpret_cast<bf16x8v *>(dst);")]
