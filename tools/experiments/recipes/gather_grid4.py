# A/B: the persistent gather kernels on 4 blocks per CU instead of 8
EDITS = [("sampler_gather.hip", "const int64_t cap = (int64_t)kNumCU * 8;", "const int64_t cap = (int64_t)kNumCU * 4;")]
