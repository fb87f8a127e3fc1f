# A/B: the persistent gather kernels on up to 16 blocks per CU instead of 8
EDITS = [("sampler_gather.hip", "const int64_t cap = (int64_t)kNumCU * 8;", "const int64_t cap = (int64_t)kNumCU * 16;")]
