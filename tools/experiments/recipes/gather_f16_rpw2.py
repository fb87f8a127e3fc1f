# A/B: the fp16 fused gather with two rows in flight per wave instead of four
EDITS = [("sampler_gather.hip", "#define CDML_GATHER_ROWS_PER_WAVE_F16 4", "#define CDML_GATHER_ROWS_PER_WAVE_F16 2")]
