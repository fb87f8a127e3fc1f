# A/B: non-temporal stores of the gathered rows
EDITS = [("sampler_gather.hip", "#define CDML_GATHER_NT_STORE 0", "#define CDML_GATHER_NT_STORE 1")]
