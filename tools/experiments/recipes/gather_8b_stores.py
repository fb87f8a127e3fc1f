# A/B: the split-fp32 gather with round 3's 8-B plane stores (three per chunk) instead of the 16-B lane-pair form
EDITS = [("sampler_gather.hip", "    if ((plane & 7) == 0) {\n      // 16-B plane stores", "    if (false) {\n      // 16-B plane stores")]
