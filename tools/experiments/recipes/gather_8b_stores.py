# STATUS: the 16-B store form was measured (nothing) and reverted; this A/B switch applies to the commit that had it (19e6f..: see
# profiles/r04_stagger_and_fc1_rounds.txt item 4), not to the tree
# A/B: the split-fp32 gather with round 3's 8-B plane stores (three per chunk) instead of the 16-B lane-pair form
EDITS = [("sampler_gather.hip", "    if ((plane & 7) == 0) {\n      // 16-B plane stores", "    if (false) {\n      // 16-B plane stores")]
