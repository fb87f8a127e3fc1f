# A/B: the fp16 row norm as rounds 1-4 computed it (eight conversions + eight fmas per chunk, the pad test on every element)
EDITS = [("common.h", """  if (8 * q + 8 > F) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (8 * q + u >= F) x[u] = 0;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const cdml_half2 p = {x[2 * e], x[2 * e + 1]};
    ss = __builtin_amdgcn_fdot2(p, p, ss, false);
  }
  return ss;""", """#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (8 * q + u >= F) x[u] = 0;
    const float f = (float)x[u];
    ss += f * f;
  }
  return ss;""")]
