// Row l2-normalise (fwd/bwd) and the fused triplet hinge loss + gradient.
//
// Reference: tf.nn.l2_normalize at models.py:58,61; HingeLoss.calculate_loss
// losses.py:32-38 (squared-L2 pos/neg distance, max(pos-neg+margin,0), batch
// mean) and its autodiff (train.py:141).  TF's MaximumGrad is inclusive: a
// triplet with pos-neg+margin == 0 still passes gradient.
//
// Roofline: HBM (3*D*4 B read + 3*D*4 B written per triplet).  One wave per
// triplet / row, 16-B lane accesses, wave-shuffle reductions, no atomics: the
// batch mean is a second single-block pass in a fixed order (deterministic).
#include "common.h"

namespace cdml {
namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / kWave;
constexpr float kL2Eps = 1e-12f;

__device__ __forceinline__ float4 ld4(const float *p, int q) {
  return reinterpret_cast<const float4 *>(p)[q];
}
__device__ __forceinline__ void st4(float *p, int q, float4 v) {
  reinterpret_cast<float4 *>(p)[q] = v;
}
__device__ __forceinline__ float sq4(float4 a) { return a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// ------------------------------------------------------------------ l2 norm ---
__global__ void __launch_bounds__(kThreads)
k_l2norm_fwd(const float *__restrict__ x, int64_t ldx, int M, int N, float *__restrict__ y,
             int64_t ldy, float *__restrict__ inv_out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = N >> 2;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < M; r += gridDim.x * kWavesPerBlock) {
    const float *xr = x + (int64_t)r * ldx;
    float ss = 0.f;
    for (int q = lane; q < nq; q += kWave) ss += sq4(ld4(xr, q));
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(fmaxf(ss, kL2Eps));
    if (inv_out && lane == 0) inv_out[r] = inv;
    float *yr = y + (int64_t)r * ldy;
    for (int q = lane; q < nq; q += kWave) st4(yr, q, mul4(ld4(xr, q), inv));
  }
}

__global__ void __launch_bounds__(kThreads)
k_l2norm_bwd(const float *__restrict__ z, int64_t ldz, const float *__restrict__ g, int64_t ldg,
             int M, int N, float alpha, float *__restrict__ dz, int64_t lddz) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = N >> 2;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < M; r += gridDim.x * kWavesPerBlock) {
    const float *zr = z + (int64_t)r * ldz;
    const float *gr = g + (int64_t)r * ldg;
    float ss = 0.f, zg = 0.f;
    for (int q = lane; q < nq; q += kWave) {
      const float4 a = ld4(zr, q);
      ss += sq4(a);
      zg += dot4(a, ld4(gr, q));
    }
    ss = wave_sum(ss);
    zg = wave_sum(zg);
    const float inv = 1.0f / sqrtf(fmaxf(ss, kL2Eps));
    // y = z*inv, dot = sum(y*g) = inv*zg;  dz = inv*(g - y*dot)  (or inv*g when clamped)
    const float c = (ss > kL2Eps) ? inv * inv * zg : 0.f;
    float *dr = dz + (int64_t)r * lddz;
    for (int q = lane; q < nq; q += kWave) {
      const float4 a = ld4(zr, q), b = ld4(gr, q);
      float4 d = make_float4(inv * (b.x - a.x * c), inv * (b.y - a.y * c), inv * (b.z - a.z * c),
                             inv * (b.w - a.w * c));
      if (alpha >= 0.f) {
        d.x *= (a.x > 0.f) ? 1.f : alpha;
        d.y *= (a.y > 0.f) ? 1.f : alpha;
        d.z *= (a.z > 0.f) ? 1.f : alpha;
        d.w *= (a.w > 0.f) ? 1.f : alpha;
      }
      st4(dr, q, d);
    }
  }
}

// -------------------------------------------------------------- hinge loss ----
// Squared distance pair of one triplet, reduced over the wave.  Kept in one
// function so every wave that recomputes a triplet gets bit-identical values.
__device__ __forceinline__ void dist_pair(const float *a, const float *p, const float *n, int nq,
                                          int lane, float &pos, float &neg) {
  float sp = 0.f, sn = 0.f;
  for (int q = lane; q < nq; q += kWave) {
    const float4 va = ld4(a, q);
    sp += sq4(sub4(va, ld4(p, q)));
    sn += sq4(sub4(va, ld4(n, q)));
  }
  pos = wave_sum(sp);
  neg = wave_sum(sn);
}

__global__ void __launch_bounds__(kThreads)
k_triplet_hinge(const float *__restrict__ e, int64_t lde, int B, int D, float margin,
                float *__restrict__ pos_o, float *__restrict__ neg_o, float *__restrict__ hinge_o,
                float *__restrict__ de, int64_t ldde) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  const float two_over_b = 2.0f / (float)B;
  for (int i = blockIdx.x * kWavesPerBlock + wave; i < B; i += gridDim.x * kWavesPerBlock) {
    const float *a = e + (int64_t)(3 * i) * lde, *p = a + lde, *n = p + lde;
    float pos, neg;
    dist_pair(a, p, n, nq, lane, pos, neg);
    const float t = pos - neg + margin;
    if (lane == 0) {
      pos_o[i] = pos;
      neg_o[i] = neg;
      hinge_o[i] = fmaxf(t, 0.f);
    }
    if (de) {
      const float s = (t >= 0.f) ? two_over_b : 0.f;
      float *da = de + (int64_t)(3 * i) * ldde, *dp = da + ldde, *dn = dp + ldde;
      for (int q = lane; q < nq; q += kWave) {
        const float4 va = ld4(a, q), vp = ld4(p, q), vn = ld4(n, q);
        st4(da, q, mul4(sub4(vn, vp), s));
        st4(dp, q, mul4(sub4(vp, va), s));
        st4(dn, q, mul4(sub4(va, vn), s));
      }
    }
  }
}

// In-batch negatives: slot i owns rows a_i (2i) and p_i (2i+1).  p_i is the
// positive of triplet i and the negative of triplet k = (i - shift) mod B, so the
// wave of slot i recomputes triplet k's activity and writes both rows' complete
// gradients: no atomics, fixed summation order.
__global__ void __launch_bounds__(kThreads)
k_triplet_hinge_inbatch(const float *__restrict__ e, int64_t lde, const int32_t *__restrict__ rows,
                        const int32_t *__restrict__ shift_p, int B, int D, float margin,
                        float *__restrict__ pos_o, float *__restrict__ neg_o,
                        float *__restrict__ hinge_o, uint8_t *__restrict__ valid_o,
                        float *__restrict__ de, int64_t ldde) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  const int shift = *shift_p;
  const float two_over_b = 2.0f / (float)B;
  for (int i = blockIdx.x * kWavesPerBlock + wave; i < B; i += gridDim.x * kWavesPerBlock) {
    const int j = (i + shift) % B;
    const int k = (i - shift % B + B) % B;
    const float *a = e + (int64_t)(2 * i) * lde, *p = a + lde;
    const float *n = e + (int64_t)(2 * j + 1) * lde;
    const int32_t va_id = rows[2 * i], vp_id = rows[2 * i + 1], vn_id = rows[2 * j + 1];
    const bool valid_i = (vn_id != va_id) && (vn_id != vp_id);
    float pos, neg;
    dist_pair(a, p, n, nq, lane, pos, neg);
    const float t = pos - neg + margin;
    if (lane == 0) {
      pos_o[i] = pos;
      neg_o[i] = neg;
      hinge_o[i] = valid_i ? fmaxf(t, 0.f) : 0.f;
      if (valid_o) valid_o[i] = valid_i ? 1 : 0;
    }
    if (de) {
      const float *ak = e + (int64_t)(2 * k) * lde, *pk = ak + lde;  // triplet k: (a_k,p_k,p_i)
      const bool valid_k = (vp_id != rows[2 * k]) && (vp_id != rows[2 * k + 1]);
      float posk, negk;
      dist_pair(ak, pk, p, nq, lane, posk, negk);
      const float si = (valid_i && t >= 0.f) ? two_over_b : 0.f;
      const float sk = (valid_k && (posk - negk + margin) >= 0.f) ? two_over_b : 0.f;
      float *da = de + (int64_t)(2 * i) * ldde, *dp = da + ldde;
      for (int q = lane; q < nq; q += kWave) {
        const float4 va = ld4(a, q), vp = ld4(p, q), vn = ld4(n, q), vak = ld4(ak, q);
        st4(da, q, mul4(sub4(vn, vp), si));
        const float4 g1 = mul4(sub4(vp, va), si);    // as positive of triplet i
        const float4 g2 = mul4(sub4(vak, vp), sk);   // as negative of triplet k
        st4(dp, q, make_float4(g1.x + g2.x, g1.y + g2.y, g1.z + g2.z, g1.w + g2.w));
      }
    }
  }
}

// ------------------------------------------------------------ fused tower tail ----
// tf.nn.l2_normalize of the output layer (models.py:61) -> HingeLoss.calculate_loss
// (losses.py:32-38) -> its gradient -> l2-normalise backward -> leaky-relu' of the output
// layer (train.py:141), in ONE launch instead of three + the statistics pass: one wave per
// triplet slot keeps the rows of z it needs in registers (NCH float4 per lane per row).  The
// arithmetic is that of k_l2norm_fwd / dist_pair / k_triplet_hinge{,_inbatch} / k_l2norm_bwd,
// operation for operation.  The step's scalars (stats[0..3] = mean hinge, mean pos, mean neg, active
// fraction) come from k_loss_stats in a second small launch, stats[4] = calc_var of the [B,3,D]
// triplet tensor (train.py:67-71) from per-block partials + k_tail_var_final when var_ws is given.
// (Folding them into this launch's last block through ticket words and write-through partial sums
// was built and measured: 21 us against 15 + 6 us for the two launches, i.e. nothing, and it
// leaned on sc1 loads seeing another XCD's sc1 stores -- dropped.)
// MODE 0: rows 3i, 3i+1, 3i+2 = anchor, positive, negative.  MODE 1 (in-batch negatives): slot i
// owns rows 2i (a_i), 2i+1 (p_i); its negative is p_j, j = (i+shift) mod B, and p_i is also the
// negative of slot k = (i-shift) mod B, so the wave of slot i recomputes triplet k's activity
// and writes both of its rows' complete gradients (no atomics, fixed order).
template <int NCH>
struct TailRow {
  float4 v[NCH];
  float ss, inv;
  __device__ __forceinline__ void load(const float *zr, int nq, int lane) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      v[c] = (q < nq) ? ld4(zr, q) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += sq4(v[c]);
    }
    ss = wave_sum(s);
    inv = 1.0f / sqrtf(fmaxf(ss, kL2Eps));
  }
  __device__ __forceinline__ float4 e(int c) const { return mul4(v[c], inv); }
};

template <int NCH>
__device__ __forceinline__ void tail_dist(const TailRow<NCH> &a, const TailRow<NCH> &p, const TailRow<NCH> &n,
                                          float &pos, float &neg) {
  float sp = 0.f, sn = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float4 va = a.e(c);
    sp += sq4(sub4(va, p.e(c)));
    sn += sq4(sub4(va, n.e(c)));
  }
  pos = wave_sum(sp);
  neg = wave_sum(sn);
}

// dz = inv * (g - z*c) * lrelu'(z), c = inv^2 <z,g>  (k_l2norm_bwd); also stores e
template <int NCH>
__device__ __forceinline__ void tail_store_row(const TailRow<NCH> &r, const float4 (&g)[NCH], float alpha,
                                               float *er, float *dr, uint16_t *br, int nq, int lane,
                                               int64_t plane = 0, float h2_scale = 0.f) {
  float zg = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) zg += dot4(r.v[c], g[c]);
  zg = wave_sum(zg);
  const float inv = r.inv;
  const float cc = (r.ss > kL2Eps) ? inv * inv * zg : 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int q = lane + kWave * c;
    if (q >= nq) continue;
    const float4 a = r.v[c], b = g[c];
    float4 d = make_float4(inv * (b.x - a.x * cc), inv * (b.y - a.y * cc), inv * (b.z - a.z * cc),
                           inv * (b.w - a.w * cc));
    if (alpha >= 0.f) {
      d.x *= (a.x > 0.f) ? 1.f : alpha;
      d.y *= (a.y > 0.f) ? 1.f : alpha;
      d.z *= (a.z > 0.f) ? 1.f : alpha;
      d.w *= (a.w > 0.f) ? 1.f : alpha;
    }
    st4(er, q, r.e(c));
    st4(dr, q, d);
    if (br && h2_scale > 0.f) {   // precision "f16x2": the two fp16 planes hi | lo of d * h2_scale (saturating), `plane` apart
      using half4v = __attribute__((ext_vector_type(4))) _Float16;
      const float dv[4] = {d.x, d.y, d.z, d.w};
      half4v hi, lo;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float sv = __builtin_amdgcn_fmed3f(dv[u] * h2_scale, -65504.f, 65504.f);
        hi[u] = (_Float16)sv;
        lo[u] = (_Float16)(sv - (float)hi[u]);
      }
      reinterpret_cast<half4v *>(br)[q] = hi;
      reinterpret_cast<half4v *>(br + plane)[q] = lo;
    } else if (br) {   // bf16 copy (round to nearest even) for the reduced-precision GEMMs
      auto rn = [](float x) -> uint32_t {
        const uint32_t u = __float_as_uint(x);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
      };
      uint2 w;
      w.x = rn(d.x) | (rn(d.y) << 16);
      w.y = rn(d.z) | (rn(d.w) << 16);
      reinterpret_cast<uint2 *>(br)[q] = w;
      if (plane) {   // precision "f32x3": the mid and lo planes too (hi + mid + lo == d exactly), `plane` elements apart
        float r[4] = {d.x, d.y, d.z, d.w};
        uint32_t hb[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
#pragma unroll
        for (int pl = 1; pl < 3; ++pl) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            r[u] -= __uint_as_float(hb[u] << 16);
            hb[u] = rn(r[u]);
          }
          uint2 m;
          m.x = hb[0] | (hb[1] << 16);
          m.y = hb[2] | (hb[3] << 16);
          reinterpret_cast<uint2 *>(br + pl * plane)[q] = m;
        }
      }
    }
  }
}

template <int MODE, int NCH>
__global__ void __launch_bounds__(kThreads)
k_vnet_tail(const float *__restrict__ z, int64_t ldz, const int32_t *__restrict__ rows,
            const int32_t *__restrict__ shift_p, int B, int D, float margin, float alpha,
            float *__restrict__ e, int64_t lde, float *__restrict__ pos_o, float *__restrict__ neg_o,
            float *__restrict__ hinge_o, uint8_t *__restrict__ valid_o, float *__restrict__ dz2,
            int64_t lddz, uint16_t *__restrict__ dz2_bf, int64_t ldbf, float *__restrict__ var_ws,
            int64_t plane_bf, float h2_scale) {
  __shared__ __attribute__((aligned(16))) float s_red[kWavesPerBlock][8];
  extern __shared__ __attribute__((aligned(16))) float s_col[];                 // [kWavesPerBlock][D] when var_ws
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  const int shift = (MODE == 1) ? *shift_p : 0;
  const float two_over_b = 2.0f / (float)B;
  float4 csum[NCH];
  float tsq = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) csum[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = blockIdx.x * kWavesPerBlock + wave; i < B; i += gridDim.x * kWavesPerBlock) {
    TailRow<NCH> A, P, N;
    float4 ga[NCH], gp[NCH];
    int64_t ra, rp;
    bool valid_i = true;
    float pos, neg;
    if (MODE == 0) {
      ra = 3 * (int64_t)i; rp = ra + 1;
      A.load(z + ra * ldz, nq, lane);
      P.load(z + rp * ldz, nq, lane);
      N.load(z + (ra + 2) * ldz, nq, lane);
      tail_dist<NCH>(A, P, N, pos, neg);
      const float t = pos - neg + margin;
      const float s = (t >= 0.f) ? two_over_b : 0.f;       // MaximumGrad: inclusive at 0
      float4 gn[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4 va = A.e(c), vp = P.e(c), vn = N.e(c);
        ga[c] = mul4(sub4(vn, vp), s);
        gp[c] = mul4(sub4(vp, va), s);
        gn[c] = mul4(sub4(va, vn), s);
      }
      if (lane == 0) { pos_o[i] = pos; neg_o[i] = neg; hinge_o[i] = fmaxf(t, 0.f); }
      tail_store_row<NCH>(N, gn, alpha, e + (ra + 2) * lde, dz2 + (ra + 2) * lddz,
                          dz2_bf ? dz2_bf + (ra + 2) * ldbf : nullptr, nq, lane, plane_bf, h2_scale);
    } else {
      const int j = (i + shift) % B;
      const int k = (i - shift % B + B) % B;
      ra = 2 * (int64_t)i; rp = ra + 1;
      const int32_t va_id = rows[2 * i], vp_id = rows[2 * i + 1], vn_id = rows[2 * j + 1];
      valid_i = (vn_id != va_id) && (vn_id != vp_id);
      const bool valid_k = (vp_id != rows[2 * k]) && (vp_id != rows[2 * k + 1]);
      TailRow<NCH> AK, PK;
      A.load(z + ra * ldz, nq, lane);
      P.load(z + rp * ldz, nq, lane);
      N.load(z + (int64_t)(2 * j + 1) * ldz, nq, lane);
      AK.load(z + (int64_t)(2 * k) * ldz, nq, lane);
      PK.load(z + (int64_t)(2 * k + 1) * ldz, nq, lane);
      tail_dist<NCH>(A, P, N, pos, neg);
      float posk, negk;
      tail_dist<NCH>(AK, PK, P, posk, negk);               // triplet k = (a_k, p_k, p_i)
      const float t = pos - neg + margin;
      const float si = (valid_i && t >= 0.f) ? two_over_b : 0.f;
      const float sk = (valid_k && (posk - negk + margin) >= 0.f) ? two_over_b : 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4 va = A.e(c), vp = P.e(c), vn = N.e(c), vak = AK.e(c);
        ga[c] = mul4(sub4(vn, vp), si);
        const float4 g1 = mul4(sub4(vp, va), si);          // as positive of triplet i
        const float4 g2 = mul4(sub4(vak, vp), sk);         // as negative of triplet k
        gp[c] = make_float4(g1.x + g2.x, g1.y + g2.y, g1.z + g2.z, g1.w + g2.w);
      }
      if (lane == 0) {
        pos_o[i] = pos; neg_o[i] = neg;
        hinge_o[i] = valid_i ? fmaxf(t, 0.f) : 0.f;
        if (valid_o) valid_o[i] = valid_i ? 1 : 0;
      }
    }
    tail_store_row<NCH>(A, ga, alpha, e + ra * lde, dz2 + ra * lddz, dz2_bf ? dz2_bf + ra * ldbf : nullptr, nq, lane, plane_bf, h2_scale);
    tail_store_row<NCH>(P, gp, alpha, e + rp * lde, dz2 + rp * lddz, dz2_bf ? dz2_bf + rp * ldbf : nullptr, nq, lane, plane_bf, h2_scale);
    if (var_ws) {       // column sums and sum of squares of the [B,3,D] triplet tensor
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4 va = A.e(c), vp = P.e(c), vn = N.e(c);
        csum[c].x += va.x + vp.x + vn.x; csum[c].y += va.y + vp.y + vn.y;
        csum[c].z += va.z + vp.z + vn.z; csum[c].w += va.w + vp.w + vn.w;
        tsq += sq4(va) + sq4(vp) + sq4(vn);
      }
    }
  }
  if (!var_ws) return;
  // ---- block partial of the variance summary: D column sums of the [B,3,D] triplet tensor and its
  // sum of squares (waves in order: deterministic); k_tail_var_final adds the blocks in order ----
  float *mine = s_col + wave * D;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int q = lane + kWave * c;
    if (q < nq) st4(mine, q, csum[c]);
  }
  tsq = wave_sum(tsq);
  if (lane == 0) s_red[wave][0] = tsq;
  __syncthreads();
  float *out = var_ws + (int64_t)blockIdx.x * (D + 4);
  for (int d = threadIdx.x; d < D; d += kThreads) {
    float v = 0.f;
    for (int w = 0; w < kWavesPerBlock; ++w) v += s_col[w * D + d];
    out[d] = v;
  }
  if (threadIdx.x == 0) {
    float v = 0.f;
    for (int w = 0; w < kWavesPerBlock; ++w) v += s_red[w][0];
    out[D] = v;
  }
}

// stats[4] = calc_var (train.py:67-71) = [sum t^2 - n_rows * sum_d mean_d^2] / (n_rows * D), n_rows = 3B,
// from the nb block partials of k_vnet_tail (fixed order, double accumulation).
__global__ void __launch_bounds__(kThreads)
k_tail_var_final(const float *__restrict__ var_ws, int nb, int B, int D, float *__restrict__ stats) {
  __shared__ double s_v[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const double n_rows = 3.0 * (double)B;
  double vsum = 0.0;
  for (int d = threadIdx.x; d < D; d += kThreads) {
    double col = 0.0;
    for (int b = 0; b < nb; ++b) col += (double)var_ws[(int64_t)b * (D + 4) + d];
    const double mean = col / n_rows;
    vsum -= n_rows * mean * mean;
  }
  for (int b = threadIdx.x; b < nb; b += kThreads) vsum += (double)var_ws[(int64_t)b * (D + 4) + D];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vsum += __shfl_xor(vsum, off, 64);
  if (lane == 0) s_v[wave] = vsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w = 0; w < kWavesPerBlock; ++w) v += s_v[w];
    stats[4] = (float)(v / (n_rows * (double)D));
  }
}

// ---------------------------------------------------- semi-hard mining (config 2) --
// Build-defined (the reference only draws uniform negatives); spec in
// oracle/tower.py semihard_select.  sqn[r] = |e_r|^2.
__global__ void __launch_bounds__(kThreads)
k_row_sumsq(const float *__restrict__ e, int64_t lde, int R, int D, float *__restrict__ sqn) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < R; r += gridDim.x * kWavesPerBlock) {
    float ss = 0.f;
    for (int q = lane; q < nq; q += kWave) ss += sq4(ld4(e + (int64_t)r * lde, q));
    ss = wave_sum(ss);
    if (lane == 0) sqn[r] = ss;
  }
}

struct Cand { float d; int c; };
__device__ __forceinline__ bool closer(float d, int c, const Cand &b) {   // min, ties -> smaller index
  return d < b.d || (d == b.d && c < b.c);
}
__device__ __forceinline__ bool farther(float d, int c, const Cand &b) {  // max, ties -> smaller index
  return d > b.d || (d == b.d && c < b.c);
}

// One wave per anchor scans its row of S = E_a . E^T (dot products with every
// embedded row): dist = sqn[a] + sqn[c] - 2 S; eligible = other videos.
__global__ void __launch_bounds__(kThreads)
k_semihard_select(const float *__restrict__ S, int64_t ldS, const float *__restrict__ sqn,
                  const int32_t *__restrict__ rows, int B, int32_t *__restrict__ neg_row) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int R = 2 * B;
  const float inf = __builtin_huge_valf();
  for (int i = blockIdx.x * kWavesPerBlock + wave; i < B; i += gridDim.x * kWavesPerBlock) {
    const float *Si = S + (int64_t)i * ldS;
    const float sa = sqn[2 * i];
    const int32_t va = rows[2 * i], vp = rows[2 * i + 1];
    const float dp = sa + sqn[2 * i + 1] - 2.0f * Si[2 * i + 1];
    Cand out{inf, 0x7fffffff}, in{-inf, 0x7fffffff};
    // four candidates per lane and load: the S row, the video ids and the squared norms all as 16-B
    // accesses (R = 2B is a multiple of 4)
    auto scan4 = [&](int q, const float4 s4, const int4 r4, const float4 n4) {
      const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
      const int32_t rv[4] = {r4.x, r4.y, r4.z, r4.w};
      const float nv[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = 4 * q + u;
        if (rv[u] != va && rv[u] != vp) {
          const float d = sa + nv[u] - 2.0f * sv[u];
          if (d > dp) { if (closer(d, c, out)) out = Cand{d, c}; }
          if (farther(d, c, in)) in = Cand{d, c};
        }
      }
    };
    const int nq = R >> 2;
    const int4 *rows4 = reinterpret_cast<const int4 *>(rows);
    int q = lane;
    for (; q + 3 * kWave < nq; q += 4 * kWave) {     // four groups (16 candidates) per lane in flight
      float4 sv[4], nv[4];
      int4 rv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        const f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(Si) + q + u * kWave);   // S is read once
        sv[u] = make_float4(t4.x, t4.y, t4.z, t4.w);
        rv[u] = rows4[q + u * kWave];
        nv[u] = ld4(sqn, q + u * kWave);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) scan4(q + u * kWave, sv[u], rv[u], nv[u]);
    }
    for (; q < nq; q += kWave) scan4(q, ld4(Si, q), rows4[q], ld4(sqn, q));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_xor(out.d, off, 64); const int oc = __shfl_xor(out.c, off, 64);
      if (closer(od, oc, out)) out = Cand{od, oc};
      const float id = __shfl_xor(in.d, off, 64); const int ic = __shfl_xor(in.c, off, 64);
      if (farther(id, ic, in)) in = Cand{id, ic};
    }
    if (lane == 0) neg_row[i] = (out.c != 0x7fffffff) ? out.c : (in.c != 0x7fffffff ? in.c : -1);
  }
}

// Indexed triplets: anchor row 2i, positive 2i+1, negative neg_row[i] (any row of
// e, or -1 = masked).  Forward: one wave per triplet; scale[i] = 2/B if active.
__global__ void __launch_bounds__(kThreads)
k_hinge_indexed_fwd(const float *__restrict__ e, int64_t lde, const int32_t *__restrict__ neg_row,
                    int B, int D, float margin, float *__restrict__ pos_o, float *__restrict__ neg_o,
                    float *__restrict__ hinge_o, float *__restrict__ scale) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  const float two_over_b = 2.0f / (float)B;
  for (int i = blockIdx.x * kWavesPerBlock + wave; i < B; i += gridDim.x * kWavesPerBlock) {
    const int nr = neg_row[i];
    const bool valid = nr >= 0;
    const float *a = e + (int64_t)(2 * i) * lde, *p = a + lde;
    const float *n = e + (int64_t)(valid ? nr : 2 * i) * lde;
    float pos, neg;
    dist_pair(a, p, n, nq, lane, pos, neg);
    const float t = pos - neg + margin;
    if (lane == 0) {
      pos_o[i] = pos;
      neg_o[i] = neg;
      hinge_o[i] = valid ? fmaxf(t, 0.f) : 0.f;
      // round 6: the scratch word of a triplet is its KEY for the backward scan -- the mined row when the triplet is
      // active (its gradient scale is then 2 / B), -1 when it is not (scale 0): one int per triplet to scan instead of two
      reinterpret_cast<int32_t *>(scale)[i] = (valid && t >= 0.f) ? nr : -1;
    }
  }
}

// Backward: the gradient of every embedded row r = its own triplet's term, then the terms of every triplet that mined r
// as its negative in ascending triplet order (no atomics: bit-reproducible).
// Round 6.  Rounds 2-5 gave every ROW a wave that scanned all B triplets for its row in global memory (2B waves x 8 B x B =
// 1 GB of L2 reads at B = 8 192) and added its hits one dependent round trip after the other: 62 us of BASELINE config
// 2's step.  Now: the forward pass leaves ONE word per triplet, its key (the mined row if the triplet is active, else
// -1); a block copies the keys into LDS once (32 KB at B = 8 192; longer batches in chunks) and each of its waves -- four
// consecutive rows to a wave, their gradients in registers -- scans them THERE (ds_read_b128, no global latency in the
// loop), appends its hits in ascending triplet order to a list of its own and adds them in that order with the loads of
// kBatch hits in flight together (semi-hard mining sends many anchors to the same few "hub" rows: 27 on one row after 30
// steps of config 2; a hub is a chain on one wave).  Intermediate forms of this round, measured and replaced: hits of a
// 16- / 64-row block compacted through LDS with __syncthreads per 1 024 triplets (bound by that chain of barriers and
// global round trips: 30 us without any hub); 16 rows to a wave (4 x slower than the scan it replaced: the rows' own
// dependent loads); the wave scanning global memory itself (8 iterations x ~2 us of load latency).  Same terms, same
// order per row as ever: the same bits.
// Fused tail (z != null): once a row's gradient is complete its wave runs k_l2norm_bwd's arithmetic on it (dz2 = the
// gradient of the output layer's pre-activation, leaky-relu' included) and, on request, writes dz2's bf16 copy or its
// three exact planes -- two more launches of config 2's step (l2norm_bwd, split) folded into this one.
constexpr int kIdxRowsPerWave = 4;
constexpr int kIdxRows = kIdxRowsPerWave * kWavesPerBlock;      // rows per block
constexpr int kIdxKeyChunk = 8192;                              // keys held in LDS at a time
template <int NCH>
__global__ void __launch_bounds__(kThreads)
k_hinge_indexed_bwd(const float *__restrict__ e, int64_t lde, const int32_t *__restrict__ neg_row,
                    const float *__restrict__ scale, int B, int D, float *__restrict__ de,
                    int64_t ldde, const float *__restrict__ z, int64_t ldz, float alpha,
                    float *__restrict__ dz2, int64_t lddz, uint16_t *__restrict__ dz2_bf, int64_t ldbf,
                    int64_t plane_bf) {
  constexpr int kListCap = 512;
  __shared__ __attribute__((aligned(16))) int s_key[kIdxKeyChunk];
  __shared__ int s_mine[kWavesPerBlock][kListCap];   // per wave: (triplet << 2) | row within the wave's four
  const int32_t *__restrict__ key = reinterpret_cast<const int32_t *>(scale);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  const int R = 2 * B;
  const int rw0 = (blockIdx.x * kWavesPerBlock + wave) * kIdxRowsPerWave;       // this wave's rows: rw0 .. rw0 + 3
  const bool live = rw0 < R;                         // (a wave past the last row still helps to copy the keys)
  const float two_over_b = 2.0f / (float)B;          // the forward pass's scale of an active triplet
  // the rows' gradients live in REGISTERS (NCH float4 per lane and row) until they are complete
  float4 acc[kIdxRowsPerWave][NCH];
#pragma unroll
  for (int u = 0; u < kIdxRowsPerWave; ++u) {        // own triplet: anchor or positive role
    const int r = min(rw0 + u, R - 1);
    const int i = r >> 1;
    const float si = key[i] >= 0 ? two_over_b : 0.f;
    const float *a = e + (int64_t)(2 * i) * lde, *p = a + lde;
    const int nr = neg_row[i];
    const float *n = e + (int64_t)(nr >= 0 ? nr : 2 * i) * lde;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      if (q < nq) {
        const float4 va = ld4(a, q), vp = ld4(p, q), vn = ld4(n, q);
        acc[u][c] = (r & 1) ? mul4(sub4(vp, va), si) : mul4(sub4(vn, vp), si);
      } else {
        acc[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  constexpr int kBatch = NCH == 1 ? 16 : (NCH == 2 ? 4 : 2);
  int n_mine = 0;                                    // wave-uniform
  auto flush = [&]() {                               // add the listed hits, in list order, kBatch loads in flight
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int h0 = 0; h0 < n_mine; h0 += kBatch) {
      float4 ga[kBatch][NCH], ge[kBatch][NCH];
      int ub[kBatch];
#pragma unroll
      for (int b = 0; b < kBatch; ++b) {
        const bool on = h0 + b < n_mine;
        const int hv = on ? s_mine[wave][h0 + b] : 0;
        const int jj = hv >> 2, u = hv & 3;
        ub[b] = on ? u : -1;
        const float *aj = e + (int64_t)(2 * jj) * lde;
        const float *er = e + (int64_t)min(rw0 + u, R - 1) * lde;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int q = lane + kWave * c;
          const bool ld = on && q < nq;
          ga[b][c] = ld ? ld4(aj, q) : make_float4(0.f, 0.f, 0.f, 0.f);
          ge[b][c] = ld ? ld4(er, q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int b = 0; b < kBatch; ++b) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const float4 g = mul4(sub4(ga[b][c], ge[b][c]), two_over_b);
#pragma unroll
          for (int uu = 0; uu < kIdxRowsPerWave; ++uu)   // (ub is wave-uniform: a select, not a dynamic register index)
            if (uu == ub[b]) acc[uu][c] = make_float4(acc[uu][c].x + g.x, acc[uu][c].y + g.y, acc[uu][c].z + g.z, acc[uu][c].w + g.w);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    n_mine = 0;
  };
  for (int k0 = 0; k0 < B; k0 += kIdxKeyChunk) {
    const int nk = min(kIdxKeyChunk, B - k0);        // keys of this chunk (the whole batch up to 8 192 triplets)
    if (k0) __syncthreads();                         // every wave is done with the previous chunk
    for (int t = (int)threadIdx.x * 4; t < nk; t += kThreads * 4) {
      if (t + 3 < nk) {
        *reinterpret_cast<int4 *>(s_key + t) = *reinterpret_cast<const int4 *>(key + k0 + t);
      } else {
        for (int u = 0; u < 4 && t + u < nk; ++u) s_key[t + u] = key[k0 + t + u];
      }
    }
    __syncthreads();
    if (!live) continue;
    // the scan: lane l of a step looks at keys j .. j + 3, j = step * 256 + 4 l
    for (int j0 = 0; j0 < nk; j0 += 4 * kWave) {
      const int j = j0 + 4 * lane;
      int4 k4 = make_int4(-1, -1, -1, -1);
      if (j + 3 < nk) {
        k4 = *reinterpret_cast<const int4 *>(s_key + j);
      } else if (j < nk) {
        k4.x = s_key[j];
        if (j + 1 < nk) k4.y = s_key[j + 1];
        if (j + 2 < nk) k4.z = s_key[j + 2];
      }
      const unsigned r0_ = (unsigned)(k4.x - rw0), r1_ = (unsigned)(k4.y - rw0), r2_ = (unsigned)(k4.z - rw0), r3_ = (unsigned)(k4.w - rw0);
      const unsigned h = (unsigned)(r0_ < (unsigned)kIdxRowsPerWave) | ((unsigned)(r1_ < (unsigned)kIdxRowsPerWave) << 1) |
                         ((unsigned)(r2_ < (unsigned)kIdxRowsPerWave) << 2) | ((unsigned)(r3_ < (unsigned)kIdxRowsPerWave) << 3);
      if (__ballot(h != 0) == 0ull) continue;        // (wave-uniform; the usual case)
      // position of a lane's hits in the list: the hits of the lanes below it, then its own in triplet order
      int below = 0, total = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned long long bl = __ballot((h >> u) & 1u);
        below += __popcll(bl & ((1ull << lane) - 1ull));
        total += __popcll(bl);
      }
      if (n_mine + total > kListCap) flush();        // (total <= 256 <= kListCap)
      int pos = n_mine + below;
      const unsigned rl[4] = {r0_, r1_, r2_, r3_};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if ((h >> u) & 1u) s_mine[wave][pos++] = ((k0 + j + u) << 2) | (int)rl[u];
      n_mine += total;
    }
  }
  if (!live) return;
  flush();
#pragma unroll
  for (int u = 0; u < kIdxRowsPerWave; ++u) {
    const int r = rw0 + u;
    if (r >= R) break;
    float *dr = de + (int64_t)r * ldde;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      if (q < nq) st4(dr, q, acc[u][c]);
    }
  }
  if (!z) return;
  // k_l2norm_bwd on the finished rows, operation for operation (its sums run over q = lane, lane + 64, ...: the order
  // of the register chunks)
  float4 zr[kIdxRowsPerWave][NCH];
  float inv[kIdxRowsPerWave], cc[kIdxRowsPerWave];
#pragma unroll
  for (int u = 0; u < kIdxRowsPerWave; ++u) {
    const int r = min(rw0 + u, R - 1);
    const float *zp = z + (int64_t)r * ldz;
    float ss = 0.f, zg = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      zr[u][c] = (q < nq) ? ld4(zp, q) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < nq) {
        ss += sq4(zr[u][c]);
        zg += dot4(zr[u][c], acc[u][c]);
      }
    }
    ss = wave_sum(ss);
    zg = wave_sum(zg);
    inv[u] = 1.0f / sqrtf(fmaxf(ss, kL2Eps));
    cc[u] = (ss > kL2Eps) ? inv[u] * inv[u] * zg : 0.f;
  }
#pragma unroll
  for (int u = 0; u < kIdxRowsPerWave; ++u) {
    const int r = rw0 + u;
    if (r >= R) break;
    float *dr = dz2 + (int64_t)r * lddz;
    uint16_t *br = dz2_bf ? dz2_bf + (int64_t)r * ldbf : nullptr;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int q = lane + kWave * c;
      if (q >= nq) continue;
      const float4 a = zr[u][c], b = acc[u][c];
      const float iv = inv[u], cv = cc[u];
      float4 d = make_float4(iv * (b.x - a.x * cv), iv * (b.y - a.y * cv), iv * (b.z - a.z * cv), iv * (b.w - a.w * cv));
      if (alpha >= 0.f) {
        d.x *= (a.x > 0.f) ? 1.f : alpha;
        d.y *= (a.y > 0.f) ? 1.f : alpha;
        d.z *= (a.z > 0.f) ? 1.f : alpha;
        d.w *= (a.w > 0.f) ? 1.f : alpha;
      }
      st4(dr, q, d);
      if (br) {                                      // bf16 copy (round to nearest even); plane_bf: the mid and lo planes too
        auto rn = [](float x) -> uint32_t {
          const uint32_t u32 = __float_as_uint(x);
          return (u32 + 0x7fffu + ((u32 >> 16) & 1u)) >> 16;
        };
        float rr[4] = {d.x, d.y, d.z, d.w};
        uint32_t hb[4] = {rn(d.x), rn(d.y), rn(d.z), rn(d.w)};
        reinterpret_cast<uint2 *>(br)[q] = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
        if (plane_bf) {
#pragma unroll
          for (int pl = 1; pl < 3; ++pl) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              rr[t] -= __uint_as_float(hb[t] << 16);
              hb[t] = rn(rr[t]);
            }
            reinterpret_cast<uint2 *>(br + pl * plane_bf)[q] = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
          }
        }
      }
    }
  }
}

// Evaluation.mean_dist / mean_cos_dist (evaluate.py:57-90): per co-watch pair the
// squared L2 distance and the dot product of the two embeddings.
__global__ void __launch_bounds__(kThreads)
k_pair_dist(const float *__restrict__ e, int64_t lde, const int32_t *__restrict__ pairs, int P, int D,
            float *__restrict__ sqdist, float *__restrict__ dot) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  for (int i = blockIdx.x * kWavesPerBlock + wave; i < P; i += gridDim.x * kWavesPerBlock) {
    const float *a = e + (int64_t)pairs[2 * i] * lde, *b = e + (int64_t)pairs[2 * i + 1] * lde;
    float sd = 0.f, sp = 0.f;
    for (int q = lane; q < nq; q += kWave) {
      const float4 va = ld4(a, q), vb = ld4(b, q);
      sd += sq4(sub4(va, vb));
      sp += dot4(va, vb);
    }
    sd = wave_sum(sd);
    sp = wave_sum(sp);
    if (lane == 0) { sqdist[i] = sd; dot[i] = sp; }
  }
}

// ------------------------------------------- fusion towers (models.py:65-157) ----
// Elementwise pieces between the FC layers of MultiplyNet / MlpNet / ResNet: the
// multiply fusion tf.multiply(visual, doc) (models.py:90,117,148), ResNet's
// residual sums (models.py:150-154), the leaky-relu derivative applied to a
// gradient that reached an FC output through several consumers.  HBM-bound, 16-B
// lane accesses over [M][N] views with leading dimensions.
// mode 0: out = a*b          mode 1: out = a*b + a + b        mode 2: out = a + b
__global__ void __launch_bounds__(kThreads)
k_ew_combine(int mode, const float *__restrict__ a, int64_t lda, const float *__restrict__ b,
             int64_t ldb, int M, int N, float *__restrict__ out, int64_t ldo) {
  const int nq = N >> 2;
  const int64_t total = (int64_t)M * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nq;
    const int q = (int)(i - r * nq);
    const float4 x = ld4(a + r * lda, q), y = ld4(b + r * ldb, q);
    float4 o;
    if (mode == 0) o = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
    else if (mode == 1) o = make_float4(x.x * y.x + x.x + y.x, x.y * y.y + x.y + y.y, x.z * y.z + x.z + y.z,
                                        x.w * y.w + x.w + y.w);
    else o = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    st4(out + r * ldo, q, o);
  }
}

// gradient of k_ew_combine modes 0/1 wrt both inputs, each followed by the
// leaky-relu derivative of the FC layer that produced that input (a, b are
// post-activations): da = g*(b + res) * lrelu'(a), db = g*(a + res) * lrelu'(b).
__global__ void __launch_bounds__(kThreads)
k_ew_fusion_bwd(int residual, const float *__restrict__ g, int64_t ldg, const float *__restrict__ a,
                int64_t lda, const float *__restrict__ b, int64_t ldb, int M, int N, float alpha,
                float *__restrict__ da, int64_t ldda, float *__restrict__ db, int64_t lddb) {
  const int nq = N >> 2;
  const int64_t total = (int64_t)M * nq;
  const float res = residual ? 1.f : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nq;
    const int q = (int)(i - r * nq);
    const float4 gg = ld4(g + r * ldg, q), x = ld4(a + r * lda, q), y = ld4(b + r * ldb, q);
#define CDML_FB(c, OUTA, OUTB)                                   \
  OUTA.c = gg.c * (y.c + res) * ((x.c > 0.f) ? 1.f : alpha);     \
  OUTB.c = gg.c * (x.c + res) * ((y.c > 0.f) ? 1.f : alpha);
    float4 oa, ob;
    CDML_FB(x, oa, ob) CDML_FB(y, oa, ob) CDML_FB(z, oa, ob) CDML_FB(w, oa, ob)
#undef CDML_FB
    st4(da + r * ldda, q, oa);
    st4(db + r * lddb, q, ob);
  }
}

// dpre = g * lrelu'(y)  (y = post-activation)
__global__ void __launch_bounds__(kThreads)
k_lrelu_bwd(const float *__restrict__ g, int64_t ldg, const float *__restrict__ y, int64_t ldy, int M,
            int N, float alpha, float *__restrict__ out, int64_t ldo) {
  const int nq = N >> 2;
  const int64_t total = (int64_t)M * nq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / nq;
    const int q = (int)(i - r * nq);
    const float4 gg = ld4(g + r * ldg, q), v = ld4(y + r * ldy, q);
    st4(out + r * ldo, q, make_float4(gg.x * ((v.x > 0.f) ? 1.f : alpha), gg.y * ((v.y > 0.f) ? 1.f : alpha),
                                      gg.z * ((v.z > 0.f) ? 1.f : alpha), gg.w * ((v.w > 0.f) ? 1.f : alpha)));
  }
}

// stats[0..3] = mean hinge, mean pos, mean neg, fraction of triplets with hinge > 0.
__global__ void __launch_bounds__(1024)
k_loss_stats(const float *__restrict__ pos, const float *__restrict__ neg,
             const float *__restrict__ hinge, int B, float *__restrict__ stats) {
  __shared__ float s[4][1024 / kWave];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < B; i += 1024) {
    const float h = hinge[i];
    acc[0] += h;
    acc[1] += pos[i];
    acc[2] += neg[i];
    acc[3] += (h > 0.f) ? 1.f : 0.f;
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float v = wave_sum(acc[c]);
    if (lane == 0) s[c][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.f;
    for (int w = 0; w < 1024 / kWave; ++w) v += s[threadIdx.x][w];
    stats[threadIdx.x] = v / (float)B;
  }
}

int grid_rows(int rows) {
  int64_t b = ((int64_t)rows + kWavesPerBlock - 1) / kWavesPerBlock;
  if (b > kNumCU * 8) b = kNumCU * 8;
  return (int)(b < 1 ? 1 : b);
}

int check_rows(const char *who, const void *p, int64_t ld, int N) {
  CDML_REQUIRE(p, CDML_E_BADARG, "%s: null pointer", who);
  CDML_REQUIRE((N & 3) == 0 && ld >= N && (ld & 3) == 0 && aligned16(p), CDML_E_ALIGN,
               "%s: width and leading dimension must be multiples of 4, base 16-B aligned", who);
  return CDML_OK;
}

}  // namespace
}  // namespace cdml

using namespace cdml;

extern "C" int cdml_l2norm_fwd(const float *x, int64_t ldx, int M, int N, float *y, int64_t ldy,
                               float *inv_out, cdml_stream_t stream) {
  CDML_REQUIRE(M > 0 && N > 0, CDML_E_BADARG, "l2norm_fwd: bad shape");
  int rc;
  if ((rc = check_rows("l2norm_fwd", x, ldx, N))) return rc;
  if ((rc = check_rows("l2norm_fwd", y, ldy, N))) return rc;
  hipLaunchKernelGGL(k_l2norm_fwd, dim3(grid_rows(M)), dim3(kThreads), 0, (hipStream_t)stream, x, ldx,
                     M, N, y, ldy, inv_out);
  return check_launch("l2norm_fwd");
}

extern "C" int cdml_l2norm_bwd(const float *z, int64_t ldz, const float *g, int64_t ldg, int M, int N,
                               float lrelu_alpha, float *dz, int64_t lddz, cdml_stream_t stream) {
  CDML_REQUIRE(M > 0 && N > 0, CDML_E_BADARG, "l2norm_bwd: bad shape");
  int rc;
  if ((rc = check_rows("l2norm_bwd", z, ldz, N))) return rc;
  if ((rc = check_rows("l2norm_bwd", g, ldg, N))) return rc;
  if ((rc = check_rows("l2norm_bwd", dz, lddz, N))) return rc;
  hipLaunchKernelGGL(k_l2norm_bwd, dim3(grid_rows(M)), dim3(kThreads), 0, (hipStream_t)stream, z, ldz,
                     g, ldg, M, N, lrelu_alpha, dz, lddz);
  return check_launch("l2norm_bwd");
}

extern "C" int cdml_triplet_hinge(const float *e, int64_t lde, int B, int D, float margin, float *pos,
                                  float *neg, float *hinge, float *stats, float *de, int64_t ldde,
                                  cdml_stream_t stream) {
  CDML_REQUIRE(B > 0 && D > 0 && pos && neg && hinge, CDML_E_BADARG, "triplet_hinge: bad argument");
  int rc;
  if ((rc = check_rows("triplet_hinge", e, lde, D))) return rc;
  if (de && (rc = check_rows("triplet_hinge", de, ldde, D))) return rc;
  hipLaunchKernelGGL(k_triplet_hinge, dim3(grid_rows(B)), dim3(kThreads), 0, (hipStream_t)stream, e,
                     lde, B, D, margin, pos, neg, hinge, de, ldde);
  if ((rc = check_launch("triplet_hinge"))) return rc;
  if (stats) {
    hipLaunchKernelGGL(k_loss_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, pos, neg, hinge, B,
                       stats);
    rc = check_launch("triplet_hinge stats");
  }
  return rc;
}

extern "C" int cdml_triplet_hinge_inbatch(const float *e, int64_t lde, const int32_t *rows,
                                          const int32_t *shift, int B, int D, float margin,
                                          float *pos, float *neg, float *hinge, uint8_t *valid_out,
                                          float *stats, float *de, int64_t ldde,
                                          cdml_stream_t stream) {
  CDML_REQUIRE(B >= 2 && D > 0 && rows && shift && pos && neg && hinge, CDML_E_BADARG,
               "triplet_hinge_inbatch: bad argument");
  int rc;
  if ((rc = check_rows("triplet_hinge_inbatch", e, lde, D))) return rc;
  if (de && (rc = check_rows("triplet_hinge_inbatch", de, ldde, D))) return rc;
  hipLaunchKernelGGL(k_triplet_hinge_inbatch, dim3(grid_rows(B)), dim3(kThreads), 0,
                     (hipStream_t)stream, e, lde, rows, shift, B, D, margin, pos, neg, hinge,
                     valid_out, de, ldde);
  if ((rc = check_launch("triplet_hinge_inbatch"))) return rc;
  if (stats) {
    hipLaunchKernelGGL(k_loss_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, pos, neg, hinge, B,
                       stats);
    rc = check_launch("triplet_hinge_inbatch stats");
  }
  return rc;
}

constexpr int kTailVarBlocks = 256;   // grid cap when the variance is wanted (last block sums the partials)

extern "C" size_t cdml_vnet_tail_workspace(int B, int D) {
  if (B <= 0 || D <= 0) return 0;
  return (size_t)kTailVarBlocks * (size_t)(D + 4) * sizeof(float);
}

static int vnet_tail_impl(int mode, const float *z, int64_t ldz, const int32_t *rows,
                          const int32_t *shift, int B, int D, float margin, float lrelu_alpha,
                          float *e, int64_t lde, float *pos, float *neg, float *hinge,
                          uint8_t *valid_out, float *dz2, int64_t lddz2, uint16_t *dz2_bf16,
                          int64_t ldbf, int64_t plane_bf, float *stats, float *var_ws, cdml_stream_t stream,
                          float h2_scale = 0.f) {
  CDML_REQUIRE(mode == 0 || mode == 1, CDML_E_BADARG, "vnet_tail: mode must be 0 (a,p,n rows) or 1 (in-batch)");
  CDML_REQUIRE(B >= (mode == 1 ? 2 : 1) && D > 0 && pos && neg && hinge, CDML_E_BADARG, "vnet_tail: bad argument");
  CDML_REQUIRE(mode == 0 || (rows && shift), CDML_E_BADARG, "vnet_tail: in-batch mode needs rows and shift");
  CDML_REQUIRE(!var_ws || stats, CDML_E_BADARG, "vnet_tail: the variance is written to stats[4]");
  CDML_REQUIRE(D <= 1024, CDML_E_UNSUPPORTED, "vnet_tail: embedding size %d > 1024", D);
  int rc;
  if ((rc = check_rows("vnet_tail z", z, ldz, D))) return rc;
  if ((rc = check_rows("vnet_tail e", e, lde, D))) return rc;
  if ((rc = check_rows("vnet_tail dz2", dz2, lddz2, D))) return rc;
  CDML_REQUIRE(!dz2_bf16 || (ldbf >= D && (ldbf & 3) == 0 && (reinterpret_cast<uintptr_t>(dz2_bf16) & 7) == 0),
               CDML_E_ALIGN, "vnet_tail: bf16 copy needs an 8-B aligned base and a leading dimension multiple of 4");
  CDML_REQUIRE(plane_bf == 0 || (dz2_bf16 && plane_bf >= D && (plane_bf & 3) == 0 && ldbf >= (h2_scale > 0.f ? 1 : 2) * plane_bf + D),
               CDML_E_ALIGN, "vnet_tail: planes need plane >= D (a multiple of 4) and a leading dimension >= 2 plane + D (fp16 planes: plane + D)");
  int grid = grid_rows(B);
  if (var_ws && grid > kTailVarBlocks) grid = kTailVarBlocks;
  const size_t lds = var_ws ? (size_t)kWavesPerBlock * D * sizeof(float) : 0;
  const int nch = ((D >> 2) + kWave - 1) / kWave;
#define CDML_LAUNCH_TAIL(M, N)                                                                       \
  hipLaunchKernelGGL((k_vnet_tail<M, N>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, z, ldz, \
                     rows, shift, B, D, margin, lrelu_alpha, e, lde, pos, neg, hinge, valid_out, dz2,  \
                     lddz2, dz2_bf16, ldbf, var_ws, plane_bf, h2_scale)
  if (mode == 0) {
    if (nch <= 1) CDML_LAUNCH_TAIL(0, 1); else if (nch <= 2) CDML_LAUNCH_TAIL(0, 2); else CDML_LAUNCH_TAIL(0, 4);
  } else {
    if (nch <= 1) CDML_LAUNCH_TAIL(1, 1); else if (nch <= 2) CDML_LAUNCH_TAIL(1, 2); else CDML_LAUNCH_TAIL(1, 4);
  }
#undef CDML_LAUNCH_TAIL
  if ((rc = check_launch("vnet_tail"))) return rc;
  if (stats) {
    hipLaunchKernelGGL(k_loss_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, pos, neg, hinge, B, stats);
    if ((rc = check_launch("vnet_tail stats"))) return rc;
  }
  if (var_ws) {
    hipLaunchKernelGGL(k_tail_var_final, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, var_ws, grid, B, D, stats);
    rc = check_launch("vnet_tail variance");
  }
  return rc;
}

extern "C" int cdml_vnet_tail(int mode, const float *z, int64_t ldz, const int32_t *rows,
                              const int32_t *shift, int B, int D, float margin, float lrelu_alpha,
                              float *e, int64_t lde, float *pos, float *neg, float *hinge,
                              uint8_t *valid_out, float *dz2, int64_t lddz2, uint16_t *dz2_bf16,
                              int64_t ldbf, float *stats, float *var_ws, cdml_stream_t stream) {
  return vnet_tail_impl(mode, z, ldz, rows, shift, B, D, margin, lrelu_alpha, e, lde, pos, neg, hinge, valid_out, dz2, lddz2,
                        dz2_bf16, ldbf, 0, stats, var_ws, stream);
}

// cdml_vnet_tail writing dz2 also as its three bf16 planes hi | mid | lo (precision "f32x3": the split-fp32 GEMMs of
// the backward pass read them): dz2_planes = bf16 [rows][ldbf], plane p at columns p * plane_bf.
extern "C" int cdml_vnet_tail_planes(int mode, const float *z, int64_t ldz, const int32_t *rows,
                                     const int32_t *shift, int B, int D, float margin, float lrelu_alpha,
                                     float *e, int64_t lde, float *pos, float *neg, float *hinge,
                                     uint8_t *valid_out, float *dz2, int64_t lddz2, uint16_t *dz2_planes,
                                     int64_t ldbf, int64_t plane_bf, float *stats, float *var_ws,
                                     cdml_stream_t stream) {
  CDML_REQUIRE(dz2_planes && plane_bf > 0, CDML_E_BADARG, "vnet_tail_planes: the plane buffer and its plane stride are required");
  return vnet_tail_impl(mode, z, ldz, rows, shift, B, D, margin, lrelu_alpha, e, lde, pos, neg, hinge, valid_out, dz2, lddz2,
                        dz2_planes, ldbf, plane_bf, stats, var_ws, stream);
}

// cdml_vnet_tail writing dz2 also as the two fp16 planes hi | lo of dz2 * scale (precision "f16x2"; values beyond fp16's
// range saturate): dz2_planes = fp16 [rows][ldbf], plane p at columns p * plane_h (plane_h >= D, ldbf >= plane_h + D).
extern "C" int cdml_vnet_tail_h2(int mode, const float *z, int64_t ldz, const int32_t *rows,
                                 const int32_t *shift, int B, int D, float margin, float lrelu_alpha,
                                 float *e, int64_t lde, float *pos, float *neg, float *hinge,
                                 uint8_t *valid_out, float *dz2, int64_t lddz2, uint16_t *dz2_planes,
                                 int64_t ldbf, int64_t plane_h, float scale, float *stats, float *var_ws,
                                 cdml_stream_t stream) {
  CDML_REQUIRE(dz2_planes && plane_h > 0 && scale > 0.f, CDML_E_BADARG, "vnet_tail_h2: the plane buffer, its plane stride and a positive scale are required");
  return vnet_tail_impl(mode, z, ldz, rows, shift, B, D, margin, lrelu_alpha, e, lde, pos, neg, hinge, valid_out, dz2, lddz2,
                        dz2_planes, ldbf, plane_h, stats, var_ws, stream, scale);
}

extern "C" int cdml_semihard_select(const float *S, int64_t ldS, const float *e, int64_t lde,
                                    const int32_t *rows, int B, int D, float *sqn_scratch,
                                    int32_t *neg_row_out, cdml_stream_t stream) {
  CDML_REQUIRE(B >= 1 && D > 0 && rows && sqn_scratch && neg_row_out, CDML_E_BADARG,
               "semihard_select: bad argument");
  int rc;
  if ((rc = check_rows("semihard_select e", e, lde, D))) return rc;
  if ((rc = check_rows("semihard_select S", S, ldS, 2 * B))) return rc;
  CDML_REQUIRE(aligned16(rows) && aligned16(sqn_scratch), CDML_E_ALIGN,
               "semihard_select: rows and sqn_scratch must be 16-B aligned");
  hipLaunchKernelGGL(k_row_sumsq, dim3(grid_rows(2 * B)), dim3(kThreads), 0, (hipStream_t)stream, e,
                     lde, 2 * B, D, sqn_scratch);
  hipLaunchKernelGGL(k_semihard_select, dim3(grid_rows(B)), dim3(kThreads), 0, (hipStream_t)stream, S,
                     ldS, sqn_scratch, rows, B, neg_row_out);
  return check_launch("semihard_select");
}

extern "C" int cdml_triplet_hinge_indexed_tail(const float *e, int64_t lde, const int32_t *neg_row, int B, int D,
                                               float margin, float *pos, float *neg, float *hinge, float *stats,
                                               float *scale_scratch, float *de, int64_t ldde, const float *z,
                                               int64_t ldz, float lrelu_alpha, float *dz2, int64_t lddz,
                                               uint16_t *dz2_bf16, int64_t ldbf, int64_t plane_bf,
                                               cdml_stream_t stream) {
  CDML_REQUIRE(B >= 1 && D > 0 && neg_row && pos && neg && hinge && scale_scratch, CDML_E_BADARG,
               "triplet_hinge_indexed: bad argument");
  int rc;
  if ((rc = check_rows("triplet_hinge_indexed", e, lde, D))) return rc;
  if (de && (rc = check_rows("triplet_hinge_indexed", de, ldde, D))) return rc;
  CDML_REQUIRE(!de || (aligned16(neg_row) && aligned16(scale_scratch)), CDML_E_ALIGN,
               "triplet_hinge_indexed: neg_row and scale_scratch must be 16-B aligned");
  if (z) {
    CDML_REQUIRE(de && dz2, CDML_E_BADARG, "triplet_hinge_indexed_tail: the fused tail needs de and dz2");
    if ((rc = check_rows("triplet_hinge_indexed_tail z", z, ldz, D))) return rc;
    if ((rc = check_rows("triplet_hinge_indexed_tail dz2", dz2, lddz, D))) return rc;
    CDML_REQUIRE(!dz2_bf16 || ((reinterpret_cast<uintptr_t>(dz2_bf16) & 7) == 0 && (ldbf & 3) == 0 && (plane_bf & 3) == 0 &&
                               ldbf >= (plane_bf ? 2 * plane_bf + D : D) && plane_bf >= 0),
                 CDML_E_ALIGN, "triplet_hinge_indexed_tail: dz2_bf16 8-B aligned, ldbf and plane_bf multiples of 4, ldbf >= 2 plane_bf + D");
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_hinge_indexed_fwd, dim3(grid_rows(B)), dim3(kThreads), 0, s, e, lde, neg_row, B,
                     D, margin, pos, neg, hinge, scale_scratch);
  if (de) {
    const int nch = ((D >> 2) + kWave - 1) / kWave;   // float4 chunks per lane and row (D <= 256: 1)
    CDML_REQUIRE(nch <= 4, CDML_E_UNSUPPORTED, "triplet_hinge_indexed: embeddings of at most 1024 dimensions (got %d)", D);
    const dim3 grid((2 * B + kIdxRows - 1) / kIdxRows);
#define CDML_LAUNCH_HIB(N)                                                                                          \
    hipLaunchKernelGGL((k_hinge_indexed_bwd<N>), grid, dim3(kThreads), 0, s, e, lde, neg_row, scale_scratch, B, D, de, \
                       ldde, z, ldz, lrelu_alpha, dz2, lddz, dz2_bf16, ldbf, plane_bf)
    if (nch <= 1) CDML_LAUNCH_HIB(1);
    else if (nch == 2) CDML_LAUNCH_HIB(2);
    else CDML_LAUNCH_HIB(4);
#undef CDML_LAUNCH_HIB
  }
  if ((rc = check_launch("triplet_hinge_indexed"))) return rc;
  if (stats) {
    hipLaunchKernelGGL(k_loss_stats, dim3(1), dim3(1024), 0, s, pos, neg, hinge, B, stats);
    rc = check_launch("triplet_hinge_indexed stats");
  }
  return rc;
}

extern "C" int cdml_triplet_hinge_indexed(const float *e, int64_t lde, const int32_t *neg_row, int B,
                                          int D, float margin, float *pos, float *neg, float *hinge,
                                          float *stats, float *scale_scratch, float *de, int64_t ldde,
                                          cdml_stream_t stream) {
  return cdml_triplet_hinge_indexed_tail(e, lde, neg_row, B, D, margin, pos, neg, hinge, stats, scale_scratch, de, ldde,
                                         nullptr, 0, -1.0f, nullptr, 0, nullptr, 0, 0, stream);
}

extern "C" int cdml_pair_dist(const float *e, int64_t lde, int n_rows, const int32_t *pairs, int P,
                              int D, float *sqdist, float *dot, float *means, cdml_stream_t stream) {
  CDML_REQUIRE(P >= 1 && D > 0 && n_rows > 0 && pairs && sqdist && dot, CDML_E_BADARG,
               "pair_dist: bad argument");
  int rc;
  if ((rc = check_rows("pair_dist", e, lde, D))) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pair_dist, dim3(grid_rows(P)), dim3(kThreads), 0, s, e, lde, pairs, P, D, sqdist, dot);
  if ((rc = check_launch("pair_dist"))) return rc;
  if (means) {  // means[1] = mean squared distance, means[2] = mean dot product
    hipLaunchKernelGGL(k_loss_stats, dim3(1), dim3(1024), 0, s, sqdist, dot, sqdist, P, means);
    rc = check_launch("pair_dist means");
  }
  return rc;
}

static int ew_grid(int M, int N) {
  int64_t b = ((int64_t)M * (N >> 2) + kThreads - 1) / kThreads;
  if (b > kNumCU * 8) b = kNumCU * 8;
  return (int)(b < 1 ? 1 : b);
}

extern "C" int cdml_ew_combine(int mode, const float *a, int64_t lda, const float *b, int64_t ldb, int M,
                               int N, float *out, int64_t ldo, cdml_stream_t stream) {
  CDML_REQUIRE(mode >= 0 && mode <= 2 && M > 0 && N > 0, CDML_E_BADARG, "ew_combine: bad argument");
  int rc;
  if ((rc = check_rows("ew_combine a", a, lda, N))) return rc;
  if ((rc = check_rows("ew_combine b", b, ldb, N))) return rc;
  if ((rc = check_rows("ew_combine out", out, ldo, N))) return rc;
  hipLaunchKernelGGL(k_ew_combine, dim3(ew_grid(M, N)), dim3(kThreads), 0, (hipStream_t)stream, mode, a,
                     lda, b, ldb, M, N, out, ldo);
  return check_launch("ew_combine");
}

extern "C" int cdml_ew_fusion_bwd(int residual, const float *g, int64_t ldg, const float *a, int64_t lda,
                                  const float *b, int64_t ldb, int M, int N, float alpha, float *da,
                                  int64_t ldda, float *db, int64_t lddb, cdml_stream_t stream) {
  CDML_REQUIRE(M > 0 && N > 0, CDML_E_BADARG, "ew_fusion_bwd: bad argument");
  int rc;
  if ((rc = check_rows("ew_fusion_bwd g", g, ldg, N))) return rc;
  if ((rc = check_rows("ew_fusion_bwd a", a, lda, N))) return rc;
  if ((rc = check_rows("ew_fusion_bwd b", b, ldb, N))) return rc;
  if ((rc = check_rows("ew_fusion_bwd da", da, ldda, N))) return rc;
  if ((rc = check_rows("ew_fusion_bwd db", db, lddb, N))) return rc;
  hipLaunchKernelGGL(k_ew_fusion_bwd, dim3(ew_grid(M, N)), dim3(kThreads), 0, (hipStream_t)stream,
                     residual, g, ldg, a, lda, b, ldb, M, N, alpha, da, ldda, db, lddb);
  return check_launch("ew_fusion_bwd");
}

extern "C" int cdml_lrelu_bwd(const float *g, int64_t ldg, const float *y, int64_t ldy, int M, int N,
                              float alpha, float *out, int64_t ldo, cdml_stream_t stream) {
  CDML_REQUIRE(M > 0 && N > 0, CDML_E_BADARG, "lrelu_bwd: bad argument");
  int rc;
  if ((rc = check_rows("lrelu_bwd g", g, ldg, N))) return rc;
  if ((rc = check_rows("lrelu_bwd y", y, ldy, N))) return rc;
  if ((rc = check_rows("lrelu_bwd out", out, ldo, N))) return rc;
  hipLaunchKernelGGL(k_lrelu_bwd, dim3(ew_grid(M, N)), dim3(kThreads), 0, (hipStream_t)stream, g, ldg, y,
                     ldy, M, N, alpha, out, ldo);
  return check_launch("lrelu_bwd");
}
