// Probe: do LDS reads of one wave overlap the MFMAs of another wave on the same SIMD?
// One block of 8 waves per CU; waves 0-3 issue MFMAs, waves 4-7 (same SIMDs) issue LDS reads.
// Variants: MFMA waves alone, LDS waves alone, both; with a barrier every PERIOD MFMAs or none.
// build: hipcc --offload-arch=gfx950 -O3 -o pingpong_probe pingpong_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bit0: MFMA waves active, bit1: LDS waves active, bit2: barriers between chunks (ping-pong)
template <int MODE, int BF16, int VALU = 0>
__global__ void __launch_bounds__(512) k_probe(const float *in, float *out, int iters) {
  __shared__ f32x4 lds[4096];   // 64 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, grp = wave >> 2;
  for (int i = t; i < 4096; i += 512) lds[i] = f32x4{in[i & 2047], 1.f, 2.f, 3.f};
  __syncthreads();
  f32x16 acc[2] = {(f32x16)(0.f), (f32x16)(0.f)};
  f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = in[lane], b = in[lane + 64];
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 ah, bh;
  for (int e = 0; e < 8; ++e) { ah[e] = (__bf16)in[lane + e]; bh[e] = (__bf16)in[lane + 8 + e]; }
  auto mfma_chunk = [&]() {     // 2048 cycles of MFMA (fp32) / 512 (bf16: 16 MFMAs)
    if (BF16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i & 1], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 1], 0, 0, 0);
    }
  };
  const int base = (lane * 7 + wave * 13) & 1023;
  auto lds_chunk = [&]() {      // 24 ds_read_b128 into 24 registers, then wait for all
    if (VALU) {                 // ... or folded into a sum: 4 VALU adds per read beside the partner's MFMAs
#pragma unroll
      for (int i = 0; i < 24; ++i) sum += lds[base + i * 64];
      asm volatile("" : "+v"(sum));
      return;
    }
    f32x4 r[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) r[i] = lds[base + i * 64];
#pragma unroll
    for (int i = 0; i < 24; ++i) asm volatile("" :: "v"(r[i]));
  };
  for (int it = 0; it < iters; ++it) {
    if (MODE & 4) {
      // ping-pong: group 0 MFMA while group 1 LDS, then swap
      if (grp == 0) { if (MODE & 1) mfma_chunk(); } else { if (MODE & 2) lds_chunk(); }
      asm volatile("s_barrier" ::: "memory");
      if (grp == 1) { if (MODE & 1) mfma_chunk(); } else { if (MODE & 2) lds_chunk(); }
      asm volatile("s_barrier" ::: "memory");
    } else {
      if (grp == 0) { if (MODE & 1) { mfma_chunk(); mfma_chunk(); } } else { if (MODE & 2) { lds_chunk(); lds_chunk(); } }
    }
  }
  float s = sum.x + sum.y + sum.z + sum.w;
  for (int j = 0; j < 16; ++j) s += acc[0][j] + acc[1][j];
  out[blockIdx.x * 512 + t] = s;
}

template <int MODE, int BF16, int VALU = 0>
void run(const char *tag, const float *in, float *out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<MODE, BF16, VALU>), dim3(256), dim3(512), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<MODE, BF16, VALU>), dim3(256), dim3(512), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // MFMA work per iteration: with barriers both groups do one chunk each (all 8 waves), without only group 0 does two
  const double chunks = 2.0 * 4 * 256 * iters;               // chunk executions chip-wide
  const double cyc_per_chunk = BF16 ? 512.0 : 2048.0;
  const double ideal_ms = (MODE & 1) ? (double)iters * 2 * cyc_per_chunk / 2.4e6 : 0.0;   // one SIMD's MFMA time at 2.4 GHz
  printf("%-44s %8.3f ms   ideal MFMA-only %.3f ms  -> %.3f\n", tag, ms, ideal_ms, ideal_ms > 0 ? ideal_ms / ms : 0.0);
  (void)chunks;
}

int main() {
  float *in, *out; hipMalloc(&in, 2048 * 4); hipMalloc(&out, 256 * 512 * 4);
  float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = 0.001f * (i % 97);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int it = 4000;
  printf("fp32 32x32x2 (chunk = 32 MFMAs = 2048 cycles)\n");
  run<1, 0>("MFMA waves only, no barriers", in, out, it);
  run<2, 0>("LDS waves only, no barriers", in, out, it);
  run<3, 0>("both, no barriers (free-running)", in, out, it);
  run<5, 0>("MFMA only, ping-pong barriers", in, out, it);
  run<6, 0>("LDS only, ping-pong barriers", in, out, it);
  run<7, 0>("both, ping-pong barriers", in, out, it);
  run<6, 0, 1>("LDS + 4 adds/read only, ping-pong barriers", in, out, it);
  run<7, 0, 1>("both, LDS + 4 adds/read, ping-pong barriers", in, out, it);
  printf("bf16 32x32x16 (chunk = 16 MFMAs = 512 cycles)\n");
  run<1, 1>("MFMA waves only, no barriers", in, out, it);
  run<3, 1>("both, no barriers (free-running)", in, out, it);
  run<5, 1>("MFMA only, ping-pong barriers", in, out, it);
  run<6, 1>("LDS only, ping-pong barriers", in, out, it);
  run<7, 1>("both, ping-pong barriers", in, out, it);
  return 0;
}
