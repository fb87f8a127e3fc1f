// Calibration microbenchmark: what v_mfma_f32_32x32x2_f32 sustains on this box when
// nothing else is in the way (operands in registers, no LDS, no barriers).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(const float *in, float *out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x16)(0.f);
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[threadIdx.x + 256 * i];
    b[i] = in[threadIdx.x + 256 * (i + 4)];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + i) & 3], b[k], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Same loop with LIVE operands: 16 A and 16 B registers of random data, a different pair for
// every MFMA, so the operand buses and multipliers toggle as they do in a GEMM (the loop
// above re-reads 4+4 constant registers).  What the chip sustains here is the power-limited
// ceiling of an fp32 GEMM on real data, before any LDS / L2 / HBM traffic is paid for.
__global__ void __launch_bounds__(256) k_mfma_live(const float *in, float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) {
    a[i] = in[(threadIdx.x + 67 * i) & 2047];
    b[i] = in[(threadIdx.x + 131 * i + 977) & 2047];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 5 * i) & 15], b[(k + 3 * i) & 15], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run_live(const char *tag, const float *in, float *out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma_live, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_mfma_live, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flop = (double)blocks * 4 * iters * 64 * (2.0 * 32 * 32 * 2);
  printf("%-28s blocks %5d live   %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", tag, blocks, ms, flop / ms / 1e9,
         flop / ms / 1e9 / 157.3);
}

template <int NACC>
void run(const char *tag, const float *in, float *out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flop = (double)blocks * 4 * iters * 4 * NACC * (2.0 * 32 * 32 * 2);
  printf("%-28s blocks %5d acc %d  %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", tag, blocks, NACC, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 2048 * 4); hipMalloc(&out, 256 * 2048 * 4 * 4);
  std::vector<float> h(2048);
  for (int mode = 0; mode < 3; ++mode) {
    for (auto &v : h) v = mode == 0 ? 0.f : mode == 1 ? 0.02f * rand() / RAND_MAX : (2.f * rand() / RAND_MAX - 1.f);
    hipMemcpy(in, h.data(), 2048 * 4, hipMemcpyHostToDevice);
    const char *tag = mode == 0 ? "zeros" : mode == 1 ? "U[0,0.02) (step-like)" : "U[-1,1)";
    printf("-- operands: %s\n", tag);
    run<4>("1 wave/SIMD", in, out, 256, 20000);
    run<4>("2 waves/SIMD", in, out, 512, 20000);
    run<4>("4 waves/SIMD", in, out, 1024, 10000);
    run<2>("2 waves/SIMD, 2 acc", in, out, 512, 40000);
    run_live("1 wave/SIMD, live operands", in, out, 256, 5000);
    run_live("2 waves/SIMD, live operands", in, out, 512, 5000);
  }
  return 0;
}
