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

// Operands FRESH every iteration: 32 a/b values per lane re-read from a 64-KiB LDS image (filled
// from `in`), two waves per SIMD so the reads hide under the partner's MFMAs.  With a random
// image every MFMA multiplies values it has not seen; with a zero image the schedule is the same
// and only the data differ -- the difference between the two is the data-dependent power.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k_mfma_fresh(const float *in, float *out, int iters) {
  __shared__ f32x4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512)
    lds[i] = f32x4{in[(4 * i) & 2047], in[(4 * i + 1) & 2047], in[(4 * i + 2) & 2047], in[(4 * i + 3) & 2047]};
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
  const int lane = threadIdx.x & 63;
  int pos = (threadIdx.x * 5) & 4095;
  for (int it = 0; it < iters; ++it) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = lds[(pos + 64 * j) & 4095];
      b[j] = lds[(pos + 64 * j + 256 + lane) & 4095];
    }
    pos = (pos + 517) & 4095;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[2 * (j & 1) + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][e], b[(j + i) & 3][e], acc[2 * (j & 1) + i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

void run_fresh(const char *tag, const float *in, float *out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma_fresh, dim3(256), dim3(512), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_mfma_fresh, dim3(256), dim3(512), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flop = 256.0 * 8 * iters * 32 * (2.0 * 32 * 32 * 2);
  printf("%-28s blocks   256 fresh  %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", tag, ms, flop / ms / 1e9,
         flop / ms / 1e9 / 157.3);
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
  hipMalloc(&in, 2048 * 4); hipMalloc(&out, 1024 * 512 * 4 * 4);
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
    run_fresh("2 waves/SIMD, fresh from LDS", in, out, 10000);
  }
  return 0;
}
