// Microbenchmark: how much of the fp32 MFMA rate survives when every operand comes fresh from LDS, as a
// function of the register blocking of the wave tile (R x C accumulators of 32 x 32: R + C fragment reads of
// 16 B per lane feed 4 R C MFMAs) and of the waves per SIMD.  No global loads, no barriers: what is lost here
// is lost to the LDS reads alone (issue slots, waits, and the clock the chip holds with the LDS busy).
// The shipped fp32 GEMMs block 2 x 2 at two waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_blocking mfma_lds_blocking.hip ; run: ./mfma_lds_blocking
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R, int C, int THREADS>
__global__ void __launch_bounds__(THREADS) k_blocked(const float *in, float *out, int iters) {
  __shared__ f32x4 lds[4096];                                  // 64 KiB
  for (int i = threadIdx.x; i < 4096; i += THREADS)
    lds[i] = f32x4{in[(4 * i) & 2047], in[(4 * i + 1) & 2047], in[(4 * i + 2) & 2047], in[(4 * i + 3) & 2047]};
  __syncthreads();
  f32x16 acc[R][C];
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) acc[r][c] = (f32x16)(0.f);
  const int lane = threadIdx.x & 63;
  int pos = (threadIdx.x * 5) & 4095;
  f32x4 a[R], b[C];
#pragma unroll
  for (int r = 0; r < R; ++r) a[r] = lds[(pos + 64 * r) & 4095];
#pragma unroll
  for (int c = 0; c < C; ++c) b[c] = lds[(pos + 64 * (R + c) + lane) & 4095];
  for (int it = 0; it < iters; ++it) {
    pos = (pos + 517) & 4095;
    f32x4 an[R], bn[C];                                        // the next k-group's fragments, read under these MFMAs
#pragma unroll
    for (int r = 0; r < R; ++r) an[r] = lds[(pos + 64 * r) & 4095];
#pragma unroll
    for (int c = 0; c < C; ++c) bn[c] = lds[(pos + 64 * (R + c) + lane) & 4095];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][e], b[c][e], acc[r][c], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = an[r];
#pragma unroll
    for (int c = 0; c < C; ++c) b[c] = bn[c];
  }
  float s = 0.f;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c)
      for (int j = 0; j < 16; ++j) s += acc[r][c][j];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int R, int C, int THREADS>
void run(const char *tag, const float *in, float *out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_blocked<R, C, THREADS>), dim3(256), dim3(THREADS), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_blocked<R, C, THREADS>), dim3(256), dim3(THREADS), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = 256.0 * (THREADS / 64) * iters * (4.0 * R * C) * (2.0 * 32 * 32 * 2);
  printf("%-44s %d x %d, %d waves/SIMD, %4.2f LDS B/lane/MFMA  %8.3f ms  %6.1f TFLOP/s (%.3f of 157.3)\n", tag, R, C,
         THREADS / 256, 16.0 * (R + C) / (4.0 * R * C), ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 2048 * 4); hipMalloc(&out, 1024 * 512 * 4 * 4);
  std::vector<float> h(2048);
  for (int mode = 1; mode < 3; ++mode) {
    for (auto &v : h) v = mode == 1 ? 0.02f * rand() / RAND_MAX : (2.f * rand() / RAND_MAX - 1.f);
    hipMemcpy(in, h.data(), 2048 * 4, hipMemcpyHostToDevice);
    printf("-- operands: %s\n", mode == 1 ? "U[0,0.02) (step-like)" : "U[-1,1)");
    for (int rep = 0; rep < 2; ++rep) {
      run<2, 2, 512>("shipped blocking", in, out, 20000);
      run<2, 2, 256>("shipped blocking, one wave per SIMD", in, out, 20000);
      run<4, 2, 512>("128 x 64 per wave", in, out, 10000);
      run<4, 2, 256>("128 x 64 per wave, one wave per SIMD", in, out, 10000);
      run<4, 4, 256>("128 x 128 per wave, one wave per SIMD", in, out, 5000);
    }
  }
  return 0;
}
