// Calibration microbenchmark for the config-4 precision path: what v_mfma_f32_32x32x16_bf16
// sustains on this box (a) with operands in registers and (b) with operands re-read from LDS
// every iteration, for zero / small / N(0,1)-like data.  The dense bf16 peak is 2.5 PFLOP/s at
// 2.4 GHz; what these loops reach is the ceiling a bf16 GEMM's inner loop can approach before
// any global traffic, barrier or epilogue is paid for (the matrix pipe's power draw moves the
// shader clock, so the ceiling depends on the data).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak_bf16 mfma_peak_bf16.hip ; run: ./mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ bf16x8 load8(const u16 *p) { return *reinterpret_cast<const bf16x8 *>(p); }

// live operands in registers: 8 A and 8 B fragments per lane, a different pair for every MFMA
__global__ void __launch_bounds__(256) k_reg(const u16 *in, float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = load8(in + ((threadIdx.x * 8 + 536 * i) & 16383));
    b[i] = load8(in + ((threadIdx.x * 8 + 1048 * i + 7816) & 16383));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 5 * i) & 7], b[(k + 3 * i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// fragments re-read from a 64-KiB LDS image every iteration (8 ds_read_b128 per 16 MFMAs per wave:
// the read/MFMA ratio of a 128x64 wave tile); 2 waves per SIMD so one wave's reads hide under
// the other's MFMAs
__global__ void __launch_bounds__(512) k_lds(const u16 *in, float *out, int iters) {
  __shared__ __attribute__((aligned(16))) u16 lds[32768];
  for (int i = threadIdx.x; i < 4096; i += 512)
    *reinterpret_cast<bf16x8 *>(lds + 8 * i) = load8(in + ((8 * i) & 16383));
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.f);
  int pos = (threadIdx.x * 5) & 4095;
  for (int it = 0; it < iters; ++it) {
    bf16x8 a[4], b[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const bf16x8 *>(lds + 8 * ((pos + 64 * j) & 4095));
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(lds + 8 * ((pos + 64 * j + 331) & 4095));
    pos = (pos + 517) & 4095;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[2 * j + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[i], acc[2 * j + i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename F>
float time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

static u16 to_bf16(float x) { unsigned u; memcpy(&u, &x, 4); return (u16)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
  u16 *in; float *out;
  hipMalloc(&in, 16384 * 2 + 64); hipMalloc(&out, 1024 * 512 * 4);
  std::vector<u16> h(16384 + 32);
  const double mf = 2.0 * 32 * 32 * 16;
  for (int mode = 0; mode < 3; ++mode) {
    for (auto &v : h) {
      float x = 0.f;
      if (mode == 1) x = 0.02f * rand() / RAND_MAX;
      if (mode == 2) { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (float)rand() / RAND_MAX; x = sqrtf(-2 * logf(u1)) * cosf(6.2831853f * u2); }
      v = to_bf16(x);
    }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("-- operands: %s\n", mode == 0 ? "zeros" : mode == 1 ? "U[0,0.02) (step-like)" : "N(0,1)");
    for (int blocks : {256, 512}) {
      const int iters = 20000;
      float ms = time_ms([&] { hipLaunchKernelGGL(k_reg, dim3(blocks), dim3(256), 0, 0, in, out, iters); });
      double tf = (double)blocks * 4 * iters * 32 * mf / ms / 1e9;
      printf("registers only, %d wave(s)/SIMD      %.3f ms  %7.1f TFLOP/s (%.3f of 2500)\n", blocks / 256, ms, tf, tf / 2500);
    }
    {
      const int iters = 20000;
      float ms = time_ms([&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 0, 0, in, out, iters); });
      double tf = 256.0 * 8 * iters * 8 * mf / ms / 1e9;
      printf("fragments from LDS, 2 waves/SIMD    %.3f ms  %7.1f TFLOP/s (%.3f of 2500)\n", ms, tf, tf / 2500);
    }
  }
  return 0;
}
