// What does feeding LDS cost the wave that issues it, next to a stream of bf16 MFMAs?
// One wave per SIMD, 16 MFMAs (v_mfma_f32_32x32x16_bf16, register operands) per iteration, plus per
// iteration (= per 512 MFMA cycles) P pieces of 1 KiB moved from an L2-resident buffer into LDS:
//   mode 0: nothing            mode 1: LDS-DMA (buffer_load_dwordx4 ... lds)
//   mode 2: global_load_dwordx4 into registers + ds_write_b128 one iteration later
//   mode 3 / 4: LDS-DMA the way a GEMM panel streams: rows of a [rows][1536] bf16 matrix (3072-B
//           row stride, 64 MB: beyond L2), each piece = 16 rows x 64 B (mode 3: the 32-deep K-tile
//           of tools/experiments/gemm_bf16_w4.hip) or 8 rows x 128 B (mode 4: whole lines, a
//           64-deep K-tile), advancing along the row every iteration
// placed one piece after every MFMA (first P gaps).  The time per iteration over the MFMA-only
// time is what the pieces cost the issuing wave.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_issue_probe dma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma(i32x4 srd, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_base), "v"(voff), "s"(srd) : "memory", "m0");
}

template <int MODE, int P>
__global__ void __launch_bounds__(256, 1) k_probe(const float *src, float *out, int iters) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const bf16x8 *>(src + ((threadIdx.x * 4 + 1024 * i) & 16383));
    b[i] = *reinterpret_cast<const bf16x8 *>(src + ((threadIdx.x * 4 + 1024 * i + 4096) & 16383));
  }
  const uint64_t base = (uint64_t)(uintptr_t)src;
  i32x4 srd;
  srd.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
  srd.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((base >> 32) & 0xffff));
  srd.z = (MODE >= 3) ? (64 << 20) : (1 << 20);
  srd.w = 0x00020000;
  const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + wave * 16384);
  f32x4 stage[P > 0 ? P : 1];
  for (int i = 0; i < (P > 0 ? P : 1); ++i) stage[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t off = (blockIdx.x * 4 + wave) * 4096 + lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[(j >> 2) & 3], acc[j & 3], 0, 0, 0);
      if (j < P) {
        const uint32_t o = (off + j * 1024) & ((1 << 20) - 1);
        if (MODE == 1) dma(srd, o, lbase + j * 1024);
        if (MODE == 3 || MODE == 4) {
          constexpr int RB = MODE == 3 ? 64 : 128;              // bytes of a row per piece
          const int rows_pp = 1024 / RB;                        // rows per piece
          const uint32_t row = ((blockIdx.x * 4 + wave) * P + j) * rows_pp + lane / (RB / 16);
          const uint32_t kb = ((uint32_t)it * RB) % 3072;
          dma(srd, (row % 21000) * 3072 + kb + (lane % (RB / 16)) * 16, lbase + j * 1024);
        }
        if (MODE == 2) {
          *reinterpret_cast<f32x4 *>(lds + wave * 16384 + j * 1024 + lane * 16) = stage[j];   // last iteration's piece
          stage[j] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const unsigned char *>(src) + o);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    off += 65536;
    if (MODE == 1 || MODE == 3 || MODE == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P > 0 ? P : 0) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float s = lds[threadIdx.x];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (MODE == 2) for (int i = 0; i < (P > 0 ? P : 1); ++i) s += stage[i].x;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int P>
void run(const char *tag, const float *src, float *out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<MODE, P>), dim3(256), dim3(256), 0, 0, src, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_probe<MODE, P>), dim3(256), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double tf = 256.0 * 4 * iters * 16 * (2.0 * 32 * 32 * 16) / ms / 1e9;
  printf("%-44s %2d pieces / 16 MFMAs  %.3f ms  %7.1f TFLOP/s (%.3f of 2500)  %.1f ns per iteration\n", tag, P, ms, tf,
         tf / 2500, ms * 1e6 / iters);
}

int main() {
  float *src, *out;
  hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20); hipMalloc(&out, 256 * 256 * 4);
  std::vector<float> h(1 << 18);
  for (auto &v : h) v = 0.01f * (rand() % 200 - 100);
  hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  run<0, 0>("MFMAs only", src, out);
  run<1, 2>("LDS-DMA (buffer_load ... lds)", src, out);
  run<1, 4>("LDS-DMA (buffer_load ... lds)", src, out);
  run<1, 8>("LDS-DMA (buffer_load ... lds)", src, out);
  run<2, 2>("global_load_dwordx4 + ds_write_b128", src, out);
  run<2, 4>("global_load_dwordx4 + ds_write_b128", src, out);
  run<2, 8>("global_load_dwordx4 + ds_write_b128", src, out);
  run<3, 4>("LDS-DMA, panel rows, 16 rows x 64 B per piece", src, out);
  run<4, 4>("LDS-DMA, panel rows, 8 rows x 128 B per piece", src, out);
  run<3, 2>("LDS-DMA, panel rows, 16 rows x 64 B per piece", src, out);
  run<4, 2>("LDS-DMA, panel rows, 8 rows x 128 B per piece", src, out);
  return 0;
}
