// Microbenchmark: does the LAYOUT of a streamed bf16 activation matter to HBM?  One block per 256-row tile walks
// the contraction in 64-element K-tiles, the way the 256 x 256 GEMM streams its k-contiguous A operand:
//   row-major : a K-tile = 256 pieces of 128 B, one per row, 10 KB apart (what FC2 / dW2 read of h1 today)
//   tiled     : a K-tile = one contiguous 32 KB block ([row tile][K-tile][256 rows][64])
// Loads only (16 B per lane, 2 K-tiles = 8 loads per thread in flight), nothing else in the way.
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_tile_pattern hbm_tile_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool TILED>
__global__ void __launch_bounds__(512) k_stream(const i32x4 *a, int rows, int kbytes, int splits, int *out) {
  const int tile = blockIdx.x / splits, split = blockIdx.x % splits;
  const int ktiles = kbytes / 128 / splits, kt0 = split * ktiles;
  const int t = threadIdx.x;
  i32x4 acc = {0, 0, 0, 0};
  for (int kt = kt0; kt < kt0 + ktiles; kt += 2) {
    i32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kt + (j >> 2), p = (j & 3) * 512 + t;       // piece p of the K-tile: row p / 8, 16-B chunk p % 8
      const int r = p >> 3, c = p & 7;
      const int64_t byte = TILED ? (((int64_t)tile * (kbytes / 128) + k) * 256 + r) * 128 + c * 16
                                 : ((int64_t)tile * 256 + r) * kbytes + (int64_t)k * 128 + c * 16;
      v[j] = __builtin_nontemporal_load(a + byte / 16);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}

template <bool TILED>
void run(const char *tag, const i32x4 *a, int rows, int kbytes, int splits, int *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = rows / 256 * splits;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_stream<TILED>, dim3(blocks), dim3(512), 0, 0, a, rows, kbytes, splits, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_stream<TILED>, dim3(blocks), dim3(512), 0, 0, a, rows, kbytes, splits, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = (double)rows * kbytes;
  printf("%-10s %5d blocks (%d K-splits): %.1f MB in %7.1f us = %.2f TB/s\n", tag, blocks, splits, bytes / 1e6, ms * 1e3, bytes / ms / 1e9);
}

int main() {
  // two sizes: 252 MB (config 4's h1: 24576 x 5120 bf16, about the Infinity Cache) and 1 GB (HBM for certain)
  for (int rows : {24576, 98304}) {
    const int kbytes = 10240;
    void *a; int *out;
    hipMalloc(&a, (size_t)rows * kbytes); hipMalloc(&out, 4);
    hipMemset(a, 1, (size_t)rows * kbytes);
    printf("-- %d rows x %d B\n", rows, kbytes);
    for (int rep = 0; rep < 2; ++rep)
      for (int splits : {2, 4}) {
        run<false>("row-major", (const i32x4 *)a, rows, kbytes, splits, out);
        run<true>("tiled", (const i32x4 *)a, rows, kbytes, splits, out);
      }
    hipFree(a); hipFree(out);
  }
  return 0;
}
