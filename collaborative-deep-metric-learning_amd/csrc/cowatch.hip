// Co-watch graph statistics of the reference's ETL on the device:
//   get_cowatch_graph (parse_data.py:221-254): multiplicity of every UNDIRECTED co-watch edge
//     over all (a, p) pairs; a pair with a == p is an error there (RuntimeError);
//   select_cowatch    (parse_data.py:256-289): keep the pairs whose edge was seen at least
//     `threshold` times -- all occurrences, in input order (unique = 0), or every qualifying
//     edge once (unique = 1; the reference then orients and orders them at random, here they
//     come out as (min, max) in ascending edge order).
// Integer work, bit-exact against oracle/etl.py and the reference's own outputs
// (tests/golden/cowatch_graph_seed7.npz).  HBM-bound: 64-bit edge keys (min << 32 | max),
// one radix sort, a run-length encode, a binary search per pair and a stable compaction; the
// sort / RLE / compaction are rocPRIM's device primitives, the rest are the kernels below.
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace cdml {
namespace {

constexpr int kThreads = 256;
using u64 = unsigned long long;

__global__ void __launch_bounds__(kThreads)
k_edge_keys(const int32_t *__restrict__ pairs, int64_t P, u64 *__restrict__ keys, int32_t *__restrict__ self_flag) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const uint32_t a = (uint32_t)pairs[2 * i], b = (uint32_t)pairs[2 * i + 1];
    if (a == b) atomicOr(self_flag, 1);
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    keys[i] = ((u64)lo << 32) | hi;
  }
}

// flag[i] = multiplicity of pair i's edge >= threshold (binary search in the distinct edges)
__global__ void __launch_bounds__(kThreads)
k_flag_pairs(const u64 *__restrict__ keys, int64_t P, const u64 *__restrict__ uniq,
             const unsigned *__restrict__ counts, const unsigned *__restrict__ n_uniq, unsigned threshold,
             unsigned char *__restrict__ flags) {
  const int64_t U = *n_uniq;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads) {
    const u64 k = keys[i];
    int64_t lo = 0, hi = U - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (uniq[mid] < k) lo = mid + 1; else hi = mid;
    }
    flags[i] = (U > 0 && uniq[lo] == k && counts[lo] >= threshold) ? 1 : 0;
  }
}

__global__ void __launch_bounds__(kThreads)
k_flag_edges(const unsigned *__restrict__ counts, const unsigned *__restrict__ n_uniq, int64_t P,
             unsigned threshold, unsigned char *__restrict__ flags) {
  const int64_t U = *n_uniq;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < P; i += (int64_t)gridDim.x * kThreads)
    flags[i] = (i < U && counts[i] >= threshold) ? 1 : 0;
}

// edge keys -> (min, max) int32 pairs, in place over the first *n entries; counts copied if asked
__global__ void __launch_bounds__(kThreads)
k_decode_edges(const u64 *__restrict__ keys, const u64 *__restrict__ n, int32_t *__restrict__ out_pairs) {
  const int64_t N = (int64_t)*n;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < N; i += (int64_t)gridDim.x * kThreads) {
    const u64 k = keys[i];
    out_pairs[2 * i] = (int32_t)(uint32_t)(k >> 32);
    out_pairs[2 * i + 1] = (int32_t)(uint32_t)(k & 0xffffffffu);
  }
}

__global__ void k_widen_count(const unsigned *__restrict__ in, u64 *__restrict__ out) { *out = *in; }

int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > kNumCU * 16) b = kNumCU * 16;
  return (int)(b < 1 ? 1 : b);
}

struct Layout {
  size_t keys, sorted, uniq, counts, flags, n_uniq, n_sel, temp, temp_bytes, total;
};

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

Layout layout_for(int64_t P) {
  Layout L{};
  size_t b1 = 0, b2 = 0, b3 = 0;
  u64 *k = nullptr;
  unsigned *c = nullptr;
  unsigned char *f = nullptr;
  rocprim::radix_sort_keys(nullptr, b1, k, k, (size_t)P, 0, 64, (hipStream_t)0);
  rocprim::run_length_encode(nullptr, b2, k, (size_t)P, k, c, c, (hipStream_t)0);
  rocprim::select(nullptr, b3, k, f, k, k, (size_t)P, (hipStream_t)0);
  L.temp_bytes = b1 > b2 ? (b1 > b3 ? b1 : b3) : (b2 > b3 ? b2 : b3);
  size_t off = 0;
  L.keys = off; off += align_up((size_t)P * 8);
  L.sorted = off; off += align_up((size_t)P * 8);
  L.uniq = off; off += align_up((size_t)P * 8);
  L.counts = off; off += align_up((size_t)P * 4);
  L.flags = off; off += align_up((size_t)P);
  L.n_uniq = off; off += 256;
  L.n_sel = off; off += 256;
  L.temp = off; off += align_up(L.temp_bytes);
  L.total = off;
  return L;
}

#define CDML_PRIM(call)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) return fail(CDML_E_HIP, "cowatch: %s", hipGetErrorString(e_)); \
  } while (0)

// shared front half: keys, sort, run-length encode
int build_graph(const int32_t *pairs, int64_t P, int32_t *self_flag, unsigned char *ws, const Layout &L,
                hipStream_t s) {
  u64 *keys = reinterpret_cast<u64 *>(ws + L.keys), *sorted = reinterpret_cast<u64 *>(ws + L.sorted);
  u64 *uniq = reinterpret_cast<u64 *>(ws + L.uniq);
  unsigned *counts = reinterpret_cast<unsigned *>(ws + L.counts);
  unsigned *n_uniq = reinterpret_cast<unsigned *>(ws + L.n_uniq);
  size_t tb = L.temp_bytes;
  hipLaunchKernelGGL(k_edge_keys, dim3(grid_for(P)), dim3(kThreads), 0, s, pairs, P, keys, self_flag);
  int rc = check_launch("cowatch keys");
  if (rc) return rc;
  CDML_PRIM(rocprim::radix_sort_keys(ws + L.temp, tb, keys, sorted, (size_t)P, 0, 64, s));
  tb = L.temp_bytes;
  CDML_PRIM(rocprim::run_length_encode(ws + L.temp, tb, sorted, (size_t)P, uniq, counts, n_uniq, s));
  return CDML_OK;
}

}  // namespace
}  // namespace cdml

using namespace cdml;

extern "C" size_t cdml_cowatch_workspace(int64_t n_pairs) {
  return n_pairs > 0 ? layout_for(n_pairs).total : 0;
}

extern "C" int cdml_cowatch_graph(const int32_t *pairs, int64_t n_pairs, int32_t *edges_out,
                                  int32_t *counts_out, int64_t *n_edges_out, int32_t *self_pair_flag,
                                  void *workspace, size_t workspace_bytes, cdml_stream_t stream) {
  CDML_REQUIRE(pairs && edges_out && counts_out && n_edges_out && self_pair_flag && n_pairs > 0, CDML_E_BADARG,
               "cowatch_graph: bad argument");
  const Layout L = layout_for(n_pairs);
  CDML_REQUIRE(workspace && workspace_bytes >= L.total && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
               CDML_E_BADARG, "cowatch_graph: workspace of %zu bytes (256-B aligned) required", L.total);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  hipStream_t s = (hipStream_t)stream;
  int rc = build_graph(pairs, n_pairs, self_pair_flag, ws, L, s);
  if (rc) return rc;
  u64 *n_sel = reinterpret_cast<u64 *>(ws + L.n_sel);
  hipLaunchKernelGGL(k_widen_count, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned *>(ws + L.n_uniq), n_sel);
  hipLaunchKernelGGL(k_decode_edges, dim3(grid_for(n_pairs)), dim3(kThreads), 0, s,
                     reinterpret_cast<const u64 *>(ws + L.uniq), n_sel, edges_out);
  CDML_PRIM(hipMemcpyAsync(counts_out, ws + L.counts, (size_t)n_pairs * 4, hipMemcpyDeviceToDevice, s));
  CDML_PRIM(hipMemcpyAsync(n_edges_out, n_sel, 8, hipMemcpyDeviceToDevice, s));
  return check_launch("cowatch_graph");
}

extern "C" int cdml_cowatch_select(const int32_t *pairs, int64_t n_pairs, int threshold, int unique,
                                   int32_t *out_pairs, int64_t *out_count, int32_t *self_pair_flag,
                                   void *workspace, size_t workspace_bytes, cdml_stream_t stream) {
  CDML_REQUIRE(pairs && out_pairs && out_count && self_pair_flag && n_pairs > 0, CDML_E_BADARG,
               "cowatch_select: bad argument");
  CDML_REQUIRE((reinterpret_cast<uintptr_t>(pairs) & 7) == 0 && (reinterpret_cast<uintptr_t>(out_pairs) & 7) == 0,
               CDML_E_ALIGN, "cowatch_select: pair buffers must be 8-B aligned");
  const Layout L = layout_for(n_pairs);
  CDML_REQUIRE(workspace && workspace_bytes >= L.total && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0,
               CDML_E_BADARG, "cowatch_select: workspace of %zu bytes (256-B aligned) required", L.total);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  hipStream_t s = (hipStream_t)stream;
  int rc = build_graph(pairs, n_pairs, self_pair_flag, ws, L, s);
  if (rc) return rc;
  const unsigned thr = threshold < 1 ? 1u : (unsigned)threshold;
  unsigned char *flags = ws + L.flags;
  const unsigned *n_uniq = reinterpret_cast<const unsigned *>(ws + L.n_uniq);
  const unsigned *counts = reinterpret_cast<const unsigned *>(ws + L.counts);
  u64 *n_sel = reinterpret_cast<u64 *>(ws + L.n_sel);
  size_t tb = L.temp_bytes;
  if (unique) {
    hipLaunchKernelGGL(k_flag_edges, dim3(grid_for(n_pairs)), dim3(kThreads), 0, s, counts, n_uniq, n_pairs, thr,
                       flags);
    u64 *sel = reinterpret_cast<u64 *>(ws + L.sorted);       // the sorted keys are no longer needed
    CDML_PRIM(rocprim::select(ws + L.temp, tb, reinterpret_cast<const u64 *>(ws + L.uniq), flags, sel, n_sel,
                              (size_t)n_pairs, s));
    hipLaunchKernelGGL(k_decode_edges, dim3(grid_for(n_pairs)), dim3(kThreads), 0, s, sel, n_sel, out_pairs);
  } else {
    hipLaunchKernelGGL(k_flag_pairs, dim3(grid_for(n_pairs)), dim3(kThreads), 0, s,
                       reinterpret_cast<const u64 *>(ws + L.keys), n_pairs,
                       reinterpret_cast<const u64 *>(ws + L.uniq), counts, n_uniq, thr, flags);
    // a pair is one 8-byte item: the stable compaction keeps (a, p) together and in input order
    CDML_PRIM(rocprim::select(ws + L.temp, tb, reinterpret_cast<const u64 *>(pairs), flags,
                              reinterpret_cast<u64 *>(out_pairs), n_sel, (size_t)n_pairs, s));
  }
  CDML_PRIM(hipMemcpyAsync(out_count, n_sel, 8, hipMemcpyDeviceToDevice, s));
  return check_launch("cowatch_select");
}
