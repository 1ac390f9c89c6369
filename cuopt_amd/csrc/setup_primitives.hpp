// Device-side primitives shared by the set-up's translation units (kernels_setup.hip: sort, scan, transposition, the analysis pass;
// kernels_layout_build.hip: the layouts' constructions from the device-resident CSR).  gfx950, wave64.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

namespace {

constexpr int kT = 256;  // threads per workgroup of the set-up kernels

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane)
{
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan over the THREADS values of a workgroup; every thread gets its prefix, *total (optional, same for all) the sum.
// scratch: THREADS / 64 + 1 ints of LDS.  Ends with a barrier (scratch may be reused right after).
template <int THREADS>
__device__ __forceinline__ int block_exclusive_scan(int v, int* scratch, int* total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inc = wave_inclusive_scan(v, lane);
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  int base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) {
    const int s = scratch[w];
    if (w < wave) base += s;
    sum += s;
  }
  __syncthreads();
  if (total) *total = sum;
  return base + inc - v;
}

inline int grid_of(int64_t n, int cap = 4096) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + kT - 1) / kT, cap)); }

// ---- an open-addressing table of column indices in LDS (2 x capacity slots) -------------------------------------------------------
constexpr int kEstChunk = 64;
constexpr int32_t kEstEmpty = -1;

__device__ __forceinline__ bool est_insert(int32_t* tab, uint32_t mask, int32_t c)
{
  uint32_t h = ((uint32_t)c * 2654435761u) & mask;
  for (;;) {
    const int32_t seen = atomicCAS(&tab[h], kEstEmpty, c);
    if (seen == kEstEmpty) return true;
    if (seen == c) return false;
    h = (h + 1) & mask;
  }
}
__device__ __forceinline__ bool est_contains(const int32_t* tab, uint32_t mask, int32_t c)
{
  uint32_t h = ((uint32_t)c * 2654435761u) & mask;
  for (;;) {
    const int32_t seen = tab[h];
    if (seen == kEstEmpty) return false;
    if (seen == c) return true;
    h = (h + 1) & mask;
  }
}


// ascending bitonic sort of n (a power of two) words in LDS by the whole workgroup
template <int T>
__device__ __forceinline__ void lds_bitonic_sort(uint32_t* w, int n)
{
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += T) {
        const int x = i ^ j;
        if (x > i) {
          const uint32_t a = w[i], b = w[x];
          if ((a > b) == ((i & k) == 0)) w[i] = b, w[x] = a;
        }
      }
      __syncthreads();
    }
}
// exclusive max-scan over the workgroup's threads (values >= 0)
template <int T>
__device__ __forceinline__ int block_exclusive_max(int v, int* scratch)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc = o > inc ? o : inc;
  }
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < T / 64; ++w)
    if (w < wave) base = scratch[w] > base ? scratch[w] : base;
  __syncthreads();
  int ex = __shfl_up(inc, 1, 64);
  if (lane == 0) ex = 0;
  return ex > base ? ex : base;
}

}  // namespace

// exclusive scan of int32 on the device: out[i] = sum_{j < i} in[j] for i in [0, n] (n + 1 outputs); block_sums: >= (n + 1) / 4096 + 1 ints
int dev_exclusive_scan(hipStream_t s, const int32_t* in, int32_t* out, int64_t n, int32_t* block_sums);
