// Cost of a grid-wide barrier on MI355X (one workgroup per CU, all co-resident): is a persistent whole-chip PDHG loop for
// mid-size LPs (1e5..1e6 nonzeros, the matrices resident in the chip's aggregate LDS) cheaper per phase than a kernel boundary
// inside a hipGraph (~1.7 us gap + ramp)?   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_payload.hip -o grid_barrier_payload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>  // 0: relaxed atomics only, 1: release/acquire fences at agent scope, 2: + a 64-byte payload exchanged per block
__global__ void __launch_bounds__(256) k_barrier(unsigned* counter, double* payload, int iters, double* out)
{
  const unsigned G = gridDim.x;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2 && threadIdx.x < 8) payload[(size_t)blockIdx.x * 8 + threadIdx.x] = acc + it + threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it + 1) * G;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (MODE == 2) {  // every block reads every other block's payload (G * 64 bytes): the reduction a PDHG decision needs
      double s = 0.0;
      for (unsigned b = threadIdx.x; b < G * 8; b += 256) s += __builtin_nontemporal_load(payload + b);
      acc += s;
    }
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main()
{
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  unsigned* counter; double *payload, *out;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&payload, 1024 * 64)); CK(hipMalloc(&out, 1024 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int G : {64, 256, 512}) {
    if (G > 2 * p.multiProcessorCount) continue;
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 4));
        CK(hipEventRecord(e0, 0));
        if (mode == 0) k_barrier<0><<<G, 256>>>(counter, payload, iters, out);
        if (mode == 1) k_barrier<1><<<G, 256>>>(counter, payload, iters, out);
        if (mode == 2) k_barrier<2><<<G, 256>>>(counter, payload, iters, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("grid %3d workgroups, mode %d (%s): %.2f us per barrier\n", G, mode,
             mode == 0 ? "relaxed atomics" : mode == 1 ? "release/acquire fences" : "fences + all-to-all 64 B payload", best * 1e3 / iters);
    }
  }
  // for comparison: an empty kernel per phase inside a hipGraph
  return 0;
}
