// Feasibility probe for a persistent multi-workgroup PDHG loop: cost of a grid-wide barrier on MI355X
// (G workgroups, agent-scope atomics + fences so that data written before the barrier is visible across XCDs).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/grid_barrier tools/grid_barrier.hip && tools/bin/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// sense-reversing counter barrier; returns false on timeout (someone is not resident)
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned G, unsigned& epoch, int* abort_flag)
{
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    epoch += 1;
    const unsigned target = epoch * G;
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1L << 24) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

// two-level variant: 16 workgroups share a group counter (their arrivals proceed in parallel with the other groups'),
// the last arriver of a group bumps the top counter that everybody polls.  counters: [0] top, [32*(1+grp)] groups
__device__ __forceinline__ bool grid_barrier2(unsigned* counters, unsigned G, unsigned& epoch, int* abort_flag)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    epoch += 1;
    const unsigned grp = blockIdx.x >> 4, ngroups = (G + 15) >> 4;
    const unsigned size = grp + 1 < ngroups ? 16u : G - 16u * (ngroups - 1);
    const unsigned prev = __hip_atomic_fetch_add(counters + 32 * (1 + grp), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == epoch * size) __hip_atomic_fetch_add(counters, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = epoch * ngroups;
    long spins = 0;
    while (__hip_atomic_load(counters, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1L << 24) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
}

// each round: every WG writes its slot, barrier, every WG sums all slots (checks cross-XCD visibility), barrier
template <int MODE>
__global__ void k_probe(unsigned* counter, int* abort_flag, double* slots, double* out, int rounds, int with_data)
{
  __shared__ double red[16];
  const unsigned G = gridDim.x;
  unsigned epoch = 0;
  double check = 0.0;
  for (int r = 0; r < rounds; ++r) {
    if (with_data && threadIdx.x == 0) slots[blockIdx.x] = (double)(r + 1) * (blockIdx.x + 1);
    if (!(MODE ? grid_barrier2(counter, G, epoch, abort_flag) : grid_barrier(counter, G, epoch, abort_flag))) return;
    if (with_data) {
      double s = 0.0;
      for (unsigned i = threadIdx.x; i < G; i += blockDim.x) s += slots[i];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) for (unsigned w = 0; w < blockDim.x / 64; ++w) check += red[w];
      if (!(MODE ? grid_barrier2(counter, G, epoch, abort_flag) : grid_barrier(counter, G, epoch, abort_flag))) return;
    }
  }
  if (threadIdx.x == 0) out[blockIdx.x] = check;
}

int main()
{
  unsigned* counter; int* abort_flag; double *slots, *out;
  CHECK(hipMalloc(&counter, 8192)); CHECK(hipMalloc(&abort_flag, 256));
  CHECK(hipMalloc(&slots, 4096 * 8)); CHECK(hipMalloc(&out, 4096 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int mode : {0, 1})
  for (int T : {512})
    for (int G : {8, 32, 64, 128, 200, 256})
      for (int with_data : {0, 1}) {
        const int rounds = 2000;
        CHECK(hipMemset(counter, 0, 8192)); CHECK(hipMemset(abort_flag, 0, 256));
        if (mode) k_probe<1><<<G, T>>>(counter, abort_flag, slots, out, 10, with_data); else k_probe<0><<<G, T>>>(counter, abort_flag, slots, out, 10, with_data);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(counter, 0, 8192));
        CHECK(hipEventRecord(e0));
        if (mode) k_probe<1><<<G, T>>>(counter, abort_flag, slots, out, rounds, with_data); else k_probe<0><<<G, T>>>(counter, abort_flag, slots, out, rounds, with_data);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int ab; CHECK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
        std::vector<double> h(G); CHECK(hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost));
        // expected check: sum_r (r+1) * G(G+1)/2
        const double expect = with_data ? (double)rounds * (rounds + 1) / 2 * ((double)G * (G + 1) / 2) : 0.0;
        bool good = true;
        for (int g = 0; g < G; ++g) good = good && h[g] == expect;
        printf("%s T=%4d G=%3d %s: %.2f us per barrier%s%s\n", mode ? "two-level" : "flat     ", T, G, with_data ? "data+2 barriers/round" : "barrier only       ",
               1e3 * ms / rounds / (with_data ? 2 : 1), ab ? "  ABORTED" : "", good ? "" : "  WRONG DATA");
      }
  return 0;
}
