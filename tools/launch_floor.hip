// What a kernel costs on this part before it does any work (round-3 review item 4: is BASELINE config 2 -- 4 kernels of 5-11 us per
// iteration -- above its floor?).  A hipGraph of 400 back-to-back kernels of each kind on one stream, timed with events around
// five replays and, per kernel, with hipExtLaunchKernel's start/stop events (the dispatch's own duration, what rocprofv3 reports):
//   empty      : one workgroup, returns at once
//   wide_empty : 391 workgroups x 256 threads, returns at once (the shape of k_primal at n = 1e5)
//   one_trip   : 391 x 256, load 8 bytes -> store 8 bytes (one global round trip)
//   two_trips  : 391 x 256, load a control word, then load what it selects -> store (two DEPENDENT round trips: k_primal)
//   three_trips: ... -> gather through the loaded index -> store (the chain of a cache-resident SpMV workgroup, without its LDS phase)
//   hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_empty(const int* ctl, const double* a, const int* idx, double* out) {}
__global__ void k_one(const int* ctl, const double* a, const int* idx, double* out) { const int i = blockIdx.x * 256 + threadIdx.x; out[i] = a[i] + 1.0; }
__global__ void k_two(const int* ctl, const double* a, const int* idx, double* out)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int sel = ctl[0];
  out[i] = a[i + sel * 131072] + 1.0;
}
__global__ void k_three(const int* ctl, const double* a, const int* idx, double* out)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int sel = ctl[0];
  const int j = idx[i + sel * 131072];
  out[i] = a[j] + 1.0;
}
int main()
{
  const int n = 1 << 18, wgs = 391, reps = 400;
  int *ctl, *idx; double *a, *out;
  OK(hipMalloc((void**)&ctl, 64)); OK(hipMalloc((void**)&idx, n * 4)); OK(hipMalloc((void**)&a, n * 8)); OK(hipMalloc((void**)&out, n * 8));
  OK(hipMemset(ctl, 0, 64)); OK(hipMemset(a, 0, n * 8));
  std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (int)((i * 7919u) % 100000u);
  OK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  hipStream_t s; OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  struct { const char* name; void (*k)(const int*, const double*, const int*, double*); int g; } kinds[] = {
    {"empty", k_empty, 1}, {"wide_empty", k_empty, wgs}, {"one_trip", k_one, wgs}, {"two_trips", k_two, wgs}, {"three_trips", k_three, wgs}};
  std::printf("%-12s %24s %24s\n", "kernel", "us per kernel in a graph", "us dispatch duration");
  for (auto& kd : kinds) {
    hipGraph_t g; hipGraphExec_t x;
    OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kd.k, dim3(kd.g), dim3(256), 0, s, ctl, a, idx, out);
    OK(hipStreamEndCapture(s, &g)); OK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    OK(hipGraphLaunch(x, s)); OK(hipStreamSynchronize(s));
    OK(hipEventRecord(e0, s)); for (int r = 0; r < 5; ++r) OK(hipGraphLaunch(x, s)); OK(hipEventRecord(e1, s)); OK(hipEventSynchronize(e1));
    float ms = 0; OK(hipEventElapsedTime(&ms, e0, e1));
    double dur = 0;
    for (int r = 0; r < 50; ++r) {
      hipEvent_t a0, a1; OK(hipEventCreate(&a0)); OK(hipEventCreate(&a1));
      hipExtLaunchKernelGGL(kd.k, dim3(kd.g), dim3(256), 0, s, a0, a1, 0, ctl, a, idx, out);
      OK(hipEventSynchronize(a1)); float t = 0; OK(hipEventElapsedTime(&t, a0, a1)); dur += t;
      (void)hipEventDestroy(a0); (void)hipEventDestroy(a1);
    }
    std::printf("%-12s %24.2f %24.2f\n", kd.name, 1e3 * ms / (5.0 * reps), 1e3 * dur / 50);
    (void)hipGraphExecDestroy(x); (void)hipGraphDestroy(g);
  }
  return 0;
}
