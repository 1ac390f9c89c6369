// Stand-alone probe (round-3 review item 3): can the collectives of a sharded PDLP attempt be captured into a hipGraph?
// One rank (all a one-GPU box offers), the RCCL the product binds (dlopen("librccl.so.1"), the one PyTorch bundles when LD_LIBRARY_PATH
// points at torch/lib): ncclAllGather + ncclAllReduce on a non-blocking stream, first eagerly, then inside
// hipStreamBeginCapture(ThreadLocal | Global | Relaxed) -> instantiate -> three replays -> check the data.
//   hipcc -O1 tools/rccl_capture_repro.cpp -o /tmp/rccl_capture_repro -ldl && /tmp/rccl_capture_repro [mode 0|1|2]
// Prints one line per stage; a crash inside the capture shows as a missing "captured" line (run it under `timeout`).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct unique_id { char internal[128]; };
typedef void* comm_t;
#define HIPOK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define NCOK(e) do { int e_ = (e); if (e_ != 0) { std::printf("RCCL error %d at line %d\n", e_, __LINE__); return 3; } } while (0)
int main(int argc, char** argv)
{
  const int mode = argc > 1 ? std::atoi(argv[1]) : 2;  // 0 global, 1 thread local, 2 relaxed
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { std::printf("no librccl: %s\n", dlerror()); return 1; }
  auto GetUniqueId  = (int (*)(unique_id*))dlsym(lib, "ncclGetUniqueId");
  auto CommInitRank = (int (*)(comm_t*, int, unique_id, int))dlsym(lib, "ncclCommInitRank");
  auto AllGather    = (int (*)(const void*, void*, size_t, int, comm_t, hipStream_t))dlsym(lib, "ncclAllGather");
  auto AllReduce    = (int (*)(const void*, void*, size_t, int, int, comm_t, hipStream_t))dlsym(lib, "ncclAllReduce");
  auto GetVersion   = (int (*)(int*))dlsym(lib, "ncclGetVersion");
  auto CommDestroy  = (int (*)(comm_t))dlsym(lib, "ncclCommDestroy");
  int version = 0;
  if (GetVersion) GetVersion(&version);
  std::printf("rccl version code %d, capture mode %d\n", version, mode);
  HIPOK(hipSetDevice(0));
  hipStream_t s;
  HIPOK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unique_id id;
  NCOK(GetUniqueId(&id));
  comm_t comm;
  NCOK(CommInitRank(&comm, 1, id, 0));
  const size_t n = 1 << 20;
  double *buf, *sc;
  HIPOK(hipMalloc((void**)&buf, n * sizeof(double)));
  HIPOK(hipMalloc((void**)&sc, 8 * sizeof(double)));
  std::vector<double> h(n, 1.5);
  HIPOK(hipMemcpy(buf, h.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIPOK(hipMemcpy(sc, h.data(), 8 * sizeof(double), hipMemcpyHostToDevice));
  const int kDouble = 8 /* ncclDouble */, kSum = 0;
  NCOK(AllGather(buf, buf, n, kDouble, comm, s));  // in place, as the product calls it
  NCOK(AllReduce(sc, sc, 3, kDouble, kSum, comm, s));
  HIPOK(hipStreamSynchronize(s));
  std::printf("eager collectives ok\n");
  std::fflush(stdout);
  hipGraph_t graph;
  HIPOK(hipStreamBeginCapture(s, (hipStreamCaptureMode)mode));
  NCOK(AllGather(buf, buf, n, kDouble, comm, s));
  NCOK(AllReduce(sc, sc, 3, kDouble, kSum, comm, s));
  HIPOK(hipStreamEndCapture(s, &graph));
  std::printf("captured\n");
  std::fflush(stdout);
  hipGraphExec_t exec;
  HIPOK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  std::printf("instantiated\n");
  std::fflush(stdout);
  for (int i = 0; i < 3; ++i) HIPOK(hipGraphLaunch(exec, s));
  HIPOK(hipStreamSynchronize(s));
  HIPOK(hipMemcpy(h.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost));
  std::printf("replayed x3, data %s\n", h[0] == 1.5 && h[n - 1] == 1.5 ? "intact" : "CHANGED");
  if (CommDestroy) CommDestroy(comm);
  return 0;
}
