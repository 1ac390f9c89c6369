#pragma once
// HIP/gfx950 device layer of the MI355X-native PDLP solver: kernels + the `pdlpdev_*` C-ABI
// declared in include/cuopt_amd/pdlp_device.h (which lists the reference code each entry point
// replaces).  Hand-written for CDNA4: wave64, LDS-staged CSR stream SpMV with fused PDHG epilogues,
// device-resident step acceptance (no host round trip per PDHG step), hipGraph replay.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <tuple>
#include <utility>

#include <dlfcn.h>
#include <unistd.h>
#include <mutex>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "host_parallel.hpp"
#include "pdlp_kernels.hpp"
#include "pdlp_epilogues.hpp"
#include "pdlp_kernel_decls.hpp"

using namespace pdlp;

// ================================================================================================
// error plumbing
// ================================================================================================
inline thread_local std::string g_err;
inline int fail(int code, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ================================================================================================
// RCCL, bound lazily so that the library loads (and the single-GPU path runs) without it
// ================================================================================================
namespace rccl {
typedef struct ncclComm* comm_t;
typedef struct { char internal[128]; } unique_id;
enum { kFloat64 = 8 };           // ncclDouble
enum { kSum = 0, kMax = 2 };     // ncclSum / ncclMax
inline void* lib;
inline int (*GetUniqueId)(unique_id*);
inline int (*CommInitRank)(comm_t*, int, unique_id, int);
inline int (*CommDestroy)(comm_t);
inline int (*CommAbort)(comm_t);
inline int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
inline int (*ReduceScatter)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
inline int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t);
inline int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t);  // halo exchange of a sharded structured LP
inline int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t);
inline int (*GroupStart)();
inline int (*GroupEnd)();
inline const char* (*GetErrorString)(int);
inline std::mutex load_mutex;
inline int load()
{
  std::lock_guard<std::mutex> guard(load_mutex);
  if (lib) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(-3, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
  *(void**)&GetUniqueId    = dlsym(lib, "ncclGetUniqueId");
  *(void**)&CommInitRank   = dlsym(lib, "ncclCommInitRank");
  *(void**)&CommDestroy    = dlsym(lib, "ncclCommDestroy");
  *(void**)&CommAbort      = dlsym(lib, "ncclCommAbort");
  *(void**)&AllReduce      = dlsym(lib, "ncclAllReduce");
  *(void**)&ReduceScatter  = dlsym(lib, "ncclReduceScatter");
  *(void**)&AllGather      = dlsym(lib, "ncclAllGather");
  *(void**)&Send           = dlsym(lib, "ncclSend");
  *(void**)&Recv           = dlsym(lib, "ncclRecv");
  *(void**)&GroupStart     = dlsym(lib, "ncclGroupStart");
  *(void**)&GroupEnd       = dlsym(lib, "ncclGroupEnd");
  *(void**)&GetErrorString = dlsym(lib, "ncclGetErrorString");
  if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce)
    return fail(-3, "RCCL symbols missing");
  return 0;
}
}  // namespace rccl
#define RCCL_TRY(expr)                                                                     \
  do {                                                                                     \
    int r_ = (expr);                                                                       \
    if (r_ != 0)                                                                           \
      return fail(-4, "%s failed: %s", #expr,                                              \
                  rccl::GetErrorString ? rccl::GetErrorString(r_) : "rccl error");         \
  } while (0)


// RCCL communicators of this process: one per (unique id, rank), shared by the solvers created with the same pair while
// any of them is alive (reference count); the last solver to go destroys it -- a unique id bootstraps exactly one communicator
// per rank, so a caller that creates a second solver after the first one is gone draws a new id.
namespace comm_cache {
struct Entry {
  rccl::comm_t comm;
  int refs;
  bool aborted = false;
};
inline std::mutex mu;
inline std::map<std::string, Entry> map;
inline void release(const std::string& key)
{
  rccl::comm_t dead = nullptr;
  bool aborted      = false;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = map.find(key);
    if (it == map.end() || --it->second.refs > 0) return;
    dead = it->second.comm, aborted = it->second.aborted;
    map.erase(it);
  }
  if (dead && !aborted && rccl::CommDestroy) (void)rccl::CommDestroy(dead);  // (ncclCommAbort already freed an aborted one)
}
}  // namespace comm_cache

// ================================================================================================
// in-process "soft" communicator (verification of the sharded path at world > 1 on one GPU)
// ================================================================================================
#include <condition_variable>
#include <mutex>
namespace softcomm {
constexpr char kMagic[8] = {'C', 'U', 'O', 'P', 'T', 'S', 'F', 'T'};
struct Comm {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0, generation = 0;
  int refs = 0;                  // solvers attached (the last one to leave frees the communicator)
  bool aborted = false;          // a rank failed: every barrier returns false from now on, nobody waits for the missing rank
  std::vector<double*> bufs;     // this round's buffer of every rank
  std::vector<double*> scratch;  // per-rank result staging
  std::vector<size_t> scratch_size;
  std::vector<void*> p2p_base;   // direct-peer transport: every rank's landing block (same device, same process)
  bool barrier()
  {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const int gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != generation || aborted; });
    }
    return !aborted;
  }
  void abort()
  {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    cv.notify_all();
  }
};
#define SOFT_BARRIER(c)                                                                                   \
  do {                                                                                                    \
    if (!(c)->barrier()) return fail(-6, "in-process communicator: another rank failed, solve abandoned"); \
  } while (0)
struct Peers {
  const double* p[16];
};
static __global__ void __launch_bounds__(256) k_combine(Peers peers, int world, size_t count, int op, double* __restrict__ out)
{
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    double acc = peers.p[0][i];
    for (int r = 1; r < world; ++r) {  // fixed rank order -> every rank computes the same bits
      const double v = peers.p[r][i];
      acc            = op == 0 ? acc + v : (v > acc ? v : acc);
    }
    out[i] = acc;
  }
}
}  // namespace softcomm

// ---- roctx ranges (the reference marks every phase with NVTX: LP/pdhg.cu:75,168,241, LP/pdlp.cu:541,1227) --------------
// bound lazily like RCCL: without the library (or with CUOPT_AMD_ROCTX=0) the calls are no-ops
namespace roctx {
inline int (*Push)(const char*) = nullptr;
inline int (*Pop)()             = nullptr;
inline std::once_flag once;
inline void load()
{
  std::call_once(once, [] {
    if (cuopt_amd::tune_int("roctx", 1) == 0) return;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        Push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        Pop  = (int (*)())dlsym(h, "roctxRangePop");
        if (Push && Pop) return;
        Push = nullptr, Pop = nullptr;
      }
    }
  });
}
struct Range {
  explicit Range(const char* name)
  {
    load();
    if (Push) Push(name);
  }
  ~Range()
  {
    if (Pop) Pop();
  }
};
}  // namespace roctx

// ================================================================================================
// direct peer transport (owner-computes dataflow): push / pull kernels
// ================================================================================================
namespace p2pdev {
// wait until every rank's flag of exchange `kind` shows this rank's epoch, then landing -> dst (`count` doubles) EXCEPT this rank's
// own share [own0, own0 + own_count), which its producing kernel stored straight into dst.  Every workgroup polls for itself (local
// memory, one load per rank per poll); patience is bounded: a peer that never arrives sets the step error and the fault flag instead
// of hanging the device.  Four 16-byte requests per thread in flight (the landing block is fine-grained: every read is a trip to
// memory, and one request at a time ran at 0.75 TB/s).
static __global__ void __launch_bounds__(256) k_pull(pdlpdev_ctl* __restrict__ ctl, double* __restrict__ dst, const double* __restrict__ land, int count,
                                              int own0, int own_count, const unsigned long long* __restrict__ flags, int world, int kind,
                                              const unsigned long long* __restrict__ epoch, int* __restrict__ fault, const Push* __restrict__ push)
{
  if (!active(ctl)) return;
  raise(push);  // the producing kernel before this one in the stream is complete: its exchange is published here
  if (!wait_flags(flags, world, kind, epoch)) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *fault = 1, ctl->error = 1;
    return;
  }
  // (count, own0 and own_count are multiples of 16 entries and both buffers are 256-byte aligned: 16-byte requests)
  typedef double v2 __attribute__((ext_vector_type(2)));
  const v2* __restrict__ src2 = reinterpret_cast<const v2*>(land);
  v2* __restrict__ dst2       = reinterpret_cast<v2*>(dst);
  const int n2 = (count - own_count) >> 1, o2 = own0 >> 1, skip2 = own_count >> 1;  // pairs to copy; the own share is stepped over
  constexpr int U = 4;
  for (int i = blockIdx.x * 256 * U + threadIdx.x; i < n2; i += gridDim.x * 256 * U) {
    v2 v[U];
    int at[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = i + u * 256;
      at[u]       = k < n2 ? (k < o2 ? k : k + skip2) : -1;
      if (at[u] >= 0) v[u] = __builtin_nontemporal_load(src2 + at[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (at[u] >= 0) dst2[at[u]] = v[u];
  }
}
// the halo exchange's consumer: per peer ONE range [off, off + cnt) of the landed vector (buffer coordinates) instead of everything
struct PullRanges {
  int off[16], cnt[16];
};
static __global__ void __launch_bounds__(256) k_pull_ranges(pdlpdev_ctl* __restrict__ ctl, double* __restrict__ dst, const double* __restrict__ land, PullRanges R,
                                                     const unsigned long long* __restrict__ flags, int world, int kind, const unsigned long long* __restrict__ epoch,
                                                     int* __restrict__ fault, const Push* __restrict__ push)
{
  if (!active(ctl)) return;
  raise(push);
  if (!wait_flags(flags, world, kind, epoch)) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *fault = 1, ctl->error = 1;
    return;
  }
  for (int q = 0; q < world; ++q)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < R.cnt[q]; i += gridDim.x * 256) dst[R.off[q] + i] = __builtin_nontemporal_load(land + R.off[q] + i);
}
}  // namespace p2pdev

// ================================================================================================
// context
// ================================================================================================
constexpr size_t kSlicePad = 512;  // spare entries of the vectors that are exchanged in equal slices (sliced-primal dataflow)
struct pdlpdev_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int32_t m = 0, n = 0;
  int64_t nnz = 0;
  // matrices (values are scaled in place by pdlpdev_scale_problem)
  int32_t *a_off = nullptr, *a_idx = nullptr, *at_off = nullptr, *at_idx = nullptr;
  double *a_val = nullptr, *at_val = nullptr;
  int32_t *a_rb = nullptr, *at_rb = nullptr;  // row-block boundaries of the stream kernels
  int a_nb = 0, at_nb = 0;
  // slab-major row panels (optional second layout of the same nonzeros, see pdlp_kernels.hpp)
  struct Panels {
    bool on = false;
    PanelView v{};
    int32_t* perm = nullptr;  // position in CSR order of each panel-order nonzero
    double* val   = nullptr;
    int64_t nent  = 0;        // nonzeros inside the panels (rows with a workgroup of their own are read from the CSR)
  } pa, pat;
  int cus = 256;  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  // rows of A / of A^T with more than kLongRow nonzeros (set-up kernels give each a workgroup instead of a lane)
  int32_t *a_long = nullptr, *at_long = nullptr;
  int a_nlong = 0, at_nlong = 0;
  // sorted jagged rows with LDS column sets (third layout, structured matrices; pdlp_kernels.hpp)
  struct Jag {
    bool on = false;
    JagView v{};
    int32_t* perm = nullptr;  // position in CSR order of each jagged-order entry
    double* val   = nullptr;
    int64_t nent  = 0;
    double saving = 0.0;      // share of the global gathers the LDS column sets save (build_jag)
  } ja, jat;
  // Dense row segments (runs of >= kDenseMin consecutive columns inside a row: budget / convexity / linking constraints that
  // run through a block of variables) are stored INDEX-FREE, 8 bytes per entry instead of 12, and multiplied by two streaming
  // kernels of their own (k_dense_rows: lane <-> entry, the vector read coalesced; k_dense_cols: lane <-> column, the rows that
  // cover it in ascending order); the four layouts then work on the sparse remainder ("hot" CSR: ha_* / hat_*, the matrices
  // without the segments' entries) and add what the segments contribute ahead of their fused epilogues (dense_plus).  The
  // set-up kernels (norms, scaling) keep running on the full CSR.  Rows / columns a segment touches are compared with the
  // oracle at the long-row tolerance (their sums are split in two).
  struct Dense {
    bool on = false;
    int nrows = 0, nseg = 0, ntiles = 0;
    int64_t nent = 0;
    int32_t *row = nullptr, *row_seg = nullptr;  // rows that own segments, their segment ranges
    int32_t *seg_row = nullptr, *seg_c0 = nullptr, *seg_len = nullptr, *seg_ptr = nullptr;  // nseg (+1)
    int32_t *ch_seg = nullptr, *ch_k0 = nullptr, *row_ch = nullptr;  // chunks of the segments (k_dense_rows), per owning row
    double* ch_part = nullptr;
    int nchunks = 0;
    int32_t *tile_ptr = nullptr, *tile_seg = nullptr, *tile_id = nullptr;  // per 256-column tile some segment overlaps: those segments, ascending rows
    int32_t *perm = nullptr, *s_perm_a = nullptr, *s_perm_at = nullptr;  // positions in the FULL CSR of A / A / A^T
    double* val = nullptr;                   // nent: the segments' values, row after row
    double *add_m = nullptr, *add_n = nullptr;  // what the segments contribute to A v (per row) / A^T v (per column)
    int32_t* tile_slot = nullptr;               // per 256-column tile: its index in tile_ptr, -1 where no segment overlaps it
    bool fused_a = false, fused_at = false;     // the panel kernels of that side add the segments themselves (no launches in front)
    int64_t hot_nnz = 0;
  } dense;
  int64_t hot_nnz_at = 0;
  int32_t *ha_off = nullptr, *ha_idx = nullptr, *hat_off = nullptr, *hat_idx = nullptr;  // the CSR the hot loop multiplies:
  double *ha_val = nullptr, *hat_val = nullptr;                                          // a_* / at_* unless dense.on
  // gather-free layout (fourth layout: huge unstructured matrices; pdlp_kernels.hpp "pb")
  struct Pb {
    bool on = false;
    PbView v{};
    int32_t* perm = nullptr;  // position in CSR order of each padded P-order entry (-1: padding)
    double* val   = nullptr;
    int64_t np    = 0;        // padded entries
    int p_threads = 512;      // phase P workgroup: 512 (8192-column panels) or 1024 (16384)
    double pad    = 1.0;      // padded entries / nonzeros
  } pba, pbat;
  // problem vectors: scaled working copies and the unscaled originals
  double *c = nullptr, *lb = nullptr, *ub = nullptr, *lo = nullptr, *hi = nullptr;
  // all lower (upper) bounds are the same 0 or infinity: k_primal takes the constant instead of streaming the array
  struct UniformBounds {
    int lb_same = 0, ub_same = 0;
    double lb = 0.0, ub = 0.0;
  } ubd;
  void note_uniform_bounds(const double* lb_host, const double* ub_host)  // (null: that side is unchanged)
  {
    auto same = [&](const double* v, int* flag, double* value) {
      if (!v) return;
      *flag = 0;
      if (n <= 0) return;
      const double v0 = v[0];
      if (!(v0 == 0.0 || std::isinf(v0))) return;
      // (16 MB of bounds at n = 1e6: the scan runs on the host pool's threads)
      constexpr int kParts = 16;
      bool differs[kParts] = {false};
      cuopt_amd::parallel_tasks(kParts, [&](int t) {
        const int32_t a = (int32_t)((int64_t)n * t / kParts), b = (int32_t)((int64_t)n * (t + 1) / kParts);
        for (int32_t j = a; j < b; ++j)
          if (v[j] != v0) {
            differs[t] = true;
            return;
          }
      }, (int64_t)n * 4);
      for (bool d : differs)
        if (d) return;
      *flag = 1, *value = v0;
    };
    same(lb_host, &ubd.lb_same, &ubd.lb);
    same(ub_host, &ubd.ub_same, &ubd.ub);
  }
  double *c_u = nullptr, *lb_u = nullptr, *ub_u = nullptr, *lo_u = nullptr, *hi_u = nullptr;
  double *dr = nullptr, *dc = nullptr;
  bool scaled = false;
  // iterate state
  double *x[2] = {nullptr, nullptr}, *y[2] = {nullptr, nullptr}, *aty[2] = {nullptr, nullptr};
  double *xbar = nullptr, *sumx = nullptr, *sumy = nullptr, *avgx = nullptr, *avgy = nullptr;
  double *lrx = nullptr, *lry = nullptr, *rc[2] = {nullptr, nullptr};
  double *tmp_n = nullptr, *tmp_m = nullptr;
  double *ax_u[3] = {nullptr, nullptr, nullptr}, *aty_u[3] = {nullptr, nullptr, nullptr};  // unscaled A x / A^T y of pdlpdev_eval(which)
  double* rc_scratch = nullptr;  // reduced costs of eval(LAST_RESTART): never returned
  double *bestx = nullptr, *besty = nullptr, *bestrc = nullptr;  // save_best_primal_so_far snapshot (scaled x, y)
  // reductions
  double *part_a = nullptr, *part_at = nullptr;  // per-row-block partials (8 quantities each)
  double *part_g = nullptr;                      // generic grid-stride partials
  double *scal = nullptr;                        // device scalars (outputs of finalize kernels)
  double *scal_h = nullptr;                      // pinned mirror
  pdlpdev_ctl *ctl = nullptr, *ctl_h = nullptr;  // device control block + pinned mirror
  pdlpdev_step_params sp = {0.3, 0.6, 0.5, 0.5};
  // multi-GPU
  rccl::comm_t comm = nullptr;  // non-null also marks "sharded mode" when the soft communicator is used
  softcomm::Comm* soft = nullptr;
  std::string comm_key;  // RCCL: this solver's entry of the communicator cache
  int rank = 0, world = 1;
  double* ar_buf = nullptr;  // n + pad doubles: A^T y partial + packed scalars
  // "sliced primal" dataflow of a sharded solve (CUOPT_AMD_SHARD_DATAFLOW=rsag): inside the attempt loop a rank updates only
  // its slice [rank * slice, rank * slice + slice) of the primal-side vectors; reduce-scatter(A^T y' partials) -> slice of
  // A^T y', all-gather(xbar slices) -> the gathered vector of the next A xbar.  Outside the loop everything is replicated.
  bool rsag = false;
  int slice = 0;               // entries per rank, a multiple of 16; slice * world >= n
  double* rs_buf = nullptr;    // slice + 8: this rank's part of the reduced A^T y'
  double* rs_scal = nullptr;   // ||dy||^2, interaction, ||dx||^2 partial sums of this rank -> all-reduced
  // "owner computes" dataflow (CUOPT_AMD_SHARD_DATAFLOW=owner): on top of the sliced primal update a rank also holds ITS
  // COLUMNS of A (rows [rank * slice, ...) of A^T over ALL rows of A), so that A^T y' of its slice is complete on the rank:
  // all-gather(xbar slices) -> local rows of A -> y' -> all-gather(y' row blocks) -> local columns of A -> slice of A^T y'
  // and of the step-size sums.  No partial products travel, nothing is reduced but three scalars, and every column is summed
  // over all rows in row order exactly as on one GPU.
  bool owner = false;
  int ypad = 0;               // entries per rank in the gathered dual vector: the largest row block, a multiple of 16
  double* ygather = nullptr;  // world * ypad: rank q's y' at [q * ypad, ...); also the gather vector of the column block
  int32_t oc_rows = 0;        // columns of A this rank owns (= rows of the column block)
  int64_t oc_nnz = 0;
  int32_t *oc_off = nullptr, *oc_idx = nullptr, *oc_rb = nullptr, *oc_long = nullptr;
  double* oc_val = nullptr;
  int oc_nb = 0, oc_nlong = 0;
  Panels poc;
  Jag joc;
  double* part_oc = nullptr;
  // HALO exchange of the owner-computes dataflow (round 5).  A rank's rows touch only the columns of its own slice plus, on a
  // structured LP (a band, a staircase -- as given or as the set-up's reordering found it), a few thousand columns at the edges of
  // its neighbours' slices; its columns likewise touch its own rows plus the edges of the neighbours' row blocks.  Then the two
  // all-gathers of an attempt (7/8 of n + m doubles received per rank at 8 ranks) shrink to neighbour messages of the ranges
  // actually referenced: per peer ONE contiguous range of xbar (global column numbers) and one of the gathered y' -- kilobytes.
  // Decided at pdlpdev_owner_setup from the matrices themselves (every rank learns every rank's needs through one all-gather of
  // the range table); used when the ranges sum to at most a quarter of the all-gathers' volume (CUOPT_AMD_TUNE=shard_halo=0|1
  // forces it off / on).  Values are the same doubles either way: iterates are bit-identical to the all-gather's.
  struct Halo {
    bool on = false;
    std::vector<int32_t> recv_off[2], recv_cnt[2], send_off[2], send_cnt[2];  // [kind 0 xbar | kind 1 y'][peer]: offsets into the buffer
    int64_t bytes = 0, bytes_allgather = 0;  // received per attempt by this rank: the halo's ranges / the two all-gathers
  } halo;
  // direct peer transport of the owner-computes dataflow (CUOPT_AMD_SHARD_TRANSPORT=p2p): every rank owns one fine-grained
  // LANDING block [xbar of all ranks | y' of all ranks | 4 step-size scalars per rank | 3 * world epoch flags]; a producer
  // stores its slice into every rank's block (peer-mapped: same process -> the pointer itself after
  // hipDeviceEnablePeerAccess, other process -> hipIpcOpenMemHandle) and then raises its flag there; a consumer waits for the
  // world flags of the exchange, then copies the landed data into its ordinary vectors.  No collective call, no host between
  // the kernels of an attempt -> the attempt graph replays as on one GPU.
  struct P2P {
    bool on = false;
    char* base = nullptr;            // this rank's landing block
    size_t bytes = 0;
    size_t off_x = 0, off_y = 0, off_s = 0, off_f = 0;  // byte offsets inside every rank's block
    p2pdev::Peers peers{};           // base of every rank's block as THIS process addresses it
    std::vector<void*> opened;       // hipIpcOpenMemHandle mappings to close
    unsigned long long* epoch = nullptr;  // device: epochs of the three exchanges (xbar, y', scalars) as this rank counts them
    int* fault = nullptr;            // device: set when a wait ran out of patience (peer died)
    p2pdev::Push* push_dev = nullptr;  // device: the three exchanges' descriptors (xbar, y', scalars) for the producing kernels
    p2pdev::Push push_host[p2pdev::kKinds];  // their host copies (the halo set-up narrows the ranges afterwards)
  } p2p;
  // pdlpdev_time_kernel: the next launch through launch_k carries these events (kernel start / stop timestamps of the
  // dispatch itself, what rocprofv3 --kernel-trace reports)
  // (a call site may consist of several launches -- dense segments, phase P, phase R: each gets its own pair, the durations add up)
  bool prof_armed = false;
  static constexpr int kProfPairs = 8;
  hipEvent_t prof_ev[2 * kProfPairs] = {};
  int prof_used = 0;
  int rejected_in_a_row = 0;  // attempts enqueued since the last accepted step (pdlpdev_run's guard against endless rejections)
  // graphs
  int use_graph = 1;
  bool comm_warm = false;          // one attempt went out eagerly over this communicator (before the first capture)
  bool graph_comm_failed = false;  // capturing the RCCL collectives into an attempt graph failed once: plain launches from then on
  // one allocation for the ~40 problem / iterate vectors of a large LP (each hipMalloc + memset pair costs ~0.1 ms: 3 ms of a 25 ms
  // set-up at 1e6 x 1e6); dev_alloc draws from it while it lasts
  char* slab = nullptr;
  size_t slab_cap = 0, slab_used = 0;
  char* arena = nullptr;  // current small-buffer chunk (dev_alloc)
  char* first_chunk = nullptr;  // recycled with the stream, not in `allocs`
  bool stream_borrowed = false; // pdlpdev_create_share_stream: the stream is another context's
  size_t arena_used = 0;
  bool small_resident = false;  // whole attempt batches inside one workgroup (k_pdhg_small)
  bool shared_with_parent = false;  // pdlpdev_clone_shared: matrices, layouts, scaling vectors, c and the stream are another context's
  pdlpdev_ctx* parent = nullptr;    // ... that one
  bool rows_aliased = false;        // ... and so are the row bounds (lo, hi and their unscaled twins) until a reset brings other ones:
                                    // K clones that differ in their VARIABLE bounds read one copy (the batched dual step then fetches
                                    // lo / hi once for all LPs -- the caches see the same addresses)
  int clones_alive = 0;             // contexts that alias this one's arrays
  bool rows_private = false;        // a parent whose lo / hi were re-allocated by a reset AFTER its last clone was made: no clone aliases them
  int batches_alive = 0;            // pdlpdev_batch / pdlpdev_small_batch objects that hold this context's pointer
  std::map<int, hipGraphExec_t> graphs;  // attempts-per-replay -> executable graph
  std::vector<void*> allocs;
  int64_t bytes = 0;
  // Scratch of the layouts' device constructions (kernels_layout_build.hip): the A side's temporaries serve the A^T side again -- a
  // hipMalloc of a few GB right after a hipFree of the same size took 1.9 s at 1e9 nonzeros (the first one of the process: 2 ms) --
  // and go back to the runtime when the context has been created (scratch_free, pdlp_create.hip).
  struct Scratch {
    void* p;
    size_t bytes;
    bool busy;
  };
  std::vector<Scratch> scratch;
};
inline int scratch_take(pdlpdev_ctx* c, void** p, size_t bytes)
{
  bytes = std::max<size_t>(bytes, 256);
  for (auto& b : c->scratch)
    if (!b.busy && b.bytes >= bytes && b.bytes <= 2 * bytes + 4096) {
      b.busy = true;
      *p     = b.p;
      return 0;
    }
  HIP_TRY(hipMalloc(p, bytes));
  c->scratch.push_back(pdlpdev_ctx::Scratch{*p, bytes, true});
  return 0;
}
inline void scratch_release(pdlpdev_ctx* c, void* p)
{
  for (auto& b : c->scratch)
    if (b.p == p) b.busy = false;
}
inline void scratch_free(pdlpdev_ctx* c)
{
  for (auto& b : c->scratch) (void)hipFree(b.p);
  c->scratch.clear();
}

// one LP's arguments of k_step_decision_batch (pdlp_device.hip; launched by kernels_batch.hip)
struct pdlpdev_decision_args {
  pdlpdev_ctl* ctl;
  const double* part_dy;
  int nb_dy;
  const double* part_t;
  int nb_t;
  pdlpdev_step_params sp;
};

constexpr int kGenericBlocks = 1024;
constexpr int kScalars       = 64;

// Streams (an HSA queue each: ~2 ms to create), the pinned read-back block and the first arena chunk are handed from
// a destroyed context to the next one created on the same device: back-to-back small solves (cuOptSolve in a loop,
// MIP-style re-solves) otherwise spend more time in these three calls than in PDHG.  Never freed (a few per device).
// hipFuncSetAttribute is per device and must happen before the first launch that asks for > 64 KiB of LDS:
// one flag per (kernel instantiation, device), taken under a lock (batch solves create contexts from many threads)
struct PerDeviceOnce {
  std::mutex m;
  bool done[64] = {};
  template <class F>
  int run(int device, F&& f)
  {
    std::lock_guard<std::mutex> lock(m);
    if (device < 0 || device >= 64) return f();
    if (done[device]) return 0;
    const int rc = f();
    if (rc == 0) done[device] = true;
    return rc;
  }
};

struct Recycled {
  int device;
  hipStream_t stream;
  double* pinned;
  char* chunk;
};
inline std::mutex g_recycle_mutex;
inline std::vector<Recycled> g_recycled;
inline bool take_recycled(int device, Recycled* out)
{
  std::lock_guard<std::mutex> lock(g_recycle_mutex);
  for (size_t i = 0; i < g_recycled.size(); ++i)
    if (g_recycled[i].device == device) {
      *out = g_recycled[i];
      g_recycled.erase(g_recycled.begin() + i);
      return true;
    }
  return false;
}
inline bool give_recycled(const Recycled& r)
{
  std::lock_guard<std::mutex> lock(g_recycle_mutex);
  if (g_recycled.size() >= 16) return false;
  g_recycled.push_back(r);
  return true;
}

// Zero-filled device memory.  Buffers under 256 KiB are carved out of 1 MiB chunks: a small LP (the MIP-style
// re-solve case) needs ~60 buffers, and 60 hipMalloc + hipFree calls cost more than its whole solve.
constexpr size_t kArenaChunk = (size_t)1 << 20, kArenaMaxItem = (size_t)256 << 10;
template <class T>
inline int dev_alloc(pdlpdev_ctx* c, T** p, size_t count)
{
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  c->bytes += (int64_t)bytes;
  if (bytes <= kArenaMaxItem) {
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (c->arena == nullptr || c->arena_used + need > kArenaChunk) {
      HIP_TRY(hipMalloc((void**)&c->arena, kArenaChunk));
      HIP_TRY(hipMemsetAsync(c->arena, 0, kArenaChunk, c->stream));
      c->allocs.push_back(c->arena);
      c->arena_used = 0;
    }
    *p = (T*)(c->arena + c->arena_used);
    c->arena_used += need;
    return 0;
  }
  {
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (c->slab && c->slab_used + need <= c->slab_cap) {
      *p = (T*)(c->slab + c->slab_used);
      c->slab_used += need;
      return 0;
    }
  }
  static const bool timing = getenv("CUOPT_AMD_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipMalloc((void**)p, bytes));
  const auto t1 = std::chrono::steady_clock::now();
  HIP_TRY(hipMemsetAsync(*p, 0, bytes, c->stream));
  if (timing && std::chrono::duration<double>(t1 - t0).count() > 1e-3)  // only the surprising ones
    fprintf(stderr, "[cuopt_amd setup]     hipMalloc %10zu B: %.2f ms\n", bytes, 1e3 * std::chrono::duration<double>(t1 - t0).count());
  c->allocs.push_back(*p);
  return 0;
}
#define TRY(expr)          \
  do {                     \
    int rc_ = (expr);      \
    if (rc_ != 0) return rc_; \
  } while (0)

// the stream kernels remap blockIdx so that each XCD owns a contiguous range of row blocks
// (xcd_remap); the grid is padded to a multiple of 8 so the remap is a bijection.
static inline int stream_grid(int nb) { return std::max(8, ((nb + 7) / 8) * 8); }
static inline int grid_for(int64_t n, int per_thread = 1)
{
  int64_t g = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, 2048));
}

// ---- defined in pdlp_comm.hip (communicators, collectives, the peer transport's set-up) ------------------------------------------
int setup_dataflow(pdlpdev_ctx* ctx);
int reduce_scatter(pdlpdev_ctx* ctx, const double* send, double* recv, size_t count);
int all_gather(pdlpdev_ctx* ctx, double* buf, size_t count);
int allreduce(pdlpdev_ctx* ctx, double* buf, size_t count, int op);
int p2p_setup(pdlpdev_ctx* ctx);
int p2p_push_ranges(pdlpdev_ctx* ctx);
int halo_setup(pdlpdev_ctx* ctx, const int32_t* need /* [world][4]: xbar lo, hi (columns), y' lo, hi (positions in ygather) */);
int halo_exchange(pdlpdev_ctx* ctx, int kind, double* buf);
