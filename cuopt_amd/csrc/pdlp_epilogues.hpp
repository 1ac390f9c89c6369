// Device-side pieces shared by the kernel translation units (kernels_stream / _panel / _jag / _pb / _dense .hip) and the core
// (pdlp_device.hip): the direct peer transport's device structs, the fused epilogues of the SpMV skeletons, the dense-segment view.
#pragma once
#include "pdlp_kernels.hpp"

using namespace pdlp;

namespace p2pdev {
struct Peers {
  char* base[16];
};
constexpr int kKinds = 3;  // exchanges per attempt: xbar slices, y' row blocks, step-size scalars
__device__ __forceinline__ bool active(const pdlpdev_ctl* ctl) { return ctl->error == 0 && ctl->steps_taken < ctl->target_steps; }

// What a PRODUCING kernel (primal step, dual step, packing of the step-size sums) needs to store its results straight into
// every rank's landing block and to raise this rank's flag there; lives in device memory (one per exchange), kernels take a
// pointer (null: no peer transport) and read the table with scalar loads.
struct Push {
  Peers P;
  int world, rank, kind;
  size_t dst_off, flag_off;  // bytes inside every rank's block: this rank's slot of the exchange / the flag area
  unsigned long long* epoch;
  // halo exchange through the peer stores (round 6): rank q needs the entries [lo[q], hi[q]) of this rank's share only -- what its rows
  // (kind 0: xbar) or columns (kind 1: y') reference, found at set-up (halo_setup); everything: [0, INT_MAX)
  int lo[16], hi[16];
  __device__ __forceinline__ double* slot(int q) const { return reinterpret_cast<double*>(P.base[q] + dst_off); }
  __device__ __forceinline__ bool wants(int q, int i) const { return q != rank && i >= lo[q] && i < hi[q]; }
};
// A producing kernel's store into a landing block: system scope = write-through, so that no cache write-back is needed before the
// flag goes up
__device__ __forceinline__ void put(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// Called once per workgroup at the end of a producing kernel: ONE thread of the grid counts the exchange.  The flag itself is raised
// by the first workgroup of the NEXT kernel of the stream (raise, below): a kernel boundary is what guarantees that every store of
// the producer has landed, for nothing.  (Round 3 published from the producer's tail -- every thread waited for its own stores, the
// workgroup took a ticket, the last one raised the flags: at one rank that made k_primal 31.8 us instead of 13.)
__device__ __forceinline__ void count_exchange(const Push* T)
{
  if (blockIdx.x == 0 && threadIdx.x == 0) T->epoch[T->kind] = T->epoch[T->kind] + 1;
}
// first thread of the consuming kernel: this rank's flag of the exchange goes up in every rank's block
__device__ __forceinline__ void raise(const Push* T)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const unsigned long long e = T->epoch[T->kind];
  for (int q = 0; q < T->world; ++q) {
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(T->P.base[q] + T->flag_off) + (size_t)T->kind * T->world + T->rank;
    __hip_atomic_store(flag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// lane q waits for rank q's flag of exchange `kind` (relaxed polls), then ONE lane acquires for the workgroup; false when
// patience ran out (5 s: a peer died)
__device__ __forceinline__ bool wait_flags(const unsigned long long* flags, int world, int kind, const unsigned long long* epoch)
{
  bool ok = true;
  if ((int)threadIdx.x < world) {
    const unsigned long long want = epoch[kind];
    const unsigned long long* f   = flags + (size_t)kind * world + threadIdx.x;
    const unsigned long long t0   = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 500000000ull) {  // 100 MHz
        ok = false;
        break;
      }
    }
  }
  ok = __syncthreads_and(ok);
  if (threadIdx.x == 0) __threadfence_system();
  __syncthreads();
  return ok;
}
}  // namespace p2pdev

// (2) rows of A: v = A xbar (stream SpMV) -> dual projection (utils.cuh:97-112) -> ||dy||^2 partial,
//     plus the deferred dual averaging.
struct DualEpilogue {
  static constexpr int NQ = 1;
  using Op = SumOp;
  const double* __restrict__ y;
  double* __restrict__ yn;
  const double* __restrict__ lo;
  const double* __restrict__ hi;
  double* __restrict__ sumy;
  double sigma, weight;
  bool pend;
  double* __restrict__ copy = nullptr;  // sharded solves, owner-computes dataflow: y' also goes to this rank's slot of the gathered dual
  const p2pdev::Push* __restrict__ push = nullptr;  // ... or, with the direct peer transport, into every rank's landing block
  // the row's operands, separable from the arithmetic so that a layout can request them before its row sums are ready
  struct Ops {
    double y, lo, hi, sum;
  };
  __device__ __forceinline__ Ops load(int i) const { return Ops{y[i], lo[i], hi[i], pend ? sumy[i] : 0.0}; }
  __device__ __forceinline__ void apply(int i, double v, const Ops& o, double (&acc)[1])
  {
    const double yi = o.y;
    double next     = yi - (sigma * v);
    const double low = next + sigma * o.lo;
    const double up  = next + sigma * o.hi;
    next            = dmax(low, dmin(up, 0.0));
    yn[i]           = next;
    if (copy) copy[i] = next;
    if (push)
      for (int q = 0; q < push->world; ++q)
        if (push->wants(q, i)) p2pdev::put(push->slot(q) + i, next);  // (this rank's own rows: `copy`, an ordinary store)
    const double dy = next - yi;
    acc[0] += dy * dy;
    if (pend) sumy[i] = o.sum + weight * yi;
  }
  __device__ __forceinline__ void row(int i, double v, double (&acc)[1]) { apply(i, v, load(i), acc); }
};

// (3) rows of A^T: AtY' = A^T y' (stream SpMV) fused with the step-size statistics
//     interaction = dx . (AtY' - AtY), ||dx||^2  (adaptive_step_size_strategy.cu:278-340)
struct StepEpilogue {
  static constexpr int NQ = 2;
  using Op = SumOp;
  const double* __restrict__ x;
  const double* __restrict__ xn;
  const double* __restrict__ aty;
  double* __restrict__ atyn;
  struct Ops {
    double x, xn, aty;
  };
  __device__ __forceinline__ Ops load(int j) const { return Ops{x[j], xn[j], aty[j]}; }
  __device__ __forceinline__ void apply(int j, double v, const Ops& o, double (&acc)[2])
  {
    atyn[j]         = v;
    const double dx = o.xn - o.x;
    const double t  = v - o.aty;
    acc[0] += t * dx;
    acc[1] += dx * dx;
  }
  __device__ __forceinline__ void row(int j, double v, double (&acc)[2]) { apply(j, v, load(j), acc); }
};

// (plain SpMV: A^T y at start / after restart-to-average; parity hook; multi-GPU partial products)
struct StoreEpilogue {
  static constexpr int NQ = 0;
  using Op = SumOp;
  double* __restrict__ out;
  __device__ __forceinline__ void row(int r, double v, double (&)[1]) { out[r] = v; }
};

// Convergence information, primal side (convergence_information.cu:221-248 + row part of :323-366):
// rows of the SCALED A against the SCALED iterate; (A x)_i = (A^ x^)_i / D_r,i and y_i = y^_i D_r,i
// recover the unscaled quantities without a second copy of the matrix.
struct EvalPrimalEpilogue {
  static constexpr int NQ = 3;
  using Op = SumOp;
  const double* __restrict__ yhat;
  const double* __restrict__ dr;
  const double* __restrict__ lo_u;
  const double* __restrict__ hi_u;
  double eps_rel;
  double* __restrict__ linf_rows;  // per-row r_p,i - eps*bcomb_i (max-reduced by a second pass)
  double* __restrict__ ax_out;     // (A x)_i of the unscaled problem, kept for the infeasibility pass
  __device__ __forceinline__ void row(int i, double v, double (&acc)[3])
  {
    const double d  = dr[i];
    const double ax = v / d;
    ax_out[i]       = ax;
    const double yi = yhat[i] * d;
    const double lo = lo_u[i], hi = hi_u[i];
    const double rp = violation(ax, lo, hi);
    acc[0] += rp * rp;
    acc[1] += bound_value_product(yi, lo, hi);
    acc[2] += yi * yi;
    if (linf_rows) linf_rows[i] = rp - eps_rel * combine_bounds(lo, hi);  // relative_residual_t, utils.cuh:385-409
  }
};

// dual side (convergence_information.cu:261-320,369-422): one column j per lane
struct EvalDualCore {
  const double* __restrict__ xhat;
  const double* __restrict__ dc;
  const double* __restrict__ c_u;
  const double* __restrict__ lb_u;
  const double* __restrict__ ub_u;
  double eps_rel;
  int rule_finite;
  double* __restrict__ rc_out;
  double* __restrict__ linf_rows;
  double* __restrict__ aty_out;  // (A^T y)_j of the unscaled problem, kept for the infeasibility pass
  // acc: 0 ||r_d||^2, 1 sum B(rc,lb,ub), 2 c.x, 3 ||x||^2
  __device__ __forceinline__ void col(int j, double aty_scaled, double (&acc)[4])
  {
    const double d    = dc[j];
    const double aty  = aty_scaled / d;
    aty_out[j]        = aty;
    const double cj   = c_u[j];
    const double g    = cj - aty;
    const double xj   = xhat[j] * d;
    const double lb   = lb_u[j], ub = ub_u[j];
    const double bv   = g > 0.0 ? lb : ub;  // bound_value_gradient, utils.cuh:195-202
    double rc;
    if (g == 0.0)
      rc = g;
    else if (rule_finite)  // copy_gradient_if_finite_bounds, utils.cuh:231-239
      rc = dfinite(bv) ? g : 0.0;
    else  // copy_gradient_if_should_be_reduced_cost, utils.cuh:221-229
      rc = fabs(xj - bv) <= fabs(xj) ? g : 0.0;
    const double rd = g - rc;
    rc_out[j]       = rc;
    acc[0] += rd * rd;
    acc[1] += bound_value_product(rc, lb, ub);
    acc[2] += cj * xj;
    acc[3] += xj * xj;
    if (linf_rows) linf_rows[j] = rd - eps_rel * cj;  // the dual "rhs" is c_j itself (signed), :204-208
  }
};

struct EvalDualEpilogue {
  static constexpr int NQ = 4;
  using Op = SumOp;
  EvalDualCore core;
  __device__ __forceinline__ void row(int j, double v, double (&acc)[4]) { core.col(j, v, acc); }
};

// ---- dense row segments: index-free storage (pdlpdev_ctx::Dense) ---------------------------------------------------------
struct DenseView {
  const int32_t* __restrict__ row;
  const int32_t* __restrict__ row_seg;
  const int32_t* __restrict__ seg_row;
  const int32_t* __restrict__ seg_c0;
  const int32_t* __restrict__ seg_len;
  const int32_t* __restrict__ seg_ptr;
  const int32_t* __restrict__ tile_ptr;
  const int32_t* __restrict__ tile_seg;
  const int32_t* __restrict__ tile_id;  // the 256-column tiles some segment overlaps
  const double* __restrict__ val;
  const int32_t* __restrict__ ch_seg;   // chunks of <= kDenseChunk entries of a segment: one workgroup each ...
  const int32_t* __restrict__ ch_k0;
  const int32_t* __restrict__ row_ch;   // ... and per owning row its chunk range (added up in this order)
  double* __restrict__ ch_part;
};

constexpr int kDenseChunk = 4096;

// the gathered vector of a call site, picked on the device like the layouts do (see k_pb_products)
__device__ __forceinline__ const double* pick_vector(const pdlpdev_ctl* ctl, const double* v0, const double* v1, int mode)
{
  if (mode == 0) return v0;
  const bool cur = ctl->cur != 0;
  return (cur == (mode == 1)) ? v0 : v1;
}
