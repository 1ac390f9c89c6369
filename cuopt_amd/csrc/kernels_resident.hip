// The resident small-LP path (BASELINE config 5: LP relaxations of MIP size, re-solved over and over) -- its own translation unit
// since round 6: the one-workgroup PDHG loop, the one-workgroup head of a major iteration, and the BATCH of K such LPs in K
// workgroups of one launch (pdlpdev_small_batch_*: the MI355X-first answer to the reference's thread pool of independent solves,
// cpp/src/linear_programming/utilities/cython_solve.cu:264-296, and to the MIP heuristics' streams of relaxations,
// cpp/src/mip/relaxed_lp/relaxed_lp.cu:53-127 -- one LP keeps ONE of the chip's 256 CUs busy, 256 LPs keep all of them).
#include "pdlp_ctx.hpp"
#include "pdlp_layouts.hpp"

// ------------------------------------------------------------------------------------------------
// Small LPs (MIP-style repeated re-solves, BASELINE config 5): the whole batch of PDHG attempts between two
// major iterations runs inside ONE workgroup, with the LP on chip.  At this size a 4-launch attempt is pure
// launch latency (~15 us) and even L2 round trips (3-4 dependent ones per phase) cost more than the arithmetic,
// so nothing is re-read from memory inside the loop:
//   * lane t keeps nonzeros t, t+T, ... of A and of A^T (value + column) in registers;
//   * lane t owns rows / columns t, t+T, ...: their CSR extents and every per-element vector
//     (x, A^T y, c, bounds, running sums ...) live in its registers;
//   * the two gathered vectors (xbar, y'), the nonzero products and the constant vectors (c, bounds) sit in LDS.
// Products are val * vec[col] and every row is added up by its owner in CSR order, so x', y', A^T y' are
// bit-identical to the multi-launch kernels (and the oracle); the three step-size sums use a different,
// fixed reduction tree.  T lanes, Q elements and U nonzeros per lane: m, n <= Q*T, nnz <= U*T.
// ------------------------------------------------------------------------------------------------
struct SmallView {
  int m, n, nnz;
  const int32_t *a_off, *a_idx, *at_off, *at_idx;
  const double *a_val, *at_val, *c, *lb, *ub, *lo, *hi;
  double *x0, *x1, *y0, *y1, *aty0, *aty1, *sumx, *sumy;
};
// prod[a..b) added up strictly left to right; eight LDS reads are in flight before the first add
__device__ __forceinline__ double lds_row_sum(const double* prod, int a, int b)
{
  double acc = 0.0;
  for (int k = a; k < b; k += 8) {
    double p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = prod[k + i < b ? k + i : a];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = k + i < b ? acc + p[i] : acc;
  }
  return acc;
}
// LDS hazards of the resident loop (round-6 audit), barriers B1 ... B5 of an attempt (B5 sits inside block_sum_fast):
//   xbar_s  written in the primal phase (own columns) before B1, gathered before B2; next written after B5 of the same attempt.
//   prod    written before B2 (A xbar products, own slots) and before B4 (A^T y' products); read by the row sums before B3 / B5;
//           each rewrite is behind the barrier that ends the previous readers (B3 -> B4's writes, B5 -> the next attempt's B2 writes).
//   yn_s    written before B3 (own rows), gathered before B4; next written behind B1 ... B2 of the next attempt.
//   red[2], pw[2]  by attempt parity: written before B5 / B3, read behind B5; the same parity is written again two attempts later,
//           ten barriers on.  Constants (c_s, bounds) are written once before the first B1.
template <int T, int Q, int U>
__device__ __forceinline__ void resident_body(const SmallView& V, pdlpdev_ctl* __restrict__ ctl, pdlpdev_ctl* __restrict__ ctl_host,
                                              const pdlpdev_step_params& sp, int target_steps, int max_attempts, double* lds)
{
  double* xbar_s = lds;               // Q*T
  double* yn_s   = xbar_s + Q * T;    // Q*T
  double* c_s    = yn_s + Q * T;      // constants, read with stride 1 by their owners
  double* lb_s   = c_s + Q * T;
  double* ub_s   = lb_s + Q * T;
  double* lo_s   = ub_s + Q * T;
  double* hi_s   = lo_s + Q * T;
  double* prod   = hi_s + Q * T;      // U*T
  // two sets (attempt parity) of cross-wave partials: a wave may start the next attempt's reduction while a slower
  // one still reads this attempt's table -- the barriers in between only order the *other* buffers
  __shared__ double red[2][3 * 16];
  __shared__ double pw[2][2];  // the two powers of the step-size rule, computed by the last wave while rows are summed
  const int t = threadIdx.x;
  // Every lane keeps its own copy of the control block and repeats the (uniform) step decision: no broadcast
  // through LDS and no barrier between the reduction and the next primal step.
  pdlpdev_ctl lc = *ctl;
  lc.target_steps = target_steps;
  double a_val[U], at_val[U];
  int a_col[U], at_col[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k  = t + u * T;
    const bool in = k < V.nnz;
    a_val[u]  = in ? V.a_val[k] : 0.0;
    a_col[u]  = in ? V.a_idx[k] : 0;
    at_val[u] = in ? V.at_val[k] : 0.0;
    at_col[u] = in ? V.at_idx[k] : 0;
  }
  const int cur0 = lc.cur;
  int r0[Q], r1[Q], c0[Q], c1[Q];  // CSR extents of the owned rows of A and of A^T (empty when out of range)
  double x[Q], xn[Q], aty[Q], atyn[Q], sumx[Q], y[Q], yn[Q], sumy[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int e = t + q * T;
    const bool row = e < V.m, col = e < V.n;
    r0[q] = row ? V.a_off[e] : 0, r1[q] = row ? V.a_off[e + 1] : 0;
    c0[q] = col ? V.at_off[e] : 0, c1[q] = col ? V.at_off[e + 1] : 0;
    c_s[e] = col ? V.c[e] : 0.0, lb_s[e] = col ? V.lb[e] : 0.0, ub_s[e] = col ? V.ub[e] : 0.0;
    x[q]    = col ? (cur0 ? V.x1 : V.x0)[e] : 0.0;
    aty[q]  = col ? (cur0 ? V.aty1 : V.aty0)[e] : 0.0;
    sumx[q] = col ? V.sumx[e] : 0.0;
    lo_s[e] = row ? V.lo[e] : 0.0, hi_s[e] = row ? V.hi[e] : 0.0;
    y[q]    = row ? (cur0 ? V.y1 : V.y0)[e] : 0.0;
    sumy[q] = row ? V.sumy[e] : 0.0;
    xn[q] = x[q], atyn[q] = aty[q], yn[q] = y[q];
  }
  const int used = (V.nnz + T - 1) / T;  // nonzero slots in use (uniform): tiny LPs skip the empty ones
  for (int attempt = 0; attempt < max_attempts; ++attempt) {
    if (lc.error != 0 || lc.steps_taken >= lc.target_steps) break;  // uniform: every lane holds the same lc
    const int cur       = lc.cur;
    const double tau = lc.tau, sigma = lc.sigma, weight = lc.step_size;
    const bool pend     = lc.pending_avg != 0;
    const double knext  = (double)(lc.k + 1) + 1.0;
    const int par       = attempt & 1;
    // primal projection (utils.cuh:80-95) + deferred averaging
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = t + q * T;
      if (j < V.n) {
        const double gradient = c_s[j] - aty[q];
        double next           = x[q] - (tau * gradient);
        next                  = dmax(dmin(next, ub_s[j]), lb_s[j]);
        xn[q]                 = next;
        xbar_s[j]             = next - x[q] + next;
        if (pend) sumx[q] = sumx[q] + weight * x[q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < used) prod[t + u * T] = a_val[u] * xbar_s[a_col[u]];
    __syncthreads();
    // y' = proj(y - sigma A xbar) (utils.cuh:97-112), ||dy||^2
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int i = t + q * T;
      if (i < V.m) {
        const double ax = lds_row_sum(prod, r0[q], r1[q]);
        double next      = y[q] - (sigma * ax);
        const double low = next + sigma * lo_s[i];
        const double up  = next + sigma * hi_s[i];
        next             = dmax(low, dmin(up, 0.0));
        yn[q]            = next;
        yn_s[i]          = next;
        const double dy  = next - y[q];
        acc[0] += dy * dy;
        if (pend) sumy[q] = sumy[q] + weight * y[q];
      }
    }
    if (t >= T - 2) pw[par][t - (T - 2)] = pow(knext, t == T - 2 ? -sp.reduction_exponent : -sp.growth_exponent);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < used) prod[t + u * T] = at_val[u] * yn_s[at_col[u]];
    __syncthreads();
    // A^T y' + interaction / ||dx||^2 (adaptive_step_size_strategy.cu:278-340)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = t + q * T;
      if (j < V.n) {
        const double v = lds_row_sum(prod, c0[q], c1[q]);
        atyn[q]          = v;
        const double dx  = xn[q] - x[q];
        const double dty = v - aty[q];
        acc[1] += dty * dx;
        acc[2] += dx * dx;
      }
    }
    block_sum_fast<3, T / 64>(acc, red[par]);
    apply_step_decision(&lc, acc[0], acc[1], acc[2], sp, pw[par]);
    if (lc.cur != cur) {  // accepted: the candidate becomes the iterate
#pragma unroll
      for (int q = 0; q < Q; ++q) x[q] = xn[q], aty[q] = atyn[q], y[q] = yn[q];
    }
  }
  {
    const int cur = lc.cur;
    double* xo    = cur ? V.x1 : V.x0;
    double* yo    = cur ? V.y1 : V.y0;
    double* atyo  = cur ? V.aty1 : V.aty0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int e = t + q * T;
      if (e < V.n) xo[e] = x[q], atyo[e] = aty[q], V.sumx[e] = sumx[q];
      if (e < V.m) yo[e] = y[q], V.sumy[e] = sumy[q];
    }
  }
  if (t == 0) *ctl = lc, *ctl_host = lc;  // the pinned mirror saves the read-back copy
}
template <int T, int Q, int U>
__global__ void __launch_bounds__(T)
k_pdhg_resident(SmallView V, pdlpdev_ctl* __restrict__ ctl, pdlpdev_ctl* __restrict__ ctl_host, pdlpdev_step_params sp,
                int target_steps, int max_attempts)
{
  extern __shared__ double lds[];
  resident_body<T, Q, U>(V, ctl, ctl_host, sp, target_steps, max_attempts, lds);
}
// K LPs, one workgroup each, ONE launch (round 6): workgroup b runs the loop of LP list[b] exactly as k_pdhg_resident would -- the
// same body, so every LP's trajectory is bit for bit the one of its own launch.  The argument records sit in pinned host memory
// (a workgroup reads ~300 bytes of it once).
struct ResidentArgs {
  SmallView V;
  pdlpdev_ctl *ctl, *ctl_host;
  pdlpdev_step_params sp;
  int target_steps, pad;
};
template <int T, int Q, int U>
__global__ void __launch_bounds__(T) k_pdhg_resident_batch(const ResidentArgs* __restrict__ args, const int* __restrict__ list, int max_attempts)
{
  extern __shared__ double lds[];
  const ResidentArgs& A = args[list[blockIdx.x]];
  resident_body<T, Q, U>(A.V, A.ctl, A.ctl_host, A.sp, A.target_steps, max_attempts, lds);
}
// the three instantiations, smallest first: (lanes, elements per lane, nonzeros per lane)
struct ResidentTier { int T, Q, U; };
constexpr ResidentTier kResidentTiers[3] = {{256, 2, 8}, {512, 2, 16}, {512, 4, 8}};
int resident_tier(int m, int n, int64_t nnz)
{
  for (int i = 0; i < 3; ++i) {
    const ResidentTier& r = kResidentTiers[i];
    if (m <= r.Q * r.T && n <= r.Q * r.T && nnz <= (int64_t)r.U * r.T) return i;
  }
  return -1;
}
static size_t resident_lds_bytes(int tier)
{
  const ResidentTier& r = kResidentTiers[tier];
  return sizeof(double) * (size_t)r.T * (7 * r.Q + r.U);
}
template <int T, int Q, int U>
static int launch_resident(hipStream_t s, int tier, const SmallView& V, pdlpdev_ctl* ctl, pdlpdev_ctl* ctl_host,
                           const pdlpdev_step_params& sp, int target_steps)
{
  static PerDeviceOnce once;  // per instantiation
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  TRY(once.run(device, [&]() -> int {
    HIP_TRY(hipFuncSetAttribute((const void*)k_pdhg_resident<T, Q, U>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)resident_lds_bytes(tier)));
    return 0;
  }));
  k_pdhg_resident<T, Q, U><<<1, T, resident_lds_bytes(tier), s>>>(V, ctl, ctl_host, sp, target_steps, 1 << 14);
  HIP_TRY(hipGetLastError());
  return 0;
}


// single-workgroup head of a major iteration (pdlpdev_major_eval) for LPs on the resident path
struct MajorSmallArgs {
  int m, n, mode, rule_finite, want_linf;
  double eps_p, eps_d;
  const int32_t *a_off, *a_idx, *at_off, *at_idx;
  const double *a_val, *at_val;
  pdlpdev_ctl* ctl;
  double *x0, *x1, *y0, *y1, *sumx, *sumy, *avgx, *avgy;
  const double *dr, *dc, *c_u, *lb_u, *ub_u, *lo_u, *hi_u;
  double *linf_m, *linf_n, *ax_cur, *ax_avg, *aty_cur, *aty_avg, *rc_cur, *rc_avg;
  double* sc;  // current at sc[0..9), average at sc[32..41)  (pinned host memory: no read-back copy)
  int guard_target = -1;  // >= 0 (evaluation enqueued right behind the attempts of a small-LP batch): only if the attempts reached this
                          // accepted-step count or raised the step-size error -- i.e. only if a major iteration is what comes next
};
constexpr int kMajorThreads = 1024;
// M vec for a matrix of <= 8192 nonzeros: all products in parallel into LDS, then every row is added up left to
// right by one lane (same order as every other SpMV here)
template <class Epi, int NQ>
__device__ __forceinline__ void small_rows(int rows, const int32_t* __restrict__ off, const int32_t* __restrict__ idx,
                                           const double* __restrict__ val, const double* vec, double* prod, Epi& e,
                                           double (&acc)[NQ])
{
  const int nnz = off[rows];
  for (int k = threadIdx.x; k < nnz; k += kMajorThreads) prod[k] = val[k] * vec[idx[k]];
  __syncthreads();
  for (int r = threadIdx.x; r < rows; r += kMajorThreads) e.row(r, lds_row_sum(prod, off[r], off[r + 1]), acc);
}
__device__ __forceinline__ void major_small_body(const MajorSmallArgs& A, double* prod /* nnz doubles of LDS */)
{
  __shared__ double red[4 * kMajorThreads / 64];
  const int t = threadIdx.x;
  const int cur = A.ctl->cur;
  double* x = cur ? A.x1 : A.x0;
  double* y = cur ? A.y1 : A.y0;
  const bool pend = A.ctl->pending_avg != 0;
  const double w = A.ctl->step_size, sw = A.ctl->sum_weights;
  for (int j = t; j < A.n; j += kMajorThreads) {
    double sx = A.sumx[j];
    if (pend) A.sumx[j] = sx = sx + w * x[j];
    A.avgx[j] = A.mode == 0 ? x[j] : (A.mode == 1 ? 0.0 : sx / sw);
  }
  for (int i = t; i < A.m; i += kMajorThreads) {
    double sy = A.sumy[i];
    if (pend) A.sumy[i] = sy = sy + w * y[i];
    A.avgy[i] = A.mode == 0 ? y[i] : (A.mode == 1 ? 0.0 : sy / sw);
  }
  __syncthreads();  // also orders the global writes above against the reads below (one workgroup)
  if (t == 0) A.ctl->pending_avg = 0;
  for (int which = 0; which < 2; ++which) {
    const double* xv = which ? A.avgx : x;
    const double* yv = which ? A.avgy : y;
    double* sc       = A.sc + 32 * which;
    {
      EvalPrimalEpilogue e{yv, A.dr, A.lo_u, A.hi_u, A.eps_p, A.want_linf ? A.linf_m : nullptr, which ? A.ax_avg : A.ax_cur};
      double acc[3] = {0.0, 0.0, 0.0};
      small_rows(A.m, A.a_off, A.a_idx, A.a_val, xv, prod, e, acc);
      block_sum_fast<3, kMajorThreads / 64>(acc, red);
      if (t == 0) sc[0] = acc[0], sc[1] = acc[1], sc[2] = acc[2];
      __syncthreads();
      if (A.want_linf) {
        double mx[1] = {0.0};
        for (int i = t; i < A.m; i += kMajorThreads) mx[0] = dmax(mx[0], A.linf_m[i]);  // own writes
        block_reduce<MaxOp, 1, kMajorThreads / 64>(mx, red);
        if (t == 0) sc[3] = mx[0];
        __syncthreads();
      }
    }
    {
      EvalDualEpilogue e{EvalDualCore{xv, A.dc, A.c_u, A.lb_u, A.ub_u, A.eps_d, A.rule_finite, which ? A.rc_avg : A.rc_cur,
                                      A.want_linf ? A.linf_n : nullptr, which ? A.aty_avg : A.aty_cur}};
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      small_rows(A.n, A.at_off, A.at_idx, A.at_val, yv, prod, e, acc);
      block_sum_fast<4, kMajorThreads / 64>(acc, red);
      if (t == 0) sc[4] = acc[0], sc[5] = acc[1], sc[6] = acc[2], sc[7] = acc[3];
      __syncthreads();
      if (A.want_linf) {
        double mx[1] = {0.0};
        for (int j = t; j < A.n; j += kMajorThreads) mx[0] = dmax(mx[0], A.linf_n[j]);
        block_reduce<MaxOp, 1, kMajorThreads / 64>(mx, red);
        if (t == 0) sc[8] = mx[0];
        __syncthreads();
      }
    }
  }
}
__global__ void __launch_bounds__(kMajorThreads) k_major_small(MajorSmallArgs A)
{
  extern __shared__ double prod[];
  major_small_body(A, prod);
}
__global__ void __launch_bounds__(kMajorThreads) k_major_small_batch(const MajorSmallArgs* __restrict__ args, const int* __restrict__ list)
{
  extern __shared__ double prod[];
  const MajorSmallArgs& A = args[list[blockIdx.x]];
  if (A.guard_target >= 0 && !(A.ctl->error != 0 || A.ctl->steps_taken >= A.guard_target)) {  // (uniform)
    if (threadIdx.x == 0) A.sc[63] = 0.0;  // "not evaluated"
    return;
  }
  major_small_body(A, prod);
  if (threadIdx.x == 0) A.sc[63] = 1.0;
}

// ---- one LP: the small-LP branches of pdlpdev_run / pdlpdev_major_eval ----------------------------------------------------------------
static SmallView small_view(const pdlpdev_ctx* ctx)
{
  return SmallView{ctx->m, ctx->n, (int)ctx->nnz, ctx->a_off, ctx->a_idx, ctx->at_off, ctx->at_idx, ctx->a_val, ctx->at_val,
                   ctx->c, ctx->lb, ctx->ub, ctx->lo, ctx->hi, ctx->x[0], ctx->x[1], ctx->y[0], ctx->y[1], ctx->aty[0],
                   ctx->aty[1], ctx->sumx, ctx->sumy};
}
static MajorSmallArgs major_args(const pdlpdev_ctx* ctx, int average_mode, int rc_rule_finite_bounds, int want_linf, double eps_rel_primal, double eps_rel_dual)
{
  return MajorSmallArgs{ctx->m, ctx->n, average_mode, rc_rule_finite_bounds, want_linf, eps_rel_primal, eps_rel_dual,
                        ctx->a_off, ctx->a_idx, ctx->at_off, ctx->at_idx, ctx->a_val, ctx->at_val, ctx->ctl,
                        ctx->x[0], ctx->x[1], ctx->y[0], ctx->y[1], ctx->sumx, ctx->sumy, ctx->avgx, ctx->avgy,
                        ctx->dr, ctx->dc, ctx->c_u, ctx->lb_u, ctx->ub_u, ctx->lo_u, ctx->hi_u, ctx->tmp_m, ctx->tmp_n,
                        ctx->ax_u[PDLPDEV_CURRENT], ctx->ax_u[PDLPDEV_AVERAGE], ctx->aty_u[PDLPDEV_CURRENT],
                        ctx->aty_u[PDLPDEV_AVERAGE], ctx->rc[0], ctx->rc[1], ctx->scal_h};
}
static int major_lds_attribute(int device)
{
  static PerDeviceOnce once;
  return once.run(device, [&]() -> int {
    HIP_TRY(hipFuncSetAttribute((const void*)k_major_small, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
    HIP_TRY(hipFuncSetAttribute((const void*)k_major_small_batch, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
    return 0;
  });
}
// one launch runs attempts until the target is reached (rejected attempts included); the cap only bounds a pathological rejection
// streak, in which case the loop relaunches.  The kernel takes the target as an argument and leaves the control block in pinned host
// memory: one launch + one synchronize per call.
int resident_run(pdlpdev_ctx* ctx, int32_t target_steps)
{
  const SmallView V = small_view(ctx);
  const int tier    = resident_tier(ctx->m, ctx->n, ctx->nnz);
  for (int guard = 0; guard < 1000; ++guard) {
    if (tier == 0) TRY((launch_resident<256, 2, 8>(ctx->stream, tier, V, ctx->ctl, ctx->ctl_h, ctx->sp, target_steps)));
    if (tier == 1) TRY((launch_resident<512, 2, 16>(ctx->stream, tier, V, ctx->ctl, ctx->ctl_h, ctx->sp, target_steps)));
    if (tier == 2) TRY((launch_resident<512, 4, 8>(ctx->stream, tier, V, ctx->ctl, ctx->ctl_h, ctx->sp, target_steps)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->ctl_h->error != 0 || ctx->ctl_h->steps_taken >= target_steps) break;
  }
  return 0;
}
int resident_major_eval(pdlpdev_ctx* ctx, int average_mode, int rc_rule_finite_bounds, int want_linf, double eps_rel_primal, double eps_rel_dual)
{
  const MajorSmallArgs A = major_args(ctx, average_mode, rc_rule_finite_bounds, want_linf, eps_rel_primal, eps_rel_dual);
  const size_t lds = sizeof(double) * (size_t)std::max<int64_t>(ctx->nnz, 1);
  TRY(major_lds_attribute(ctx->device));
  k_major_small<<<1, kMajorThreads, lds, ctx->stream>>>(A);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ================================================================================================
// K resident LPs at once: one workgroup per LP, one launch per phase of the loop
// ================================================================================================
// The phases of pdlp_solver.cpp's loop that touch the device -- attempts up to the next major iteration, the head of a major
// iteration, a restart, the scalar updates in front of the next attempts, A^T y of a fresh iterate -- each become ONE launch over the
// LPs that are in that phase (lists in pinned host memory), on the batch's own stream, with ONE synchronisation.  Every kernel body
// is the single LP's (resident_body, major_small_body, restart_block + finalize_rows, csr_stream_block), so each LP gets, bit for
// bit, the trajectory of its own solve; LPs of different resident tiers are launched tier by tier.
struct RestartBatchArgs {
  RestartView R;
  int g, pad;
  double* out;  // the LP's pinned scalar block: dist2 at [0], [1]
};
struct CtlOp {
  pdlpdev_ctl* ctl;
  int clear_error, set_weight;
  double weight;
};
__global__ void __launch_bounds__(kBlock) k_restart_batch(const RestartBatchArgs* __restrict__ args, const int2* __restrict__ blk)
{
  __shared__ double red[12];
  const int2 b = blk[blockIdx.x];
  restart_block(args[b.x].R, b.y, args[b.x].g, red);
}
__global__ void __launch_bounds__(kBlock) k_restart_finish_batch(const RestartBatchArgs* __restrict__ args, const int* __restrict__ list)
{
  __shared__ double red[8];
  const RestartBatchArgs& A = args[list[blockIdx.x]];
  finalize_rows(A.R.part, A.g, 2, 0u, A.out, red);
  if (threadIdx.x == 0) {  // k_restart_ctl
    pdlpdev_ctl* ctl       = const_cast<pdlpdev_ctl*>(A.R.ctl);
    ctl->sum_weights       = 0.0;
    ctl->its_since_restart = 0;
    ctl->pending_avg       = 0;
  }
}
__global__ void __launch_bounds__(kBlock) k_ctl_ops_batch(const CtlOp* __restrict__ ops, int n)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  pdlpdev_ctl* ctl = ops[i].ctl;
  if (ops[i].clear_error) ctl->error = 0;  // k_clear_error
  if (ops[i].set_weight) {                 // k_set_step with the step size kept
    const double w     = ops[i].weight;
    ctl->primal_weight = w;
    ctl->tau           = ctl->step_size / w;
    ctl->sigma         = ctl->step_size * w;
  }
}

// The re-solve pattern of the MIP heuristics for K LPs in one launch (relaxed_lp.cu:74-108: other variable bounds, start from the previous
// primal / dual): per LP exactly what pdlpdev_reset(lb, ub) + pdlpdev_set_step + pdlpdev_set_k + pdlpdev_set_initial +
// pdlpdev_project_primal + pdlpdev_get_ctl do one call after the other -- the same expressions, element by element.
struct SmallResetArgs {
  int n, m, k, project;
  int var, warm;          // var >= 0: the bounds of this ONE variable become [var_lb, var_ub] (a branch); warm (PDLPDEV_CURRENT / AVERAGE /
  double var_lb, var_ub;  // BEST, < 0 none): start from that iterate of the LP itself -- unscaled and scaled again like a trip through the host
  const double *bestx, *besty;
  const double *lb_new, *ub_new, *x0, *y0;  // staging (device-visible host memory); NULL: bounds unchanged / start from zero
  double *lb_u, *ub_u, *lb, *ub;
  const double *dc, *dr;
  double *x[2], *aty[2], *rc[2], *y[2], *xbar, *sumx, *avgx, *lrx, *sumy, *avgy, *lry;
  pdlpdev_ctl *ctl, *ctl_host;
  double step, weight;
};
__global__ void __launch_bounds__(512) k_small_reset_batch(const SmallResetArgs* __restrict__ args, const int* __restrict__ list)
{
  const SmallResetArgs& A = args[list[blockIdx.x]];
  const int cur    = A.ctl->cur;  // (read by every thread before thread 0 rewrites the control block: the loops below end in a barrier-free
  __syncthreads();                //  store by thread 0 only after this barrier)
  const double* wx = A.warm == PDLPDEV_BEST ? A.bestx : A.warm == PDLPDEV_AVERAGE ? A.avgx : A.x[cur];
  const double* wy = A.warm == PDLPDEV_BEST ? A.besty : A.warm == PDLPDEV_AVERAGE ? A.avgy : A.y[cur];
  for (int j = threadIdx.x; j < A.n; j += 512) {
    if (A.lb_new) A.lb_u[j] = A.lb_new[j], A.lb[j] = A.lb_new[j] / A.dc[j];  // k_scale_bounds
    if (A.ub_new) A.ub_u[j] = A.ub_new[j], A.ub[j] = A.ub_new[j] / A.dc[j];
    if (j == A.var) {
      A.lb_u[j] = A.var_lb, A.lb[j] = A.var_lb / A.dc[j];
      A.ub_u[j] = A.var_ub, A.ub[j] = A.var_ub / A.dc[j];
    }
    const double lo = A.lb[j], hi = A.ub[j];
    double x = A.x0 ? A.x0[j] / A.dc[j] : 0.0;  // k_div_inplace (set_initial) on a zeroed iterate
    if (A.warm >= 0) x = (wx[j] * A.dc[j]) / A.dc[j];  // k_unscale (what the caller would have read back), then set_initial
    double avg = 0.0;
    if (A.project) x = dmin(dmax(x, lo), hi), avg = dmin(dmax(avg, lo), hi);  // k_clamp on the iterate and on the average
    A.x[0][j] = x, A.x[1][j] = 0.0, A.aty[0][j] = 0.0, A.aty[1][j] = 0.0, A.rc[0][j] = 0.0, A.rc[1][j] = 0.0;
    A.xbar[j] = 0.0, A.sumx[j] = 0.0, A.avgx[j] = avg, A.lrx[j] = 0.0;
  }
  for (int i = threadIdx.x; i < A.m; i += 512) {
    const double y = A.warm >= 0 ? (wy[i] * A.dr[i]) / A.dr[i] : (A.y0 ? A.y0[i] / A.dr[i] : 0.0);
    A.y[0][i] = y, A.y[1][i] = 0.0;
    A.sumy[i] = 0.0, A.avgy[i] = 0.0, A.lry[i] = 0.0;
  }
  if (threadIdx.x == 0) {
    pdlpdev_ctl c;
    memset(&c, 0, sizeof(c));
    c.step_size = A.step, c.primal_weight = A.weight, c.tau = A.step / A.weight, c.sigma = A.step * A.weight;  // k_set_step
    if (A.k >= 0) c.k = A.k;
    *A.ctl = c, *A.ctl_host = c;
  }
}
// pdlpdev_get_solution for K LPs: x = x^ * D_c, y = y^ * D_r (k_unscale), reduced costs as they are -- straight into the staging block
struct SmallSolutionArgs {
  int n, m, which;
  const pdlpdev_ctl* ctl;
  const double *x0, *x1, *y0, *y1, *avgx, *avgy, *bestx, *besty, *bestrc, *rc0, *rc1, *dc, *dr;
  double *out_x, *out_y, *out_rc;  // NULL: not wanted
};
__global__ void __launch_bounds__(512) k_small_solution_batch(const SmallSolutionArgs* __restrict__ args, const int* __restrict__ list)
{
  const SmallSolutionArgs& A = args[list[blockIdx.x]];
  const int cur   = A.ctl->cur;
  const double* x = A.which == PDLPDEV_BEST ? A.bestx : A.which == PDLPDEV_AVERAGE ? A.avgx : (cur ? A.x1 : A.x0);
  const double* y = A.which == PDLPDEV_BEST ? A.besty : A.which == PDLPDEV_AVERAGE ? A.avgy : (cur ? A.y1 : A.y0);
  const double* r = A.which == PDLPDEV_BEST ? A.bestrc : A.which == PDLPDEV_AVERAGE ? A.rc1 : A.rc0;
  for (int j = threadIdx.x; j < A.n; j += 512) {
    if (A.out_x) A.out_x[j] = x[j] * A.dc[j];
    if (A.out_rc) A.out_rc[j] = r[j];
  }
  if (A.out_y)
    for (int i = threadIdx.x; i < A.m; i += 512) A.out_y[i] = y[i] * A.dr[i];
}

struct pdlpdev_small_batch {
  int device = 0, K = 0;
  hipStream_t stream = nullptr;
  std::vector<pdlpdev_ctx*> ctx;
  std::vector<int> tier;
  size_t major_lds = 8;
  int at_blocks_cap = 0;
  void* pinned = nullptr;  // everything below lives in this one block (device-visible host memory)
  ResidentArgs* run_args = nullptr;
  int* run_list = nullptr;  // 3 * K: one list per tier
  MajorSmallArgs* major = nullptr;
  int* list = nullptr;      // K
  RestartBatchArgs* restart = nullptr;
  int2* blk = nullptr;      // restart / A^T y blocks -> (LP, block of the LP)
  StreamAtCurArgs* at = nullptr;
  CtlOp* ops = nullptr;
  SmallResetArgs* reset = nullptr;
  SmallSolutionArgs* sol = nullptr;
  double* staging = nullptr;  // 3 n + m doubles per LP (new bounds + initial iterate in, solutions out), LP l at stage_off[l]
  std::vector<size_t> stage_off;
};

template <int T, int Q, int U>
static int launch_resident_batch(pdlpdev_small_batch* b, int tier, const int* list, int count)
{
  static PerDeviceOnce once;
  TRY(once.run(b->device, [&]() -> int {
    HIP_TRY(hipFuncSetAttribute((const void*)k_pdhg_resident_batch<T, Q, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)resident_lds_bytes(tier)));
    return 0;
  }));
  k_pdhg_resident_batch<T, Q, U><<<count, T, resident_lds_bytes(tier), b->stream>>>(b->run_args, list, 1 << 14);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" {

int pdlpdev_small_batch_create(pdlpdev_small_batch** out, pdlpdev_ctx** ctx, int K)
{
  if (!out || !ctx || K < 1) return fail(-1, "pdlpdev_small_batch_create: null argument");
  *out = nullptr;
  for (int l = 0; l < K; ++l) {
    const pdlpdev_ctx* c = ctx[l];
    if (!c) return fail(-1, "pdlpdev_small_batch_create: null context");
    if (!c->small_resident || c->comm || c->pat.on || c->jat.on || c->pbat.on || c->dense.on)
      return fail(-7, "pdlpdev_small_batch_create: LP %d is not on the resident small-LP path", l);
    if (c->device != ctx[0]->device) return fail(-7, "pdlpdev_small_batch_create: the LPs sit on different devices");
    for (int q = 0; q < l; ++q)
      if (ctx[q] == c) return fail(-1, "pdlpdev_small_batch_create: LP %d and LP %d are the same context", q, l);
  }
  std::unique_ptr<pdlpdev_small_batch> b(new pdlpdev_small_batch());
  b->device = ctx[0]->device, b->K = K;
  b->ctx.assign(ctx, ctx + K);
  HIP_TRY(hipSetDevice(b->device));
  int blocks = 0;
  for (int l = 0; l < K; ++l) {
    b->tier.push_back(resident_tier(ctx[l]->m, ctx[l]->n, ctx[l]->nnz));
    b->major_lds = std::max(b->major_lds, sizeof(double) * (size_t)std::max<int64_t>(ctx[l]->nnz, 1));
    blocks += std::max(ctx[l]->at_nb, std::min(grid_for(std::max(ctx[l]->n, ctx[l]->m)), kGenericBlocks));
  }
  b->at_blocks_cap = blocks;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_run = 0, o_list3 = o_run + up(sizeof(ResidentArgs) * K), o_major = o_list3 + up(sizeof(int) * 3 * K),
               o_list = o_major + up(sizeof(MajorSmallArgs) * K), o_restart = o_list + up(sizeof(int) * K),
               o_blk = o_restart + up(sizeof(RestartBatchArgs) * K), o_at = o_blk + up(sizeof(int2) * (size_t)blocks),
               o_ops = o_at + up(sizeof(StreamAtCurArgs) * K), o_reset = o_ops + up(sizeof(CtlOp) * K), o_sol = o_reset + up(sizeof(SmallResetArgs) * K),
               o_stage = o_sol + up(sizeof(SmallSolutionArgs) * K);
  size_t total = o_stage;
  for (int l = 0; l < K; ++l) {
    b->stage_off.push_back((total - o_stage) / sizeof(double));
    total += up(sizeof(double) * (3 * (size_t)ctx[l]->n + (size_t)ctx[l]->m));
  }
  HIP_TRY(hipHostMalloc(&b->pinned, total, hipHostMallocDefault));
  char* base  = (char*)b->pinned;
  b->run_args = (ResidentArgs*)(base + o_run), b->run_list = (int*)(base + o_list3), b->major = (MajorSmallArgs*)(base + o_major);
  b->list = (int*)(base + o_list), b->restart = (RestartBatchArgs*)(base + o_restart), b->blk = (int2*)(base + o_blk);
  b->at = (StreamAtCurArgs*)(base + o_at), b->ops = (CtlOp*)(base + o_ops);
  b->reset = (SmallResetArgs*)(base + o_reset), b->sol = (SmallSolutionArgs*)(base + o_sol), b->staging = (double*)(base + o_stage);
  HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  TRY(major_lds_attribute(b->device));
  for (int l = 0; l < K; ++l) HIP_TRY(hipStreamSynchronize(ctx[l]->stream));  // whatever the set-ups left in flight
  for (int l = 0; l < K; ++l) ctx[l]->batches_alive += 1;
  *out = b.release();
  return 0;
}

void pdlpdev_small_batch_destroy(pdlpdev_small_batch* b)
{
  if (!b) return;
  for (pdlpdev_ctx* c : b->ctx) c->batches_alive -= 1;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream), (void)hipStreamDestroy(b->stream);
  if (b->pinned) (void)hipHostFree(b->pinned);
  delete b;
}

// attempts until LP l holds targets[l] accepted steps (<= 0: LP l rests); ctl[l] receives its control block.
// eval_after (optional): the head of the major iteration that follows the attempts is enqueued right behind them for every LP with
// eval_after[l].mode >= 0 -- one synchronisation for both; the evaluation runs only if LP l's attempts reached their target (or raised
// the step-size error), evaluated[l] says whether it did, out_current / out_average then hold its results.
int pdlpdev_small_batch_run(pdlpdev_small_batch* b, const int32_t* targets, pdlpdev_ctl* ctl, const pdlpdev_small_eval* eval_after, double* out_current,
                            double* out_average, int32_t* evaluated)
{
  roctx::Range range("pdlp: PDHG attempts (small-LP batch)");
  HIP_TRY(hipSetDevice(b->device));
  const int K = b->K;
  std::vector<char> todo(K, 0);
  for (int l = 0; l < K; ++l) {
    if (evaluated) evaluated[l] = 0;
    if (targets[l] > 0) {
      pdlpdev_ctx* c = b->ctx[l];
      b->run_args[l] = ResidentArgs{small_view(c), c->ctl, c->ctl_h, c->sp, targets[l], 0};
      todo[l]        = 1;
    }
  }
  for (int guard = 0; guard < 1000; ++guard) {
    int count[3] = {0, 0, 0};
    for (int l = 0; l < K; ++l)
      if (todo[l]) b->run_list[b->tier[l] * K + count[b->tier[l]]++] = l;
    if (count[0] + count[1] + count[2] == 0) break;
    if (count[0]) TRY((launch_resident_batch<256, 2, 8>(b, 0, b->run_list, count[0])));
    if (count[1]) TRY((launch_resident_batch<512, 2, 16>(b, 1, b->run_list + K, count[1])));
    if (count[2]) TRY((launch_resident_batch<512, 4, 8>(b, 2, b->run_list + 2 * K, count[2])));
    int nev = 0;
    if (guard == 0 && eval_after) {
      for (int l = 0; l < K; ++l)
        if (todo[l] && eval_after[l].mode >= 0) {
          const int want_linf = eval_after[l].eps_p >= 0.0 && eval_after[l].eps_d >= 0.0;
          b->major[l] = major_args(b->ctx[l], eval_after[l].mode, eval_after[l].rule_finite, want_linf, eval_after[l].eps_p, eval_after[l].eps_d);
          b->major[l].guard_target = targets[l];
          b->list[nev++]           = l;
        }
      if (nev) {
        k_major_small_batch<<<nev, kMajorThreads, b->major_lds, b->stream>>>(b->major, b->list);
        HIP_TRY(hipGetLastError());
      }
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (int q = 0; q < nev; ++q) {
      const int l = b->list[q];
      if (b->ctx[l]->scal_h[63] != 1.0) continue;
      const bool want_linf = eval_after[l].eps_p >= 0.0 && eval_after[l].eps_d >= 0.0;
      read_eval(b->ctx[l]->scal_h, want_linf, out_current + (size_t)l * PDLPDEV_EV_COUNT);
      read_eval(b->ctx[l]->scal_h + 32, want_linf, out_average + (size_t)l * PDLPDEV_EV_COUNT);
      if (evaluated) evaluated[l] = 1;
    }
    for (int l = 0; l < K; ++l)
      if (todo[l] && (b->ctx[l]->ctl_h->error != 0 || b->ctx[l]->ctl_h->steps_taken >= targets[l])) todo[l] = 0;
  }
  if (ctl)
    for (int l = 0; l < K; ++l)
      if (targets[l] > 0) ctl[l] = *b->ctx[l]->ctl_h;
  return 0;
}

// pdlpdev_major_eval for every LP with req[l].mode >= 0: out_current / out_average hold PDLPDEV_EV_COUNT doubles per LP
int pdlpdev_small_batch_major_eval(pdlpdev_small_batch* b, const pdlpdev_small_eval* req, double* out_current, double* out_average)
{
  roctx::Range range("pdlp: major iteration evaluation (small-LP batch)");
  HIP_TRY(hipSetDevice(b->device));
  int count = 0;
  for (int l = 0; l < b->K; ++l)
    if (req[l].mode >= 0) {
      const int want_linf = req[l].eps_p >= 0.0 && req[l].eps_d >= 0.0;
      b->major[l]         = major_args(b->ctx[l], req[l].mode, req[l].rule_finite, want_linf, req[l].eps_p, req[l].eps_d);
      b->list[count++]    = l;
    }
  if (!count) return 0;
  k_major_small_batch<<<count, kMajorThreads, b->major_lds, b->stream>>>(b->major, b->list);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  for (int l = 0; l < b->K; ++l)
    if (req[l].mode >= 0) {
      const bool want_linf = req[l].eps_p >= 0.0 && req[l].eps_d >= 0.0;
      read_eval(b->ctx[l]->scal_h, want_linf, out_current + (size_t)l * PDLPDEV_EV_COUNT);
      read_eval(b->ctx[l]->scal_h + 32, want_linf, out_average + (size_t)l * PDLPDEV_EV_COUNT);
    }
  return 0;
}

// pdlpdev_restart for every LP with which[l] >= 0 (PDLPDEV_CURRENT / PDLPDEV_AVERAGE); dist2[2 l], dist2[2 l + 1]
int pdlpdev_small_batch_restart(pdlpdev_small_batch* b, const int32_t* which, const int32_t* unscaled_distances, double* dist2)
{
  HIP_TRY(hipSetDevice(b->device));
  int count = 0, blocks = 0;
  for (int l = 0; l < b->K; ++l)
    if (which[l] >= 0) {
      pdlpdev_ctx* c = b->ctx[l];
      const int g    = std::min(grid_for(std::max(c->n, c->m)), kGenericBlocks);
      b->restart[l]  = RestartBatchArgs{RestartView{c->n, c->m, which[l], unscaled_distances ? unscaled_distances[l] : 0, c->dc, c->dr, c->ctl, c->x[0], c->x[1], c->y[0],
                                                   c->y[1], c->avgx, c->avgy, c->lrx, c->lry, c->sumx, c->sumy, c->part_g},
                                       g, 0, c->scal_h};
      for (int q = 0; q < g; ++q) b->blk[blocks++] = make_int2(l, q);
      b->list[count++] = l;
    }
  if (!count) return 0;
  k_restart_batch<<<blocks, kBlock, 0, b->stream>>>(b->restart, b->blk);
  k_restart_finish_batch<<<count, kBlock, 0, b->stream>>>(b->restart, b->list);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  for (int l = 0; l < b->K; ++l)
    if (which[l] >= 0) dist2[2 * l] = b->ctx[l]->scal_h[0], dist2[2 * l + 1] = b->ctx[l]->scal_h[1];
  return 0;
}

// in front of the next attempts: pdlpdev_clear_error where clear_error[l], pdlpdev_set_step(-1, primal_weight[l]) where
// primal_weight[l] > 0, pdlpdev_compute_aty where compute_aty[l] (any array may be NULL)
int pdlpdev_small_batch_prepare(pdlpdev_small_batch* b, const int32_t* clear_error, const double* primal_weight, const int32_t* compute_aty)
{
  HIP_TRY(hipSetDevice(b->device));
  int nops = 0, blocks = 0, nat = 0;
  for (int l = 0; l < b->K; ++l) {
    const bool ce = clear_error && clear_error[l], sw = primal_weight && primal_weight[l] > 0.0;
    if (ce || sw) b->ops[nops++] = CtlOp{b->ctx[l]->ctl, ce ? 1 : 0, sw ? 1 : 0, sw ? primal_weight[l] : 0.0};
    if (compute_aty && compute_aty[l]) {
      pdlpdev_ctx* c = b->ctx[l];
      b->at[l]       = StreamAtCurArgs{c->at_nb, c->at_rb, c->hat_off, c->hat_idx, c->hat_val, c->ctl, c->y[0], c->y[1], c->aty[0], c->aty[1]};
      for (int q = 0; q < c->at_nb; ++q) b->blk[blocks++] = make_int2(l, q);
      ++nat;
    }
  }
  if (nops) k_ctl_ops_batch<<<(nops + kBlock - 1) / kBlock, kBlock, 0, b->stream>>>(b->ops, nops);
  if (nat) TRY(launch_stream_at_cur_batch(b->stream, b->at, b->blk, blocks));
  if (nops || nat) {
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return 0;
}

// the re-solve pattern (relaxed_lp.cu:74-108) for every LP with take[l] != 0: new variable bounds (lb[l] / ub[l]: n doubles each, NULL =
// unchanged), the state of a freshly scaled problem, step size / primal weight / k (< 0: 0) as given, the initial iterate x0[l] / y0[l]
// (UNSCALED; NULL = zeros), the projection of the initial primal (project != 0); ctl[l] receives the control block.  What
// pdlpdev_reset(lb, ub, NULL, NULL), pdlpdev_set_step, pdlpdev_set_k, pdlpdev_set_initial, pdlpdev_project_primal and pdlpdev_get_ctl
// do for one LP, bit for bit, in ONE launch.
int pdlpdev_small_batch_reset(pdlpdev_small_batch* b, const int32_t* take, const double* const* lb, const double* const* ub, const double* const* x0,
                              const double* const* y0, const double* step, const double* weight, const int32_t* k, int project, pdlpdev_ctl* ctl,
                              const int32_t* var, const double* var_lb, const double* var_ub, const int32_t* warm)
{
  HIP_TRY(hipSetDevice(b->device));
  int count = 0;
  for (int l = 0; l < b->K; ++l) {
    if (!take[l]) continue;
    pdlpdev_ctx* c = b->ctx[l];
    if (!c->scaled) return fail(-1, "pdlpdev_small_batch_reset: LP %d has not been scaled yet", l);
    if (c->rows_aliased || c->clones_alive > 0) return fail(-7, "pdlpdev_small_batch_reset: LP %d shares arrays with clones", l);
    c->rejected_in_a_row = 0;
    const size_t n = (size_t)c->n, m = (size_t)c->m;
    double* st     = b->staging + b->stage_off[l];
    const double *slb = nullptr, *sub = nullptr, *sx = nullptr, *sy = nullptr;
    if (lb && lb[l]) memcpy(st, lb[l], n * sizeof(double)), slb = st;
    if (ub && ub[l]) memcpy(st + n, ub[l], n * sizeof(double)), sub = st + n;
    if (x0 && x0[l]) memcpy(st + 2 * n, x0[l], n * sizeof(double)), sx = st + 2 * n;
    if (y0 && y0[l]) memcpy(st + 3 * n, y0[l], m * sizeof(double)), sy = st + 3 * n;
    c->note_uniform_bounds(lb ? lb[l] : nullptr, ub ? ub[l] : nullptr);
    const int v = var ? var[l] : -1, wm = warm ? warm[l] : -1;
    if (v >= c->n) return fail(-1, "pdlpdev_small_batch_reset: LP %d has no variable %d", l, v);
    if (v >= 0) c->ubd.lb_same = 0, c->ubd.ub_same = 0;  // (one bound differs from the others now, or may)
    if (wm == PDLPDEV_BEST && !c->bestx) return fail(-1, "pdlpdev_small_batch_reset: LP %d saved no best iterate to start from", l);
    b->reset[l] = SmallResetArgs{c->n, c->m, k ? k[l] : -1, project, v, wm, v >= 0 ? var_lb[l] : 0.0, v >= 0 ? var_ub[l] : 0.0, c->bestx, c->besty,
                                 slb, sub, sx, sy, c->lb_u, c->ub_u, c->lb, c->ub, c->dc, c->dr,
                                 {c->x[0], c->x[1]}, {c->aty[0], c->aty[1]}, {c->rc[0], c->rc[1]}, {c->y[0], c->y[1]}, c->xbar, c->sumx, c->avgx, c->lrx,
                                 c->sumy, c->avgy, c->lry, c->ctl, c->ctl_h, step[l], weight[l]};
    b->list[count++] = l;
  }
  if (!count) return 0;
  k_small_reset_batch<<<count, 512, 0, b->stream>>>(b->reset, b->list);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (ctl)
    for (int l = 0; l < b->K; ++l)
      if (take[l]) ctl[l] = *b->ctx[l]->ctl_h;
  return 0;
}

// the same without the copies: x_view[l] / y_view[l] / rc_view[l] point INTO the batch's pinned staging block (valid until the next
// pdlpdev_small_batch_reset / _get_solutions / _solution_views call of this batch)
int pdlpdev_small_batch_solution_views(pdlpdev_small_batch* b, const int32_t* which, const double** x_view, const double** y_view, const double** rc_view)
{
  HIP_TRY(hipSetDevice(b->device));
  int count = 0;
  for (int l = 0; l < b->K; ++l) {
    if (x_view) x_view[l] = nullptr;
    if (y_view) y_view[l] = nullptr;
    if (rc_view) rc_view[l] = nullptr;
    if (which[l] < 0) continue;
    pdlpdev_ctx* c = b->ctx[l];
    if (which[l] == PDLPDEV_BEST && !c->bestx) return fail(-1, "pdlpdev_small_batch_solution_views: LP %d saved no best iterate", l);
    const size_t n = (size_t)c->n;
    double* st     = b->staging + b->stage_off[l];
    b->sol[l] = SmallSolutionArgs{c->n, c->m, which[l], c->ctl, c->x[0], c->x[1], c->y[0], c->y[1], c->avgx, c->avgy, c->bestx, c->besty, c->bestrc,
                                  c->rc[0], c->rc[1], c->dc, c->dr, x_view ? st : nullptr, y_view ? st + 2 * n : nullptr, rc_view ? st + n : nullptr};
    if (x_view) x_view[l] = st;
    if (rc_view) rc_view[l] = st + n;
    if (y_view) y_view[l] = st + 2 * n;
    b->list[count++] = l;
  }
  if (!count) return 0;
  k_small_solution_batch<<<count, 512, 0, b->stream>>>(b->sol, b->list);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

// pdlpdev_get_solution(which[l]) for every LP with which[l] >= 0: UNSCALED x[l] (n), y[l] (m), reduced costs rc[l] (n); any array or
// entry may be NULL.  One launch, one synchronisation; the copies to the caller's arrays are plain host copies.
int pdlpdev_small_batch_get_solutions(pdlpdev_small_batch* b, const int32_t* which, double* const* x, double* const* y, double* const* rc)
{
  HIP_TRY(hipSetDevice(b->device));
  int count = 0;
  for (int l = 0; l < b->K; ++l) {
    if (which[l] < 0) continue;
    pdlpdev_ctx* c = b->ctx[l];
    if (which[l] == PDLPDEV_BEST && !c->bestx) return fail(-1, "pdlpdev_small_batch_get_solutions: LP %d saved no best iterate", l);
    const size_t n = (size_t)c->n;
    double* st     = b->staging + b->stage_off[l];
    b->sol[l] = SmallSolutionArgs{c->n, c->m, which[l], c->ctl, c->x[0], c->x[1], c->y[0], c->y[1], c->avgx, c->avgy, c->bestx, c->besty, c->bestrc,
                                  c->rc[0], c->rc[1], c->dc, c->dr, (x && x[l]) ? st : nullptr, (y && y[l]) ? st + 2 * n : nullptr, (rc && rc[l]) ? st + n : nullptr};
    b->list[count++] = l;
  }
  if (!count) return 0;
  k_small_solution_batch<<<count, 512, 0, b->stream>>>(b->sol, b->list);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  for (int l = 0; l < b->K; ++l) {
    if (which[l] < 0) continue;
    const size_t n = (size_t)b->ctx[l]->n, m = (size_t)b->ctx[l]->m;
    const double* st = b->staging + b->stage_off[l];
    if (x && x[l]) memcpy(x[l], st, n * sizeof(double));
    if (rc && rc[l]) memcpy(rc[l], st + n, n * sizeof(double));
    if (y && y[l]) memcpy(y[l], st + 2 * n, m * sizeof(double));
  }
  return 0;
}

}  // extern "C"
