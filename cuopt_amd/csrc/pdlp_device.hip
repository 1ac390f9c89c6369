#include "pdlp_ctx.hpp"
#include "pdlp_layouts.hpp"
#include "pdlp_setup.hpp"


// ================================================================================================
// kernels: element-wise helpers of the set-up (the scaling kernels live in pdlp_scaling.hip)
// ================================================================================================
__global__ void __launch_bounds__(kBlock) k_fill(int64_t n, double* __restrict__ d, double v)
{
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    d[i] = v;
}
__global__ void __launch_bounds__(kBlock) k_scale_vectors(int n, int m, double* __restrict__ c,
                                                          double* __restrict__ lb,
                                                          double* __restrict__ ub,
                                                          const double* __restrict__ dc,
                                                          double* __restrict__ lo,
                                                          double* __restrict__ hi,
                                                          const double* __restrict__ dr)
{
  const int tot = n > m ? n : m;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < tot; i += gridDim.x * kBlock) {
    if (i < n) {
      c[i]  = c[i] * dc[i];
      lb[i] = lb[i] / dc[i];
      ub[i] = ub[i] / dc[i];
    }
    if (i < m) {
      lo[i] = lo[i] * dr[i];
      hi[i] = hi[i] * dr[i];
    }
  }
}
// the bound part of k_scale_vectors alone (pdlpdev_reset): same expressions, so a re-solve with new bounds is
// bit-identical to a fresh solver; NULL = that vector is unchanged
__global__ void __launch_bounds__(kBlock) k_scale_bounds(int n, int m, double* lb, double* ub,
                                                         const double* __restrict__ dc, double* lo, double* hi,
                                                         const double* __restrict__ dr)
{
  const int tot = n > m ? n : m;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < tot; i += gridDim.x * kBlock) {
    if (i < n) {
      if (lb) lb[i] = lb[i] / dc[i];
      if (ub) ub[i] = ub[i] / dc[i];
    }
    if (i < m) {
      if (lo) lo[i] = lo[i] * dr[i];
      if (hi) hi[i] = hi[i] * dr[i];
    }
  }
}
__global__ void __launch_bounds__(kBlock) k_div_inplace(int n, double* __restrict__ v,
                                                        const double* __restrict__ d)
{
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) v[i] = v[i] / d[i];
}
__global__ void __launch_bounds__(kBlock) k_div_to(int n, double* __restrict__ out, const double* __restrict__ v,
                                                   const double* __restrict__ d)
{
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = v[i] / d[i];
}
__global__ void __launch_bounds__(kBlock) k_clamp(int n, double* __restrict__ x,
                                                  const double* __restrict__ lb,
                                                  const double* __restrict__ ub)
{
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    x[i] = dmin(dmax(x[i], lb[i]), ub[i]);  // clamp, utils.cuh:131-137
}

// generic grid-stride reductions -> part[q * gridDim + block]
// mode 0: max |v| ; 1: sum v^2 ; 2: sum combine_bounds(a,b)^2 ; 3: sum (a-b)^2
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_reduce(int64_t n, const double* __restrict__ a,
                                                   const double* __restrict__ b,
                                                   double* __restrict__ part)
{
  __shared__ double red[8];
  double acc[1] = {0.0};
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (MODE == 0) {
      const double v = fabs(a[i]);
      acc[0] = v > acc[0] ? v : acc[0];
    } else if (MODE == 1) {
      acc[0] += a[i] * a[i];
    } else if (MODE == 2) {
      const double v = combine_bounds(a[i], b[i]);
      acc[0] += v * v;
    } else {
      const double v = a[i] - b[i];
      acc[0] += v * v;
    }
  }
  if (MODE == 0)
    block_reduce<MaxOp, 1>(acc, red);
  else
    block_reduce<SumOp, 1>(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}
__global__ void __launch_bounds__(kBlock) k_dot(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                double* __restrict__ part)
{
  __shared__ double red[8];
  double acc[1] = {0.0};
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) acc[0] += a[i] * b[i];
  block_reduce<SumOp, 1>(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}
// out[q] = reduce(part[q*nb .. q*nb+nb)) ; op_mask bit q set => max
__global__ void __launch_bounds__(kBlock) k_finalize(const double* __restrict__ part, int nb, int nq,
                                                     unsigned op_mask, double* __restrict__ out)
{
  __shared__ double red[8];
  finalize_rows(part, nb, nq, op_mask, out, red);
}

// ================================================================================================
// kernels: the PDHG attempt (4 launches, no host interaction)
// ================================================================================================
// (1) primal projection + extrapolation (primal_projection functor, utils.cuh:80-95) fused with the
//     deferred averaging of the previously accepted iterate (weighted_average_solution.cu:73-108).
__global__ void __launch_bounds__(kBlock)
k_primal(int n, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ x0, double* __restrict__ x1,
         const double* __restrict__ aty0, const double* __restrict__ aty1,
         const double* __restrict__ c, const double* __restrict__ lb, const double* __restrict__ ub,
         double* __restrict__ xbar, double* __restrict__ sumx, const p2pdev::Push* __restrict__ push, pdlpdev_ctx::UniformBounds ubd)
{
  // (ubd: every lower / upper bound is the same 0 or infinity -- x >= 0 is the usual LP -- so the bound arrays, 16 of the kernel's
  //  72 bytes per column, are not read at all; scaling keeps 0 and infinity what they are)
  if (!loop_active(ctl)) return;
  const int cur       = ctl->cur;
  const double tau    = ctl->tau;
  const double weight = ctl->step_size;
  const bool pend     = ctl->pending_avg != 0;
  const double* __restrict__ x   = cur ? x1 : x0;
  double* __restrict__ xn        = cur ? x0 : x1;
  const double* __restrict__ aty = cur ? aty1 : aty0;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    const double xj       = x[j];
    const double gradient = c[j] - aty[j];
    double next           = xj - (tau * gradient);
    next                  = dmax(dmin(next, ubd.ub_same ? ubd.ub : ub[j]), ubd.lb_same ? ubd.lb : lb[j]);
    xn[j]                 = next;
    const double xb       = next - xj + next;
    xbar[j]               = xb;
    if (push)  // sharded solve, direct peer transport: this rank's slice of xbar lands in every OTHER rank's block (its own copy is
               // the ordinary store above: write-through stores cost 17 us per attempt at C3 and buy nothing at home)
      for (int q = 0; q < push->world; ++q)
        if (push->wants(q, j)) p2pdev::put(push->slot(q) + j, xb);
    if (pend) sumx[j] = sumx[j] + weight * xj;
  }
  if (push) p2pdev::count_exchange(push);
}



// multi-GPU variant of (3): after the all-reduce of the A^T y' partial products
__global__ void __launch_bounds__(kBlock)
k_step_stats(int n, int nbg, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ reduced,
             const double* __restrict__ x0, const double* __restrict__ x1, double* __restrict__ aty0,
             double* __restrict__ aty1, double* __restrict__ part)
{
  __shared__ double red[12];
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  const double* __restrict__ x   = cur ? x1 : x0;
  const double* __restrict__ xn  = cur ? x0 : x1;
  const double* __restrict__ aty = cur ? aty1 : aty0;
  double* __restrict__ atyn      = cur ? aty0 : aty1;
  double acc[2] = {0.0, 0.0};
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    const double v  = reduced[j];
    atyn[j]         = v;
    const double dx = xn[j] - x[j];
    const double t  = v - aty[j];
    acc[0] += t * dx;
    acc[1] += dx * dx;
  }
  block_reduce<SumOp, 2>(acc, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x]       = acc[0];
    part[nbg + blockIdx.x] = acc[1];
  }
}

constexpr int kDecisionThreads = 1024;  // one wide workgroup: every partial is one independent load
// Latency is all that matters here (one workgroup on the critical path of every attempt): the control block is
// read once into registers (uniform -> scalar loads) and written back once, the two pow() of the step-size rule
// are evaluated by the last wave while the others fetch partials, the sums use DPP lane permutes.
__device__ __forceinline__ void step_decision_workgroup(pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ part_dy, int nb_dy,
                                                        const double* __restrict__ part_t, int nb_t, const double* __restrict__ dy2_reduced,
                                                        const pdlpdev_step_params& sp)
{
  __shared__ double red[3 * 16];
  __shared__ double pw[2];
  pdlpdev_ctl lc = *ctl;
  if (!(lc.error == 0 && lc.steps_taken < lc.target_steps)) return;
  const int t = threadIdx.x;
  if (t >= kDecisionThreads - 2) {
    const double knext = (double)(lc.k + 1) + 1.0;
    pw[t - (kDecisionThreads - 2)] = pow(knext, t == kDecisionThreads - 2 ? -sp.reduction_exponent : -sp.growth_exponent);
  }
  double acc[3] = {0.0, 0.0, 0.0};
  if (dy2_reduced == nullptr) {
#pragma unroll 4
    for (int i = t; i < nb_dy; i += kDecisionThreads) acc[0] += part_dy[i];
  }
#pragma unroll 4
  for (int i = t; i < nb_t; i += kDecisionThreads) {
    acc[1] += part_t[i];
    acc[2] += part_t[nb_t + i];
  }
  block_sum_fast<3, kDecisionThreads / 64>(acc, red);
  if (t != 0) return;
  apply_step_decision(&lc, dy2_reduced ? dy2_reduced[0] : acc[0], acc[1], acc[2], sp, pw);
  *ctl = lc;
}
__global__ void __launch_bounds__(kDecisionThreads)
k_step_decision(pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ part_dy, int nb_dy,
                const double* __restrict__ part_t, int nb_t, const double* __restrict__ dy2_reduced,
                pdlpdev_step_params sp)
{
  step_decision_workgroup(ctl, part_dy, nb_dy, part_t, nb_t, dy2_reduced, sp);
}
// the decisions of the K LPs of a shared-matrix batch (kernels_batch.hip): workgroup <-> LP, each exactly k_step_decision
__global__ void __launch_bounds__(kDecisionThreads)
k_step_decision_batch(const pdlpdev_decision_args* __restrict__ args)
{
  const pdlpdev_decision_args a = args[blockIdx.x];
  step_decision_workgroup(a.ctl, a.part_dy, a.nb_dy, a.part_t, a.nb_t, nullptr, a.sp);
}

// direct peer transport of a sharded solve: wait for every rank's three step-size sums (landed in this rank's block), add them
// up in rank order -- the same bits on every rank -- and take the decision
// (one workgroup, so the whole scalar exchange lives in this kernel: its own three sums from the partials of the two SpMV kernels,
// write-through stores into every rank's block, the flag, the wait for the others -- no ticket, no kernel boundary in between)
__global__ void __launch_bounds__(kBlock)
k_step_decision_p2p(pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ part_dy, int nb_dy, const double* __restrict__ part_t, int nb_t,
                    const double* __restrict__ landed, const unsigned long long* __restrict__ flags, int world,
                    const unsigned long long* __restrict__ epoch, int* __restrict__ fault, pdlpdev_step_params sp, const p2pdev::Push* __restrict__ push)
{
  if (!loop_active(ctl)) return;
  {
    __shared__ double red[3 * 8];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nb_dy; i += kBlock) acc[0] += part_dy[i];
    for (int i = threadIdx.x; i < nb_t; i += kBlock) {
      acc[1] += part_t[i];
      acc[2] += part_t[nb_t + i];
    }
    block_reduce<SumOp, 3>(acc, red);
    if (threadIdx.x == 0) {
      for (int q = 0; q < push->world; ++q) {
        double* d = push->slot(q);
        p2pdev::put(d, acc[0]), p2pdev::put(d + 1, acc[1]), p2pdev::put(d + 2, acc[2]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      push->epoch[push->kind] = push->epoch[push->kind] + 1;
      p2pdev::raise(push);
    }
    __syncthreads();
  }
  if (!p2pdev::wait_flags(flags, world, 2, epoch)) {
    if (threadIdx.x == 0) *fault = 1, ctl->error = 1;
    return;
  }
  if (threadIdx.x != 0) return;
  double sum[3] = {0.0, 0.0, 0.0};
  for (int q = 0; q < world; ++q)
    for (int k = 0; k < 3; ++k) sum[k] += __hip_atomic_load(landed + 4 * q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  pdlpdev_ctl lc = *ctl;
  apply_step_decision(&lc, sum[0], sum[1], sum[2], sp);
  *ctl = lc;
}

__global__ void __launch_bounds__(kBlock)
k_permute_pad(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ src, double* __restrict__ dst)
{
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int32_t k = perm[i];
    dst[i]          = k >= 0 ? src[k] : 0.0;
  }
}
__global__ void __launch_bounds__(kBlock)
k_permute(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ src, double* __restrict__ dst)
{
  for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    dst[i] = src[perm[i]];
}

// deferred averaging made explicit (called before a major iteration consumes the sums)
__global__ void __launch_bounds__(kBlock)
k_flush_average(int n, int m, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0,
                const double* __restrict__ x1, const double* __restrict__ y0,
                const double* __restrict__ y1, double* __restrict__ sumx, double* __restrict__ sumy)
{
  if (ctl->pending_avg == 0) return;
  const int cur           = ctl->cur;
  const double w          = ctl->step_size;
  const double* __restrict__ x = cur ? x1 : x0;
  const double* __restrict__ y = cur ? y1 : y0;
  const int tot = n > m ? n : m;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < tot; i += gridDim.x * kBlock) {
    if (i < n) sumx[i] = sumx[i] + w * x[i];
    if (i < m) sumy[i] = sumy[i] + w * y[i];
  }
}
__global__ void k_clear_pending(pdlpdev_ctl* ctl) { ctl->pending_avg = 0; }

__global__ void __launch_bounds__(kBlock)
k_sum_partials_to(const double* __restrict__ part, int nb, double* __restrict__ out)
{
  __shared__ double red[8];
  double acc[1] = {0.0};
  for (int i = threadIdx.x; i < nb; i += kBlock) acc[0] += part[i];
  block_reduce<SumOp, 1>(acc, red);
  if (threadIdx.x == 0) out[0] = acc[0];
}

// sliced-primal dataflow: this rank's three step-size partial sums, side by side, for ONE small all-reduce
__global__ void __launch_bounds__(kBlock)
k_pack_step_sums(const double* __restrict__ part_dy, int nb_dy, const double* __restrict__ part_t, int nb_t, double* __restrict__ out)
{
  __shared__ double red[3 * 8];
  double acc[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nb_dy; i += kBlock) acc[0] += part_dy[i];
  for (int i = threadIdx.x; i < nb_t; i += kBlock) {
    acc[1] += part_t[i];
    acc[2] += part_t[nb_t + i];
  }
  block_reduce<SumOp, 3>(acc, red);
  if (threadIdx.x == 0) {
    out[0] = acc[0], out[1] = acc[1], out[2] = acc[2];
  }
}

__global__ void k_restart_ctl(pdlpdev_ctl* ctl)
{
  ctl->sum_weights       = 0.0;
  ctl->its_since_restart = 0;
  ctl->pending_avg       = 0;
}
__global__ void k_set_step(pdlpdev_ctl* ctl, double step, double w)
{
  if (step >= 0.0) ctl->step_size = step;
  ctl->primal_weight = w;
  ctl->tau           = ctl->step_size / w;
  ctl->sigma         = ctl->step_size * w;
}
__global__ void k_set_target(pdlpdev_ctl* ctl, int target) { ctl->target_steps = target; }
__global__ void k_set_k(pdlpdev_ctl* ctl, int k) { ctl->k = k; }
__global__ void k_clear_error(pdlpdev_ctl* ctl) { ctl->error = 0; }
__global__ void k_set_error(pdlpdev_ctl* ctl) { ctl->error = 1; }
__global__ void k_set_loop_state(pdlpdev_ctl* ctl, double sum_weights, int its_since_restart, int k)
{
  ctl->sum_weights       = sum_weights;
  ctl->its_since_restart = its_since_restart;
  ctl->k                 = k;
  ctl->pending_avg       = 0;
}

// unscale for output: x = x^ * D_c, y = y^ * D_r (unscale_solutions, initial_scaling.cu:460-484)
__global__ void __launch_bounds__(kBlock)
k_unscale(int n, const double* __restrict__ v, const double* __restrict__ d, double* __restrict__ out)
{
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = v[i] * d[i];
}






// Every hot-loop launch goes through here so that pdlpdev_time_kernel can ask for the dispatch's own start / stop
// timestamps (hipExtLaunchKernel) without putting event records between the kernels of an attempt.
#include "pdlp_launch.hpp"
#include "pdlp_core_internal.hpp"

// panel values <- current CSR values (after upload and again after scale_problem)
int sync_panel_values(pdlpdev_ctx* c)
{
  const int64_t hot = c->dense.hot_nnz, hot_t = c->hot_nnz_at;
  // first the hot copies of the matrices, the segments' and the extracted rows' values, from the full (just scaled) CSR
  if (c->ha_val != c->a_val) k_permute<<<grid_for(hot), kBlock, 0, c->stream>>>(hot, c->dense.s_perm_a, c->a_val, c->ha_val);
  if (c->hat_val != c->at_val) k_permute<<<grid_for(hot_t), kBlock, 0, c->stream>>>(hot_t, c->dense.s_perm_at, c->at_val, c->hat_val);
  if (c->dense.on) k_permute<<<grid_for(c->dense.nent), kBlock, 0, c->stream>>>(c->dense.nent, c->dense.perm, c->a_val, c->dense.val);
  (void)hot, (void)hot_t;
  if (c->pa.on) k_permute<<<grid_for(c->pa.nent), kBlock, 0, c->stream>>>(c->pa.nent, c->pa.perm, c->ha_val, c->pa.val);
  if (c->pat.on) k_permute<<<grid_for(c->pat.nent), kBlock, 0, c->stream>>>(c->pat.nent, c->pat.perm, c->hat_val, c->pat.val);
  if (c->ja.on) k_permute<<<grid_for(c->ja.nent), kBlock, 0, c->stream>>>(c->ja.nent, c->ja.perm, c->ha_val, c->ja.val);
  if (c->jat.on) k_permute<<<grid_for(c->jat.nent), kBlock, 0, c->stream>>>(c->jat.nent, c->jat.perm, c->hat_val, c->jat.val);
  if (c->pba.on) k_permute_pad<<<grid_for(c->pba.np), kBlock, 0, c->stream>>>(c->pba.np, c->pba.perm, c->ha_val, c->pba.val);
  if (c->pbat.on) k_permute_pad<<<grid_for(c->pbat.np), kBlock, 0, c->stream>>>(c->pbat.np, c->pbat.perm, c->hat_val, c->pbat.val);
  if (c->poc.on) k_permute<<<grid_for(c->poc.nent), kBlock, 0, c->stream>>>(c->poc.nent, c->poc.perm, c->oc_val, c->poc.val);
  if (c->joc.on) k_permute<<<grid_for(c->joc.nent), kBlock, 0, c->stream>>>(c->joc.nent, c->joc.perm, c->oc_val, c->joc.val);
  HIP_TRY(hipGetLastError());
  return 0;
}


// per slice of the gathered vector: the smallest and largest index this matrix references there (halo exchange, pdlp_ctx.hpp Halo)
__global__ void __launch_bounds__(kBlock) k_slice_ranges(int64_t nnz, const int32_t* __restrict__ idx, int32_t slice, int world,
                                                         int32_t* __restrict__ lo, int32_t* __restrict__ hi)
{
  __shared__ int32_t slo[16], shi[16];
  if (threadIdx.x < 16) slo[threadIdx.x] = 0x7fffffff, shi[threadIdx.x] = -1;
  __syncthreads();
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * kBlock) {
    const int32_t j = idx[k];
    const int q     = min(j / slice, world - 1);
    if (j < slo[q]) atomicMin(&slo[q], j);
    if (j > shi[q]) atomicMax(&shi[q], j);
  }
  __syncthreads();
  if ((int)threadIdx.x < world && shi[threadIdx.x] >= 0) atomicMin(&lo[threadIdx.x], slo[threadIdx.x]), atomicMax(&hi[threadIdx.x], shi[threadIdx.x]);
}

extern "C" {

static inline int oc_partials(const pdlpdev_ctx* ctx) { return ctx->joc.on ? ctx->joc.v.nblk + ctx->joc.v.nlong : ctx->poc.on ? ctx->poc.v.W : ctx->oc_nb; }
int pdlpdev_owner_slice(pdlpdev_ctx* ctx, int32_t* col_begin, int32_t* ncols)
{
  if (!ctx->owner) return fail(-1, "pdlpdev_owner_slice: the solver does not run the owner-computes dataflow");
  const int64_t cs = (int64_t)ctx->rank * ctx->slice;
  *col_begin       = (int32_t)std::min<int64_t>(cs, ctx->n);
  *ncols           = (int32_t)std::max<int64_t>(0, std::min<int64_t>(ctx->slice, (int64_t)ctx->n - cs));
  return 0;
}
// The column block of the owner-computes dataflow: rows [col_begin, col_begin + ncols) of the GLOBAL A^T (column indices =
// global row numbers of A, ascending: every column is then summed over the rows in the order one GPU uses), unscaled
// values; row_bounds[world + 1] = the ranks' row blocks.  Call after pdlpdev_scale_problem: the values are scaled here with
// this rank's D_c and the gathered D_r, by the expression the transposed copy of an unsharded solve goes through.
int pdlpdev_owner_setup(pdlpdev_ctx* ctx, const int32_t* off, const int32_t* idx, const double* val, const int32_t* row_bounds)
{
  if (!ctx->owner) return fail(-1, "pdlpdev_owner_setup: the solver does not run the owner-computes dataflow");
  if (!ctx->scaled) return fail(-1, "pdlpdev_owner_setup: call after pdlpdev_scale_problem");
  HIP_TRY(hipSetDevice(ctx->device));
  int32_t cb = 0, nc = 0;
  TRY(pdlpdev_owner_slice(ctx, &cb, &nc));
  if (row_bounds[ctx->rank + 1] - row_bounds[ctx->rank] != ctx->m) return fail(-1, "pdlpdev_owner_setup: row bounds disagree with this rank's block");
  int maxrows = 0;
  for (int q = 0; q < ctx->world; ++q) maxrows = std::max(maxrows, row_bounds[q + 1] - row_bounds[q]);
  ctx->ypad = (maxrows + 15) & ~15;
  const int64_t gcols = (int64_t)ctx->world * ctx->ypad;
  if (gcols >= ((int64_t)1 << 31)) return fail(-1, "pdlpdev_owner_setup: gathered dual too long");
  const int64_t nnz = nc > 0 ? off[nc] : 0;
  ctx->oc_rows = nc, ctx->oc_nnz = nnz;
  // global row -> position in the gathered dual (rank q's rows at [q * ypad, ...))
  std::vector<int32_t> ridx((size_t)std::max<int64_t>(nnz, 1));
  {
    std::vector<int32_t> shift(ctx->world);
    for (int q = 0; q < ctx->world; ++q) shift[q] = q * ctx->ypad - row_bounds[q];
    for (int32_t r = 0; r < nc; ++r) {
      int q = 0;
      for (int k = off[r]; k < off[r + 1]; ++k) {  // ascending rows: the owner only moves forward
        while (idx[k] >= row_bounds[q + 1]) ++q;
        ridx[k] = idx[k] + shift[q];
      }
    }
  }
  TRY(upload_i32(ctx, &ctx->oc_off, off, (size_t)nc + 1));
  TRY(upload_i32(ctx, &ctx->oc_idx, ridx.data(), (size_t)nnz, 8));
  TRY(upload_f64(ctx, &ctx->oc_val, val, (size_t)nnz, 8));
  TRY(dev_alloc(ctx, &ctx->ygather, (size_t)gcols + kSlicePad));
  std::vector<int32_t> longs;
  for (int32_t r = 0; r < nc; ++r)
    if (off[r + 1] - off[r] > kLongRow) longs.push_back(r);
  ctx->oc_nlong = (int)longs.size();
  if (ctx->oc_nlong) TRY(upload_i32(ctx, &ctx->oc_long, longs.data(), longs.size()));
  std::vector<int32_t> rb = build_row_blocks(nc, off);
  ctx->oc_nb = (int)rb.size() / 2 - 1;
  TRY(upload_i32(ctx, &ctx->oc_rb, rb.data(), rb.size()));
  // scaling: D_r of every rank's rows in the gathered layout (ygather doubles as the staging buffer), then (val * D_c[j]) * D_r[i]
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->ygather + (size_t)ctx->rank * ctx->ypad, ctx->dr, (size_t)ctx->m * sizeof(double), hipMemcpyDeviceToDevice, s));
  TRY(all_gather(ctx, ctx->ygather, (size_t)ctx->ypad));
  k_scale_matrix<<<grid_for(nc), kBlock, 0, s>>>(nc, ctx->oc_off, ctx->oc_idx, ctx->oc_val, ctx->dc + cb, ctx->ygather);
  if (ctx->oc_nlong) k_scale_matrix_long<<<ctx->oc_nlong, kBlock, 0, s>>>(ctx->oc_long, ctx->oc_off, ctx->oc_idx, ctx->oc_val, ctx->dc + cb, ctx->ygather);
  HIP_TRY(hipGetLastError());
  // layouts, by the rules of the two other matrices (timed mode counts as auto here)
  {
    const char* mode_env = getenv("CUOPT_AMD_SPMV_LAYOUT");
    std::string mode     = mode_env ? mode_env : "auto";
    if (mode == "timed") mode = "auto";
    const int64_t slab_bytes = std::max<int64_t>(64, cuopt_amd::tune_int("slab_bytes", 1398102));
    const int64_t ws_limit   = cuopt_amd::tune_int("panel_ws_bytes", kPanelWorkingSetBytes);
    if (nc > 0 && (mode == "auto" || mode == "jag")) {
      JagHost j = build_jag(nc, (int32_t)gcols, off, ridx.data(), mode == "jag" ? 1 : 0, ctx->cus);
      TRY(upload_jag(ctx, &ctx->joc, j, ctx->oc_off, ctx->oc_idx, ctx->oc_val));
    }
    const bool panels = mode == "panel" || (mode == "auto" && gcols * 8 > ws_limit && gather_working_set(nc, (int32_t)gcols, off, ridx.data()) > ws_limit);
    if (nc > 0 && !ctx->joc.on && panels) {
      PanelHost h = build_panels(nc, (int32_t)gcols, off, ridx.data(), slab_bytes, true);
      TRY(upload_panels(ctx, &ctx->poc, h, ctx->oc_off, ctx->oc_idx, ctx->oc_val));
    }
  }
  TRY(dev_alloc(ctx, &ctx->part_oc, (size_t)8 * std::max(oc_partials(ctx), 1)));
  {
    const char* tr = getenv("CUOPT_AMD_SHARD_TRANSPORT");
    if (tr && std::string(tr) == "p2p") TRY(p2p_setup(ctx));
    else if (tr && std::string(tr) != "collective") return fail(-1, "CUOPT_AMD_SHARD_TRANSPORT must be collective or p2p");
  }
  if (ctx->poc.on) k_permute<<<grid_for(ctx->poc.nent), kBlock, 0, s>>>(ctx->poc.nent, ctx->poc.perm, ctx->oc_val, ctx->poc.val);
  if (ctx->joc.on) k_permute<<<grid_for(ctx->joc.nent), kBlock, 0, s>>>(ctx->joc.nent, ctx->joc.perm, ctx->oc_val, ctx->joc.val);
  HIP_TRY(hipGetLastError());
  {
    // what this rank references outside its own slice / rows: per peer one range of xbar (the rank's rows of A, on the device) and
    // one of the gathered y' (the column block, here on the host)
    const int W = ctx->world;
    std::vector<int32_t> need((size_t)4 * W, 0), xl(W, 0x7fffffff), xh(W, -1);
    int32_t *d_lo = nullptr, *d_hi = nullptr;
    TRY(dev_alloc(ctx, &d_lo, 16));
    TRY(dev_alloc(ctx, &d_hi, 16));
    HIP_TRY(hipMemcpyAsync(d_lo, xl.data(), W * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_hi, xh.data(), W * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (ctx->nnz > 0) k_slice_ranges<<<grid_for(ctx->nnz, 8), kBlock, 0, s>>>(ctx->nnz, ctx->a_idx, ctx->slice, W, d_lo, d_hi);
    HIP_TRY(hipMemcpyAsync(xl.data(), d_lo, W * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(xh.data(), d_hi, W * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<int32_t> yl(W, 0x7fffffff), yh(W, -1);
    for (int64_t k = 0; k < nnz; ++k) {
      const int32_t pos = ridx[(size_t)k];
      const int q       = pos / ctx->ypad;
      yl[q] = std::min(yl[q], pos), yh[q] = std::max(yh[q], pos);
    }
    for (int q = 0; q < W; ++q) {
      need[(size_t)4 * q + 0] = xh[q] >= 0 ? xl[q] : 0, need[(size_t)4 * q + 1] = xh[q] >= 0 ? xh[q] + 1 : 0;
      need[(size_t)4 * q + 2] = yh[q] >= 0 ? yl[q] : 0, need[(size_t)4 * q + 3] = yh[q] >= 0 ? yh[q] + 1 : 0;
    }
    TRY(halo_setup(ctx, need.data()));
  }
  HIP_TRY(hipStreamSynchronize(s));  // the host arrays are the caller's
  return 0;
}

// ---- helpers --------------------------------------------------------------------------------------
int fetch_scalars(pdlpdev_ctx* ctx, int count)
{
  HIP_TRY(hipMemcpyAsync(ctx->scal_h, ctx->scal, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}
int fetch_ctl(pdlpdev_ctx* ctx, pdlpdev_ctl* out)
{
  HIP_TRY(hipMemcpyAsync(ctx->ctl_h, ctx->ctl, sizeof(pdlpdev_ctl), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (out) *out = *ctx->ctl_h;
  return 0;
}


static int reduce_vec(pdlpdev_ctx* ctx, int mode, int64_t n, const double* a, const double* b, int slot)
{
  const int g = std::min(grid_for(n), kGenericBlocks);
  switch (mode) {
    case 0: k_reduce<0><<<g, kBlock, 0, ctx->stream>>>(n, a, b, ctx->part_g); break;
    case 1: k_reduce<1><<<g, kBlock, 0, ctx->stream>>>(n, a, b, ctx->part_g); break;
    case 2: k_reduce<2><<<g, kBlock, 0, ctx->stream>>>(n, a, b, ctx->part_g); break;
    default: k_reduce<3><<<g, kBlock, 0, ctx->stream>>>(n, a, b, ctx->part_g); break;
  }
  k_finalize<<<1, kBlock, 0, ctx->stream>>>(ctx->part_g, g, 1, mode == 0 ? 1u : 0u, ctx->scal + slot);
  LAUNCH_CHECK();
  return 0;
}

// max over the ranks of one host scalar (identity without a communicator): lets every rank of a sharded solve
// take the same wall-clock decision (time limit), which they must -- the collectives have to match up
int pdlpdev_agree_max(pdlpdev_ctx* ctx, double* value)
{
  if (!ctx->comm) return 0;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(ctx->scal + 60, value, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  TRY(allreduce(ctx, ctx->scal + 60, 1, rccl::kMax));
  HIP_TRY(hipMemcpyAsync(ctx->scal_h + 60, ctx->scal + 60, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *value = ctx->scal_h[60];
  return 0;
}

int pdlpdev_init_norms(pdlpdev_ctx* ctx, double out[3])
{
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(reduce_vec(ctx, 0, ctx->nnz, ctx->a_val, nullptr, 0));
  TRY(reduce_vec(ctx, 1, ctx->n, ctx->c, nullptr, 1));
  TRY(reduce_vec(ctx, 2, ctx->m, ctx->lo, ctx->hi, 2));
  TRY(allreduce(ctx, ctx->scal + 0, 1, rccl::kMax));
  TRY(allreduce(ctx, ctx->scal + 2, 1, rccl::kSum));
  TRY(fetch_scalars(ctx, 3));
  out[0] = ctx->scal_h[0], out[1] = ctx->scal_h[1], out[2] = ctx->scal_h[2];
  return 0;
}

// New bounds for the SAME matrix and objective, iterate and all loop state back to their values right after
// pdlpdev_scale_problem: what a MIP heuristic needs between two relaxations (relaxed_lp.cu:53-127 builds a whole
// new pdlp_solver_t instead).  D_r, D_c depend on A only (initial_scaling.cu:125-307), so nothing is rescaled.
int pdlpdev_reset(pdlpdev_ctx* ctx, const double* lb, const double* ub, const double* lo, const double* hi)
{
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->scaled) return fail(-1, "pdlpdev_reset: the problem has not been scaled yet");
  ctx->rejected_in_a_row = 0;
  hipStream_t s = ctx->stream;
  const size_t nb = (size_t)ctx->n * sizeof(double), mb = (size_t)ctx->m * sizeof(double);
  auto put = [&](const double* src, double* unscaled, double* scaled, size_t bytes) -> int {
    if (!src || bytes == 0) return 0;
    HIP_TRY(hipMemcpyAsync(unscaled, src, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(scaled, unscaled, bytes, hipMemcpyDeviceToDevice, s));
    return 0;
  };
  TRY(put(lb, ctx->lb_u, ctx->lb, nb));
  TRY(put(ub, ctx->ub_u, ctx->ub, nb));
  {
    const pdlpdev_ctx::UniformBounds before = ctx->ubd;
    ctx->note_uniform_bounds(lb, ub);
    const pdlpdev_ctx::UniformBounds& now = ctx->ubd;
    if (before.lb_same != now.lb_same || before.ub_same != now.ub_same || (now.lb_same && before.lb != now.lb) || (now.ub_same && before.ub != now.ub)) {
      // the attempt graphs carry k_primal's arguments by value: captured again on the next run
      for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
      ctx->graphs.clear();
    }
  }
  if ((lo || hi) && (ctx->rows_aliased || (ctx->clones_alive > 0 && !ctx->rows_private))) {
    // row bounds shared with clones (or with the parent): this context gets arrays of its own first; the others keep the old ones
    for (double** v : {&ctx->lo, &ctx->lo_u, &ctx->hi, &ctx->hi_u}) {
      const double* src = *v;
      TRY(dev_alloc(ctx, v, (size_t)ctx->m));
      HIP_TRY(hipMemcpyAsync(*v, src, mb, hipMemcpyDeviceToDevice, s));
    }
    ctx->rows_aliased = false;
    ctx->rows_private = true;  // (until pdlpdev_clone_shared makes another clone of this context: that one aliases the new arrays)
    for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);  // (the attempt graphs carry the old arrays' addresses)
    ctx->graphs.clear();
  }
  TRY(put(lo, ctx->lo_u, ctx->lo, mb));
  TRY(put(hi, ctx->hi_u, ctx->hi, mb));
  if (lb || ub || lo || hi)
    k_scale_bounds<<<grid_for(std::max(ctx->m, ctx->n)), kBlock, 0, s>>>(ctx->n, ctx->m, lb ? ctx->lb : nullptr, ub ? ctx->ub : nullptr,
                                                                      ctx->dc, lo ? ctx->lo : nullptr, hi ? ctx->hi : nullptr, ctx->dr);
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipMemsetAsync(ctx->x[i], 0, nb, s));
    HIP_TRY(hipMemsetAsync(ctx->aty[i], 0, nb, s));
    HIP_TRY(hipMemsetAsync(ctx->rc[i], 0, nb, s));
    HIP_TRY(hipMemsetAsync(ctx->y[i], 0, mb, s));
  }
  for (double* v : {ctx->xbar, ctx->sumx, ctx->avgx, ctx->lrx}) HIP_TRY(hipMemsetAsync(v, 0, nb, s));
  for (double* v : {ctx->sumy, ctx->avgy, ctx->lry}) HIP_TRY(hipMemsetAsync(v, 0, mb, s));
  HIP_TRY(hipMemsetAsync(ctx->ctl, 0, sizeof(pdlpdev_ctl), s));
  LAUNCH_CHECK();
  return 0;
}
// out[0] = sum c_j^2, out[1] = sum bcomb_i^2 of the scaled (unscaled != 0: the user's) problem
int pdlpdev_weight_norms(pdlpdev_ctx* ctx, int unscaled, double out[2])
{
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(reduce_vec(ctx, 1, ctx->n, unscaled ? ctx->c_u : ctx->c, nullptr, 1));
  TRY(reduce_vec(ctx, 2, ctx->m, unscaled ? ctx->lo_u : ctx->lo, unscaled ? ctx->hi_u : ctx->hi, 2));
  TRY(allreduce(ctx, ctx->scal + 2, 1, rccl::kSum));
  TRY(fetch_scalars(ctx, 3));
  out[0] = ctx->scal_h[1], out[1] = ctx->scal_h[2];
  return 0;
}


int pdlpdev_problem_norms(pdlpdev_ctx* ctx, double* norm_c, double* norm_b)
{
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(reduce_vec(ctx, 1, ctx->n, ctx->c_u, nullptr, 0));
  TRY(reduce_vec(ctx, 2, ctx->m, ctx->lo_u, ctx->hi_u, 1));
  TRY(allreduce(ctx, ctx->scal + 1, 1, rccl::kSum));
  TRY(fetch_scalars(ctx, 2));
  if (norm_c) *norm_c = sqrt(ctx->scal_h[0]);
  if (norm_b) *norm_b = sqrt(ctx->scal_h[1]);
  return 0;
}

int pdlpdev_set_step_params(pdlpdev_ctx* ctx, const pdlpdev_step_params* p)
{
  ctx->sp = *p;
  for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);  // parameters are baked in
  ctx->graphs.clear();
  return 0;
}
int pdlpdev_set_step(pdlpdev_ctx* ctx, double step_size, double primal_weight)
{
  HIP_TRY(hipSetDevice(ctx->device));
  k_set_step<<<1, 1, 0, ctx->stream>>>(ctx->ctl, step_size, primal_weight);
  LAUNCH_CHECK();
  return 0;
}
int pdlpdev_set_k(pdlpdev_ctx* ctx, int32_t k)
{
  HIP_TRY(hipSetDevice(ctx->device));
  k_set_k<<<1, 1, 0, ctx->stream>>>(ctx->ctl, k);
  LAUNCH_CHECK();
  return 0;
}
int pdlpdev_set_initial(pdlpdev_ctx* ctx, const double* x, const double* y)
{
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(fetch_ctl(ctx, nullptr));
  const int cur = ctx->ctl_h->cur;
  if (x) {
    HIP_TRY(hipMemcpyAsync(ctx->x[cur], x, (size_t)ctx->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    k_div_inplace<<<grid_for(ctx->n), kBlock, 0, ctx->stream>>>(ctx->n, ctx->x[cur], ctx->dc);
  }
  if (y) {
    HIP_TRY(hipMemcpyAsync(ctx->y[cur], y, (size_t)ctx->m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    k_div_inplace<<<grid_for(ctx->m), kBlock, 0, ctx->stream>>>(ctx->m, ctx->y[cur], ctx->dr);
  }
  LAUNCH_CHECK();
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}
}  // extern "C"
extern "C" {
// update_step_size_on_initial_solution (pdlp.cu:878-948): the quantities of one compute_step_sizes call with
// delta_primal = x0, delta_dual = y' = y0 and a zero A^T y.  out = {interaction x0.(A^T y0), ||x0||^2, ||y0||^2, max|x0|,
// max|y0|}.
//   x0_unscaled == y0_unscaled == nullptr: on the SCALED iterate that set_initial left in the current buffers; leaves A^T y0
//     in the current A^T y buffer (which is what the loop needs next anyway).
//   both given (compute_initial_step_size_before_scaling, pdlp.cu:929-947): the caller's vectors as they came, against the
//     SCALED matrix; they pass through the next-iterate buffers, which the first attempt overwrites before it reads them,
//     and the current A^T y buffer is left alone.
int pdlpdev_initial_solution_stats(pdlpdev_ctx* ctx, double out[5], const double* x0_unscaled, const double* y0_unscaled)
{
  HIP_TRY(hipSetDevice(ctx->device));
  if ((x0_unscaled == nullptr) != (y0_unscaled == nullptr)) return fail(-1, "initial_solution_stats: both unscaled vectors or none");
  hipStream_t s = ctx->stream;
  const int n = ctx->n, m = ctx->m;
  TRY(fetch_ctl(ctx, nullptr));
  const int cur = ctx->ctl_h->cur;
  const double *xv = ctx->x[cur], *yv = ctx->y[cur], *atyv = ctx->aty[cur];
  if (x0_unscaled) {
    const int nxt = 1 - cur;
    HIP_TRY(hipMemcpyAsync(ctx->x[nxt], x0_unscaled, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->y[nxt], y0_unscaled, (size_t)m * sizeof(double), hipMemcpyHostToDevice, s));
    double* prod = ctx->comm ? ctx->ar_buf : ctx->aty[nxt];
    launch_plain(ctx, 1, ctx->y[nxt], prod);
    LAUNCH_CHECK();
    if (ctx->comm) TRY(allreduce(ctx, prod, (size_t)n, rccl::kSum));
    xv = ctx->x[nxt], yv = ctx->y[nxt], atyv = prod;
  } else {
    TRY(pdlpdev_compute_aty(ctx));
  }
  const int g = std::min(grid_for(n), kGenericBlocks);
  k_dot<<<g, kBlock, 0, s>>>(n, xv, atyv, ctx->part_g);
  k_finalize<<<1, kBlock, 0, s>>>(ctx->part_g, g, 1, 0u, ctx->scal + 48);
  TRY(reduce_vec(ctx, 1, n, xv, nullptr, 49));
  TRY(reduce_vec(ctx, 1, m, yv, nullptr, 50));
  TRY(reduce_vec(ctx, 0, n, xv, nullptr, 51));
  TRY(reduce_vec(ctx, 0, m, yv, nullptr, 52));
  LAUNCH_CHECK();
  if (ctx->comm) {  // the dual side is sharded
    TRY(allreduce(ctx, ctx->scal + 50, 1, rccl::kSum));
    TRY(allreduce(ctx, ctx->scal + 52, 1, rccl::kMax));
  }
  HIP_TRY(hipMemcpyAsync(ctx->scal_h + 48, ctx->scal + 48, 5 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < 5; ++i) out[i] = ctx->scal_h[48 + i];
  return 0;
}

int pdlpdev_project_primal(pdlpdev_ctx* ctx)
{
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(fetch_ctl(ctx, nullptr));
  const int cur = ctx->ctl_h->cur;
  k_clamp<<<grid_for(ctx->n), kBlock, 0, ctx->stream>>>(ctx->n, ctx->x[cur], ctx->lb, ctx->ub);
  k_clamp<<<grid_for(ctx->n), kBlock, 0, ctx->stream>>>(ctx->n, ctx->avgx, ctx->lb, ctx->ub);
  LAUNCH_CHECK();
  return 0;
}

// ---- hot loop -------------------------------------------------------------------------------------
}  // extern "C"
extern "C" {
int pdlpdev_compute_aty(pdlpdev_ctx* ctx)
{
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (!ctx->comm) {
    launch_at_cur(ctx, nullptr, 0);
    LAUNCH_CHECK();
  } else {
    launch_at_cur(ctx, ctx->ar_buf, 0);
    LAUNCH_CHECK();
    TRY(allreduce(ctx, ctx->ar_buf, (size_t)ctx->n, rccl::kSum));
    TRY(fetch_ctl(ctx, nullptr));
    HIP_TRY(hipMemcpyAsync(ctx->aty[ctx->ctl_h->cur], ctx->ar_buf, (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  return 0;
}


// dense row segments: their share of A v (transpose = 0) / A^T v lands in dense.add_m / add_n right before the layout's kernel adds it
void dense_part(pdlpdev_ctx* ctx, int transpose, const double* v0, const double* v1, int mode, int in_loop)
{
  const pdlpdev_ctx::Dense& D = ctx->dense;
  const bool fused            = transpose ? (ctx->pat.on && ctx->pat.v.dn_pan_ptr != nullptr) : (ctx->pa.on && ctx->pa.v.dn_own_seg != nullptr);
  if (D.on && !fused) {
    DenseView V{D.row, D.row_seg, D.seg_row, D.seg_c0, D.seg_len, D.seg_ptr, D.tile_ptr, D.tile_seg, D.tile_id, D.val, D.ch_seg, D.ch_k0, D.row_ch, D.ch_part};
    if (transpose) {
      launch_k(ctx, k_dense_cols, D.ntiles, kBlock, 0, V, ctx->n, ctx->ctl, v0, v1, mode, in_loop, D.add_n);
    } else {
      launch_k(ctx, k_dense_rows, D.nchunks, kBlock, 0, V, ctx->ctl, v0, v1, mode, in_loop);
      launch_k(ctx, k_dense_rows_finish, (D.nrows + kBlock - 1) / kBlock, kBlock, 0, V, D.nrows, ctx->ctl, in_loop, D.add_m);
    }
  }
}
// launch helpers: pick the layout (jagged rows with LDS column sets, slab-major panels, CSR stream)
static void launch_a_dual(pdlpdev_ctx* ctx, double* ycopy = nullptr, const p2pdev::Push* push = nullptr)
{
  dense_part(ctx, 0, ctx->xbar, nullptr, 0, 1);
  if (ctx->pba.on) {
    (void)pb_products(ctx, ctx->pba, ctx->xbar, nullptr, 0, 1);
    (void)pb_rows(ctx, k_pb_a_dual, ctx->pba, ctx->ctl, ctx->y[0], ctx->y[1], ctx->lo, ctx->hi, ctx->sumy, ctx->part_a, ycopy, push);
  } else if (ctx->ja.on)
    (void)JAG_LAUNCH(ctx, k_jag_a_dual, ctx->ja.v, ctx->ctl, ctx->xbar, ctx->y[0], ctx->y[1], ctx->lo, ctx->hi, ctx->sumy, ctx->part_a, ycopy, push);
  else if (ctx->pa.on)
    launch_k(ctx, ctx->pa.v.seg ? k_panel_a_dual<true> : k_panel_a_dual<false>, ctx->pa.v.W, kPanelThreads, 0, ctx->pa.v, ctx->ctl, ctx->xbar, ctx->y[0], ctx->y[1], ctx->lo, ctx->hi, ctx->sumy, ctx->part_a, ycopy, push);
  else
    launch_k(ctx, k_spmv_a_dual, stream_grid(ctx->a_nb), kBlock, 0, ctx->a_nb, ctx->a_rb, ctx->ha_off, ctx->ha_idx, ctx->ha_val, ctx->ctl, ctx->xbar, ctx->y[0], ctx->y[1], ctx->lo, ctx->hi, ctx->sumy, ctx->part_a, ycopy, push, ctx->dense.add_m);
}
static void launch_at_step(pdlpdev_ctx* ctx)
{
  dense_part(ctx, 1, ctx->y[0], ctx->y[1], 1, 1);
  if (ctx->pbat.on) {
    (void)pb_products(ctx, ctx->pbat, ctx->y[0], ctx->y[1], 1, 1);  // y' = the trial dual
    (void)pb_rows(ctx, k_pb_at_step, ctx->pbat, ctx->ctl, ctx->x[0], ctx->x[1], ctx->aty[0], ctx->aty[1], ctx->part_at);
  } else if (ctx->jat.on)
    (void)JAG_LAUNCH(ctx, k_jag_at_step, ctx->jat.v, ctx->ctl, ctx->y[0], ctx->y[1], ctx->x[0], ctx->x[1], ctx->aty[0], ctx->aty[1], ctx->part_at);
  else if (ctx->pat.on)
    launch_k(ctx, ctx->pat.v.seg ? k_panel_at_step<true> : k_panel_at_step<false>, ctx->pat.v.W, kPanelThreads, 0, ctx->pat.v, ctx->ctl, ctx->y[0], ctx->y[1], ctx->x[0], ctx->x[1], ctx->aty[0], ctx->aty[1], ctx->part_at);
  else
    launch_k(ctx, k_spmv_at_step, stream_grid(ctx->at_nb), kBlock, 0, ctx->at_nb, ctx->at_rb, ctx->hat_off, ctx->hat_idx, ctx->hat_val, ctx->ctl, ctx->y[0], ctx->y[1], ctx->x[0], ctx->x[1], ctx->aty[0], ctx->aty[1], ctx->part_at, ctx->dense.add_n);
}
void launch_at_cur(pdlpdev_ctx* ctx, double* out_override, int use_next)
{
  dense_part(ctx, 1, ctx->y[0], ctx->y[1], use_next ? 1 : 2, 0);
  if (ctx->pbat.on) {
    (void)pb_products(ctx, ctx->pbat, ctx->y[0], ctx->y[1], use_next ? 1 : 2, 0);
    (void)pb_rows(ctx, k_pb_at_cur, ctx->pbat, ctx->ctl, ctx->aty[0], ctx->aty[1], out_override, use_next);
  } else if (ctx->jat.on)
    (void)JAG_LAUNCH(ctx, k_jag_at_cur, ctx->jat.v, ctx->ctl, ctx->y[0], ctx->y[1], ctx->aty[0], ctx->aty[1], out_override, use_next);
  else if (ctx->pat.on)
    launch_k(ctx, ctx->pat.v.seg ? k_panel_at_cur<true> : k_panel_at_cur<false>, ctx->pat.v.W, kPanelThreads, 0, ctx->pat.v, ctx->ctl, ctx->y[0], ctx->y[1], ctx->aty[0], ctx->aty[1], out_override, use_next);
  else
    launch_k(ctx, k_spmv_at_cur, stream_grid(ctx->at_nb), kBlock, 0, ctx->at_nb, ctx->at_rb, ctx->hat_off, ctx->hat_idx, ctx->hat_val, ctx->ctl, ctx->y[0], ctx->y[1], ctx->aty[0], ctx->aty[1], out_override, use_next, ctx->dense.add_n);
}
// plain y = A x (transpose = 0) or y = A^T x through the layout the solver iterates with
void launch_plain(pdlpdev_ctx* ctx, int transpose, const double* vec, double* out)
{
  dense_part(ctx, transpose, vec, nullptr, 0, 0);
  if (transpose) {
    if (ctx->pbat.on) {
      (void)pb_products(ctx, ctx->pbat, vec, nullptr, 0, 0);
      (void)pb_rows(ctx, k_pb_plain, ctx->pbat, out);
    } else if (ctx->jat.on)
      (void)JAG_LAUNCH(ctx, k_jag_plain, ctx->jat.v, vec, out);
    else if (ctx->pat.on)
      launch_k(ctx, ctx->pat.v.seg ? k_panel_plain<true> : k_panel_plain<false>, ctx->pat.v.W, kPanelThreads, 0, ctx->pat.v, vec, out);
    else
      launch_k(ctx, k_spmv_plain, stream_grid(ctx->at_nb), kBlock, 0, ctx->at_nb, ctx->at_rb, ctx->hat_off, ctx->hat_idx, ctx->hat_val, vec, out, ctx->dense.add_n);
  } else {
    if (ctx->pba.on) {
      (void)pb_products(ctx, ctx->pba, vec, nullptr, 0, 0);
      (void)pb_rows(ctx, k_pb_plain, ctx->pba, out);
    } else if (ctx->ja.on)
      (void)JAG_LAUNCH(ctx, k_jag_plain, ctx->ja.v, vec, out);
    else if (ctx->pa.on)
      launch_k(ctx, ctx->pa.v.seg ? k_panel_plain<true> : k_panel_plain<false>, ctx->pa.v.W, kPanelThreads, 0, ctx->pa.v, vec, out);
    else
      launch_k(ctx, k_spmv_plain, stream_grid(ctx->a_nb), kBlock, 0, ctx->a_nb, ctx->a_rb, ctx->ha_off, ctx->ha_idx, ctx->ha_val, vec, out, ctx->dense.add_m);
  }
}
static void launch_decision(pdlpdev_ctx* ctx)
{
  launch_k(ctx, k_step_decision, 1, kDecisionThreads, 0, ctx->ctl, ctx->part_a, dual_partials(ctx), ctx->part_at, step_partials(ctx), nullptr, ctx->sp);
}

// owner-computes dataflow: A^T y' of this rank's columns from the gathered y' (complete sums) + the step statistics of the slice
static void launch_oc_step(pdlpdev_ctx* ctx)
{
  const size_t cs = (size_t)ctx->rank * ctx->slice;
  double *x0 = ctx->x[0] + cs, *x1 = ctx->x[1] + cs, *t0 = ctx->aty[0] + cs, *t1 = ctx->aty[1] + cs;
  const double* yg = ctx->ygather;  // the trial dual of every rank, whichever ping-pong buffer it lives in
  if (ctx->joc.on)
    (void)JAG_LAUNCH(ctx, k_jag_at_step, ctx->joc.v, ctx->ctl, yg, yg, x0, x1, t0, t1, ctx->part_oc);
  else if (ctx->poc.on)
    launch_k(ctx, ctx->poc.v.seg ? k_panel_at_step<true> : k_panel_at_step<false>, ctx->poc.v.W, kPanelThreads, 0, ctx->poc.v, ctx->ctl, yg, yg, x0, x1, t0, t1, ctx->part_oc);
  else
    launch_k(ctx, k_spmv_at_step, stream_grid(ctx->oc_nb), kBlock, 0, ctx->oc_nb, ctx->oc_rb, ctx->oc_off, ctx->oc_idx, ctx->oc_val, ctx->ctl, yg, yg, x0, x1, t0, t1, ctx->part_oc, (const double*)nullptr);
}

// one PDHG attempt = 4 launches (single GPU) on ctx->stream
static int enqueue_attempt(pdlpdev_ctx* ctx)
{
  const int n = ctx->n;
  if (ctx->owner) {
    if (!ctx->oc_off) return fail(-1, "owner-computes dataflow: pdlpdev_owner_setup was not called");
    const size_t cs = (size_t)ctx->rank * ctx->slice;
    const int len   = (int)std::max<int64_t>(0, std::min<int64_t>(ctx->slice, (int64_t)n - (int64_t)cs));
    launch_k(ctx, k_primal, grid_for(len), kBlock, 0, len, ctx->ctl, ctx->x[0] + cs, ctx->x[1] + cs, ctx->aty[0] + cs, ctx->aty[1] + cs,
             ctx->c + cs, ctx->lb + cs, ctx->ub + cs, ctx->xbar + cs, ctx->sumx + cs, ctx->p2p.on ? ctx->p2p.push_dev : (const p2pdev::Push*)nullptr, ctx->ubd);
    LAUNCH_CHECK();
    if (ctx->p2p.on) {
      // direct peer stores + epoch flags instead of collectives: the producing kernels (the primal step above, the dual step,
      // the packing of the step-size sums) store into every rank's landing block and raise this rank's flag from their last
      // workgroup; a consumer waits for the world flags and copies the landed vector into ordinary memory.  Kernels only,
      // nothing between them but stream order: 7 launches per attempt.
      pdlpdev_ctx::P2P& P = ctx->p2p;
      const int W = ctx->world;
      const unsigned long long* flags = reinterpret_cast<const unsigned long long*>(P.base + P.off_f);
      auto pull = [&](int kind, double* dst, size_t land_off, int per_rank) {
        const int count = W * per_rank, remote = count - per_rank;
        const int g     = std::max(1, std::min((remote + 4095) / 4096, 1024));
        launch_k(ctx, p2pdev::k_pull, g, 256, 0, ctx->ctl, dst, reinterpret_cast<const double*>(P.base + land_off), count, ctx->rank * per_rank, per_rank,
                 flags, W, kind, P.epoch, P.fault, P.push_dev + kind);
      };
      // (halo exchange: only the ranges this rank's rows / columns reference were stored into its block, and only they are copied)
      auto pull_ranges = [&](int kind, double* dst, size_t land_off) {
        p2pdev::PullRanges R{};
        int most = 0;
        for (int q = 0; q < W; ++q) R.off[q] = ctx->halo.recv_off[kind][q], R.cnt[q] = ctx->halo.recv_cnt[kind][q], most = std::max(most, R.cnt[q]);
        launch_k(ctx, p2pdev::k_pull_ranges, std::max(1, std::min((most + 255) / 256, 64)), 256, 0, ctx->ctl, dst, reinterpret_cast<const double*>(P.base + land_off), R, flags, W,
                 kind, P.epoch, P.fault, P.push_dev + kind);
      };
      if (ctx->halo.on) pull_ranges(0, ctx->xbar, P.off_x);
      else pull(0, ctx->xbar, P.off_x, ctx->slice);  // (this rank's slice: k_primal's ordinary store, above)
      launch_a_dual(ctx, ctx->ygather + (size_t)ctx->rank * ctx->ypad, P.push_dev + 1);
      if (ctx->halo.on) pull_ranges(1, ctx->ygather, P.off_y);
      else pull(1, ctx->ygather, P.off_y, ctx->ypad);
      launch_oc_step(ctx);
      launch_k(ctx, k_step_decision_p2p, 1, kBlock, 0, ctx->ctl, ctx->part_a, dual_partials(ctx), ctx->part_oc, oc_partials(ctx),
               reinterpret_cast<const double*>(P.base + P.off_s), flags, W, P.epoch, P.fault, ctx->sp, P.push_dev + 2);
      LAUNCH_CHECK();
      return 0;
    }
    if (ctx->halo.on) TRY(halo_exchange(ctx, 0, ctx->xbar));  // (a structured LP: the ranges the rows reference, from their owners)
    else TRY(all_gather(ctx, ctx->xbar, (size_t)ctx->slice));
    launch_a_dual(ctx, ctx->ygather + (size_t)ctx->rank * ctx->ypad);
    LAUNCH_CHECK();
    if (ctx->halo.on) TRY(halo_exchange(ctx, 1, ctx->ygather));
    else TRY(all_gather(ctx, ctx->ygather, (size_t)ctx->ypad));
    launch_oc_step(ctx);
    launch_k(ctx, k_pack_step_sums, 1, kBlock, 0, ctx->part_a, dual_partials(ctx), ctx->part_oc, oc_partials(ctx), ctx->rs_scal);
    LAUNCH_CHECK();
    TRY(allreduce(ctx, ctx->rs_scal, 3, rccl::kSum));
    launch_k(ctx, k_step_decision, 1, kDecisionThreads, 0, ctx->ctl, nullptr, 0, ctx->rs_scal + 1, 1, ctx->rs_scal, ctx->sp);
    LAUNCH_CHECK();
    return 0;
  }
  if (ctx->rsag) {
    // sliced primal: primal step on this rank's columns -> all-gather(xbar) -> local rows of A -> partial A^T y' ->
    // reduce-scatter -> this rank's columns of A^T y' and of the step-size sums -> ONE 3-scalar all-reduce -> the same
    // decision on every rank.  Same wire bytes as the all-reduce of the replicated dataflow, 1/world of its element-wise work.
    const size_t cs = (size_t)ctx->rank * ctx->slice;
    const int len   = (int)std::max<int64_t>(0, std::min<int64_t>(ctx->slice, (int64_t)n - (int64_t)cs));
    launch_k(ctx, k_primal, grid_for(len), kBlock, 0, len, ctx->ctl, ctx->x[0] + cs, ctx->x[1] + cs, ctx->aty[0] + cs, ctx->aty[1] + cs,
             ctx->c + cs, ctx->lb + cs, ctx->ub + cs, ctx->xbar + cs, ctx->sumx + cs, (const p2pdev::Push*)nullptr, ctx->ubd);
    LAUNCH_CHECK();
    TRY(all_gather(ctx, ctx->xbar, (size_t)ctx->slice));
    launch_a_dual(ctx);
    launch_at_cur(ctx, ctx->ar_buf, 1);
    LAUNCH_CHECK();
    TRY(reduce_scatter(ctx, ctx->ar_buf, ctx->rs_buf, (size_t)ctx->slice));
    const int g = std::min(grid_for(len), kGenericBlocks);
    launch_k(ctx, k_step_stats, g, kBlock, 0, len, g, ctx->ctl, ctx->rs_buf, ctx->x[0] + cs, ctx->x[1] + cs, ctx->aty[0] + cs, ctx->aty[1] + cs, ctx->part_g);
    launch_k(ctx, k_pack_step_sums, 1, kBlock, 0, ctx->part_a, dual_partials(ctx), ctx->part_g, g, ctx->rs_scal);
    LAUNCH_CHECK();
    TRY(allreduce(ctx, ctx->rs_scal, 3, rccl::kSum));
    launch_k(ctx, k_step_decision, 1, kDecisionThreads, 0, ctx->ctl, nullptr, 0, ctx->rs_scal + 1, 1, ctx->rs_scal, ctx->sp);
    LAUNCH_CHECK();
    return 0;
  }
  launch_k(ctx, k_primal, grid_for(n), kBlock, 0, n, ctx->ctl, ctx->x[0], ctx->x[1], ctx->aty[0], ctx->aty[1], ctx->c, ctx->lb, ctx->ub, ctx->xbar, ctx->sumx, (const p2pdev::Push*)nullptr, ctx->ubd);
  launch_a_dual(ctx);
  if (!ctx->comm) {
    launch_at_step(ctx);
    launch_decision(ctx);
  } else {
    // partial A^T y' of this row block -> ar_buf[0..n), ||dy||^2 partial -> ar_buf[n]; ONE all-reduce
    launch_at_cur(ctx, ctx->ar_buf, 1);
    launch_k(ctx, k_sum_partials_to, 1, kBlock, 0, ctx->part_a, dual_partials(ctx), ctx->ar_buf + n);
    LAUNCH_CHECK();
    TRY(allreduce(ctx, ctx->ar_buf, (size_t)n + 1, rccl::kSum));
    const int g = std::min(grid_for(n), kGenericBlocks);
    launch_k(ctx, k_step_stats, g, kBlock, 0, n, g, ctx->ctl, ctx->ar_buf, ctx->x[0], ctx->x[1], ctx->aty[0], ctx->aty[1], ctx->part_g);
    launch_k(ctx, k_step_decision, 1, kDecisionThreads, 0, ctx->ctl, nullptr, 0, ctx->part_g, g, ctx->ar_buf + n, ctx->sp);
  }
  LAUNCH_CHECK();
  return 0;
}

static int get_graph(pdlpdev_ctx* ctx, int attempts, hipGraphExec_t* out)
{
  auto it = ctx->graphs.find(attempts);
  if (it != ctx->graphs.end()) {
    *out = it->second;
    return 0;
  }
  hipGraph_t graph;
  HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int i = 0; i < attempts && rc == 0; ++i) rc = enqueue_attempt(ctx);
  hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
  if (rc != 0) return rc;
  HIP_TRY(e);
  hipGraphExec_t exec;
  HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIP_TRY(hipGraphDestroy(graph));
  ctx->graphs[attempts] = exec;
  *out                  = exec;
  return 0;
}

int pdlpdev_run(pdlpdev_ctx* ctx, int32_t target_steps, pdlpdev_ctl* ctl)
{
  roctx::Range range("pdlp: PDHG attempts");
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->small_resident && !ctx->comm) {  // the whole loop inside one workgroup (kernels_resident.hip)
    TRY(resident_run(ctx, target_steps));
    if (ctl) *ctl = *ctx->ctl_h;
    return 0;
  }
  k_set_target<<<1, 1, 0, ctx->stream>>>(ctx->ctl, target_steps);
  LAUNCH_CHECK();
  TRY(fetch_ctl(ctx, nullptr));
  // Each attempt accepts at most one step, so `remaining` attempts can never overshoot; rejected
  // attempts are made up for in the next round (one control-block read per round, not per step).
  int guard = 0;
  while (ctx->ctl_h->error == 0 && ctx->ctl_h->steps_taken < target_steps) {
    int remaining = target_steps - ctx->ctl_h->steps_taken;
    // Sharded solves over RCCL replay attempt graphs too: the collectives are captured with the kernels around them (round 4:
    // tools/rccl_capture_repro.cpp captures ncclAllGather + ncclAllReduce in every capture mode with both RCCL builds of this image,
    // and bench.py --force-comm runs the owner dataflow through captured graphs at one rank; round 3's crash inside the capture did
    // not reproduce after the prune).  A capture that fails is remembered and the solver goes on with plain launches -- the same
    // sequence of collectives, so ranks that took different paths still meet.  The in-process communicator synchronises on the
    // host and can never be captured; the direct peer transport is kernels only.
    const bool rccl_graphs = ctx->comm && !ctx->p2p.on && !ctx->soft && !ctx->graph_comm_failed;
    bool replayed = false;
    if (rccl_graphs && ctx->use_graph && !ctx->comm_warm) {
      // the very first attempt of a solver over RCCL goes out as plain launches: whatever a collective sets up lazily on its first call
      // (channels, proxies) must not happen inside a stream capture
      TRY(enqueue_attempt(ctx));
      ctx->comm_warm = true;
      remaining -= 1;
    }
    if (ctx->use_graph && (!ctx->comm || ctx->p2p.on || rccl_graphs)) {
      replayed = true;
      while (remaining > 0) {
        int chunk = 1;
        while (chunk * 2 <= remaining && chunk < 64) chunk *= 2;
        hipGraphExec_t g;
        const int grc = get_graph(ctx, chunk, &g);
        if (grc != 0 && rccl_graphs) {  // (nothing of this chunk was enqueued: a failed capture leaves the stream empty)
          ctx->graph_comm_failed = true;
          (void)hipGetLastError();
          replayed = false;
          break;
        }
        TRY(grc);
        HIP_TRY(hipGraphLaunch(g, ctx->stream));
        remaining -= chunk;
      }
    }
    if (!replayed) {
      for (int i = 0; i < remaining; ++i) TRY(enqueue_attempt(ctx));
    }
    const int before = ctx->ctl_h->steps_taken, asked = target_steps - before;
    TRY(fetch_ctl(ctx, nullptr));
    if (ctx->p2p.on && ctx->ctl_h->error != 0) {  // a wait of the direct peer transport ran out of patience: a peer is gone
      int fault = 0;
      HIP_TRY(hipMemcpy(&fault, ctx->p2p.fault, sizeof(int), hipMemcpyDeviceToHost));
      if (fault) return fail(-6, "direct peer transport: rank %d waited 5 s for another rank's data (a peer failed or is stuck)", ctx->rank);
    }
    // every rejection shrinks the step size; 64 in a row leave nothing of it: report it like the reference's invalid step
    // size instead of re-enqueueing forever.  Counted across rounds AND calls (a call asks for at most one major-iteration
    // period: 40 steps under Stable2, fewer than 64)
    ctx->rejected_in_a_row = ctx->ctl_h->steps_taken == before ? ctx->rejected_in_a_row + asked : 0;
    if (ctx->ctl_h->error == 0 && ctx->rejected_in_a_row >= 64) {
      ctx->rejected_in_a_row = 0;
      k_set_error<<<1, 1, 0, ctx->stream>>>(ctx->ctl);
      LAUNCH_CHECK();
      TRY(fetch_ctl(ctx, nullptr));
    }
    if (++guard > 100000) return fail(-6, "pdlpdev_run: no progress");
  }
  if (ctx->rsag && guard > 0) {
    // back to replicated primal vectors for everything outside the loop (major iterations, snapshots, the solution): the
    // current x, its A^T y and the running sum are complete on their owners' slices only
    const int cur = ctx->ctl_h->cur;
    TRY(all_gather(ctx, ctx->x[cur], (size_t)ctx->slice));
    TRY(all_gather(ctx, ctx->aty[cur], (size_t)ctx->slice));
    TRY(all_gather(ctx, ctx->sumx, (size_t)ctx->slice));
  }
  if (ctl) *ctl = *ctx->ctl_h;
  return 0;
}

// The reference accepts host OR device arrays at the C API (cuopt_c.cpp:119,261-403 copy with raft::copy).  The
// front end (plain C++, no HIP header) asks here: device / managed memory is copied down, anything else -- also when no
// HIP device or runtime is usable -- is an ordinary host pointer.
int pdlpdev_copy_in(void* dst, const void* src, size_t bytes)
{
  if (bytes == 0) return 0;
  if (!dst || !src) return fail(-1, "pdlpdev_copy_in: null pointer");
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  const hipError_t e = hipPointerGetAttributes(&attr, src);
  if (e == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged)) {
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
  }
  (void)hipGetLastError();  // "invalid value" for plain host memory is the normal case
  memcpy(dst, src, bytes);
  return 0;
}

void pdlpdev_range_push(const char* name)
{
  roctx::load();
  if (roctx::Push) roctx::Push(name);
}
void pdlpdev_range_pop(void)
{
  if (roctx::Pop) roctx::Pop();
}

int pdlpdev_prepare_graphs(pdlpdev_ctx* ctx)
{
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->use_graph || (ctx->comm && !ctx->p2p.on) || ctx->small_resident) return 0;
  for (int chunk = 1; chunk <= 64; chunk *= 2) {
    hipGraphExec_t g;
    TRY(get_graph(ctx, chunk, &g));
  }
  return 0;
}

int pdlpdev_get_ctl(pdlpdev_ctx* ctx, pdlpdev_ctl* ctl)
{
  HIP_TRY(hipSetDevice(ctx->device));
  return fetch_ctl(ctx, ctl);
}
int pdlpdev_clear_error(pdlpdev_ctx* ctx)
{
  HIP_TRY(hipSetDevice(ctx->device));
  k_clear_error<<<1, 1, 0, ctx->stream>>>(ctx->ctl);
  LAUNCH_CHECK();
  return 0;
}
int pdlpdev_set_graph_mode(pdlpdev_ctx* ctx, int use_graph)
{
  ctx->use_graph = use_graph;
  return 0;
}

// ---- results ----------------------------------------------------------------------------------------
int pdlpdev_get_solution(pdlpdev_ctx* ctx, int which, double* x, double* y, double* rc)
{
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(fetch_ctl(ctx, nullptr));
  const int cur = ctx->ctl_h->cur;
  hipStream_t s = ctx->stream;
  if (which == PDLPDEV_BEST) {
    if (!ctx->bestx) return fail(-1, "pdlpdev_get_solution(BEST): nothing was saved");
    if (x) {
      k_unscale<<<grid_for(ctx->n), kBlock, 0, s>>>(ctx->n, ctx->bestx, ctx->dc, ctx->tmp_n);
      HIP_TRY(hipMemcpyAsync(x, ctx->tmp_n, (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    if (y) {
      k_unscale<<<grid_for(ctx->m), kBlock, 0, s>>>(ctx->m, ctx->besty, ctx->dr, ctx->tmp_m);
      HIP_TRY(hipMemcpyAsync(y, ctx->tmp_m, (size_t)ctx->m * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    if (rc) HIP_TRY(hipMemcpyAsync(rc, ctx->bestrc, (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToHost, s));
    LAUNCH_CHECK();
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
  }
  if (x) {
    k_unscale<<<grid_for(ctx->n), kBlock, 0, s>>>(ctx->n, which == PDLPDEV_AVERAGE ? ctx->avgx : ctx->x[cur], ctx->dc, ctx->tmp_n);
    HIP_TRY(hipMemcpyAsync(x, ctx->tmp_n, (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (y) {
    k_unscale<<<grid_for(ctx->m), kBlock, 0, s>>>(ctx->m, which == PDLPDEV_AVERAGE ? ctx->avgy : ctx->y[cur], ctx->dr, ctx->tmp_m);
    HIP_TRY(hipMemcpyAsync(y, ctx->tmp_m, (size_t)ctx->m * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (rc)
    HIP_TRY(hipMemcpyAsync(rc, ctx->rc[which == PDLPDEV_AVERAGE ? 1 : 0], (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToHost, s));
  LAUNCH_CHECK();
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

static int64_t locate_buffer(pdlpdev_ctx* ctx, int id, double** ptr)
{
  if (hipSetDevice(ctx->device) != hipSuccess) return -2;
  if (fetch_ctl(ctx, nullptr) != 0) return -2;
  const int cur = ctx->ctl_h->cur;
  double* src   = nullptr;
  int64_t count = 0;
  const int64_t n = ctx->n, m = ctx->m, nnz = ctx->nnz;
  switch (id) {
    case PDLPDEV_BUF_X: src = ctx->x[cur], count = n; break;
    case PDLPDEV_BUF_Y: src = ctx->y[cur], count = m; break;
    case PDLPDEV_BUF_X_OTHER: src = ctx->x[cur ^ 1], count = n; break;
    case PDLPDEV_BUF_Y_OTHER: src = ctx->y[cur ^ 1], count = m; break;
    case PDLPDEV_BUF_ATY: src = ctx->aty[cur], count = n; break;
    case PDLPDEV_BUF_ATY_OTHER: src = ctx->aty[cur ^ 1], count = n; break;
    case PDLPDEV_BUF_XBAR: src = ctx->xbar, count = n; break;
    case PDLPDEV_BUF_SUM_X: src = ctx->sumx, count = n; break;
    case PDLPDEV_BUF_SUM_Y: src = ctx->sumy, count = m; break;
    case PDLPDEV_BUF_AVG_X: src = ctx->avgx, count = n; break;
    case PDLPDEV_BUF_AVG_Y: src = ctx->avgy, count = m; break;
    case PDLPDEV_BUF_DROW: src = ctx->dr, count = m; break;
    case PDLPDEV_BUF_DCOL: src = ctx->dc, count = n; break;
    case PDLPDEV_BUF_A_VALUES: src = ctx->a_val, count = nnz; break;
    case PDLPDEV_BUF_AT_VALUES: src = ctx->at_val, count = nnz; break;
    case PDLPDEV_BUF_C: src = ctx->c, count = n; break;
    case PDLPDEV_BUF_LB: src = ctx->lb, count = n; break;
    case PDLPDEV_BUF_UB: src = ctx->ub, count = n; break;
    case PDLPDEV_BUF_LO: src = ctx->lo, count = m; break;
    case PDLPDEV_BUF_HI: src = ctx->hi, count = m; break;
    case PDLPDEV_BUF_RC_CURRENT: src = ctx->rc[0], count = n; break;
    case PDLPDEV_BUF_RC_AVERAGE: src = ctx->rc[1], count = n; break;
    case PDLPDEV_BUF_LAST_RESTART_X: src = ctx->lrx, count = n; break;
    case PDLPDEV_BUF_LAST_RESTART_Y: src = ctx->lry, count = m; break;
    default: fail(-1, "unknown buffer %d", id); return -1;
  }
  *ptr = src;
  return count;
}

int64_t pdlpdev_download(pdlpdev_ctx* ctx, int id, void* host, int64_t max_elements)
{
  double* src   = nullptr;
  int64_t count = locate_buffer(ctx, id, &src);
  if (count < 0) return count;
  count = std::min(count, max_elements);
  if (count > 0) {
    if (hipMemcpyAsync(host, src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return -2;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return -2;
  }
  return count;
}

int64_t pdlpdev_upload(pdlpdev_ctx* ctx, int id, const void* host, int64_t elements)
{
  double* dst   = nullptr;
  int64_t count = locate_buffer(ctx, id, &dst);
  if (count < 0) return count;
  if (id == PDLPDEV_BUF_A_VALUES || id == PDLPDEV_BUF_AT_VALUES) {
    fail(-1, "matrix values cannot be overwritten");
    return -1;
  }
  count = std::min(count, elements);
  if ((id == PDLPDEV_BUF_LB && ctx->ubd.lb_same) || (id == PDLPDEV_BUF_UB && ctx->ubd.ub_same)) {
    // bounds written behind the solver's back: the arrays are read again (and the attempt graphs, which carry the flags by value,
    // are captured again)
    if (id == PDLPDEV_BUF_LB) ctx->ubd.lb_same = 0;
    else ctx->ubd.ub_same = 0;
    for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
    ctx->graphs.clear();
  }
  if (count > 0) {
    if (hipMemcpyAsync(dst, host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return -2;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return -2;
  }
  return count;
}

int pdlpdev_set_loop_state(pdlpdev_ctx* ctx, double sum_weights, int32_t its_since_restart, int32_t k)
{
  HIP_TRY(hipSetDevice(ctx->device));
  k_set_loop_state<<<1, 1, 0, ctx->stream>>>(ctx->ctl, sum_weights, its_since_restart, k);
  LAUNCH_CHECK();
  return 0;
}

// ---- measurement / parity hooks ------------------------------------------------------------------------
int pdlpdev_spmv(pdlpdev_ctx* ctx, int transpose, const double* x, double* y)
{
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s  = ctx->stream;
  const int rows = transpose ? ctx->n : ctx->m, cols = transpose ? ctx->m : ctx->n;
  double* in     = transpose ? ctx->tmp_m : ctx->tmp_n;
  double* outv   = transpose ? ctx->tmp_n : ctx->tmp_m;
  HIP_TRY(hipMemcpyAsync(in, x, (size_t)cols * sizeof(double), hipMemcpyHostToDevice, s));
  launch_plain(ctx, transpose, in, outv);
  LAUNCH_CHECK();
  HIP_TRY(hipMemcpyAsync(y, outv, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

int pdlpdev_time_kernel(pdlpdev_ctx* ctx, int kernel_id, int reps, double* avg_ms)
{
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (reps < 1) reps = 1;
  // save what the timed launches may touch: control block and the running sums
  TRY(fetch_ctl(ctx, nullptr));
  pdlpdev_ctl saved = *ctx->ctl_h;
  pdlpdev_ctl forced = saved;
  forced.pending_avg  = 1;                       // time the kernels WITH their averaging traffic
  forced.target_steps = saved.steps_taken + 1;   // and not as no-ops
  forced.error        = 0;
  double *sx = nullptr, *sy = nullptr;
  HIP_TRY(hipMalloc((void**)&sx, std::max<size_t>(ctx->n, 1) * sizeof(double)));
  HIP_TRY(hipMalloc((void**)&sy, std::max<size_t>(ctx->m, 1) * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(sx, ctx->sumx, (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(sy, ctx->sumy, (size_t)ctx->m * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->ctl, &forced, sizeof(forced), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (hipEvent_t& e : ctx->prof_ev) HIP_TRY(hipEventCreate(&e));
  auto one = [&]() {
    switch (kernel_id) {
      case PDLPDEV_K_PRIMAL:
        launch_k(ctx, k_primal, grid_for(ctx->n), kBlock, 0, ctx->n, ctx->ctl, ctx->x[0], ctx->x[1], ctx->aty[0], ctx->aty[1], ctx->c, ctx->lb, ctx->ub, ctx->xbar, ctx->sumx, (const p2pdev::Push*)nullptr, ctx->ubd);
        break;
      case PDLPDEV_K_SPMV_A_DUAL: launch_a_dual(ctx); break;
      case PDLPDEV_K_SPMV_AT_STEP: launch_at_step(ctx); break;
      case PDLPDEV_K_STEP_DECISION: launch_decision(ctx); break;
      case PDLPDEV_K_SPMV_A_PLAIN:
        launch_plain(ctx, 0, ctx->xbar, ctx->tmp_m);
        break;
      case PDLPDEV_K_SPMV_AT_PLAIN:
        launch_plain(ctx, 1, ctx->y[saved.cur], ctx->tmp_n);
        break;
      default: return fail(-1, "pdlpdev_time_kernel: unknown kernel %d", kernel_id);
    }
    return 0;
  };
  // The kernels of an attempt evict each other's streams from L2 and the 256 MiB Infinity Cache; timing one of them
  // in a tight loop of its own would let its matrix sit in those caches and flatter it.  So every repetition enqueues
  // the whole attempt in the solver's order and brackets only the kernel of interest with events.
  const bool loop_kernel = kernel_id == PDLPDEV_K_PRIMAL || kernel_id == PDLPDEV_K_SPMV_A_DUAL ||
                           kernel_id == PDLPDEV_K_SPMV_AT_STEP || kernel_id == PDLPDEV_K_STEP_DECISION;
  auto attempt = [&](bool timed) -> int {
    const int order[4] = {PDLPDEV_K_PRIMAL, PDLPDEV_K_SPMV_A_DUAL, PDLPDEV_K_SPMV_AT_STEP, PDLPDEV_K_STEP_DECISION};
    const int wanted   = kernel_id;
    // a plain SpMV takes the place of its fused twin
    const int slot = loop_kernel ? wanted : (wanted == PDLPDEV_K_SPMV_A_PLAIN ? PDLPDEV_K_SPMV_A_DUAL : PDLPDEV_K_SPMV_AT_STEP);
    int rc = 0;
    for (int id : order) {
      const bool mine = id == slot;
      kernel_id       = mine ? wanted : id;
      ctx->prof_armed = mine && timed;
      if (ctx->prof_armed) ctx->prof_used = 0;
      rc              = one();
      ctx->prof_armed = false;
      if (rc != 0) break;
      if (id == PDLPDEV_K_STEP_DECISION)  // the decision ended the forced step: arm the next repetition
        HIP_TRY(hipMemcpyAsync(ctx->ctl, &forced, sizeof(forced), hipMemcpyHostToDevice, s));
    }
    kernel_id = wanted;
    return rc;
  };
  TRY(attempt(false));  // warm
  float ms = 0.f;
  for (int i = 0; i < reps; ++i) {
    TRY(attempt(true));
    HIP_TRY(hipStreamSynchronize(s));
    for (int q = 0; q < ctx->prof_used; ++q) {  // the launches of the call site: their own durations, added up
      float one_ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&one_ms, ctx->prof_ev[2 * q], ctx->prof_ev[2 * q + 1]));
      ms += one_ms;
    }
  }
  if (avg_ms) *avg_ms = (double)ms / reps;
  // restore
  HIP_TRY(hipMemcpyAsync(ctx->sumx, sx, (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->sumy, sy, (size_t)ctx->m * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->ctl, &saved, sizeof(saved), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (hipEvent_t& e : ctx->prof_ev) (void)hipEventDestroy(e), e = nullptr;
  (void)hipFree(sx), (void)hipFree(sy);
  return 0;
}

int pdlpdev_synchronize(pdlpdev_ctx* ctx)
{
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}
int64_t pdlpdev_device_bytes(pdlpdev_ctx* ctx) { return ctx->bytes; }
int pdlpdev_shard_dataflow(pdlpdev_ctx* ctx) { return !ctx->comm ? 0 : ctx->owner ? 3 : ctx->rsag ? 2 : 1; }
int pdlpdev_shard_transport(pdlpdev_ctx* ctx) { return ctx->p2p.on ? 1 : 0; }
int pdlpdev_shard_wire_bytes(pdlpdev_ctx* ctx, int64_t out[3])
{
  out[0] = ctx->halo.on ? 1 : 0, out[1] = ctx->halo.on ? ctx->halo.bytes : ctx->halo.bytes_allgather, out[2] = ctx->halo.bytes_allgather;
  return 0;
}
int pdlpdev_dense_info(pdlpdev_ctx* ctx, int64_t out[3])
{
  out[0] = ctx->dense.on ? 1 : 0, out[1] = ctx->dense.nseg, out[2] = ctx->dense.nent;
  return 0;
}
int pdlpdev_layout_info(pdlpdev_ctx* ctx, int32_t out[8])
{
  // per matrix: layout (0 CSR stream, 1 slab-major panels, 2 resident single-workgroup loop, 3 jagged rows + LDS column
  // sets), workgroups, slabs (panels) or percent of the global gathers the LDS sets save (jagged)
  out[0] = ctx->pba.on ? 4 : ctx->ja.on ? 3 : ctx->pa.on ? 1 : 0;
  out[1] = ctx->pba.on ? ctx->pba.v.B : ctx->ja.on ? ctx->ja.v.nblk : ctx->pa.on ? ctx->pa.v.W : ctx->a_nb;
  out[2] = ctx->pba.on ? (int)(100.0 * (ctx->pba.pad - 1.0) + 0.5) : ctx->ja.on ? (int)(100.0 * ctx->ja.saving + 0.5) : ctx->pa.on ? ctx->pa.v.S : 1;
  out[3] = ctx->pbat.on ? 4 : ctx->jat.on ? 3 : ctx->pat.on ? 1 : 0;
  out[4] = ctx->pbat.on ? ctx->pbat.v.B : ctx->jat.on ? ctx->jat.v.nblk : ctx->pat.on ? ctx->pat.v.W : ctx->at_nb;
  out[5] = ctx->pbat.on ? (int)(100.0 * (ctx->pbat.pad - 1.0) + 0.5) : ctx->jat.on ? (int)(100.0 * ctx->jat.saving + 0.5) : ctx->pat.on ? ctx->pat.v.S : 1;
  out[6] = ctx->pa.on && ctx->pa.v.seg, out[7] = ctx->pat.on && ctx->pat.v.seg;  // panels: the long-tail variant (row sums by nonzero)
  if (ctx->small_resident) out[0] = out[3] = 2;
  return 0;
}

// FNV-1a of a device array (parity tests: the device-side set-up must produce the host constructions' arrays bit for bit)
static uint64_t checksum_device(pdlpdev_ctx* ctx, const void* dev, size_t bytes)
{
  if (!dev || bytes == 0) return 0;
  std::vector<unsigned char> h(bytes);
  if (hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ~0ull;
  uint64_t f = 1469598103934665603ull;
  for (unsigned char b : h) f = (f ^ b) * 1099511628211ull;
  (void)ctx;
  return f;
}
int pdlpdev_debug_layout_checksums(pdlpdev_ctx* ctx, uint64_t out[16])
{
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 16; ++i) out[i] = 0;
  const size_t nnz = (size_t)ctx->nnz;
  out[0] = checksum_device(ctx, ctx->at_off, ((size_t)ctx->n + 1) * 4);
  out[1] = checksum_device(ctx, ctx->at_idx, nnz * 4);
  out[2] = checksum_device(ctx, ctx->at_val, nnz * 8);
  auto panels = [&](const pdlpdev_ctx::Panels& P, int32_t rows, uint64_t* o) {
    if (!P.on) return;
    const int NP = P.v.NP ? P.v.NP : P.v.W;
    o[0] = checksum_device(ctx, P.v.row0, ((size_t)NP + 1) * 4);
    o[1] = checksum_device(ctx, P.v.tile_ptr, ((size_t)NP * P.v.S + 1) * 4);
    o[2] = P.v.seg ? 0 : checksum_device(ctx, P.v.rowptr, (size_t)P.v.S * ((size_t)rows + NP) * 2);
    o[3] = checksum_device(ctx, P.v.col, (size_t)P.nent * 4);
    o[4] = checksum_device(ctx, P.perm, (size_t)P.nent * 4);
  };
  panels(ctx->pa, ctx->m, out + 3);
  panels(ctx->pat, ctx->n, out + 8);
  auto gather_free = [&](const pdlpdev_ctx::Pb& P, int64_t side_nnz, uint64_t* o) {  // (a side is in ONE layout: the panels' slots)
    if (!P.on) return;
    const PbView& v = P.v;
    o[0] = checksum_device(ctx, P.perm, (size_t)P.np * 4);
    o[1] = checksum_device(ctx, v.lidx, (size_t)P.np * 2);
    o[2] = checksum_device(ctx, v.piece_dst, (size_t)(P.np >> v.gshift) * 4);
    o[4] = checksum_device(ctx, v.wg_e0, ((size_t)v.nwg + 1) * 4) ^ (checksum_device(ctx, v.wg_panel, (size_t)v.nwg * 4) * 3) ^
           (checksum_device(ctx, v.bin_row0, ((size_t)v.B + 1) * 4) * 5) ^ (checksum_device(ctx, v.bin_e0, ((size_t)v.B + 1) * 4) * 7) ^
           (uint64_t)v.S * 17 ^ (uint64_t)v.gshift * 19 ^ (uint64_t)v.panel_shift * 23 ^ (uint64_t)P.p_threads * 29 ^ (uint64_t)v.wide * 31;
    if (v.wide) {  // the slot words and the steps' levels instead of the rows by length and the jagged diagonals
      o[3] = checksum_device(ctx, v.rib, (size_t)P.np * 2) ^ (checksum_device(ctx, v.step_lv, (size_t)(P.np >> 10)) * 3) ^ (uint64_t)v.nser * 37;
      if (v.nser) {
        int32_t ser_nnz = 0;
        (void)hipMemcpy(&ser_nnz, v.ser_eptr + v.nser, sizeof(int32_t), hipMemcpyDeviceToHost);
        o[3] ^= (checksum_device(ctx, v.ser_ptr, ((size_t)v.B + 1) * 4) * 5) ^ (checksum_device(ctx, v.ser_row, (size_t)v.nser * 4) * 7) ^
                (checksum_device(ctx, v.ser_eptr, ((size_t)v.nser + 1) * 4) * 11) ^ (checksum_device(ctx, v.ser_slot, (size_t)ser_nnz * 4) * 13);
      }
      return;
    }
    int32_t groups = 0;
    (void)hipMemcpy(&groups, v.bin_grp + v.B, sizeof(int32_t), hipMemcpyDeviceToHost);
    o[3] = checksum_device(ctx, v.pos, (size_t)side_nnz * 2) ^ (checksum_device(ctx, v.sr, (size_t)v.rows * 4) * 3);
    o[4] ^= (checksum_device(ctx, v.bin_grp, ((size_t)v.B + 1) * 4) * 11) ^ (checksum_device(ctx, v.grp_pos, ((size_t)groups + 1) * 4) * 13);
  };
  gather_free(ctx->pba, ctx->dense.on ? ctx->dense.hot_nnz : ctx->nnz, out + 3);
  gather_free(ctx->pbat, ctx->dense.on ? ctx->hot_nnz_at : ctx->nnz, out + 8);
  auto jag = [&](const pdlpdev_ctx::Jag& J) -> uint64_t {
    if (!J.on) return 0;
    int32_t nsr = 0, nset = 0;
    (void)hipMemcpy(&nsr, J.v.tile_sr + J.v.ngroups, sizeof(int32_t), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&nset, J.v.set_ptr + J.v.nblk, sizeof(int32_t), hipMemcpyDeviceToHost);
    return checksum_device(ctx, J.v.slot, (size_t)J.nent * 2) ^ (checksum_device(ctx, J.perm, (size_t)J.nent * 4) * 3) ^
           (checksum_device(ctx, J.v.row0, ((size_t)J.v.nblk + 1) * 4) * 5) ^ (checksum_device(ctx, J.v.sr, (size_t)nsr * 4) * 7) ^
           (checksum_device(ctx, J.v.tile_e, ((size_t)J.v.ngroups + 1) * 4) * 11) ^ (checksum_device(ctx, J.v.tile_sr, ((size_t)J.v.ngroups + 1) * 4) * 13) ^
           (checksum_device(ctx, J.v.win, (size_t)J.v.nblk * 8) * 17) ^ (checksum_device(ctx, J.v.set_ptr, ((size_t)J.v.nblk + 1) * 4) * 19) ^
           (checksum_device(ctx, J.v.set_col, (size_t)nset * 4) * 23) ^ (checksum_device(ctx, J.v.lr_ptr, ((size_t)J.v.nblk + 1) * 4) * 29) ^
           (checksum_device(ctx, J.v.lr_row, (size_t)J.v.nlong * 4) * 31) ^ (uint64_t)J.v.waves * 37;
  };
  out[13] = jag(ctx->ja), out[14] = jag(ctx->jat);
  out[15] = (uint64_t)ctx->pa.on | (uint64_t)ctx->pat.on << 1 | (uint64_t)ctx->ja.on << 2 | (uint64_t)ctx->jat.on << 3 |
            (uint64_t)(ctx->pa.on && ctx->pa.v.seg) << 4 | (uint64_t)(ctx->pat.on && ctx->pat.v.seg) << 5 | (uint64_t)ctx->pba.on << 6 |
            (uint64_t)ctx->pbat.on << 7;
  return 0;
}

}  // extern "C"
