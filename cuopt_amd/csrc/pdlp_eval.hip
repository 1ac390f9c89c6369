// The major iteration of the device layer (split out of pdlp_device.hip in round 6): average / flush kernels, the fused convergence
// evaluation (convergence_information.cu), infeasibility information (infeasibility_information.cu), the trust-region restart's
// bound search (Methodical1), the KKT restart's device side -- kernels and the pdlpdev_* entry points that enqueue them.
#include "pdlp_ctx.hpp"
#include "pdlp_layouts.hpp"
#include "pdlp_setup.hpp"
#include "pdlp_launch.hpp"
#include "pdlp_core_internal.hpp"

// ================================================================================================
// kernels: major iteration
// ================================================================================================
// mode 0 copy current, 1 zero, 2 sum / sum_weights (weighted_average_solution.cu:114-142)
__global__ void __launch_bounds__(kBlock)
k_make_average(int n, int m, int mode, const pdlpdev_ctl* __restrict__ ctl,
               const double* __restrict__ x0, const double* __restrict__ x1,
               const double* __restrict__ y0, const double* __restrict__ y1,
               const double* __restrict__ sumx, const double* __restrict__ sumy,
               double* __restrict__ avgx, double* __restrict__ avgy)
{
  const int cur = ctl->cur;
  const double* __restrict__ x = cur ? x1 : x0;
  const double* __restrict__ y = cur ? y1 : y0;
  const double sw = ctl->sum_weights;
  const int tot   = n > m ? n : m;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < tot; i += gridDim.x * kBlock) {
    if (i < n) avgx[i] = mode == 0 ? x[i] : (mode == 1 ? 0.0 : sumx[i] / sw);
    if (i < m) avgy[i] = mode == 0 ? y[i] : (mode == 1 ? 0.0 : sumy[i] / sw);
  }
}


// multi-GPU: same per-column rule after the all-reduce of A^T y
__global__ void __launch_bounds__(kBlock)
k_eval_dual_elementwise(int n, int nbg, const pdlpdev_ctl* __restrict__ ctl, int which,
                        const double* __restrict__ x0, const double* __restrict__ x1,
                        const double* __restrict__ avgx, const double* __restrict__ reduced,
                        EvalDualCore core, double* __restrict__ part)
{
  __shared__ double red[20];
  const int cur = ctl->cur;
  core.xhat     = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock)
    core.col(j, reduced[j], acc);
  block_reduce<SumOp, 4>(acc, red);
  if (threadIdx.x == 0)
    for (int q = 0; q < 4; ++q) part[(size_t)q * nbg + blockIdx.x] = acc[q];
}
// max over a vector, clipped at 0 from below (thrust::transform_reduce(max, init 0) in the
// reference, convergence_information.cu:164-208)
__global__ void __launch_bounds__(kBlock)
k_max_partials(int n, const double* __restrict__ v, double* __restrict__ part)
{
  __shared__ double red[8];
  double acc[1] = {0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    acc[0] = v[i] > acc[0] ? v[i] : acc[0];
  block_reduce<MaxOp, 1>(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}


// ---- infeasibility information (infeasibility_information.cu:176-223): the iterate is the ray estimate.
// rows: max_i violation((A x)_i, homogenous bounds), max |y_i|, sum_i B(y_i, lo_i, hi_i)
__global__ void __launch_bounds__(kBlock)
k_infeas_rows(int m, int nbg, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ ax,
              const double* __restrict__ y0, const double* __restrict__ y1, const double* __restrict__ avgy,
              const double* __restrict__ dr, const double* __restrict__ lo_u, const double* __restrict__ hi_u,
              double* __restrict__ part)
{
  __shared__ double red[12];
  const int cur = ctl->cur;
  const double* __restrict__ yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  double mx[2] = {0.0, 0.0}, sm[1] = {0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
    const double lo = lo_u[i], hi = hi_u[i];
    const double hl = dfinite(lo) ? 0.0 : lo, hu = dfinite(hi) ? 0.0 : hi;  // zero_if_is_finite, :86-98
    const double r  = fabs(violation(ax[i], hl, hu));
    const double yi = yv[i] * dr[i];
    mx[0] = r > mx[0] ? r : mx[0];
    mx[1] = fabs(yi) > mx[1] ? fabs(yi) : mx[1];
    sm[0] += bound_value_product(yi, lo, hi);
  }
  block_reduce<MaxOp, 2>(mx, red);
  __syncthreads();
  block_reduce<SumOp, 1>(sm, red + 8);
  if (threadIdx.x == 0) {
    part[blockIdx.x]           = mx[0];
    part[nbg + blockIdx.x]     = mx[1];
    part[2 * nbg + blockIdx.x] = sm[0];
  }
}
// columns: g = -A^T y ; rc by the preset's rule ; max |g - rc|, max |rc|, max |x|, max bound violation of x,
//          sum_j B(rc_j, lb_j, ub_j), c.x
__global__ void __launch_bounds__(kBlock)
k_infeas_cols(int n, int nbg, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ aty,
              const double* __restrict__ x0, const double* __restrict__ x1, const double* __restrict__ avgx,
              const double* __restrict__ dc, const double* __restrict__ c_u, const double* __restrict__ lb_u,
              const double* __restrict__ ub_u, int rule_finite, double* __restrict__ part)
{
  __shared__ double red[28];
  const int cur = ctl->cur;
  const double* __restrict__ xv = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  double mx[4] = {0.0, 0.0, 0.0, 0.0}, sm[2] = {0.0, 0.0};
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
    const double g  = -1.0 * aty[j];
    const double xj = xv[j] * dc[j];
    const double lb = lb_u[j], ub = ub_u[j];
    const double bv = g > 0.0 ? lb : ub;
    double rc;
    if (g == 0.0)
      rc = g;
    else if (rule_finite)
      rc = dfinite(bv) ? g : 0.0;
    else
      rc = fabs(xj - bv) <= fabs(xj) ? g : 0.0;
    const double rd = fabs(g - rc);
    double viol     = 0.0;  // max_violation, utils.cuh:181-193
    if (dfinite(lb)) viol = dmax(viol, -xj);
    if (dfinite(ub)) viol = dmax(viol, xj);
    mx[0] = rd > mx[0] ? rd : mx[0];
    mx[1] = fabs(rc) > mx[1] ? fabs(rc) : mx[1];
    mx[2] = fabs(xj) > mx[2] ? fabs(xj) : mx[2];
    mx[3] = viol > mx[3] ? viol : mx[3];
    sm[0] += bound_value_product(rc, lb, ub);
    sm[1] += c_u[j] * xj;
  }
  block_reduce<MaxOp, 4>(mx, red);
  __syncthreads();
  block_reduce<SumOp, 2>(sm, red + 16);
  if (threadIdx.x == 0) {
    for (int q = 0; q < 4; ++q) part[(size_t)q * nbg + blockIdx.x] = mx[q];
    part[(size_t)4 * nbg + blockIdx.x] = sm[0];
    part[(size_t)5 * nbg + blockIdx.x] = sm[1];
  }
}


// ---- trust-region restart support (Methodical1) -------------------------------------------------------
// Virtual element k of the joint vector z = (x, y) of the UNSCALED problem at point `which`:
//   k <  n : center x_k, objective g_k = c_k - (A^T y)_k, bounds [lb, ub], weight wp
//   k >= n : center y_i, objective -(subgradient_i - (A x)_i), transformed bounds, weight wd
// (solve_bound_constrained_trust_region :1400-1465; compute_subgradient_kernel :1739-1780;
//  compute_direction_and_threshold utils.cuh:291-322; transformed bounds utils.cuh:242-255)
struct TrPoint {
  const double* __restrict__ xhat;
  const double* __restrict__ yhat;
  const double* __restrict__ lrx;
  const double* __restrict__ lry;
  const double* __restrict__ dc;
  const double* __restrict__ dr;
  const double* __restrict__ aty;
  const double* __restrict__ ax;
  const double* __restrict__ c_u;
  const double* __restrict__ lb_u;
  const double* __restrict__ ub_u;
  const double* __restrict__ lo_u;
  const double* __restrict__ hi_u;
  int n, m;
  double wp, wd;
  int first;  // 0, or n on the ranks of a sharded solve that leave the (replicated) primal part to rank 0
};
struct TrElem {
  double center, obj, lb, ub, w, dir, thr, grad, sub;  // grad: primal / dual gradient; sub: dual subgradient
};
__device__ __forceinline__ TrElem tr_element(const TrPoint& P, int k)
{
  TrElem e;
  if (k < P.n) {
    e.center = P.dc ? P.xhat[k] * P.dc[k] : P.xhat[k];
    e.grad   = P.c_u[k] - P.aty[k];
    e.obj    = e.grad;
    e.sub    = 0.0;
    e.lb = P.lb_u[k], e.ub = P.ub_u[k], e.w = P.wp;
  } else {
    const int i      = k - P.n;
    const double yi  = P.dr ? P.yhat[i] * P.dr[i] : P.yhat[i];
    const double lo = P.lo_u[i], hi = P.hi_u[i], pp = P.ax[i];
    double sub;
    if (yi < 0.0)
      sub = hi;
    else if (yi > 0.0)
      sub = lo;
    else if (!dfinite(hi) && !dfinite(lo))
      sub = 0.0;
    else if (!dfinite(hi) && dfinite(lo))
      sub = lo;
    else if (dfinite(hi) && !dfinite(lo))
      sub = hi;
    else
      sub = pp < lo ? lo : (pp > hi ? hi : pp);
    e.center = yi;
    e.sub    = sub;
    e.grad   = sub - pp;
    e.obj    = -e.grad;
    e.lb     = dfinite(hi) ? -__builtin_huge_val() : 0.0;
    e.ub     = dfinite(lo) ? __builtin_huge_val() : 0.0;
    e.w      = P.wd;
  }
  e.dir = 0.0, e.thr = 0.0;
  if (e.center >= e.ub && e.obj <= 0.0) return e;
  if (e.center <= e.lb && e.obj >= 0.0) return e;
  if (e.obj == 0.0) {
    e.thr = __builtin_huge_val();
    return e;
  }
  e.dir = -e.obj / e.w;
  if (e.dir > 0.0)
    e.thr = (e.ub - e.center) / e.dir;
  else if (e.dir < 0.0)
    e.thr = (e.lb - e.center) / e.dir;
  return e;
}
// pass 0: distances to the last-restart anchors, the three Lagrangian dot products, ||objective||^2,
//         sum w dir^2 over everything, largest finite threshold
__global__ void __launch_bounds__(kBlock) k_tr_stats(TrPoint P, int nbg, double* __restrict__ part)
{
  __shared__ double red[40];
  double sm[7] = {0, 0, 0, 0, 0, 0, 0}, mx[1] = {0.0};
  const int N = P.n + P.m;
  for (int k = P.first + blockIdx.x * kBlock + threadIdx.x; k < N; k += gridDim.x * kBlock) {
    const TrElem e = tr_element(P, k);
    if (k < P.n) {
      const double d = P.dc ? (P.lrx[k] - P.xhat[k]) * P.dc[k] : P.lrx[k] - P.xhat[k];
      sm[0] += d * d;
      sm[2] += P.c_u[k] * e.center;
      sm[3] += e.center * P.aty[k];
    } else {
      const int i    = k - P.n;
      const double d = P.dr ? (P.lry[i] - P.yhat[i]) * P.dr[i] : P.lry[i] - P.yhat[i];
      sm[1] += d * d;
      sm[4] += e.center * e.sub;  // y . subgradient
    }
    sm[5] += e.obj * e.obj;
    sm[6] += e.w * e.dir * e.dir;
    if (dfinite(e.thr) && e.thr > mx[0]) mx[0] = e.thr;
  }
  block_reduce<SumOp, 7>(sm, red);
  __syncthreads();
  block_reduce<MaxOp, 1>(mx, red + 32);
  if (threadIdx.x == 0) {
    for (int q = 0; q < 7; ++q) part[(size_t)q * nbg + blockIdx.x] = sm[q];
    part[(size_t)7 * nbg + blockIdx.x] = mx[0];
  }
}
// pass(t): low(t) = sum_{thr <= t} w (clamp(center + t dir) - center)^2 ; high(t) = sum_{thr > t} w dir^2
__global__ void __launch_bounds__(kBlock) k_tr_pass(TrPoint P, double t, int nbg, double* __restrict__ part)
{
  __shared__ double red[12];
  double sm[2] = {0.0, 0.0};
  const int N = P.n + P.m;
  for (int k = P.first + blockIdx.x * kBlock + threadIdx.x; k < N; k += gridDim.x * kBlock) {
    const TrElem e = tr_element(P, k);
    if (e.dir == 0.0) continue;
    if (e.thr <= t) {
      const double tp = dmin(dmax(e.center + t * e.dir, e.lb), e.ub);
      const double d  = tp - e.center;
      sm[0] += (d * d) * e.w;
    } else {
      sm[1] += (e.dir * e.dir) * e.w;
    }
  }
  block_reduce<SumOp, 2>(sm, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x]       = sm[0];
    part[nbg + blockIdx.x] = sm[1];
  }
}
// final: sum g_x (x_tr - x), sum g_y (y_tr - y) with z_tr = clamp(center + t dir)  (compute_bound :1052-1076)
__global__ void __launch_bounds__(kBlock) k_tr_final(TrPoint P, double t, int nbg, double* __restrict__ part)
{
  __shared__ double red[12];
  double sm[2] = {0.0, 0.0};
  const int N = P.n + P.m;
  for (int k = P.first + blockIdx.x * kBlock + threadIdx.x; k < N; k += gridDim.x * kBlock) {
    const TrElem e = tr_element(P, k);
    double tr      = e.center;
    if (e.dir != 0.0) tr = dmin(dmax(e.center + t * e.dir, e.lb), e.ub);
    sm[k < P.n ? 0 : 1] += (tr - e.center) * e.grad;
  }
  block_reduce<SumOp, 2>(sm, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x]       = sm[0];
    part[nbg + blockIdx.x] = sm[1];
  }
}

// restart (restart_block, pdlp_kernels.hpp)
__global__ void __launch_bounds__(kBlock)
k_restart(int n, int m, int which, int unscaled, const double* __restrict__ dc, const double* __restrict__ dr,
          const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ x0,
          double* __restrict__ x1, double* __restrict__ y0, double* __restrict__ y1,
          const double* __restrict__ avgx, const double* __restrict__ avgy, double* __restrict__ lrx,
          double* __restrict__ lry, double* __restrict__ sumx, double* __restrict__ sumy,
          double* __restrict__ part)
{
  __shared__ double red[12];
  const RestartView R{n, m, which, unscaled, dc, dr, ctl, x0, x1, y0, y1, avgx, avgy, lrx, lry, sumx, sumy, part};
  restart_block(R, blockIdx.x, gridDim.x, red);
}

extern "C" {

// ---- major iteration --------------------------------------------------------------------------------
int pdlpdev_flush_average(pdlpdev_ctx* ctx)
{
  HIP_TRY(hipSetDevice(ctx->device));
  k_flush_average<<<grid_for(std::max(ctx->n, ctx->m)), kBlock, 0, ctx->stream>>>(ctx->n, ctx->m, ctx->ctl, ctx->x[0], ctx->x[1], ctx->y[0], ctx->y[1], ctx->sumx, ctx->sumy);
  k_clear_pending<<<1, 1, 0, ctx->stream>>>(ctx->ctl);
  LAUNCH_CHECK();
  return 0;
}
int pdlpdev_make_average(pdlpdev_ctx* ctx, int mode)
{
  HIP_TRY(hipSetDevice(ctx->device));
  k_make_average<<<grid_for(std::max(ctx->n, ctx->m)), kBlock, 0, ctx->stream>>>(ctx->n, ctx->m, mode, ctx->ctl, ctx->x[0], ctx->x[1], ctx->y[0], ctx->y[1], ctx->sumx, ctx->sumy, ctx->avgx, ctx->avgy);
  LAUNCH_CHECK();
  return 0;
}

// the launches of one convergence evaluation; results land in sc[0..9) (layout below), nothing is read back
static int enqueue_eval(pdlpdev_ctx* ctx, int which, int rc_rule_finite_bounds, double eps_rel_primal,
                        double eps_rel_dual, double* sc)
{
  hipStream_t s = ctx->stream;
  const int n = ctx->n, m = ctx->m;
  // LAST_RESTART is evaluated like the average, with the anchors in the "alternative iterate" slots
  const double* altx = which == PDLPDEV_LAST_RESTART ? ctx->lrx : ctx->avgx;
  const double* alty = which == PDLPDEV_LAST_RESTART ? ctx->lry : ctx->avgy;
  const int kw       = which == PDLPDEV_CURRENT ? PDLPDEV_CURRENT : PDLPDEV_AVERAGE;
  // the per-constraint (l-infinity) residuals are only consumed when per_constraint_residual is set: the host
  // driver passes negative eps_rel otherwise and the two extra vectors + four reduction launches are skipped
  const bool want_linf = eps_rel_primal >= 0.0 && eps_rel_dual >= 0.0;
  double* linf_m = want_linf ? ctx->tmp_m : nullptr;
  double* linf_n = want_linf ? ctx->tmp_n : nullptr;
  // layout of sc: [0..2] primal sums, [3] primal linf, [4..7] dual sums, [8] dual linf
  if (kw == PDLPDEV_AVERAGE) dense_part(ctx, 0, altx, nullptr, 0, 0);
  else dense_part(ctx, 0, ctx->x[0], ctx->x[1], 2, 0);
  if (ctx->pba.on) {
    if (kw == PDLPDEV_AVERAGE) TRY(pb_products(ctx, ctx->pba, altx, nullptr, 0, 0));
    else TRY(pb_products(ctx, ctx->pba, ctx->x[0], ctx->x[1], 2, 0));
    TRY(pb_rows(ctx, k_pb_eval_primal, ctx->pba, ctx->ctl, kw, ctx->y[0], ctx->y[1], alty, ctx->dr, ctx->lo_u, ctx->hi_u, eps_rel_primal, linf_m, ctx->ax_u[which], ctx->part_a));
  } else if (ctx->ja.on)
    (void)JAG_LAUNCH(ctx, k_jag_eval_primal, ctx->ja.v, ctx->ctl, kw, ctx->x[0], ctx->x[1], altx, ctx->y[0], ctx->y[1], alty, ctx->dr, ctx->lo_u, ctx->hi_u, eps_rel_primal, linf_m, ctx->ax_u[which], ctx->part_a);
  else if (ctx->pa.on)
    (ctx->pa.v.seg ? k_panel_eval_primal<true> : k_panel_eval_primal<false>)<<<ctx->pa.v.W, kPanelThreads, 0, s>>>(ctx->pa.v, ctx->ctl, kw, ctx->x[0], ctx->x[1], altx, ctx->y[0], ctx->y[1], alty, ctx->dr, ctx->lo_u, ctx->hi_u, eps_rel_primal, linf_m, ctx->ax_u[which], ctx->part_a);
  else
    k_eval_primal<<<stream_grid(ctx->a_nb), kBlock, 0, s>>>(ctx->a_nb, ctx->a_rb, ctx->ha_off, ctx->ha_idx, ctx->ha_val, ctx->ctl, kw, ctx->x[0], ctx->x[1], altx, ctx->y[0], ctx->y[1], alty, ctx->dr, ctx->lo_u, ctx->hi_u, eps_rel_primal, linf_m, ctx->ax_u[which], ctx->part_a, ctx->dense.add_m);
  k_finalize<<<1, kBlock, 0, s>>>(ctx->part_a, dual_partials(ctx), 3, 0u, sc + 0);
  if (want_linf) {
    const int g = std::min(grid_for(m), kGenericBlocks);
    k_max_partials<<<g, kBlock, 0, s>>>(m, ctx->tmp_m, ctx->part_g);
    k_finalize<<<1, kBlock, 0, s>>>(ctx->part_g, g, 1, 1u, sc + 3);
  }
  EvalDualCore core{nullptr, ctx->dc, ctx->c_u, ctx->lb_u, ctx->ub_u, eps_rel_dual, rc_rule_finite_bounds, which == PDLPDEV_LAST_RESTART ? ctx->rc_scratch : ctx->rc[which == PDLPDEV_AVERAGE ? 1 : 0], linf_n, ctx->aty_u[which]};
  if (!ctx->comm) {
    if (kw == PDLPDEV_AVERAGE) dense_part(ctx, 1, alty, nullptr, 0, 0);
    else dense_part(ctx, 1, ctx->y[0], ctx->y[1], 2, 0);
    if (ctx->pbat.on) {
      if (kw == PDLPDEV_AVERAGE) TRY(pb_products(ctx, ctx->pbat, alty, nullptr, 0, 0));
      else TRY(pb_products(ctx, ctx->pbat, ctx->y[0], ctx->y[1], 2, 0));
      TRY(pb_rows(ctx, k_pb_eval_dual, ctx->pbat, ctx->ctl, kw, ctx->x[0], ctx->x[1], altx, core, ctx->part_at));
    } else if (ctx->jat.on)
      (void)JAG_LAUNCH(ctx, k_jag_eval_dual, ctx->jat.v, ctx->ctl, kw, ctx->x[0], ctx->x[1], altx, ctx->y[0], ctx->y[1], alty, core, ctx->part_at);
    else if (ctx->pat.on)
      (ctx->pat.v.seg ? k_panel_eval_dual<true> : k_panel_eval_dual<false>)<<<ctx->pat.v.W, kPanelThreads, 0, s>>>(ctx->pat.v, ctx->ctl, kw, ctx->x[0], ctx->x[1], altx, ctx->y[0], ctx->y[1], alty, core, ctx->part_at);
    else
      k_eval_dual<<<stream_grid(ctx->at_nb), kBlock, 0, s>>>(ctx->at_nb, ctx->at_rb, ctx->hat_off, ctx->hat_idx, ctx->hat_val, ctx->ctl, kw, ctx->x[0], ctx->x[1], altx, ctx->y[0], ctx->y[1], alty, core, ctx->part_at, ctx->dense.add_n);
    k_finalize<<<1, kBlock, 0, s>>>(ctx->part_at, step_partials(ctx), 4, 0u, sc + 4);
  } else {
    // partial A^T y of this row block, all-reduced together with the three dual-side row sums
    if (kw == PDLPDEV_AVERAGE) {
      launch_plain(ctx, 1, alty, ctx->ar_buf);
    } else {
      launch_at_cur(ctx, ctx->ar_buf, 0);
    }
    HIP_TRY(hipMemcpyAsync(ctx->ar_buf + n, sc, 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    TRY(allreduce(ctx, ctx->ar_buf, (size_t)n + 3, rccl::kSum));
    HIP_TRY(hipMemcpyAsync(sc, ctx->ar_buf + n, 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (want_linf) TRY(allreduce(ctx, sc + 3, 1, rccl::kMax));
    const int g = std::min(grid_for(n), kGenericBlocks);
    k_eval_dual_elementwise<<<g, kBlock, 0, s>>>(n, g, ctx->ctl, kw, ctx->x[0], ctx->x[1], altx, ctx->ar_buf, core, ctx->part_g);
    k_finalize<<<1, kBlock, 0, s>>>(ctx->part_g, g, 4, 0u, sc + 4);
  }
  if (want_linf) {
    const int g = std::min(grid_for(n), kGenericBlocks);
    k_max_partials<<<g, kBlock, 0, s>>>(n, ctx->tmp_n, ctx->part_g);
    k_finalize<<<1, kBlock, 0, s>>>(ctx->part_g, g, 1, 1u, sc + 8);
  }
  LAUNCH_CHECK();
  return 0;
}
int pdlpdev_eval(pdlpdev_ctx* ctx, int which, int rc_rule_finite_bounds, double eps_rel_primal,
                 double eps_rel_dual, double out[PDLPDEV_EV_COUNT])
{
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(enqueue_eval(ctx, which, rc_rule_finite_bounds, eps_rel_primal, eps_rel_dual, ctx->scal));
  TRY(fetch_scalars(ctx, 9));
  read_eval(ctx->scal_h, eps_rel_primal >= 0.0 && eps_rel_dual >= 0.0, out);
  return 0;
}

// ---- the head of a major iteration in one go ------------------------------------------------------
// flush the deferred average, form the average iterate, evaluate the current and the average iterate
// (pdlp.cu:1102-1160): one read-back instead of two; for LPs on the resident path one launch instead of ~12.
int pdlpdev_major_eval(pdlpdev_ctx* ctx, int average_mode, int rc_rule_finite_bounds, double eps_rel_primal,
                       double eps_rel_dual, double out_current[PDLPDEV_EV_COUNT], double out_average[PDLPDEV_EV_COUNT])
{
  roctx::Range range("pdlp: major iteration evaluation (averages + convergence information)");
  HIP_TRY(hipSetDevice(ctx->device));
  const bool want_linf = eps_rel_primal >= 0.0 && eps_rel_dual >= 0.0;
  if (ctx->small_resident && !ctx->comm) {
    TRY(resident_major_eval(ctx, average_mode, rc_rule_finite_bounds, want_linf ? 1 : 0, eps_rel_primal, eps_rel_dual));
  } else {
    TRY(pdlpdev_flush_average(ctx));
    TRY(pdlpdev_make_average(ctx, average_mode));
    TRY(enqueue_eval(ctx, PDLPDEV_CURRENT, rc_rule_finite_bounds, eps_rel_primal, eps_rel_dual, ctx->scal));
    TRY(enqueue_eval(ctx, PDLPDEV_AVERAGE, rc_rule_finite_bounds, eps_rel_primal, eps_rel_dual, ctx->scal + 32));
    TRY(fetch_scalars(ctx, 41));
  }
  read_eval(ctx->scal_h, want_linf, out_current);
  read_eval(ctx->scal_h + 32, want_linf, out_average);
  return 0;
}

int pdlpdev_eval_infeasibility(pdlpdev_ctx* ctx, int which, int rc_rule_finite_bounds, double out[4])
{
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int n = ctx->n, m = ctx->m;
  const int gr = std::min(grid_for(m), kGenericBlocks), gc = std::min(grid_for(n), kGenericBlocks);
  double* part_rows = ctx->part_g;             // 3 * gr
  double* part_cols = ctx->part_g + 3 * 2048;  // 6 * gc  (part_g holds 8 * 2048)
  k_infeas_rows<<<gr, kBlock, 0, s>>>(m, gr, ctx->ctl, which, ctx->ax_u[which], ctx->y[0], ctx->y[1], ctx->avgy, ctx->dr, ctx->lo_u, ctx->hi_u, part_rows);
  k_finalize<<<1, kBlock, 0, s>>>(part_rows, gr, 3, 0x3u, ctx->scal + 16);
  if (ctx->comm) {  // the rows are sharded: two maxima and one sum over the row blocks (infeasibility_information.cu:175-223);
    LAUNCH_CHECK();  // the column side below works on replicated vectors (A^T y was all-reduced by the evaluation)
    TRY(allreduce(ctx, ctx->scal + 16, 2, rccl::kMax));
    TRY(allreduce(ctx, ctx->scal + 18, 1, rccl::kSum));
  }
  k_infeas_cols<<<gc, kBlock, 0, s>>>(n, gc, ctx->ctl, which, ctx->aty_u[which], ctx->x[0], ctx->x[1], ctx->avgx, ctx->dc, ctx->c_u, ctx->lb_u, ctx->ub_u, rc_rule_finite_bounds, part_cols);
  k_finalize<<<1, kBlock, 0, s>>>(part_cols, gc, 6, 0xFu, ctx->scal + 24);
  LAUNCH_CHECK();
  HIP_TRY(hipMemcpyAsync(ctx->scal_h + 16, ctx->scal + 16, 16 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const double* r = ctx->scal_h + 16;  // max hom. primal residual, ||y||_inf, sum B(y)
  const double* c = ctx->scal_h + 24;  // max hom. dual residual, ||rc||_inf, ||x||_inf, max violation, sum B(rc), c.x
  // compute_remaining_stats_kernel, infeasibility_information.cu:115-172
  double max_primal = r[0], primal_obj = c[2] == 0.0 ? 0.0 : c[5] * (1.0 / c[2]);
  double max_dual = c[0], dual_obj = r[2] + c[4];
  const double scaling = std::max(r[1], c[1]);
  if (scaling != 0.0) {
    max_dual /= scaling;
    dual_obj /= scaling;
  } else {
    max_dual = 0.0, dual_obj = 0.0;
  }
  if (c[2] > 0.0) {
    max_primal = std::max(max_primal, c[3]) / c[2];
  } else {
    max_primal = 0.0, primal_obj = 0.0;
  }
  out[0] = max_primal, out[1] = primal_obj, out[2] = max_dual, out[3] = dual_obj;
  return 0;
}

int pdlpdev_trust_region_bounds(pdlpdev_ctx* ctx, int which, double wp, double wd, double pds, double dds,
                                double primal_weight, double radius, int scaled_iterates, double out[6])
{
  HIP_TRY(hipSetDevice(ctx->device));
  TRY(fetch_ctl(ctx, nullptr));
  const int cur = ctx->ctl_h->cur;
  hipStream_t s = ctx->stream;
  const double* px = which == PDLPDEV_CURRENT ? ctx->x[cur] : (which == PDLPDEV_AVERAGE ? ctx->avgx : ctx->lrx);
  const double* py = which == PDLPDEV_CURRENT ? ctx->y[cur] : (which == PDLPDEV_AVERAGE ? ctx->avgy : ctx->lry);
  if (scaled_iterates) {
    // rescale_for_restart (pdlp.cu:1144-1149): the restart strategy sees the SCALED iterates, yet its problem is the unscaled
    // one (pdlp.cu:99-103).  This context keeps the scaled matrix only: A v = D_r^-1 (A^ (D_c^-1 v)), A^T v alike.  The
    // next-iterate buffers are free between attempts.
    const int nxt = 1 - cur, n = ctx->n, m = ctx->m;
    k_div_to<<<grid_for(n), kBlock, 0, s>>>(n, ctx->x[nxt], px, ctx->dc);
    k_div_to<<<grid_for(m), kBlock, 0, s>>>(m, ctx->y[nxt], py, ctx->dr);
    launch_plain(ctx, 0, ctx->x[nxt], ctx->ax_u[which]);
    k_div_inplace<<<grid_for(m), kBlock, 0, s>>>(m, ctx->ax_u[which], ctx->dr);
    if (ctx->comm) {
      launch_plain(ctx, 1, ctx->y[nxt], ctx->ar_buf);
      LAUNCH_CHECK();
      TRY(allreduce(ctx, ctx->ar_buf, (size_t)n, rccl::kSum));
      k_div_to<<<grid_for(n), kBlock, 0, s>>>(n, ctx->aty_u[which], ctx->ar_buf, ctx->dc);
    } else {
      launch_plain(ctx, 1, ctx->y[nxt], ctx->aty_u[which]);
      k_div_inplace<<<grid_for(n), kBlock, 0, s>>>(n, ctx->aty_u[which], ctx->dc);
    }
    LAUNCH_CHECK();
  }
  TrPoint P{px, py, ctx->lrx, ctx->lry, scaled_iterates ? nullptr : ctx->dc, scaled_iterates ? nullptr : ctx->dr,
            ctx->aty_u[which], ctx->ax_u[which], ctx->c_u, ctx->lb_u,
            ctx->ub_u, ctx->lo_u, ctx->hi_u, ctx->n, ctx->m, wp, wd, (ctx->comm && ctx->rank != 0) ? ctx->n : 0};
  // sharded: the dual coordinates are this rank's rows, the primal ones are replicated and counted by rank 0 only;
  // every pass ends in a sum (and one max) over the ranks (pdlp_restart_strategy.cu:277-364 works on whole vectors)
  const int g = std::min(grid_for((int64_t)ctx->n + ctx->m), kGenericBlocks);
  k_tr_stats<<<g, kBlock, 0, s>>>(P, g, ctx->part_g);
  k_finalize<<<1, kBlock, 0, s>>>(ctx->part_g, g, 8, 0x80u, ctx->scal + 32);
  LAUNCH_CHECK();
  if (ctx->comm) {
    TRY(allreduce(ctx, ctx->scal + 32, 7, rccl::kSum));
    TRY(allreduce(ctx, ctx->scal + 39, 1, rccl::kMax));
  }
  HIP_TRY(hipMemcpyAsync(ctx->scal_h + 32, ctx->scal + 32, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const double* st = ctx->scal_h + 32;
  const double pd2 = st[0], dd2 = st[1];
  // compute_distance_traveled_last_restart_kernel :803-817
  const double own = sqrt(pd2 * pds * primal_weight + dd2 * (dds / primal_weight));
  const double T   = radius >= 0.0 ? radius : own;
  const double lagrangian = (st[2] - st[3]) + st[4];  // compute_lagrangian_value :1817-1900
  double t = 0.0;
  if (!(T == 0.0 || sqrt(st[5]) == 0.0)) {
    // Monotone fixed point on the breakpoint structure: with the partition of coordinates frozen at t the
    // radius is low(t) + s^2 high(t); its root s = F(t) satisfies t < F(t) <= t* for t < t*, and F(t*) = t*.
    // Start from the unconstrained root; stop when the partition (hence t) no longer changes.
    t = st[6] > 0.0 ? T / sqrt(st[6]) : 0.0;
    for (int it = 0; it < 200; ++it) {
      k_tr_pass<<<g, kBlock, 0, s>>>(P, t, g, ctx->part_g);
      k_finalize<<<1, kBlock, 0, s>>>(ctx->part_g, g, 2, 0u, ctx->scal + 40);
      LAUNCH_CHECK();
      if (ctx->comm) TRY(allreduce(ctx, ctx->scal + 40, 2, rccl::kSum));
      HIP_TRY(hipMemcpyAsync(ctx->scal_h + 40, ctx->scal + 40, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      const double low = ctx->scal_h[40], high = ctx->scal_h[41];
      if (high <= 0.0) {  // everything that moves is at its bound (target_threshold_determination_kernel)
        t = st[7];
        break;
      }
      const double rem = T * T - low;
      const double tn  = rem > 0.0 ? sqrt(rem / high) : t;
      if (!(tn > t)) break;
      t = tn;
    }
  }
  k_tr_final<<<g, kBlock, 0, s>>>(P, t, g, ctx->part_g);
  k_finalize<<<1, kBlock, 0, s>>>(ctx->part_g, g, 2, 0u, ctx->scal + 40);
  LAUNCH_CHECK();
  if (ctx->comm) TRY(allreduce(ctx, ctx->scal + 40, 2, rccl::kSum));
  HIP_TRY(hipMemcpyAsync(ctx->scal_h + 40, ctx->scal + 40, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  out[0] = pd2, out[1] = dd2, out[2] = own, out[3] = lagrangian;
  out[4] = lagrangian + ctx->scal_h[40];
  out[5] = lagrangian + ctx->scal_h[41];
  return 0;
}

int pdlpdev_restart(pdlpdev_ctx* ctx, int which, int unscaled_distances, double dist2[2])
{
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int g = std::min(grid_for(std::max(ctx->n, ctx->m)), kGenericBlocks);
  k_restart<<<g, kBlock, 0, s>>>(ctx->n, ctx->m, which, unscaled_distances, ctx->dc, ctx->dr, ctx->ctl, ctx->x[0], ctx->x[1], ctx->y[0], ctx->y[1], ctx->avgx, ctx->avgy, ctx->lrx, ctx->lry, ctx->sumx, ctx->sumy, ctx->part_g);
  k_finalize<<<1, kBlock, 0, s>>>(ctx->part_g, g, 2, 0u, ctx->scal);
  k_restart_ctl<<<1, 1, 0, s>>>(ctx->ctl);
  LAUNCH_CHECK();
  TRY(allreduce(ctx, ctx->scal + 1, 1, rccl::kSum));
  TRY(fetch_scalars(ctx, 2));
  dist2[0] = ctx->scal_h[0], dist2[1] = ctx->scal_h[1];
  return 0;
}

int pdlpdev_save_best(pdlpdev_ctx* ctx, int which)
{
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->bestx) {
    TRY(dev_alloc(ctx, &ctx->bestx, ctx->n));
    TRY(dev_alloc(ctx, &ctx->besty, ctx->m));
    TRY(dev_alloc(ctx, &ctx->bestrc, ctx->n));
  }
  TRY(fetch_ctl(ctx, nullptr));
  const int cur = ctx->ctl_h->cur;
  hipStream_t s = ctx->stream;
  const bool avg = which == PDLPDEV_AVERAGE;
  HIP_TRY(hipMemcpyAsync(ctx->bestx, avg ? ctx->avgx : ctx->x[cur], (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->besty, avg ? ctx->avgy : ctx->y[cur], (size_t)ctx->m * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->bestrc, ctx->rc[avg ? 1 : 0], (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToDevice, s));
  return 0;
}

}  // extern "C"
