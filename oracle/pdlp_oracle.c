/*
 * TEST INFRASTRUCTURE ONLY -- see pdlp_oracle.h.  Plain C11 restatement of cuOpt 25.08's PDLP
 * (`LP/` below = /root/reference/cpp/src/linear_programming/).  Unfused on purpose: one loop per
 * reference kernel, so each piece can be compared with one HIP kernel.
 *
 * Rounding contract (so that SpMV parity with the HIP kernels can be bit-exact): compiled with
 * -ffp-contract=off; every row sum is accumulated left-to-right in CSR order starting from 0.0;
 * long reductions (dot products, norms) use fixed 1024-element blocks summed in index order
 * (deterministic for any thread count).
 */
#include "pdlp_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_INF (1.0 / 0.0)

static double now_seconds(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* zero-filled; the pages are touched first by the threads that will work on them (static schedule, like every loop below),
 * so that on a multi-socket box an OpenMP run reads mostly local memory (round-2 review: the baseline stopped scaling at
 * 16 threads because calloc + memcpy had put every array on the first thread's NUMA node) */
static double* dalloc(size_t n)
{
  double* p = (double*)malloc((n ? n : 1) * sizeof(double));
  if (!p) return p;
  const long long cnt = (long long)(n ? n : 1);
#pragma omp parallel for schedule(static) if (cnt > 65536)
  for (long long i = 0; i < cnt; ++i) p[i] = 0.0;
  return p;
}
/* parallel copy of a matrix array by ROWS: the pages of row block b are touched by the thread that multiplies row block b */
static void copy_rows_d(int rows, const int* offsets, const double* src, double* dst)
{
#pragma omp parallel for schedule(static) if (rows > 16384)
  for (int i = 0; i < rows; ++i)
    for (int k = offsets[i]; k < offsets[i + 1]; ++k) dst[k] = src[k];
}
static int* copy_rows_i(int rows, const int* offsets, const int* src)
{
  int* dst = (int*)malloc(sizeof(int) * (size_t)(offsets[rows] > 0 ? offsets[rows] : 1));
#pragma omp parallel for schedule(static) if (rows > 16384)
  for (int i = 0; i < rows; ++i)
    for (int k = offsets[i]; k < offsets[i + 1]; ++k) dst[k] = src[k];
  return dst;
}

/* ------------------------------------------------------------------------------------------ */
/* presets: LP/solve.cu:64-199 ; defaults LP/pdlp_hyper_params.cu:22-80                       */
/* ------------------------------------------------------------------------------------------ */
void orc_hyper_preset(int mode, double* h)
{
  /* Stable2 (default) -- LP/solve.cu:99-130 */
  h[ORC_H_INITIAL_STEP_SIZE_SCALING]                  = 1.0;
  h[ORC_H_RUIZ_ITERATIONS]                            = 10;
  h[ORC_H_DO_POCK_CHAMBOLLE]                          = 1;
  h[ORC_H_DO_RUIZ]                                    = 1;
  h[ORC_H_ALPHA_POCK_CHAMBOLLE]                       = 1.0;
  h[ORC_H_ARTIFICIAL_RESTART_THRESHOLD]               = 0.36;
  h[ORC_H_STEP_SIZE_BEFORE_SCALING]                   = 0;
  h[ORC_H_PRIMAL_WEIGHT_BEFORE_SCALING]               = 0;
  h[ORC_H_PRIMAL_WEIGHT_C_SCALING]                    = 1.0;
  h[ORC_H_PRIMAL_WEIGHT_B_SCALING]                    = 1.0;
  h[ORC_H_MAJOR_ITERATION]                            = 40;
  h[ORC_H_MIN_ITERATION_RESTART]                      = 10;
  h[ORC_H_RESTART_STRATEGY]                           = 1;
  h[ORC_H_NEVER_RESTART_TO_AVERAGE]                   = 0;
  h[ORC_H_REDUCTION_EXPONENT]                         = 0.3;
  h[ORC_H_GROWTH_EXPONENT]                            = 0.6;
  h[ORC_H_PRIMAL_WEIGHT_UPDATE_SMOOTHING]             = 0.5;
  h[ORC_H_SUFFICIENT_REDUCTION]                       = 0.2;
  h[ORC_H_NECESSARY_REDUCTION]                        = 0.8;
  h[ORC_H_PRIMAL_IMPORTANCE]                          = 1.0;
  h[ORC_H_PRIMAL_DISTANCE_SMOOTHING]                  = 0.5;
  h[ORC_H_DUAL_DISTANCE_SMOOTHING]                    = 0.5;
  h[ORC_H_LAST_RESTART_BEFORE_NEW_PRIMAL_WEIGHT]      = 1;
  h[ORC_H_ARTIFICIAL_RESTART_IN_MAIN_LOOP]            = 0;
  h[ORC_H_RESCALE_FOR_RESTART]                        = 1;
  h[ORC_H_UPDATE_PRIMAL_WEIGHT_ON_INITIAL_SOLUTION]   = 0;
  h[ORC_H_UPDATE_STEP_SIZE_ON_INITIAL_SOLUTION]       = 0;
  h[ORC_H_GRADIENTS_ON_FINITE_BOUNDS_AS_RESIDUALS]    = 0;
  h[ORC_H_PROJECT_INITIAL_PRIMAL]                     = 1;
  if (mode == 0) { /* Stable1 -- LP/solve.cu:66-96 */
    h[ORC_H_INITIAL_STEP_SIZE_SCALING]               = 1.6;
    h[ORC_H_RUIZ_ITERATIONS]                         = 1;
    h[ORC_H_ALPHA_POCK_CHAMBOLLE]                    = 1.3;
    h[ORC_H_ARTIFICIAL_RESTART_THRESHOLD]            = 0.5;
    h[ORC_H_PRIMAL_WEIGHT_BEFORE_SCALING]            = 1;
    h[ORC_H_PRIMAL_WEIGHT_C_SCALING]                 = 2.2;
    h[ORC_H_PRIMAL_WEIGHT_B_SCALING]                 = 4.6;
    h[ORC_H_MAJOR_ITERATION]                         = 52;
    h[ORC_H_MIN_ITERATION_RESTART]                   = 0;
    h[ORC_H_REDUCTION_EXPONENT]                      = 0.5;
    h[ORC_H_GROWTH_EXPONENT]                         = 0.9;
    h[ORC_H_PRIMAL_WEIGHT_UPDATE_SMOOTHING]          = 0.3;
    h[ORC_H_NECESSARY_REDUCTION]                     = 0.5;
    h[ORC_H_PRIMAL_IMPORTANCE]                       = 1.8;
    h[ORC_H_PRIMAL_DISTANCE_SMOOTHING]               = 0.6;
    h[ORC_H_DUAL_DISTANCE_SMOOTHING]                 = 0.2;
    h[ORC_H_LAST_RESTART_BEFORE_NEW_PRIMAL_WEIGHT]   = 0;
    h[ORC_H_RESCALE_FOR_RESTART]                     = 0;
    h[ORC_H_GRADIENTS_ON_FINITE_BOUNDS_AS_RESIDUALS] = 1;
    h[ORC_H_PROJECT_INITIAL_PRIMAL]                  = 0;
  } else if (mode == 2) { /* Methodical1 -- LP/solve.cu:133-163 */
    h[ORC_H_RUIZ_ITERATIONS]                         = 5;
    h[ORC_H_ARTIFICIAL_RESTART_THRESHOLD]            = 0.5;
    h[ORC_H_MAJOR_ITERATION]                         = 64;
    h[ORC_H_MIN_ITERATION_RESTART]                   = 0;
    h[ORC_H_RESTART_STRATEGY]                        = 2;
    h[ORC_H_SUFFICIENT_REDUCTION]                    = 0.1;
    h[ORC_H_NECESSARY_REDUCTION]                     = 0.9;
    h[ORC_H_RESCALE_FOR_RESTART]                     = 0;
    h[ORC_H_GRADIENTS_ON_FINITE_BOUNDS_AS_RESIDUALS] = 1;
    h[ORC_H_PROJECT_INITIAL_PRIMAL]                  = 0;
  } else if (mode == 3) { /* Fast1 -- LP/solve.cu:167-197 */
    h[ORC_H_INITIAL_STEP_SIZE_SCALING]               = 0.8;
    h[ORC_H_RUIZ_ITERATIONS]                         = 6;
    h[ORC_H_DO_RUIZ]                                 = 0;
    h[ORC_H_ALPHA_POCK_CHAMBOLLE]                    = 2.0;
    h[ORC_H_ARTIFICIAL_RESTART_THRESHOLD]            = 0.3;
    h[ORC_H_PRIMAL_WEIGHT_BEFORE_SCALING]            = 1;
    h[ORC_H_PRIMAL_WEIGHT_C_SCALING]                 = 1.2;
    h[ORC_H_PRIMAL_WEIGHT_B_SCALING]                 = 1.2;
    h[ORC_H_MAJOR_ITERATION]                         = 76;
    h[ORC_H_MIN_ITERATION_RESTART]                   = 6;
    h[ORC_H_NEVER_RESTART_TO_AVERAGE]                = 1;
    h[ORC_H_REDUCTION_EXPONENT]                      = 0.4;
    h[ORC_H_SUFFICIENT_REDUCTION]                    = 0.3;
    h[ORC_H_NECESSARY_REDUCTION]                     = 0.9;
    h[ORC_H_PRIMAL_IMPORTANCE]                       = 0.8;
    h[ORC_H_PRIMAL_DISTANCE_SMOOTHING]               = 0.8;
    h[ORC_H_DUAL_DISTANCE_SMOOTHING]                 = 0.3;
    h[ORC_H_ARTIFICIAL_RESTART_IN_MAIN_LOOP]         = 1;
    h[ORC_H_GRADIENTS_ON_FINITE_BOUNDS_AS_RESIDUALS] = 1;
    h[ORC_H_PROJECT_INITIAL_PRIMAL]                  = 0;
  }
}

/* tolerances_t defaults: pdlp/solver_settings.hpp:179-188 */
void orc_default_settings(double* s)
{
  for (int i = 0; i < 6; ++i) s[i] = 1e-4;
  s[ORC_S_ITERATION_LIMIT]         = -1;
  s[ORC_S_TIME_LIMIT]              = 0;
  s[ORC_S_PER_CONSTRAINT_RESIDUAL] = 0;
  s[ORC_S_FIRST_PRIMAL_FEASIBLE]   = 0;
  s[ORC_S_NUM_THREADS]             = 0;
  s[ORC_S_INFEASIBILITY_DETECTION] = 0;
  s[ORC_S_STRICT_INFEASIBILITY]    = 0;
  s[ORC_S_PRIMAL_INFEASIBLE_TOL]   = 1e-8; /* solver_settings.cu:83-84 */
  s[ORC_S_DUAL_INFEASIBLE_TOL]     = 1e-8;
  s[ORC_S_PRIMAL_TOLERANCE_FACTOR] = -1.0;
  s[ORC_S_DUAL_TOLERANCE_FACTOR]   = -1.0;
}

/* ------------------------------------------------------------------------------------------ */
/* reductions with a fixed summation tree                                                     */
/* ------------------------------------------------------------------------------------------ */
#define RB 1024
static double blocked_sum2(int n, const double* a, const double* b) /* sum a_i*b_i */
{
  int nb = (n + RB - 1) / RB;
  if (nb <= 0) return 0.0;
  double* part = (double*)malloc(sizeof(double) * (size_t)nb);
#pragma omp parallel for schedule(static)
  for (int blk = 0; blk < nb; ++blk) {
    int s = blk * RB, e = s + RB < n ? s + RB : n;
    double acc = 0.0;
    for (int i = s; i < e; ++i) {
      double p = a[i] * b[i];
      acc      = acc + p;
    }
    part[blk] = acc;
  }
  double tot = 0.0;
  for (int blk = 0; blk < nb; ++blk) tot = tot + part[blk];
  free(part);
  return tot;
}
static double blocked_sum(int n, const double* a)
{
  int nb = (n + RB - 1) / RB;
  if (nb <= 0) return 0.0;
  double* part = (double*)malloc(sizeof(double) * (size_t)nb);
#pragma omp parallel for schedule(static)
  for (int blk = 0; blk < nb; ++blk) {
    int s = blk * RB, e = s + RB < n ? s + RB : n;
    double acc = 0.0;
    for (int i = s; i < e; ++i) acc = acc + a[i];
    part[blk] = acc;
  }
  double tot = 0.0;
  for (int blk = 0; blk < nb; ++blk) tot = tot + part[blk];
  free(part);
  return tot;
}
static double l2norm(int n, const double* a) { return sqrt(blocked_sum2(n, a, a)); }

/* ------------------------------------------------------------------------------------------ */
/* SpMV: stands in for cusparseSpMV(CSR, ALG2) at LP/pdhg.cu:87,124,                          */
/* adaptive_step_size_strategy.cu:278, convergence_information.cu:228,301 (cuSPARSE closed)   */
/* ------------------------------------------------------------------------------------------ */
void orc_spmv(int rows, const int* offsets, const int* indices, const double* values,
              const double* x, double* y)
{
#pragma omp parallel for schedule(static)
  for (int i = 0; i < rows; ++i) {
    double acc = 0.0;
    for (int k = offsets[i]; k < offsets[i + 1]; ++k) {
      double p = values[k] * x[indices[k]];
      acc      = acc + p;
    }
    y[i] = acc;
  }
}

/* explicit transpose kept as its own CSR: cpp/src/mip/problem/problem.cu:277-309
 * (raft csr_transpose -> cusparseCsr2cscEx2: stable, rows ascending inside each column) */
void orc_csr_transpose(int m, int n, const int* offsets, const int* indices, const double* values,
                       int* t_offsets, int* t_indices, double* t_values)
{
  for (int j = 0; j <= n; ++j) t_offsets[j] = 0;
  int nnz = offsets[m];
  for (int k = 0; k < nnz; ++k) t_offsets[indices[k] + 1]++;
  for (int j = 0; j < n; ++j) t_offsets[j + 1] += t_offsets[j];
  int* cur = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int j = 0; j < n; ++j) cur[j] = t_offsets[j];
  for (int i = 0; i < m; ++i)
    for (int k = offsets[i]; k < offsets[i + 1]; ++k) {
      int p        = cur[indices[k]]++;
      t_indices[p] = i;
      t_values[p]  = values[k];
    }
  free(cur);
}

/* ------------------------------------------------------------------------------------------ */
/* initial scaling vectors: LP/initial_scaling_strategy/initial_scaling.cu:36-92 (ctor),      */
/* :94-163 Ruiz, :176-307 Pock-Chambolle ; division rule LP/utils.cuh:122-129                  */
/* ------------------------------------------------------------------------------------------ */
void orc_compute_scaling(int m, int n, const int* offsets, const int* indices,
                         const double* values, const int* t_offsets, const int* t_indices,
                         const double* t_values, const double* hyper, double* d_row,
                         double* d_col)
{
  for (int i = 0; i < m; ++i) d_row[i] = 1.0;
  for (int j = 0; j < n; ++j) d_col[j] = 1.0;
  double* it_row = dalloc((size_t)m);
  double* it_col = dalloc((size_t)n);
  if (hyper[ORC_H_DO_RUIZ] != 0.0) {
    int iters = (int)hyper[ORC_H_RUIZ_ITERATIONS];
    for (int it = 0; it < iters; ++it) {
      /* inf_norm_row_and_col_kernel :94-122 -- both norms from the same snapshot */
      for (int i = 0; i < m; ++i) it_row[i] = 0.0;
      for (int j = 0; j < n; ++j) it_col[j] = 0.0;
      for (int i = 0; i < m; ++i)
        for (int k = offsets[i]; k < offsets[i + 1]; ++k) {
          int j    = indices[k];
          double v = fabs((values[k] * d_row[i]) * d_col[j]);
          if (v > it_row[i]) it_row[i] = v;
          if (v > it_col[j]) it_col[j] = v;
        }
      /* a_divides_sqrt_b_bounded, utils.cuh:122-129 */
      for (int i = 0; i < m; ++i)
        if (it_row[i] > 0.0) d_row[i] = d_row[i] / sqrt(it_row[i]);
      for (int j = 0; j < n; ++j)
        if (it_col[j] > 0.0) d_col[j] = d_col[j] / sqrt(it_col[j]);
    }
  }
  if (hyper[ORC_H_DO_POCK_CHAMBOLLE] != 0.0) {
    double alpha = hyper[ORC_H_ALPHA_POCK_CHAMBOLLE];
    /* rows from A (:176-212), columns from A^T (:215-252): separate kernels in the reference */
    for (int i = 0; i < m; ++i) {
      double acc = 0.0;
      for (int k = offsets[i]; k < offsets[i + 1]; ++k) {
        double v = fabs((values[k] * d_row[i]) * d_col[indices[k]]);
        acc      = acc + pow(v, alpha);
      }
      it_row[i] = acc;
    }
    for (int j = 0; j < n; ++j) {
      double acc = 0.0;
      for (int k = t_offsets[j]; k < t_offsets[j + 1]; ++k) {
        double v = fabs((t_values[k] * d_row[t_indices[k]]) * d_col[j]);
        acc      = acc + pow(v, 2.0 - alpha);
      }
      it_col[j] = acc;
    }
    for (int i = 0; i < m; ++i)
      if (it_row[i] > 0.0) d_row[i] = d_row[i] / sqrt(it_row[i]);
    for (int j = 0; j < n; ++j)
      if (it_col[j] > 0.0) d_col[j] = d_col[j] / sqrt(it_col[j]);
  }
  free(it_row);
  free(it_col);
}

/* ------------------------------------------------------------------------------------------ */
/* element-wise rules: LP/utils.cuh                                                            */
/* ------------------------------------------------------------------------------------------ */
static inline double dmin(double a, double b) { return a < b ? a : b; } /* raft::min */
static inline double dmax(double a, double b) { return a > b ? a : b; } /* raft::max */

/* combine_finite_abs_bounds, utils.cuh:139-148 */
static inline double combine_bounds(double lower, double upper)
{
  double val = 0.0;
  if (isfinite(upper)) val = dmax(val, fabs(upper));
  if (isfinite(lower)) val = dmax(val, fabs(lower));
  return val;
}
/* violation, utils.cuh:165-178 */
static inline double violation(double value, double lower, double upper)
{
  if (value < lower) return lower - value;
  if (value > upper) return value - upper;
  return 0.0;
}
/* bound_value_gradient, utils.cuh:195-202 (first branch is dead code; value==0 -> upper) */
static inline double bound_value_gradient(double value, double lower, double upper)
{
  return value > 0.0 ? lower : upper;
}
/* bound_value_reduced_cost_product, utils.cuh:204-219 */
static inline double bound_value_rc_product(double value, double lower, double upper)
{
  double bound_value = 0.0;
  if (value > 0.0)
    bound_value = lower;
  else if (value < 0.0)
    bound_value = upper;
  return isfinite(bound_value) ? value * bound_value : 0.0;
}

/* ------------------------------------------------------------------------------------------ */
/* convergence information on the UNSCALED problem:                                            */
/* LP/termination_strategy/convergence_information.cu:149-219 and parts :221-422               */
/* ------------------------------------------------------------------------------------------ */
void orc_eval(int m, int n, const int* offsets, const int* indices, const double* values,
              const int* t_offsets, const int* t_indices, const double* t_values, const double* c,
              const double* lo, const double* hi, const double* lb, const double* ub,
              double obj_scale, double obj_offset, int finite_bounds_rule, double rel_primal_tol,
              double rel_dual_tol, const double* x, const double* y, double* rc, double* out)
{
  double* ax   = dalloc((size_t)m);
  double* pres = dalloc((size_t)m);
  double* grad = dalloc((size_t)n);
  double* dres = dalloc((size_t)n);
  double* bv   = dalloc((size_t)(m > n ? m : n));
  /* compute_primal_residual :221-248 */
  orc_spmv(m, offsets, indices, values, x, ax);
  for (int i = 0; i < m; ++i) pres[i] = violation(ax[i], lo[i], hi[i]);
  /* compute_primal_objective :261-284 */
  double pobj = blocked_sum2(n, x, c);
  if (obj_scale != 1.0 || obj_offset != 0.0) pobj = obj_scale * pobj + obj_offset;
  double l2_pres = l2norm(m, pres);
  double linf_p  = 0.0;
  for (int i = 0; i < m; ++i) { /* relative_residual_t, utils.cuh:385-409 ; bcomb = utils.cuh:150 */
    double v = pres[i] - rel_primal_tol * combine_bounds(lo[i], hi[i]);
    if (v > linf_p) linf_p = v;
  }
  double l2_x = l2norm(n, x);
  /* compute_dual_residual :287-320 : grad = c - A^T y (SpMV alpha=-1, beta=1 on a copy of c) */
  orc_spmv(n, t_offsets, t_indices, t_values, y, grad);
  for (int j = 0; j < n; ++j) grad[j] = c[j] + (-1.0) * grad[j];
  /* compute_reduced_cost_from_primal_gradient :369-398 */
  for (int j = 0; j < n; ++j) {
    double b = bound_value_gradient(grad[j], lb[j], ub[j]);
    double r;
    if (!finite_bounds_rule) { /* copy_gradient_if_should_be_reduced_cost, utils.cuh:221-229 */
      if (grad[j] == 0.0)
        r = grad[j];
      else if (fabs(x[j] - b) <= fabs(x[j]))
        r = grad[j];
      else
        r = 0.0;
    } else { /* copy_gradient_if_finite_bounds, utils.cuh:231-239 (Stable2) */
      if (grad[j] == 0.0)
        r = grad[j];
      else if (isfinite(b))
        r = grad[j];
      else
        r = 0.0;
    }
    rc[j]   = r;
    dres[j] = grad[j] - r;
  }
  /* compute_dual_objective :323-366, :401-422 */
  for (int i = 0; i < m; ++i) bv[i] = bound_value_rc_product(y[i], lo[i], hi[i]);
  double dobj = blocked_sum(m, bv);
  for (int j = 0; j < n; ++j) bv[j] = bound_value_rc_product(rc[j], lb[j], ub[j]);
  dobj = dobj + blocked_sum(n, bv);
  if (obj_scale != 1.0 || obj_offset != 0.0) dobj = obj_scale * dobj + obj_offset;
  double l2_dres = l2norm(n, dres);
  double linf_d  = 0.0;
  for (int j = 0; j < n; ++j) { /* rhs for the dual side is c_j itself (signed), :204-208 */
    double v = dres[j] - rel_dual_tol * c[j];
    if (v > linf_d) linf_d = v;
  }
  double l2_y = l2norm(m, y);
  /* compute_remaining_stats_kernel :137-147 */
  out[0] = pobj;
  out[1] = dobj;
  out[2] = fabs(pobj - dobj);
  out[3] = fabs(pobj) + fabs(dobj);
  out[4] = l2_pres;
  out[5] = l2_dres;
  out[6] = l2_x;
  out[7] = l2_y;
  out[8] = linf_p;
  out[9] = linf_d;
  free(ax);
  free(pres);
  free(grad);
  free(dres);
  free(bv);
}


/* ------------------------------------------------------------------------------------------ */
/* infeasibility information: LP/termination_strategy/infeasibility_information.cu            */
/* ------------------------------------------------------------------------------------------ */
static double inf_norm(int n, const double* v) /* my_inf_norm */
{
  double mx = 0.0;
  for (int i = 0; i < n; ++i)
    if (fabs(v[i]) > mx) mx = fabs(v[i]);
  return mx;
}
void orc_eval_infeasibility(int m, int n, const int* offsets, const int* indices, const double* values,
                            const int* t_offsets, const int* t_indices, const double* t_values,
                            const double* c, const double* lo, const double* hi, const double* lb,
                            const double* ub, int finite_bounds_rule, const double* x, const double* y,
                            double* out)
{
  double* ax   = dalloc((size_t)m);
  double* grad = dalloc((size_t)n);
  double* rc   = dalloc((size_t)n);
  double* res  = dalloc((size_t)(m > n ? m : n));
  /* :184-198 primal ray = the primal iterate */
  const double primal_ray_inf_norm = inf_norm(n, x);
  const double inv = primal_ray_inf_norm == 0.0 ? 0.0 : 1.0 / primal_ray_inf_norm; /* DivideCheckZero */
  /* compute_homogenous_primal_residual :225-248 ; homogenous bounds :86-98 (zero_if_is_finite) */
  orc_spmv(m, offsets, indices, values, x, ax);
  for (int i = 0; i < m; ++i) {
    double hl = isfinite(lo[i]) ? 0.0 : lo[i], hu = isfinite(hi[i]) ? 0.0 : hi[i];
    res[i]    = violation(ax[i], hl, hu);
  }
  double max_primal_ray_infeasibility = inf_norm(m, res);
  /* compute_max_violation :250-270 ; max_violation functor utils.cuh:181-193 */
  double primal_ray_max_violation = 0.0;
  for (int j = 0; j < n; ++j) {
    double lm = 0.0;
    if (isfinite(lb[j])) lm = dmax(lm, -x[j]);
    if (isfinite(ub[j])) lm = dmax(lm, x[j]);
    if (lm > primal_ray_max_violation) primal_ray_max_violation = lm;
  }
  /* compute_homogenous_primal_objective :272-292 */
  double primal_ray_linear_objective = blocked_sum2(n, x, c) * inv;
  /* compute_homogenous_dual_residual :294-325 : gradient = -A^T y (c = 0 in the homogenous problem) */
  orc_spmv(n, t_offsets, t_indices, t_values, y, grad);
  for (int j = 0; j < n; ++j) grad[j] = -1.0 * grad[j];
  for (int j = 0; j < n; ++j) { /* compute_reduced_cost_from_primal_gradient :368-394 */
    double b = bound_value_gradient(grad[j], lb[j], ub[j]);
    double r;
    if (!finite_bounds_rule) {
      if (grad[j] == 0.0)
        r = grad[j];
      else if (fabs(x[j] - b) <= fabs(x[j]))
        r = grad[j];
      else
        r = 0.0;
    } else {
      if (grad[j] == 0.0)
        r = grad[j];
      else if (isfinite(b))
        r = grad[j];
      else
        r = 0.0;
    }
    rc[j]  = r;
    res[j] = grad[j] - r;
  }
  double max_dual_ray_infeasibility = inf_norm(n, res);
  /* compute_homogenous_dual_objective :327-366 + reduced cost contribution :396-417 */
  for (int i = 0; i < m; ++i) res[i] = bound_value_rc_product(y[i], lo[i], hi[i]);
  double dual_ray_linear_objective = blocked_sum(m, res);
  for (int j = 0; j < n; ++j) res[j] = bound_value_rc_product(rc[j], lb[j], ub[j]);
  dual_ray_linear_objective = dual_ray_linear_objective + blocked_sum(n, res);
  const double dual_ray_inf_norm = inf_norm(m, y), rc_inf_norm = inf_norm(n, rc);
  /* compute_remaining_stats_kernel :115-172 */
  double scaling = dmax(dual_ray_inf_norm, rc_inf_norm);
  if (scaling < 0.0 || scaling > 0.0) {
    max_dual_ray_infeasibility = max_dual_ray_infeasibility / scaling;
    dual_ray_linear_objective  = dual_ray_linear_objective / scaling;
  } else {
    max_dual_ray_infeasibility = 0.0;
    dual_ray_linear_objective  = 0.0;
  }
  if (primal_ray_inf_norm > 0.0) {
    max_primal_ray_infeasibility =
      dmax(max_primal_ray_infeasibility, primal_ray_max_violation) / primal_ray_inf_norm;
  } else {
    max_primal_ray_infeasibility = 0.0;
    primal_ray_linear_objective  = 0.0;
  }
  out[0] = max_primal_ray_infeasibility;
  out[1] = primal_ray_linear_objective;
  out[2] = max_dual_ray_infeasibility;
  out[3] = dual_ray_linear_objective;
  free(ax), free(grad), free(rc), free(res);
}
/* infeasibility part of check_termination_criteria_kernel, termination_strategy.cu:228-249 */
static int infeasibility_verdict(const double* inf, const double* s)
{
  if (inf[3] > 0.0 && inf[2] / inf[3] <= s[ORC_S_PRIMAL_INFEASIBLE_TOL]) return 2; /* PrimalInfeasible */
  if (inf[1] < 0.0 && inf[0] / -inf[1] <= s[ORC_S_DUAL_INFEASIBLE_TOL]) return 3;  /* DualInfeasible */
  return 6;
}

/* check_termination_criteria_kernel, LP/termination_strategy/termination_strategy.cu:116-250.
 * Returns Optimal(1) / PrimalFeasible(7) / "no termination" encoded as NumericalError(6). */
static int termination_verdict(const double* ev, const double* s, double norm_b, double norm_c)
{
  int optimal_gap = ev[2] <= s[ORC_S_ABS_GAP_TOL] + s[ORC_S_REL_GAP_TOL] * ev[3];
  if (s[ORC_S_PER_CONSTRAINT_RESIDUAL] != 0.0) {
    int pfeas = ev[8] <= s[ORC_S_ABS_PRIMAL_TOL];
    if (ev[9] <= s[ORC_S_ABS_DUAL_TOL] && pfeas && optimal_gap) return 1;
    if (pfeas) return 7;
  } else {
    int pfeas = ev[4] <= s[ORC_S_ABS_PRIMAL_TOL] + s[ORC_S_REL_PRIMAL_TOL] * norm_b;
    if (ev[5] <= s[ORC_S_ABS_DUAL_TOL] + s[ORC_S_REL_DUAL_TOL] * norm_c && pfeas && optimal_gap)
      return 1;
    if (pfeas) return 7;
  }
  return 6;
}

/* kernel_compute_kkt_score, LP/restart_strategy/pdlp_restart_strategy.cu:366-390 */
static double kkt_score(const double* ev, double w)
{
  double w2 = w * w;
  return sqrt(w2 * ev[4] * ev[4] + ev[5] * ev[5] / w2 + ev[2] * ev[2]);
}


/* ------------------------------------------------------------------------------------------ */
/* trust-region restart (Methodical1): LP/restart_strategy/pdlp_restart_strategy.cu            */
/*   run_trust_region_restart :277-364, compute_localized_duality_gaps :982-1030,              */
/*   bound_optimal_objective :1032-1050, solve_bound_constrained_trust_region :1290-1678,      */
/*   gradients / lagrangian :1716-1900.  The restart strategy object is built on the UNSCALED  */
/*   problem (pdlp.cu:99-103) and, with rescale_for_restart = false, sees unscaled iterates.   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  const double *x, *y;      /* the point (unscaled) */
  double pd2, dd2;          /* squared distances to the last restart point */
  double distance;          /* distance_traveled (weighted) */
  double lagrangian, lower, upper, normalized_gap;
} gap_t;

typedef struct {
  double thr, dir, lb, ub, center, w;
} tr_item;
static int tr_cmp(const void* a, const void* b)
{
  double x = ((const tr_item*)a)->thr, y = ((const tr_item*)b)->thr;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* bound_optimal_objective on the unscaled problem; radius = g->distance; wp/wd = norm weights */
static void bound_optimal_objective(int m, int n, const int* off, const int* idx, const double* val,
                                    const int* toff, const int* tidx, const double* tval,
                                    const double* c, const double* lo, const double* hi,
                                    const double* lb, const double* ub, double wp, double wd, gap_t* g)
{
  const int N = n + m;
  double* gx  = dalloc((size_t)n); /* primal gradient c - A^T y          :1716-1737 */
  double* gy  = dalloc((size_t)m); /* dual gradient  subgradient - A x   :1782-1815 */
  double* aty = dalloc((size_t)n);
  double* sub = dalloc((size_t)m);
  orc_spmv(n, toff, tidx, tval, g->y, aty);
  for (int j = 0; j < n; ++j) gx[j] = c[j] + (-1.0) * aty[j];
  orc_spmv(m, off, idx, val, g->x, gy);
  for (int i = 0; i < m; ++i) { /* compute_subgradient_kernel :1739-1780 */
    double lower = lo[i], upper = hi[i], pp = gy[i], yi = g->y[i], sc;
    if (yi < 0.0)
      sc = upper;
    else if (yi > 0.0)
      sc = lower;
    else if (!isfinite(upper) && !isfinite(lower))
      sc = 0.0;
    else if (!isfinite(upper) && isfinite(lower))
      sc = lower;
    else if (isfinite(upper) && !isfinite(lower))
      sc = upper;
    else
      sc = pp < lower ? lower : (pp > upper ? upper : pp);
    sub[i] = sc;
    gy[i]  = sc - gy[i];
  }
  /* compute_lagrangian_value :1817-1900 : c.x - x.(A^T y) + y.subgradient */
  g->lagrangian = (blocked_sum2(n, g->x, c) - blocked_sum2(n, g->x, aty)) + blocked_sum2(m, g->y, sub);

  /* solve_bound_constrained_trust_region :1391-1678 */
  double* xtr = dalloc((size_t)n);
  double* ytr = dalloc((size_t)m);
  double obj2 = 0.0;
  for (int j = 0; j < n; ++j) obj2 += gx[j] * gx[j];
  for (int i = 0; i < m; ++i) obj2 += gy[i] * gy[i];
  if (g->distance == 0.0 || sqrt(obj2) == 0.0) {
    memcpy(xtr, g->x, sizeof(double) * (size_t)n);
    memcpy(ytr, g->y, sizeof(double) * (size_t)m);
  } else {
    tr_item* it   = (tr_item*)calloc((size_t)N, sizeof(tr_item));
    double* udir  = dalloc((size_t)N); /* unsorted_direction_full_ */
    for (int k = 0; k < N; ++k) {
      double center, obj, lower, upper, wt;
      if (k < n) {
        center = g->x[k], obj = gx[k], lower = lb[k], upper = ub[k], wt = wp;
      } else {
        int i  = k - n; /* objective = -dual gradient ; transformed bounds utils.cuh:242-255 */
        center = g->y[i], obj = -gy[i], wt = wd;
        lower  = isfinite(hi[i]) ? -ORC_INF : 0.0;
        upper  = isfinite(lo[i]) ? ORC_INF : 0.0;
      }
      it[k].center = center, it[k].lb = lower, it[k].ub = upper, it[k].w = wt;
      it[k].dir = 0.0, it[k].thr = 0.0;
      /* compute_direction_and_threshold, utils.cuh:291-322 */
      if (center >= upper && obj <= 0.0) continue;
      if (center <= lower && obj >= 0.0) continue;
      if (obj == 0.0) {
        it[k].thr = ORC_INF;
        continue;
      }
      it[k].dir = -obj / wt;
      if (it[k].dir > 0.0)
        it[k].thr = (upper - center) / it[k].dir;
      else if (it[k].dir < 0.0)
        it[k].thr = (lower - center) / it[k].dir;
    }
    double high_r2 = 0.0, low_r2 = 0.0; /* weighted_l2_if_infinite, utils.cuh:325-341 */
    for (int k = 0; k < N; ++k) {
      udir[k] = it[k].dir;
      if (isinf(it[k].thr)) high_r2 += it[k].dir * it[k].dir * it[k].w;
    }
    qsort(it, (size_t)N, sizeof(tr_item), tr_cmp);
    int lowi = 0, highi = N;
    for (int k = N - 1; k >= 0; --k)
      if (it[k].thr == -ORC_INF) {
        lowi = k + 1;
        break;
      }
    for (int k = 0; k < N; ++k)
      if (it[k].thr == ORC_INF) {
        highi = k;
        break;
      }
    const double target = g->distance;
    while (lowi != highi) { /* solve_bound_constrained_trust_region_kernel :1290-1356 */
      int size    = highi - lowi;
      double test = (size & 1) == 0 ? 0.5 * (it[lowi + size / 2 - 1].thr + it[lowi + size / 2].thr)
                                    : it[lowi + size / 2].thr;
      double test_r2 = 0.0;
      for (int k = lowi; k < highi; ++k) {
        double tp = dmin(dmax(it[k].center + test * it[k].dir, it[k].lb), it[k].ub);
        double d  = tp - it[k].center;
        test_r2 += (d * d) * it[k].w;
      }
      int too_high = low_r2 + test_r2 + (test * test) * high_r2 >= target * target;
      if (too_high) {
        int nh = highi;
        for (int k = lowi; k < highi; ++k)
          if (it[k].thr >= test) {
            nh = k;
            break;
          }
        for (int k = nh; k < highi; ++k) high_r2 += (it[k].dir * it[k].dir) * it[k].w;
        highi = nh;
      } else {
        int nl = lowi;
        for (int k = highi - 1; k >= lowi; --k)
          if (it[k].thr <= test) {
            nl = k + 1;
            break;
          }
        for (int k = lowi; k < nl; ++k) {
          double tp = dmin(dmax(it[k].center + test * it[k].dir, it[k].lb), it[k].ub);
          double d  = tp - it[k].center;
          low_r2 += (d * d) * it[k].w;
        }
        lowi = nl;
      }
    }
    double tthr; /* target_threshold_determination_kernel :1109-1128 */
    if (high_r2 <= 0.0) {
      tthr = it[0].thr;
      for (int k = 1; k < N; ++k)
        if (it[k].thr > tthr) tthr = it[k].thr;
    } else {
      tthr = sqrt(((target * target) - low_r2) / high_r2);
    }
    for (int j = 0; j < n; ++j) xtr[j] = dmin(dmax(g->x[j] + tthr * udir[j], lb[j]), ub[j]);
    for (int i = 0; i < m; ++i) {
      double lower = isfinite(hi[i]) ? -ORC_INF : 0.0, upper = isfinite(lo[i]) ? ORC_INF : 0.0;
      ytr[i]       = dmin(dmax(g->y[i] + tthr * udir[n + i], lower), upper);
    }
    free(it), free(udir);
  }
  /* compute_bound :1052-1076 */
  double lsum = 0.0, usum = 0.0;
  for (int j = 0; j < n; ++j) lsum += (xtr[j] - g->x[j]) * gx[j];
  for (int i = 0; i < m; ++i) usum += (ytr[i] - g->y[i]) * gy[i];
  g->lower = lsum + g->lagrangian;
  g->upper = usum + g->lagrangian;
  free(gx), free(gy), free(aty), free(sub), free(xtr), free(ytr);
}

void orc_trust_region_bounds(int m, int n, const int* offsets, const int* indices, const double* values,
                             const int* t_offsets, const int* t_indices, const double* t_values,
                             const double* c, const double* lo, const double* hi, const double* lb,
                             const double* ub, double wp, double wd, double radius, const double* x,
                             const double* y, double* out)
{
  gap_t g;
  g.x = x, g.y = y, g.pd2 = g.dd2 = 0.0, g.distance = radius;
  bound_optimal_objective(m, n, offsets, indices, values, t_offsets, t_indices, t_values, c, lo, hi, lb, ub, wp,
                          wd, &g);
  out[0] = g.lagrangian, out[1] = g.lower, out[2] = g.upper;
}

/* ------------------------------------------------------------------------------------------ */
/* the solver                                                                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int m, n;
  const int *offsets, *indices, *t_offsets, *t_indices;
  double *A, *At;          /* scaled values      */
  const double *Au, *Atu;  /* unscaled values    */
  double *c, *lo, *hi, *lb, *ub;             /* scaled   */
  const double *cu, *lou, *hiu, *lbu, *ubu;  /* unscaled (internal min-form c) */
} prob_t;

static void fill_stats(double* stats, const double* ev, int status, int steps, int attempts,
                       double norm_b, double norm_c)
{
  stats[ORC_O_STATUS]                 = status;
  stats[ORC_O_STEPS_TAKEN]            = steps;
  stats[ORC_O_ATTEMPTED_STEPS]        = attempts;
  stats[ORC_O_PRIMAL_OBJECTIVE]       = ev[0];
  stats[ORC_O_DUAL_OBJECTIVE]         = ev[1];
  stats[ORC_O_GAP]                    = ev[2];
  stats[ORC_O_RELATIVE_GAP]           = ev[2] / (1.0 + ev[3]); /* convergence_information.cu:474 */
  stats[ORC_O_L2_PRIMAL_RESIDUAL]     = ev[4];
  stats[ORC_O_L2_DUAL_RESIDUAL]       = ev[5];
  stats[ORC_O_L2_REL_PRIMAL_RESIDUAL] = ev[4] / (1.0 + norm_b);
  stats[ORC_O_L2_REL_DUAL_RESIDUAL]   = ev[5] / (1.0 + norm_c);
}

int orc_pdlp_solve(int m, int n, const int* offsets, const int* indices, const double* values,
                   const double* c_user, const double* lo, const double* hi, const double* lb,
                   const double* ub, int maximize, double obj_offset, const double* H,
                   const double* S, const double* init_x, const double* init_y, double* x_out,
                   double* y_out, double* rc_out, double* stats)
{
  double t_start = now_seconds();
#ifdef _OPENMP
  if (S[ORC_S_NUM_THREADS] > 0) omp_set_num_threads((int)S[ORC_S_NUM_THREADS]);
#endif
  for (int i = 0; i < ORC_O_COUNT; ++i) stats[i] = 0.0;
  const int nnz = offsets[m];
  /* run_pdlp_solver: n_constraints == 0 -> NumericalError, LP/solve.cu:355-359 */
  if (m == 0) {
    stats[ORC_O_STATUS] = 6;
    return 0;
  }
  /* --- detail::problem_t, cpp/src/mip/problem/problem.cu:53-93 --- */
  double obj_scale = 1.0;
  double* cu       = dalloc((size_t)n);
  for (int j = 0; j < n; ++j) cu[j] = c_user[j];
  if (maximize) { /* problem_helpers.cuh:126-141 */
    for (int j = 0; j < n; ++j) cu[j] = -cu[j];
    obj_scale = -obj_scale;
  }
  int* t_offsets = (int*)calloc((size_t)n + 1, sizeof(int));
  int* t_indices = (int*)calloc((size_t)(nnz ? nnz : 1), sizeof(int));
  double* Atu    = dalloc((size_t)nnz);
  orc_csr_transpose(m, n, offsets, indices, values, t_offsets, t_indices, Atu);
  double* bcomb_u = dalloc((size_t)m); /* combined_bounds, utils.cuh:150-163 */
  for (int i = 0; i < m; ++i) bcomb_u[i] = combine_bounds(lo[i], hi[i]);
  /* convergence_information_t ctor :76-84 : constants of the termination test */
  /* (overridable: set_relative_{dual,primal}_tolerance_factor, pdlp.cu:209-231 / convergence_information.cu:110-123) */
  const double norm_c = S[ORC_S_DUAL_TOLERANCE_FACTOR] >= 0.0 ? S[ORC_S_DUAL_TOLERANCE_FACTOR] : l2norm(n, cu);
  const double norm_b = S[ORC_S_PRIMAL_TOLERANCE_FACTOR] >= 0.0 ? S[ORC_S_PRIMAL_TOLERANCE_FACTOR] : l2norm(m, bcomb_u);

  /* --- scaled copy (op_problem_scaled_, pdlp.cu:60-61) and scaling vectors (ctor) --- */
  prob_t P;
  P.m = m, P.n = n, P.offsets = offsets, P.indices = indices, P.t_offsets = t_offsets,
  P.t_indices = t_indices;
  P.Au = values, P.Atu = Atu, P.cu = cu, P.lou = lo, P.hiu = hi, P.lbu = lb, P.ubu = ub;
  P.A  = dalloc((size_t)nnz);
  P.At = dalloc((size_t)nnz);
  P.c = dalloc((size_t)n), P.lb = dalloc((size_t)n), P.ub = dalloc((size_t)n);
  P.lo = dalloc((size_t)m), P.hi = dalloc((size_t)m);
  copy_rows_d(m, offsets, values, P.A);
  copy_rows_d(n, t_offsets, Atu, P.At);
  /* the loop's column-index streams, placed like the values they go with */
  int* idx_local   = copy_rows_i(m, offsets, indices);
  int* t_idx_local = copy_rows_i(n, t_offsets, t_indices);
  memcpy(P.c, cu, sizeof(double) * (size_t)n);
  memcpy(P.lb, lb, sizeof(double) * (size_t)n);
  memcpy(P.ub, ub, sizeof(double) * (size_t)n);
  memcpy(P.lo, lo, sizeof(double) * (size_t)m);
  memcpy(P.hi, hi, sizeof(double) * (size_t)m);
  double* Dr = dalloc((size_t)m);
  double* Dc = dalloc((size_t)n);
  orc_compute_scaling(m, n, offsets, indices, values, t_offsets, t_indices, Atu, H, Dr, Dc);

  /* --- state: saddle_point_state_t (saddle_point.cu:26-64) zero-initialised --- */
  double *x = dalloc((size_t)n), *xn = dalloc((size_t)n), *dx = dalloc((size_t)n),
         *xbar = dalloc((size_t)n), *aty = dalloc((size_t)n), *atyn = dalloc((size_t)n),
         *tmpn = dalloc((size_t)n);
  double *y = dalloc((size_t)m), *yn = dalloc((size_t)m), *dy = dalloc((size_t)m),
         *ax = dalloc((size_t)m);
  double *sumx = dalloc((size_t)n), *sumy = dalloc((size_t)m);
  double sumw  = 0.0;
  double *avgx = dalloc((size_t)n), *avgy = dalloc((size_t)m);
  double *lrx = dalloc((size_t)n), *lry = dalloc((size_t)m); /* last_restart_duality_gap_ */
  double *rc_cur = dalloc((size_t)n), *rc_avg = dalloc((size_t)n);
  double *bcomb = dalloc((size_t)m);
  double ev_cur[10], ev_avg[10];
  double step_size = H[ORC_H_INITIAL_STEP_SIZE_SCALING], w = 0.0, tau = 0.0, sigma = 0.0;
  int k_dev = 0, total_pdhg = 0; /* d_total_pdhg_iterations_ / total_pdhg_iterations_ */
  int total_pdlp = 0, internal_it = 0, its_since_restart = 0, last_restart_was_average = 0;
  double last_candidate_kkt = 0.0, last_restart_kkt = 0.0;
  /* gap_reduction_ratio_last_trial_ is an UNINITIALISED device scalar in the reference
   * (pdlp_restart_strategy.cu:160); the published algorithm (PDLP, FirstOrderLp.jl) starts it at 1 */
  double gap_ratio_last_trial = 1.0;
  int valid_step_size = 0, num_restarts = 0, rc = 0;
  const int major     = (int)H[ORC_H_MAJOR_ITERATION];
  const int min_it    = (int)H[ORC_H_MIN_ITERATION_RESTART];
  const int rule_fin  = H[ORC_H_GRADIENTS_ON_FINITE_BOUNDS_AS_RESIDUALS] == 0.0;
  const double itlim  = S[ORC_S_ITERATION_LIMIT];
  const double tlim   = S[ORC_S_TIME_LIMIT];

  /* compute_initial_step_size, pdlp.cu:1224-1258 (eltwiseDivideCheckZero: /0 -> 0) */
#define INIT_STEP_SIZE()                                   \
  do {                                                     \
    double mx = 0.0;                                       \
    for (int k = 0; k < nnz; ++k)                          \
      if (fabs(P.A[k]) > mx) mx = fabs(P.A[k]);            \
    step_size = mx == 0.0 ? 0.0 : step_size / mx;          \
  } while (0)
  /* compute_initial_primal_weight, pdlp.cu:1260-1309 ; weighted norm utils.cuh:365-383 */
#define INIT_PRIMAL_WEIGHT()                                                      \
  do {                                                                            \
    for (int i = 0; i < m; ++i) bcomb[i] = combine_bounds(P.lo[i], P.hi[i]);      \
    double bn = sqrt(H[ORC_H_PRIMAL_WEIGHT_B_SCALING] * blocked_sum2(m, bcomb, bcomb)); \
    double cn = sqrt(H[ORC_H_PRIMAL_WEIGHT_C_SCALING] * blocked_sum2(n, P.c, P.c));     \
    if (bn > 0.0 && cn > 0.0)                                                     \
      w = H[ORC_H_PRIMAL_IMPORTANCE] * (cn / bn);                                 \
    else                                                                          \
      w = H[ORC_H_PRIMAL_IMPORTANCE];                                             \
  } while (0)

  /* ---------------- run_solver, pdlp.cu:984-1075 ---------------- */
  if (H[ORC_H_STEP_SIZE_BEFORE_SCALING] != 0.0) INIT_STEP_SIZE();
  if (H[ORC_H_PRIMAL_WEIGHT_BEFORE_SCALING] != 0.0) INIT_PRIMAL_WEIGHT();
  /* scale_problem, initial_scaling.cu:347-408 (A and A^T scaled by separate kernels :310-345) */
  for (int i = 0; i < m; ++i)
    for (int k = offsets[i]; k < offsets[i + 1]; ++k) P.A[k] = P.A[k] * Dr[i] * Dc[indices[k]];
  for (int j = 0; j < n; ++j)
    for (int k = t_offsets[j]; k < t_offsets[j + 1]; ++k)
      P.At[k] = P.At[k] * Dc[j] * Dr[t_indices[k]];
  for (int j = 0; j < n; ++j) {
    P.c[j]  = P.c[j] * Dc[j];
    P.lb[j] = P.lb[j] / Dc[j];
    P.ub[j] = P.ub[j] / Dc[j];
  }
  for (int i = 0; i < m; ++i) {
    P.lo[i] = P.lo[i] * Dr[i];
    P.hi[i] = P.hi[i] * Dr[i];
  }
  /* (x, y are zero here; scale_solutions of zeros is a no-op) */
  if (H[ORC_H_STEP_SIZE_BEFORE_SCALING] == 0.0) INIT_STEP_SIZE();
  if (H[ORC_H_PRIMAL_WEIGHT_BEFORE_SCALING] == 0.0) INIT_PRIMAL_WEIGHT();
  stats[ORC_O_INITIAL_STEP_SIZE]     = step_size;
  stats[ORC_O_INITIAL_PRIMAL_WEIGHT] = w;
  /* get_primal_and_dual_stepsizes, adaptive_step_size_strategy.cu:347-368 */
  tau   = step_size / w;
  sigma = step_size * w;
  /* update_primal_dual_solutions, pdlp.cu:857-981.  The two update_* hyper-parameters are off in every preset; the
   * reference's initial_solution_test (pdlp_test.cu:245-523) toggles them and pins the behaviour: nothing changes without
   * BOTH iterates, or when one of them is all zero; both non-zero -> new step size / new primal weight. */
  if (init_x || init_y) {
    if (init_x)
      for (int j = 0; j < n; ++j) x[j] = init_x[j];
    if (init_y)
      for (int i = 0; i < m; ++i) y[i] = init_y[i];
    if (H[ORC_H_UPDATE_STEP_SIZE_ON_INITIAL_SOLUTION] != 0.0) { /* pdlp.cu:878-948 */
      int nzx = 0, nzy = 0;
      for (int j = 0; j < n; ++j) nzx |= x[j] != 0.0;
      for (int i = 0; i < m; ++i) nzy |= y[i] != 0.0;
      if (init_x && init_y && nzx && nzy) {
        /* delta_primal = x0, delta_dual = potential_next_dual = y0, current A^T y = 0 (:905-920); scaled first (:923-931)
         * unless the step size is computed before scaling -- then the vectors as they came meet the scaled matrix and
         * are scaled afterwards (:940-947; the deltas are overwritten by the first step either way) */
        const int before = H[ORC_H_STEP_SIZE_BEFORE_SCALING] != 0.0;
        for (int j = 0; j < n; ++j) dx[j] = before ? x[j] : x[j] / Dc[j];
        for (int i = 0; i < m; ++i) dy[i] = before ? y[i] : y[i] / Dr[i];
        orc_spmv(n, t_offsets, t_indices, P.At, dy, atyn);
        /* compute_step_sizes with the device iteration counter incremented by the kernel
         * (adaptive_step_size_strategy.cu:91-188: *pdhg_iteration += 1 is NOT undone by the host's --) */
        const double inter = blocked_sum2(n, dx, atyn), dx2 = blocked_sum2(n, dx, dx), dy2 = blocked_sum2(m, dy, dy);
        const double movement = H[ORC_H_PRIMAL_DISTANCE_SMOOTHING] * w * dx2 + (H[ORC_H_DUAL_DISTANCE_SMOOTHING] / w) * dy2;
        if (movement <= 0.0 || movement >= 1.0e100) {
          valid_step_size = -1;
        } else {
          k_dev += 1;
          const double coef  = (double)k_dev;
          const double limit = fabs(inter) > 0.0 ? movement / fabs(inter) : ORC_INF;
          const double s1    = (1.0 - pow(coef + 1.0, -H[ORC_H_REDUCTION_EXPONENT])) * limit;
          const double s2    = (1.0 + pow(coef + 1.0, -H[ORC_H_GROWTH_EXPONENT])) * step_size;
          step_size          = dmin(s1, s2);
          tau                = step_size / w;
          sigma              = step_size * w;
        }
        for (int j = 0; j < n; ++j) atyn[j] = 0.0;
      }
    }
    /* the iterate is scaled before the weight update unless the weight is computed before scaling (:950-979) */
    if (H[ORC_H_PRIMAL_WEIGHT_BEFORE_SCALING] == 0.0) {
      for (int j = 0; j < n; ++j) x[j] = x[j] / Dc[j]; /* scale_solutions :410-427 */
      for (int i = 0; i < m; ++i) y[i] = y[i] / Dr[i];
    }
    if (H[ORC_H_UPDATE_PRIMAL_WEIGHT_ON_INITIAL_SOLUTION] != 0.0) { /* update_distance, pdlp_restart_strategy.cu:440-465 */
      const double pdist = sqrt(blocked_sum2(n, x, x)), ddist = sqrt(blocked_sum2(m, y, y)); /* anchors are zero */
      memcpy(lrx, x, sizeof(double) * (size_t)n); /* update_last_restart_information (as they are at this point) */
      memcpy(lry, y, sizeof(double) * (size_t)m);
      const double g = 1.0e-10; /* compute_new_primal_weight :684-750 */
      if (!(pdist < 0.0 + g || pdist >= 1.0 / g || ddist < 0.0 + g || ddist >= 1.0 / g)) {
        const double th = H[ORC_H_PRIMAL_WEIGHT_UPDATE_SMOOTHING];
        w               = exp(th * log(ddist / pdist) + (1.0 - th) * log(w));
        tau             = step_size / w;
        sigma           = step_size * w;
      }
    }
    if (H[ORC_H_PRIMAL_WEIGHT_BEFORE_SCALING] != 0.0) {
      for (int j = 0; j < n; ++j) x[j] = x[j] / Dc[j];
      for (int i = 0; i < m; ++i) y[i] = y[i] / Dr[i];
    }
    stats[ORC_O_INITIAL_STEP_SIZE]     = step_size; /* what get_step_size_h() / get_primal_weight_h() show the test */
    stats[ORC_O_INITIAL_PRIMAL_WEIGHT] = w;
  }
  if (H[ORC_H_PROJECT_INITIAL_PRIMAL] != 0.0) { /* pdlp.cu:1041-1056, clamp utils.cuh:131-137 */
    for (int j = 0; j < n; ++j) x[j] = dmin(dmax(x[j], P.lb[j]), P.ub[j]);
    for (int j = 0; j < n; ++j) avgx[j] = dmin(dmax(avgx[j], P.lb[j]), P.ub[j]);
  }

  double t_loop = now_seconds();
  int final_status = 6, returned_average = 0;
  /* ---------------- main loop, pdlp.cu:1081-1185 ---------------- */
  for (;;) {
    int is_major = ((total_pdlp % major == 0) && total_pdlp > 0) || (total_pdlp <= min_it);
    int error_occured = (valid_step_size == -1);
    int artificial    = 0;
    if (H[ORC_H_ARTIFICIAL_RESTART_IN_MAIN_LOOP] != 0.0)
      artificial = its_since_restart >= H[ORC_H_ARTIFICIAL_RESTART_THRESHOLD] * total_pdlp;
    if (is_major || artificial || error_occured) {
      /* averages: pdlp.cu:1103-1123 ; weighted_average_solution.cu:114-142 */
      if (internal_it <= 1) {
        memcpy(avgx, x, sizeof(double) * (size_t)n);
        memcpy(avgy, y, sizeof(double) * (size_t)m);
      } else if (its_since_restart == 0) {
        memset(avgx, 0, sizeof(double) * (size_t)n);
        memset(avgy, 0, sizeof(double) * (size_t)m);
      } else {
        for (int j = 0; j < n; ++j) avgx[j] = sumx[j] / sumw;
        for (int i = 0; i < m; ++i) avgy[i] = sumy[i] / sumw;
      }
      /* unscale_solutions, initial_scaling.cu:460-484 : in place, as the reference does */
      for (int j = 0; j < n; ++j) avgx[j] = avgx[j] * Dc[j];
      for (int i = 0; i < m; ++i) avgy[i] = avgy[i] * Dr[i];
      for (int j = 0; j < n; ++j) x[j] = x[j] * Dc[j];
      for (int i = 0; i < m; ++i) y[i] = y[i] * Dr[i];

      /* ---- check_termination, pdlp.cu:537-802 ---- */
      orc_eval(m, n, offsets, indices, P.Au, t_offsets, t_indices, P.Atu, cu, lo, hi, lb, ub,
               obj_scale, obj_offset, rule_fin, S[ORC_S_REL_PRIMAL_TOL], S[ORC_S_REL_DUAL_TOL], x,
               y, rc_cur, ev_cur);
      orc_eval(m, n, offsets, indices, P.Au, t_offsets, t_indices, P.Atu, cu, lo, hi, lb, ub,
               obj_scale, obj_offset, rule_fin, S[ORC_S_REL_PRIMAL_TOL], S[ORC_S_REL_DUAL_TOL],
               avgx, avgy, rc_avg, ev_avg);
      int t_cur = termination_verdict(ev_cur, S, norm_b, norm_c);
      int t_avg = termination_verdict(ev_avg, S, norm_b, norm_c);
      if (S[ORC_S_INFEASIBILITY_DETECTION] != 0.0) { /* termination_strategy.cu:80-107,228-249 */
        double inf_cur[4], inf_avg[4];
        orc_eval_infeasibility(m, n, offsets, indices, P.Au, t_offsets, t_indices, P.Atu, cu, lo, hi, lb,
                               ub, rule_fin, x, y, inf_cur);
        orc_eval_infeasibility(m, n, offsets, indices, P.Au, t_offsets, t_indices, P.Atu, cu, lo, hi, lb,
                               ub, rule_fin, avgx, avgy, inf_avg);
        if (t_cur == 6) t_cur = infeasibility_verdict(inf_cur, S);
        if (t_avg == 6) t_avg = infeasibility_verdict(inf_avg, S);
      }
      int done = 0, use_avg = 0, status = 6;
      if (total_pdlp > 1) { /* :580-583 : only limits while it <= 1 */
        if (S[ORC_S_FIRST_PRIMAL_FEASIBLE] != 0.0) { /* :587-633 */
          if (t_avg == 7 && t_cur == 7) {
            done = 1, status = 7, use_avg = !(ev_cur[4] < ev_avg[4]);
          } else if (t_cur == 7) {
            done = 1, status = 7, use_avg = 0;
          } else if (t_avg == 7) {
            done = 1, status = 7, use_avg = 1;
          }
        }
        if (!done && t_avg == 1 && t_cur == 1) { /* :636-682 ties go to the average */
          done = 1, status = 1, use_avg = !(kkt_score(ev_cur, w) < kkt_score(ev_avg, w));
        }
        if (!done && t_avg == 1) done = 1, status = 1, use_avg = 1; /* :685-700 */
        if (!done && t_cur == 1) done = 1, status = 1, use_avg = 0; /* :701-716 */
        /* infeasibility (:718-776): strict -> either iterate alone, else both must agree */
        if (!done && S[ORC_S_INFEASIBILITY_DETECTION] != 0.0) {
          if (S[ORC_S_STRICT_INFEASIBILITY] != 0.0) {
            if (t_cur == 2 || t_cur == 3)
              done = 1, status = t_cur, use_avg = 0;
            else if (t_avg == 2 || t_avg == 3)
              done = 1, status = t_avg, use_avg = 1;
          } else if ((t_cur == 2 && t_avg == 2) || (t_cur == 3 && t_avg == 3)) {
            done = 1, status = t_cur, use_avg = 0;
          }
        }
        if (!done && valid_step_size == -1) { /* :780-789 : empty solution object */
          stats[ORC_O_STATUS] = 6;
          final_status        = 6;
          stats[ORC_O_STEPS_TAKEN]     = internal_it;
          stats[ORC_O_ATTEMPTED_STEPS] = total_pdhg;
          goto finish_noiterate;
        }
      }
      /* check_limits, pdlp.cu:264-331 */
      if (!done) {
        if (tlim > 0.0 && isfinite(tlim) && (now_seconds() - t_start) >= tlim)
          done = 1, status = 5, use_avg = 0;
        else if (itlim >= 0.0 && (double)internal_it >= itlim)
          done = 1, status = 4, use_avg = 0;
      }
      if (done) { /* fill_return_problem_solution, termination_strategy.cu:268-357 */
        const double* ev = use_avg ? ev_avg : ev_cur;
        memcpy(x_out, use_avg ? avgx : x, sizeof(double) * (size_t)n);
        memcpy(y_out, use_avg ? avgy : y, sizeof(double) * (size_t)m);
        memcpy(rc_out, use_avg ? rc_avg : rc_cur, sizeof(double) * (size_t)n);
        fill_stats(stats, ev, status, internal_it, total_pdhg, norm_b, norm_c);
        final_status     = status;
        returned_average = use_avg;
        goto finish;
      }
      /* rescale (pdlp.cu:1144-1149) or only current (:1168-1175) */
      if (H[ORC_H_RESCALE_FOR_RESTART] != 0.0) {
        for (int j = 0; j < n; ++j) avgx[j] = avgx[j] / Dc[j];
        for (int i = 0; i < m; ++i) avgy[i] = avgy[i] / Dr[i];
        for (int j = 0; j < n; ++j) x[j] = x[j] / Dc[j];
        for (int i = 0; i < m; ++i) y[i] = y[i] / Dr[i];
      }
      /* ---- compute_restart -> run_kkt_restart, pdlp_restart_strategy.cu:467-641 ---- */
      if ((int)H[ORC_H_RESTART_STRATEGY] == 2) { /* run_trust_region_restart :277-364 */
        /* the restart strategy is built on the UNSCALED problem (pdlp.cu:99-103), whatever space the iterates are in: with
         * rescale_for_restart (no preset pairs it with this strategy) the scaled iterates and anchors meet the unscaled
         * matrix, objective and bounds below */
        if (its_since_restart != 0) {
          const double wp = tau == 0.0 ? 0.0 : 1.0 / tau, wd = sigma == 0.0 ? 0.0 : 1.0 / sigma;
          const double pds = H[ORC_H_PRIMAL_DISTANCE_SMOOTHING], dds = H[ORC_H_DUAL_DISTANCE_SMOOTHING];
          int do_restart = (its_since_restart >= H[ORC_H_ARTIFICIAL_RESTART_THRESHOLD] * total_pdlp);
          gap_t G[2]; /* 0 = average, 1 = current  (compute_localized_duality_gaps :982-1030) */
          G[0].x = avgx, G[0].y = avgy, G[1].x = x, G[1].y = y;
          for (int q = 0; q < 2; ++q) {
            for (int j = 0; j < n; ++j) tmpn[j] = lrx[j] - 1.0 * G[q].x[j];
            G[q].pd2 = blocked_sum2(n, tmpn, tmpn);
            for (int i = 0; i < m; ++i) ax[i] = lry[i] - 1.0 * G[q].y[i];
            G[q].dd2      = blocked_sum2(m, ax, ax);
            G[q].distance = sqrt(G[q].pd2 * pds * w + G[q].dd2 * (dds / w)); /* :803-817 */
            bound_optimal_objective(m, n, offsets, indices, P.Au, t_offsets, t_indices, P.Atu, cu, lo, hi,
                                    lb, ub, wp, wd, &G[q]);
            G[q].normalized_gap = (G[q].upper - G[q].lower) / G[q].distance;
          }
          /* pick_restart_candidate_kernel :841-856 */
          int to_avg = (G[1].normalized_gap / G[1].distance >= G[0].normalized_gap / G[0].distance);
          gap_t* cand = to_avg ? &G[0] : &G[1];
          if (!do_restart) { /* should_do_adaptive_restart_normalized_duality_gap :905-937 */
            gap_t L;
            L.x = lrx, L.y = lry;
            L.distance = sqrt(cand->pd2 * pds * w + cand->dd2 * (dds / w));
            bound_optimal_objective(m, n, offsets, indices, P.Au, t_offsets, t_indices, P.Atu, cu, lo, hi,
                                    lb, ub, wp, wd, &L);
            L.normalized_gap = (L.upper - L.lower) / L.distance;
            double ratio     = cand->normalized_gap / L.normalized_gap; /* adaptive_restart_triggered :876-903 */
            if (ratio < H[ORC_H_NECESSARY_REDUCTION] &&
                (ratio < H[ORC_H_SUFFICIENT_REDUCTION] || ratio > gap_ratio_last_trial))
              do_restart = 1;
            gap_ratio_last_trial = ratio;
          }
          if (do_restart) {
            ++num_restarts;
            int really_avg = to_avg && H[ORC_H_NEVER_RESTART_TO_AVERAGE] == 0.0;
            if (really_avg) {
              memcpy(x, avgx, sizeof(double) * (size_t)n);
              memcpy(y, avgy, sizeof(double) * (size_t)m);
              last_restart_was_average = 1;
            } else
              last_restart_was_average = 0;
            memcpy(lrx, cand->x, sizeof(double) * (size_t)n); /* update_last_restart_information */
            memcpy(lry, cand->y, sizeof(double) * (size_t)m);
            {
              double pdist = sqrt(cand->pd2), ddist = sqrt(cand->dd2); /* compute_new_primal_weight */
              const double g = 1.0e-10;
              if (!(pdist < 0.0 + g || pdist >= 1.0 / g || ddist < 0.0 + g || ddist >= 1.0 / g)) {
                double th = H[ORC_H_PRIMAL_WEIGHT_UPDATE_SMOOTHING];
                w         = exp(th * log(ddist / pdist) + (1.0 - th) * log(w));
                tau       = step_size / w;
                sigma     = step_size * w;
              }
            }
            memset(sumx, 0, sizeof(double) * (size_t)n);
            memset(sumy, 0, sizeof(double) * (size_t)m);
            sumw              = 0.0;
            its_since_restart = 0;
          }
        }
      }
      if ((int)H[ORC_H_RESTART_STRATEGY] == 1) {
        double cur_score = kkt_score(ev_cur, w);
        if (its_since_restart == 0) { /* :507-514 */
          last_candidate_kkt = cur_score;
          last_restart_kkt   = cur_score;
        } else {
          double avg_score = kkt_score(ev_avg, w);
          double cand;
          int to_avg;
          if (cur_score < avg_score)
            to_avg = 0, cand = cur_score;
          else
            to_avg = 1, cand = avg_score;
          /* kkt_restart_conditions :431-437 = artificial (:939-961) || kkt_decay (:407-429) */
          int do_restart = (its_since_restart >= H[ORC_H_ARTIFICIAL_RESTART_THRESHOLD] * total_pdlp);
          if (!do_restart) {
            if (cand < H[ORC_H_SUFFICIENT_REDUCTION] * last_restart_kkt)
              do_restart = 1;
            else if (cand < H[ORC_H_NECESSARY_REDUCTION] * last_restart_kkt &&
                     cand > last_candidate_kkt)
              do_restart = 1;
          }
          if (do_restart) {
            ++num_restarts;
            int really_avg    = to_avg && H[ORC_H_NEVER_RESTART_TO_AVERAGE] == 0.0;
            const double* zcx = really_avg ? avgx : x;
            const double* zcy = really_avg ? avgy : y;
            /* compute_distance_traveled_from_last_restart :1680-1714, :752-801 */
            for (int j = 0; j < n; ++j) tmpn[j] = lrx[j] - 1.0 * zcx[j];
            double pd2 = blocked_sum2(n, tmpn, tmpn);
            for (int i = 0; i < m; ++i) ax[i] = lry[i] - 1.0 * zcy[i];
            double dd2 = blocked_sum2(m, ax, ax);
            if (really_avg) { /* :593-605 */
              memcpy(x, avgx, sizeof(double) * (size_t)n);
              memcpy(y, avgy, sizeof(double) * (size_t)m);
              last_restart_was_average = 1;
            } else
              last_restart_was_average = 0;
            /* update_last_restart_information :819-839 (only the anchors matter for KKT) and
             * compute_new_primal_weight :684-750 ; their order does not change the result */
            memcpy(lrx, zcx, sizeof(double) * (size_t)n);
            memcpy(lry, zcy, sizeof(double) * (size_t)m);
            {
              double pdist = sqrt(pd2), ddist = sqrt(dd2);
              const double g = 1.0e-10; /* pdlp_constants.hpp:34-35 */
              if (!(pdist < 0.0 + g || pdist >= 1.0 / g || ddist < 0.0 + g || ddist >= 1.0 / g)) {
                double est = ddist / pdist;
                double th  = H[ORC_H_PRIMAL_WEIGHT_UPDATE_SMOOTHING];
                double lw  = th * log(est) + (1.0 - th) * log(w);
                w          = exp(lw);
                tau        = step_size / w;
                sigma      = step_size * w;
              }
            }
            /* reset_weighted_average_solution, weighted_average_solution.cu:51-60 */
            memset(sumx, 0, sizeof(double) * (size_t)n);
            memset(sumy, 0, sizeof(double) * (size_t)m);
            sumw              = 0.0;
            its_since_restart = 0;
            last_restart_kkt  = cand;
          }
          last_candidate_kkt = cand;
        }
      }
      if (H[ORC_H_RESCALE_FOR_RESTART] == 0.0) {
        for (int j = 0; j < n; ++j) x[j] = x[j] / Dc[j];
        for (int i = 0; i < m; ++i) y[i] = y[i] / Dr[i];
      }
    }

    /* ---------------- take_step, pdlp.cu:1187-1222 ---------------- */
    valid_step_size = 0;
    while (valid_step_size == 0) {
      /* compute_next_primal_dual_solution, pdhg.cu:160-235 */
      if (total_pdhg == 0 || (its_since_restart == 0 && last_restart_was_average))
        orc_spmv(n, t_offsets, t_idx_local, P.At, y, aty); /* compute_At_y :119-134 */
#pragma omp parallel for schedule(static) if (n > 16384) /* (small LPs: a fork per loop would cost more than the loop) */
      for (int j = 0; j < n; ++j) { /* primal_projection, utils.cuh:80-95 */
        double gradient = P.c[j] - aty[j];
        double next     = x[j] - (tau * gradient);
        next            = dmax(dmin(next, P.ub[j]), P.lb[j]);
        xn[j]           = next;
        dx[j]           = next - x[j];
        xbar[j]         = next - x[j] + next;
      }
      orc_spmv(m, offsets, idx_local, P.A, xbar, ax); /* pdhg.cu:87-97 */
#pragma omp parallel for schedule(static) if (m > 16384)
      for (int i = 0; i < m; ++i) {                 /* dual_projection, utils.cuh:97-112 */
        double next = y[i] - (sigma * ax[i]);
        double low  = next + sigma * P.lo[i];
        double up   = next + sigma * P.hi[i];
        next        = dmax(low, dmin(up, 0.0));
        yn[i]       = next;
        dy[i]       = next - y[i];
      }
      total_pdhg += 1;
      /* compute_step_sizes, adaptive_step_size_strategy.cu:231-345 then kernel :91-188 */
      orc_spmv(n, t_offsets, t_idx_local, P.At, yn, atyn);
#pragma omp parallel for schedule(static) if (n > 16384)
      for (int j = 0; j < n; ++j) tmpn[j] = atyn[j] - aty[j];
      double interaction = blocked_sum2(n, tmpn, dx);
      double ndx2        = blocked_sum2(n, dx, dx);
      double ndy2        = blocked_sum2(m, dy, dy);
      double movement    = H[ORC_H_PRIMAL_DISTANCE_SMOOTHING] * w * ndx2 +
                        (H[ORC_H_DUAL_DISTANCE_SMOOTHING] / w) * ndy2;
      if (movement <= 0.0 || movement >= 1.0e100) { /* pdlp_constants.hpp:39-47 */
        valid_step_size = -1;
      } else {
        double inter = fabs(interaction);
        k_dev += 1;
        double kc    = (double)k_dev;
        double limit = inter > 0.0 ? movement / inter : ORC_INF;
        if (step_size <= limit) valid_step_size = 1;
        double s1 = (1.0 - pow(kc + 1.0, -H[ORC_H_REDUCTION_EXPONENT])) * limit;
        double s2 = (1.0 + pow(kc + 1.0, -H[ORC_H_GROWTH_EXPONENT])) * step_size;
        step_size = dmin(s1, s2);
        tau       = step_size / w;
        sigma     = step_size * w;
      }
    }
    /* add_current_solution_to_weighted_average_solution, weighted_average_solution.cu:73-108 :
     * the weight is the step size AFTER the update above (pdlp.cu:1216-1220) */
#pragma omp parallel for schedule(static) if (n > 16384)
    for (int j = 0; j < n; ++j) sumx[j] = sumx[j] + step_size * xn[j];
#pragma omp parallel for schedule(static) if (m > 16384)
    for (int i = 0; i < m; ++i) sumy[i] = sumy[i] + step_size * yn[i];
    sumw += step_size;
    its_since_restart += 1;
    /* update_solution, pdhg.cu:237-250 : pointer swaps */
    {
      double* t;
      t = x, x = xn, xn = t;
      t = y, y = yn, yn = t;
      t = aty, aty = atyn, atyn = t;
    }
    ++total_pdlp;
    ++internal_it;
  }

finish:
finish_noiterate:
  stats[ORC_O_FINAL_STEP_SIZE]     = step_size;
  stats[ORC_O_FINAL_PRIMAL_WEIGHT] = w;
  stats[ORC_O_NUM_RESTARTS]        = num_restarts;
  stats[ORC_O_RETURNED_AVERAGE]    = returned_average;
  stats[ORC_O_LOOP_SECONDS]        = now_seconds() - t_loop;
  (void)final_status;
done:
  stats[ORC_O_SOLVE_SECONDS] = now_seconds() - t_start;
  free(cu), free(t_offsets), free(t_indices), free(Atu), free(bcomb_u), free(idx_local), free(t_idx_local);
  free(P.A), free(P.At), free(P.c), free(P.lb), free(P.ub), free(P.lo), free(P.hi);
  free(Dr), free(Dc);
  free(x), free(xn), free(dx), free(xbar), free(aty), free(atyn), free(tmpn);
  free(y), free(yn), free(dy), free(ax), free(sumx), free(sumy), free(avgx), free(avgy);
  free(lrx), free(lry), free(rc_cur), free(rc_avg), free(bcomb);
  return rc;
}

/* innermost PDHG loop with a fixed step: pdhg.cu:72-158 + the A^T y' product the step-size
 * strategy always computes (adaptive_step_size_strategy.cu:278) -- 2 SpMV per iteration */
void orc_pdhg_fixed_steps(int m, int n, const int* offsets, const int* indices,
                          const double* values, const int* t_offsets, const int* t_indices,
                          const double* t_values, const double* c, const double* lo,
                          const double* hi, const double* lb, const double* ub, double tau,
                          double sigma, int iters, double* x, double* y)
{
  double *xbar = dalloc((size_t)n), *aty = dalloc((size_t)n), *ax = dalloc((size_t)m);
  orc_spmv(n, t_offsets, t_indices, t_values, y, aty);
  for (int it = 0; it < iters; ++it) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j) {
      double gradient = c[j] - aty[j];
      double next     = x[j] - (tau * gradient);
      next            = dmax(dmin(next, ub[j]), lb[j]);
      xbar[j]         = next - x[j] + next;
      x[j]            = next;
    }
    orc_spmv(m, offsets, indices, values, xbar, ax);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) {
      double next = y[i] - (sigma * ax[i]);
      double low  = next + sigma * lo[i];
      double up   = next + sigma * hi[i];
      y[i]        = dmax(low, dmin(up, 0.0));
    }
    orc_spmv(n, t_offsets, t_indices, t_values, y, aty);
  }
  free(xbar), free(aty), free(ax);
}
