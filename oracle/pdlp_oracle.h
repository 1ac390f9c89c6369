/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's PDLP path.
 *
 * This is NOT product code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load liboracle_pdlp.so, and only as the checker / the reported CPU baseline.  The product
 * (libcuopt.so) never links or calls anything in this directory.
 *
 * What it restates: NVIDIA cuOpt 25.08 `cpp/src/linear_programming` (PDLP), literally and UNFUSED,
 * function by function; every function in pdlp_oracle.c cites the reference file:line it follows.
 *
 * Pinning status: the reference PDLP itself cannot be built here (CUDA/cuSPARSE/raft), so the
 * oracle is pinned against (tests/test_oracle_golden.py):
 *   - every known answer the reference's own tests hold for this path (afiro objective,
 *     Methodical1 initial step size 1.4893 / primal weight 0.0141652, good-max 17, max_offset 0,
 *     ranged LP 32, per-constraint residual 0.1, iteration-limit statuses, 2x1 toy LP ...),
 *   - objectives of the reference's own CPU dual simplex compiled in place (oracle/_ref).
 *   - the SOLUTION VECTOR cuOpt's own PDLP returns on afiro at default settings, held by the reference's
 *     test_lp_solver.py:386-475 (32 values, compared there with rel 1e-4): reproduced to ~1e-9, i.e. the whole
 *     trajectory (scaling, 160 iterations of accept/reject, restarts, averaging, returned iterate) follows cuOpt's,
 *   - an independent solution of the bound-constrained trust-region problem for the Methodical1 restart.
 * Bit-for-bit x_k parity with cuOpt is not attainable (cuSPARSE's reduction order is closed source); parity is
 * pinned at the level of that vector, the initial step / weight, statuses and objectives -- see DESIGN.md "parity".
 */
#ifndef PDLP_ORACLE_H
#define PDLP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* index into the `hyper` array (doubles; booleans are 0/1).  Names follow
 * cpp/include/cuopt/linear_programming/pdlp/pdlp_hyper_params.cuh:20-58 */
enum {
  ORC_H_INITIAL_STEP_SIZE_SCALING = 0,
  ORC_H_RUIZ_ITERATIONS,
  ORC_H_DO_POCK_CHAMBOLLE,
  ORC_H_DO_RUIZ,
  ORC_H_ALPHA_POCK_CHAMBOLLE,
  ORC_H_ARTIFICIAL_RESTART_THRESHOLD,
  ORC_H_STEP_SIZE_BEFORE_SCALING,
  ORC_H_PRIMAL_WEIGHT_BEFORE_SCALING,
  ORC_H_PRIMAL_WEIGHT_C_SCALING,
  ORC_H_PRIMAL_WEIGHT_B_SCALING,
  ORC_H_MAJOR_ITERATION,
  ORC_H_MIN_ITERATION_RESTART,
  ORC_H_RESTART_STRATEGY,
  ORC_H_NEVER_RESTART_TO_AVERAGE,
  ORC_H_REDUCTION_EXPONENT,
  ORC_H_GROWTH_EXPONENT,
  ORC_H_PRIMAL_WEIGHT_UPDATE_SMOOTHING,
  ORC_H_SUFFICIENT_REDUCTION,
  ORC_H_NECESSARY_REDUCTION,
  ORC_H_PRIMAL_IMPORTANCE,
  ORC_H_PRIMAL_DISTANCE_SMOOTHING,
  ORC_H_DUAL_DISTANCE_SMOOTHING,
  ORC_H_LAST_RESTART_BEFORE_NEW_PRIMAL_WEIGHT,
  ORC_H_ARTIFICIAL_RESTART_IN_MAIN_LOOP,
  ORC_H_RESCALE_FOR_RESTART,
  ORC_H_UPDATE_PRIMAL_WEIGHT_ON_INITIAL_SOLUTION,
  ORC_H_UPDATE_STEP_SIZE_ON_INITIAL_SOLUTION,
  ORC_H_GRADIENTS_ON_FINITE_BOUNDS_AS_RESIDUALS,
  ORC_H_PROJECT_INITIAL_PRIMAL,
  ORC_H_COUNT
};

/* index into the `settings` array.  Names follow pdlp/solver_settings.hpp:179-224 */
enum {
  ORC_S_ABS_GAP_TOL = 0,
  ORC_S_REL_GAP_TOL,
  ORC_S_ABS_PRIMAL_TOL,
  ORC_S_REL_PRIMAL_TOL,
  ORC_S_ABS_DUAL_TOL,
  ORC_S_REL_DUAL_TOL,
  ORC_S_ITERATION_LIMIT, /* < 0 : none */
  ORC_S_TIME_LIMIT,      /* <= 0 or inf : none */
  ORC_S_PER_CONSTRAINT_RESIDUAL,
  ORC_S_FIRST_PRIMAL_FEASIBLE,
  ORC_S_NUM_THREADS, /* 0 = leave OpenMP default */
  ORC_S_INFEASIBILITY_DETECTION,
  ORC_S_STRICT_INFEASIBILITY,
  ORC_S_PRIMAL_INFEASIBLE_TOL,
  ORC_S_DUAL_INFEASIBLE_TOL,
  ORC_S_PRIMAL_TOLERANCE_FACTOR, /* < 0: ||b|| (default); else set_relative_primal_tolerance_factor, pdlp.cu:216-220 */
  ORC_S_DUAL_TOLERANCE_FACTOR,   /* < 0: ||c|| (default); else set_relative_dual_tolerance_factor,   pdlp.cu:209-213 */
  ORC_S_COUNT
};

/* index into `stats` (output) -- additional_termination_information_t,
 * pdlp/solver_solution.hpp:63-103, plus the two scalars pdlp_test.cu:237-283 pins */
enum {
  ORC_O_STATUS = 0, /* pdlp_termination_status_t value (constants.h:65-74) */
  ORC_O_STEPS_TAKEN,
  ORC_O_ATTEMPTED_STEPS,
  ORC_O_PRIMAL_OBJECTIVE,
  ORC_O_DUAL_OBJECTIVE,
  ORC_O_GAP,
  ORC_O_RELATIVE_GAP,
  ORC_O_L2_PRIMAL_RESIDUAL,
  ORC_O_L2_DUAL_RESIDUAL,
  ORC_O_L2_REL_PRIMAL_RESIDUAL,
  ORC_O_L2_REL_DUAL_RESIDUAL,
  ORC_O_INITIAL_STEP_SIZE,
  ORC_O_INITIAL_PRIMAL_WEIGHT,
  ORC_O_FINAL_STEP_SIZE,
  ORC_O_FINAL_PRIMAL_WEIGHT,
  ORC_O_SOLVE_SECONDS,
  ORC_O_LOOP_SECONDS,
  ORC_O_NUM_RESTARTS,
  ORC_O_RETURNED_AVERAGE, /* 1 if the returned iterate is the average */
  ORC_O_COUNT
};

/* Fill `hyper` with one of the four presets of cpp/src/linear_programming/solve.cu:64-199
 * (mode numbering = constants.h:98-101: 0 Stable1, 1 Stable2, 2 Methodical1, 3 Fast1). */
void orc_hyper_preset(int mode, double* hyper);
/* default tolerances_t (all 1e-4), no limits */
void orc_default_settings(double* settings);

/* building blocks (also used by the kernel-level parity tests) */
void orc_spmv(int rows, const int* offsets, const int* indices, const double* values,
              const double* x, double* y);
void orc_csr_transpose(int m, int n, const int* offsets, const int* indices, const double* values,
                       int* t_offsets, int* t_indices, double* t_values);
void orc_compute_scaling(int m, int n, const int* offsets, const int* indices,
                         const double* values, const int* t_offsets, const int* t_indices,
                         const double* t_values, const double* hyper, double* d_row,
                         double* d_col);
/* one convergence evaluation on the UNSCALED problem; `c` is the internal (min-form) objective.
 * out[0..9] = pobj, dobj, gap, abs_obj, l2_primal_res, l2_dual_res, l2_x, l2_y,
 *             linf_rel_primal_res, linf_rel_dual_res ; rc (n) receives the reduced costs. */
void orc_eval(int m, int n, const int* offsets, const int* indices, const double* values,
              const int* t_offsets, const int* t_indices, const double* t_values, const double* c,
              const double* lo, const double* hi, const double* lb, const double* ub,
              double obj_scale, double obj_offset, int finite_bounds_rule, double rel_primal_tol,
              double rel_dual_tol, const double* x, const double* y, double* rc, double* out);

/* infeasibility_information_t::compute_infeasibility_information (the iterate itself is the ray
 * estimate), LP/termination_strategy/infeasibility_information.cu:176-223 and :115-172.
 * out[0..3] = max_primal_ray_infeasibility, primal_ray_linear_objective,
 *             max_dual_ray_infeasibility, dual_ray_linear_objective   (after compute_remaining_stats) */
void orc_eval_infeasibility(int m, int n, const int* offsets, const int* indices, const double* values,
                            const int* t_offsets, const int* t_indices, const double* t_values,
                            const double* c, const double* lo, const double* hi, const double* lb,
                            const double* ub, int finite_bounds_rule, const double* x, const double* y,
                            double* out);

/* bound_optimal_objective (pdlp_restart_strategy.cu:1032-1050) of the point (x, y) on the unscaled problem
 * with norm weights wp / wd and trust-region radius `radius`:  out = {lagrangian, lower_bound, upper_bound} */
void orc_trust_region_bounds(int m, int n, const int* offsets, const int* indices, const double* values,
                             const int* t_offsets, const int* t_indices, const double* t_values,
                             const double* c, const double* lo, const double* hi, const double* lb,
                             const double* ub, double wp, double wd, double radius, const double* x,
                             const double* y, double* out);

/* Full PDLP solve.  c/lo/hi/lb/ub are the USER's problem (maximize handled inside like
 * problem_helpers.cuh:126-141).  init_x/init_y may be NULL.  x_out (n), y_out (m), rc_out (n).
 * Returns 0, or a negative value for an unsupported configuration (e.g. trust-region restart). */
int orc_pdlp_solve(int m, int n, const int* offsets, const int* indices, const double* values,
                   const double* c, const double* lo, const double* hi, const double* lb,
                   const double* ub, int maximize, double obj_offset, const double* hyper,
                   const double* settings, const double* init_x, const double* init_y,
                   double* x_out, double* y_out, double* rc_out, double* stats);

/* `iters` raw PDHG steps with a FIXED step size on the scaled problem given directly (no scaling,
 * no restarts, no step-size adaptation): the innermost loop of pdhg.cu:72-158, used as the
 * host-core PDHG loop timed beside the GPU numbers (bench.py cpu_baseline) and by kernel tests. */
void orc_pdhg_fixed_steps(int m, int n, const int* offsets, const int* indices,
                          const double* values, const int* t_offsets, const int* t_indices,
                          const double* t_values, const double* c, const double* lo,
                          const double* hi, const double* lb, const double* ub, double tau,
                          double sigma, int iters, double* x, double* y);

#ifdef __cplusplus
}
#endif
#endif
