/*
 * C-ABI of the C++ host driver of the MI355X-native PDLP solver (cuopt_amd/csrc/pdlp_solver.cpp).
 *
 * This is the layer the libcuopt C API (cuopt_c.h -> cuOptSolve) sits on; it is exported so that
 * benches, the MIP-style warm-started re-solve loop and the parity tests can drive the solver
 * step by step.  It replaces, in the reference (cuOpt 25.08, LP/ = cpp/src/linear_programming/):
 *   cuoptamd_solver_create .... detail::problem_t ctor (cpp/src/mip/problem/problem.cu:53-136),
 *                               pdlp_solver_t ctor + the [init] block of run_solver
 *                               (LP/pdlp.cu:55-188, 984-1075)
 *   cuoptamd_solver_advance ... the while(true) loop of pdlp_solver_t::run_solver
 *                               (LP/pdlp.cu:1081-1185) incl. check_termination (:537-802) and
 *                               run_kkt_restart (LP/restart_strategy/pdlp_restart_strategy.cu:467-641)
 *   cuoptamd_hyper_preset ..... set_pdlp_solver_mode (LP/solve.cu:64-212)
 * All pointers are host pointers.  Functions return 0 or a negative code;
 * cuoptamd_last_error() holds the message.
 */
#ifndef CUOPT_AMD_PDLP_SOLVER_H
#define CUOPT_AMD_PDLP_SOLVER_H

#include <stdint.h>

#include "cuopt_amd/pdlp_device.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cuoptamd_solver cuoptamd_solver; /* opaque */

/* LP in the user's form: objective_sense * (c.x) + offset, lo <= A x <= hi, lb <= x <= ub */
typedef struct cuoptamd_lp {
  int32_t m, n;
  const int32_t* offsets; /* m+1 */
  const int32_t* indices; /* nnz */
  const double* values;   /* nnz */
  const double* c;        /* n */
  const double* lo;       /* m (may be -inf) */
  const double* hi;       /* m (may be +inf) */
  const double* lb;       /* n */
  const double* ub;       /* n */
  int32_t maximize;
  double objective_offset;
} cuoptamd_lp;

/* pdlp_hyper_params (cpp/include/cuopt/linear_programming/pdlp/pdlp_hyper_params.cuh:20-58) */
typedef struct cuoptamd_hyper {
  double initial_step_size_scaling;
  int32_t ruiz_iterations;
  int32_t do_pock_chambolle;
  int32_t do_ruiz;
  double alpha_pock_chambolle;
  double artificial_restart_threshold;
  int32_t compute_initial_step_size_before_scaling;
  int32_t compute_initial_primal_weight_before_scaling;
  double initial_primal_weight_c_scaling;
  double initial_primal_weight_b_scaling;
  int32_t major_iteration;
  int32_t min_iteration_restart;
  int32_t restart_strategy; /* 0 none, 1 KKT, 2 trust region (Methodical1) */
  int32_t never_restart_to_average;
  double reduction_exponent;
  double growth_exponent;
  double primal_weight_update_smoothing;
  double sufficient_reduction_for_restart;
  double necessary_reduction_for_restart;
  double primal_importance;
  double primal_distance_smoothing;
  double dual_distance_smoothing;
  int32_t compute_last_restart_before_new_primal_weight;
  int32_t artificial_restart_in_main_loop;
  int32_t rescale_for_restart;
  int32_t update_primal_weight_on_initial_solution;
  int32_t update_step_size_on_initial_solution;
  int32_t handle_some_primal_gradients_on_finite_bounds_as_residuals;
  int32_t project_initial_primal;
} cuoptamd_hyper;

/* pdlp_solver_settings_t subset (pdlp/solver_settings.hpp:70-224) */
typedef struct cuoptamd_settings {
  double absolute_gap_tolerance, relative_gap_tolerance;
  double absolute_primal_tolerance, relative_primal_tolerance;
  double absolute_dual_tolerance, relative_dual_tolerance;
  int32_t iteration_limit; /* INT32_MAX: none */
  double time_limit;       /* seconds; +inf: none */
  int32_t per_constraint_residual;
  int32_t first_primal_feasible;
  /* warm start overrides (pdlp.cu:1014-1021); negative = not set */
  double initial_step_size;
  double initial_primal_weight;
  int32_t initial_k;
  int32_t use_graph; /* replay the PDHG attempt through hipGraphs (default 1) */
  /* infeasibility detection (pdlp/solver_settings.hpp: detect_infeasibility, strict_infeasibility;
   * tolerances default 1e-8, solver_settings.cu:83-84) */
  int32_t detect_infeasibility;
  int32_t strict_infeasibility;
  double primal_infeasible_tolerance;
  double dual_infeasible_tolerance;
  /* on a time / iteration limit return the best primal point seen at a major iteration instead of the last
   * iterate (pdlp.cu:264-331,390-466) */
  int32_t save_best_primal_so_far;
  /* iteration log (header + one line every 1000 iterations + summary, pdlp.cu:508-535,1077-1080): to stdout
   * and/or appended to log_file (NULL or "" = none) */
  int32_t log_to_console;
  const char* log_file;
  /* The reference's verdict kernel returns PrimalFeasible BEFORE it looks at the rays (termination_strategy.cu:190-226
   * vs :228-249), so its PDLP never reports an unbounded LP whose iterates are primal feasible: it runs until the
   * step-size arithmetic overflows (NumericalError).  Non-zero: a primal-feasible iterate whose normalised ray meets
   * the dual-infeasibility test is reported as DualInfeasible (needs detect_infeasibility; both iterates must agree
   * unless strict).  Default 0 = the reference's PDLP; cuOptSolve sets it for Concurrent / DualSimplex requests, where
   * the reference's simplex would return UNBOUNDED. */
  int32_t unbounded_from_feasible_iterates;
  /* A second, looser tolerance set (abs gap, rel gap, abs primal, rel primal, abs dual, rel dual).  When enabled, the
   * first iterate seen at a major iteration that is Optimal by THESE tolerances is kept; if the run then ends on an
   * iteration or time limit before the main tolerances are met, the kept iterate is returned with status Optimal
   * (cuoptamd_result::accepted_at_looser_tolerances = 1).  cuOptSolve uses it for Concurrent / DualSimplex requests
   * on small LPs: the main tolerances are simplex-grade (1e-8), the looser set is what the caller asked for, so the
   * caller's limits are honoured exactly as the reference's Concurrent mode would.  Ignored when
   * save_best_primal_so_far is set (same snapshot buffers). */
  int32_t accept_enabled;
  double accept_tolerance[6];
  /* set_relative_{primal,dual}_tolerance_factor (pdlp.cu:209-231): the norms ||b|| / ||c|| of the termination rule
   * (eps_abs + eps_rel * factor) replaced by the caller's values -- the MIP side keeps them fixed across re-solves.
   * Negative (default): computed from the problem. */
  double relative_primal_tolerance_factor, relative_dual_tolerance_factor;
} cuoptamd_settings;

/* additional_termination_information_t (pdlp/solver_solution.hpp:63-103) + run statistics */
typedef struct cuoptamd_result {
  int32_t status; /* pdlp_termination_status_t == CUOPT_TERIMINATION_STATUS_* ; 0 = still running */
  int32_t steps_taken;
  int32_t attempted_steps;
  int32_t returned_average;
  int32_t num_restarts;
  int32_t num_major_iterations;
  double primal_objective, dual_objective, gap, relative_gap;
  double l2_primal_residual, l2_dual_residual;
  double l2_relative_primal_residual, l2_relative_dual_residual;
  double max_primal_ray_infeasibility, max_dual_ray_infeasibility; /* filled when detect_infeasibility */
  double primal_ray_linear_objective, dual_ray_linear_objective;
  double initial_step_size, initial_primal_weight;
  double step_size, primal_weight;
  double norm_b, norm_c;
  double setup_seconds; /* transpose + upload + scaling + initial step/weight */
  double loop_seconds;  /* accumulated time inside cuoptamd_solver_advance */
  int32_t accepted_at_looser_tolerances; /* 1: the returned point is the one kept under settings.accept_tolerance */
  int32_t gpus;                          /* row blocks the LP was sharded into (1 = single GPU) */
} cuoptamd_result;

/* Row-block sharded solve inside ONE process: `gpus` host threads, one solver per visible device
 * (device g for rank g), row blocks balanced by nonzeros, RCCL all-reduce over xGMI (ncclCommInitRank from every thread
 * with one unique id).  Returns rank 0's statistics, the replicated primal solution / reduced costs and the GATHERED
 * dual solution (y has lp->m entries).  `soft_communicator` != 0 runs all ranks on device 0 through the in-process
 * communicator instead of RCCL (verification on a single-GPU box; not a production mode). */
int cuoptamd_solve_sharded(const cuoptamd_lp* lp, const cuoptamd_hyper* hyper, const cuoptamd_settings* settings,
                           int gpus, int soft_communicator, cuoptamd_result* result, double* x, double* y, double* rc);

/* pdlp_warm_start_data_t (cpp/include/cuopt/linear_programming/pdlp/pdlp_warm_start_data.hpp:31-90):
 * the complete solver state at a terminating major iteration.  Vectors are CALLER-allocated host arrays
 * (n entries for the primal ones, m for the dual ones).  As in the reference (pdlp.cu:468-489) the
 * current iterate and the average are in the user's UNSCALED space, A^T y, the running sums and the
 * last-restart anchors are in the solver's scaled space, so a snapshot is only meaningful for the SAME
 * constraint matrix and hyper-parameters (the reference's own restriction). */
typedef struct cuoptamd_warm_start {
  double* current_primal_solution;                  /* n */
  double* current_dual_solution;                    /* m */
  double* initial_primal_average;                   /* n */
  double* initial_dual_average;                     /* m */
  double* current_ATY;                              /* n */
  double* sum_primal_solutions;                     /* n */
  double* sum_dual_solutions;                       /* m */
  double* last_restart_duality_gap_primal_solution; /* n */
  double* last_restart_duality_gap_dual_solution;   /* m */
  /* optional (may be NULL): the current iterate in the solver's scaled space.  (x * D) / D is not x in the last
   * bit, so a restore from the unscaled iterate alone perturbs the trajectory; when these are present the
   * restore is bit-exact and its(full) == its(coarse) + its(warm) holds exactly. */
  double* current_primal_solution_scaled;           /* n */
  double* current_dual_solution_scaled;             /* m */
  double initial_primal_weight;
  double initial_step_size;
  int32_t total_pdlp_iterations;
  int32_t total_pdhg_iterations;
  double last_candidate_kkt_score;
  double last_restart_kkt_score;
  double sum_solution_weight;
  int32_t iterations_since_last_restart;
  /* sizes of the problem the snapshot was taken from (filled by get_warm_start; 0 = unknown): a snapshot of another
   * problem is refused by set_warm_start, like the reference (test_lp_solver.py:545-565) */
  int32_t n_variables, n_constraints;
} cuoptamd_warm_start;

/* The second engine (the reference's: cpp/src/dual_simplex, LP/solve.cu:295-347): an own bounded dual simplex (sparse LU of the
 * basis with product-form updates, dual steepest-edge pricing, Harris ratio test), host code like the reference's.  LPs of up to
 * 200 000 rows and 4e6 nonzeros (CUOPT_AMD_SIMPLEX_MAX_ROWS / CUOPT_AMD_SIMPLEX_MAX_NNZ move the limits).
 * *status: 1 optimal (x, y, rc, objective filled; y and rc in the convention rc = c - A^T y), 2 primal infeasible,
 * 3 unbounded, 5 iteration limit, 6 time limit, 7 numerical trouble / the engine abstains (a vertex on the box of the infinite
 * bounds whose ray costs less than any dual tolerance), 8 too large for this engine (nothing was done), 9 cancelled
 * (*cancel became non-zero: the other engine of a Concurrent solve finished first).  time_limit <= 0 / iteration_limit <= 0: none. */
int cuoptamd_dual_simplex(const cuoptamd_lp* lp, double time_limit, int32_t iteration_limit, const volatile int32_t* cancel,
                          int32_t* status, int32_t* iterations, double* objective, double* x, double* y, double* rc);
/* The same engine started from a basis guessed from the point x0 (n entries) and, optionally, the duals y0 (m entries, same
 * convention as y; NULL: none) -- e.g. PDLP's solution: the variables and rows that sit strictly between their bounds there form
 * the basis, then the ones on a bound with the smallest reduced cost at y0, completed / repaired with slacks.  This is the
 * crossover of a first-order solution to a vertex (the reference: crossover = true, LP/solve.cu:466-547). */
int cuoptamd_dual_simplex_from(const cuoptamd_lp* lp, const double* x0, const double* y0, double time_limit, int32_t iteration_limit,
                               const volatile int32_t* cancel, int32_t* status, int32_t* iterations, double* objective, double* x,
                               double* y, double* rc);

const char* cuoptamd_last_error(void);

/* presets, mode numbering as CUOPT_PDLP_SOLVER_MODE_* (0 Stable1, 1 Stable2, 2 Methodical1, 3 Fast1) */
void cuoptamd_hyper_preset(int mode, cuoptamd_hyper* h);
void cuoptamd_default_settings(cuoptamd_settings* s);

/* Builds A^T, slices this rank's row block (rank/world; world = 1: everything), uploads, scales,
 * computes the initial step size and primal weight.  `comm_id`: 128-byte RCCL id when world > 1
 * (or world == 1 and non-NULL to exercise the collective path), else NULL.
 * init_x / init_y: optional initial primal / dual solution in the user's (unscaled) space. */
int cuoptamd_solver_create(cuoptamd_solver** out, const cuoptamd_lp* lp, const cuoptamd_hyper* hyper,
                           const cuoptamd_settings* settings, const double* init_x,
                           const double* init_y, int device, int rank, int world,
                           const uint8_t* comm_id);
void cuoptamd_solver_destroy(cuoptamd_solver* s);

/* Persistent re-solve, the MIP heuristics' call pattern (get_relaxed_lp_solution / run_lp_with_vars_fixed,
 * cpp/src/mip/relaxed_lp/relaxed_lp.cu:53-175: same A and c, tightened or fixed bounds, previous primal/dual as the
 * initial point).  The reference constructs a new pdlp_solver_t per call (transpose, scaling, cuSPARSE set-up again);
 * here the solver keeps A, A^T, D_r, D_c and its kernels' layouts and only takes the new bounds.
 * lb / ub (n) and lo / hi (m_global): user-space host arrays, NULL = unchanged.  settings: NULL = unchanged.
 * init_x / init_y as in cuoptamd_solver_create.  Afterwards the solver behaves exactly (bit for bit) like a solver
 * freshly created on the modified LP: call cuoptamd_solver_advance. */
int cuoptamd_solver_reset(cuoptamd_solver* s, const double* lb, const double* ub, const double* lo, const double* hi,
                          const cuoptamd_settings* settings, const double* init_x, const double* init_y);

/* Runs the PDLP loop until a termination criterion fires or `max_new_iterations` further PDLP
 * iterations (accepted steps) have been taken, whichever is first.  result->status == 0 means
 * "budget exhausted, not terminated"; calling again continues exactly where it stopped. */
int cuoptamd_solver_advance(cuoptamd_solver* s, int32_t max_new_iterations, cuoptamd_result* result);

/* ---- shared-matrix batch (round 5): K LPs that differ in their bounds only are solved in lockstep, the matrix streamed once per
 * attempt for all of them (pdlpdev_batch_* in pdlp_device.h; the MIP heuristics' re-solve pattern, relaxed_lp.cu:53-127).
 * cuoptamd_solver_clone: a solver for the parent's LP with other bounds (NULL = the parent's CURRENT ones; settings NULL = the
 * parent's) that shares the parent's matrices on the device; behaves bit for bit like a solver freshly created on that LP.  Destroy
 * the clones before the parent (cuoptamd_solver_reset of either is fine while both exist).  -7: empty, resident or sharded parent.
 * cuoptamd_batch_create: K = 2, 4, 8 or 16 solvers over one matrix (a parent and its clones, none advanced by hand in between is NOT
 * required: a batch may be created over solvers in any state).  -7 when the layouts are not eligible (pdlp_device.h): the caller
 * falls back to cuoptamd_solver_advance per solver or to cuoptamd_batch_solve.
 * cuoptamd_batch_advance: every solver up to max_new_iterations further iterations or to its verdict; results[l] as
 * cuoptamd_solver_advance would fill it (loop_seconds: the wall time of the batch).  Solutions through the solvers' own getters.
 * cuoptamd_batch_destroy leaves the solvers alive. */
typedef struct cuoptamd_batch cuoptamd_batch;
int cuoptamd_solver_clone(cuoptamd_solver* parent, const double* lb, const double* ub, const double* lo, const double* hi,
                          const cuoptamd_settings* settings, cuoptamd_solver** out);
int cuoptamd_batch_create(cuoptamd_solver** solvers, int K, cuoptamd_batch** out);
int cuoptamd_batch_advance(cuoptamd_batch* batch, int32_t max_new_iterations, cuoptamd_result* results);
/* ---- small-LP batch (round 6): cuoptamd_batch_create over solvers on the resident small-LP path takes ANY number of them, whatever their
 * matrices -- one workgroup per LP, one launch per phase of the loop (pdlpdev_small_batch_* in pdlp_device.h; BASELINE config 5 at
 * branch-and-bound scale, cython_solve.cu:264-296 / relaxed_lp.cu:53-127).  Bit-identical to cuoptamd_solver_advance per solver.
 * cuoptamd_batch_reset: cuoptamd_solver_reset(lb[l], ub[l], NULL, NULL, NULL, init_x[l], init_y[l]) for every solver in one launch
 * (arrays and entries may be NULL); -7 when the batch is not a small-LP batch.
 * cuoptamd_batch_get_solutions: cuoptamd_solver_get_solution for every solver in one launch (any batch; arrays / entries may be NULL). */
int cuoptamd_batch_reset(cuoptamd_batch* batch, const double* const* lb, const double* const* ub, const double* const* init_x, const double* const* init_y);
int cuoptamd_batch_get_solutions(cuoptamd_batch* batch, double* const* x, double* const* y, double* const* rc);
/* cuoptamd_batch_branch: one branch-and-bound step per node in ONE launch -- the bounds of variable var[l] of solver l become
 * [lb[l], ub[l]] (var[l] < 0: none) and the solver starts again from the primal / dual its last solve returned, all on the device
 * (relaxed_lp.cu:74-108); bit for bit cuoptamd_solver_reset(full bounds, solution of cuoptamd_solver_get_solution).
 * cuoptamd_batch_solution_views: the solutions as pointers into the batch's pinned staging block (no copies; valid until the next
 * reset / branch / solutions call of the batch). */
int cuoptamd_batch_branch(cuoptamd_batch* batch, const int32_t* var, const double* lb, const double* ub);
int cuoptamd_batch_solution_views(cuoptamd_batch* batch, const double** x, const double** y, const double** rc);
void cuoptamd_batch_destroy(cuoptamd_batch* batch);
/* the device-layer batch behind it (pdlpdev_batch_time_kernels) */
struct pdlpdev_batch* cuoptamd_batch_device(cuoptamd_batch* batch);

/* Fills *ws (get_filled_warmed_start_data, pdlp.cu:468-489) from a solver whose last advance terminated.
 * Single-GPU solvers only. */
int cuoptamd_solver_get_warm_start(cuoptamd_solver* s, cuoptamd_warm_start* ws);
/* Restores a snapshot into a freshly created solver (before its first advance), pdlp.cu:131-181:
 * iteration counts keep running from the snapshot (its(1e-2 from scratch) == its(1e-1) + its(1e-2 warm),
 * the reference's warm-start test, pdlp_test.cu:803-854). */
int cuoptamd_solver_set_warm_start(cuoptamd_solver* s, const cuoptamd_warm_start* ws);

/* set_pdlp_warm_start_data(data, var_mapping, constraint_mapping) of the reference (LP/solver_settings.cu:92-240): a
 * snapshot of an n_old x m_old problem for a problem with n_mapping variables / m_mapping constraints (0 = that side is
 * unchanged).  Shorter: new[map[i]] = old[i] for i < new size (the map must be a permutation of 0..new size-1), the
 * rest is dropped; longer: the old entries, then zeros (the map's values are not read, as in the reference).  `out`'s
 * vectors are caller-allocated at the new sizes; the scaled iterate is dropped when a size changes.  Host only. */
int cuoptamd_warm_start_remap(const cuoptamd_warm_start* in, const int32_t* var_mapping, int32_t n_mapping,
                              const int32_t* constraint_mapping, int32_t m_mapping, cuoptamd_warm_start* out);

/* x (n), y (m_global), reduced cost (n) of the returned iterate, unscaled, in the internal min-form
 * sign convention of the reference (any pointer may be NULL). Valid after a terminating advance. */
int cuoptamd_solver_get_solution(cuoptamd_solver* s, double* x, double* y, double* rc);

/* Batch solve (call_batch_solve, LP/utilities/cython_solve.cu:264-296: independent LPs solved concurrently on
 * one GPU, one host thread + one stream each, at most `max_threads` at a time; <= 0: one per available core,
 * capped at 16).  results[i] is filled for every LP; x[i] / y[i] / rc[i] (each may be NULL, as may the arrays)
 * receive the solutions.  Returns 0 if every solve ran (look at results[i].status), else the first error. */
int cuoptamd_batch_solve(int32_t count, const cuoptamd_lp* lps, const cuoptamd_hyper* hyper,
                         const cuoptamd_settings* settings, int device, int max_threads,
                         cuoptamd_result* results, double** x, double** y, double** rc);

/* the device context (for kernel timing and buffer downloads in benches/tests) */
pdlpdev_ctx* cuoptamd_solver_device(cuoptamd_solver* s);
/* rows [row_begin, row_end) of A held by this rank (of the matrix in the DEVICE's order when the set-up reordered it) */
int cuoptamd_solver_row_range(cuoptamd_solver* s, int32_t* row_begin, int32_t* row_end);
/* Set-up reordering (pdlp_device.h "device-side set-up"): info = pdlpdev_analysis_info's ten numbers (info[0] = 1: the device works
 * on P A Q; info[1]: 1 breadth-first levels, 2 seeded cells); row_new2old[m] / col_new2old[n] (either may be NULL) receive the maps
 * when the LP was reordered.  Returns 1 when reordered, 0 when the device holds the LP as given.  Everything that crosses this
 * interface -- bounds, initial iterates, solutions, warm starts -- stays in the CALLER's order either way. */
int cuoptamd_solver_reorder_info(cuoptamd_solver* s, int32_t info[10], int32_t* row_new2old, int32_t* col_new2old);

/* contiguous row-block partition balanced by nonzeros: fills bounds[0..world] */
void cuoptamd_partition_rows(int32_t m, const int32_t* offsets, int world, int32_t* bounds);
/* CSR transpose (stable: rows ascending inside each column), host, O(nnz) */
void cuoptamd_csr_transpose(int32_t m, int32_t n, const int32_t* offsets, const int32_t* indices,
                            const double* values, int32_t* t_offsets, int32_t* t_indices,
                            double* t_values);

#ifdef __cplusplus
}
#endif
#endif
