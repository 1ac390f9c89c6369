/*
 * Thin C-ABI over the HIP/gfx950 device layer of the PDLP solver.
 *
 * The C++ host driver (cuopt_amd/csrc/pdlp_solver.cpp) never includes a HIP header: everything the
 * GPU does for the PDHG hot path is reached through the `pdlpdev_*` functions below (plain
 * pointers and sizes, no HIP/torch types).  All pointer arguments are HOST pointers unless the
 * name says `dev`.  Every function returns 0 on success and a negative code on failure;
 * pdlpdev_last_error() returns the message (thread-local).
 *
 * What each group replaces in the reference (cuOpt 25.08, LP/ = cpp/src/linear_programming/):
 *   create/destroy ........ detail::problem_t device storage + saddle_point_state_t
 *                           (cpp/src/mip/problem/problem.cu:96-136, LP/saddle_point.cu:26-64),
 *                           cusparse_view_t (LP/cusparse_view.cu:127-277)
 *   scaling_* ............. pdlp_initial_scaling_strategy_t (LP/initial_scaling_strategy/
 *                           initial_scaling.cu:36-163,176-307,310-484)
 *   init_norms ............ compute_initial_step_size / compute_initial_primal_weight
 *                           (LP/pdlp.cu:1224-1309)
 *   run ................... pdlp_solver_t::take_step = pdhg_solver_t::take_step +
 *                           adaptive_step_size_strategy_t::compute_step_sizes +
 *                           weighted_average_solution_t::add_current_solution_to_weighted_average +
 *                           pdhg_solver_t::update_solution (LP/pdlp.cu:1187-1222, LP/pdhg.cu:72-281,
 *                           LP/step_size_strategy/adaptive_step_size_strategy.cu:91-345,
 *                           LP/restart_strategy/weighted_average_solution.cu:73-108)
 *   make_average .......... weighted_average_solution_t::compute_averages (…:114-142)
 *   eval .................. convergence_information_t::compute_convergence_information
 *                           (LP/termination_strategy/convergence_information.cu:149-422)
 *   restart_* ............. run_kkt_restart device parts (LP/restart_strategy/
 *                           pdlp_restart_strategy.cu:593-623,752-839,1680-1714)
 */
#ifndef CUOPT_AMD_PDLP_DEVICE_H
#define CUOPT_AMD_PDLP_DEVICE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pdlpdev_ctx pdlpdev_ctx; /* opaque */

/* Device-resident control block of the step loop, mirrored to the host on request.
 * (reference: device scalars step_size_, primal_weight_, primal/dual_step_size_,
 *  d_total_pdhg_iterations_ + pinned valid_step_size_, adaptive_step_size_strategy.hpp) */
typedef struct pdlpdev_ctl {
  double step_size;     /* eta                                              */
  double primal_weight; /* w                                                */
  double tau;           /* eta / w                                          */
  double sigma;         /* eta * w                                          */
  double sum_weights;   /* weighted-average denominator (sum of step sizes) */
  double last_interaction, last_movement, last_dx2, last_dy2; /* diagnostics of the last attempt */
  int32_t k;            /* number of step-size updates (d_total_pdhg_iterations_) */
  int32_t cur;          /* which of the two ping-pong buffers holds the iterate   */
  int32_t pending_avg;  /* 1: iterate `cur` has not been added to the sums yet    */
  int32_t steps_taken;  /* accepted PDHG steps since create                       */
  int32_t attempts;     /* attempted PDHG steps since create                      */
  int32_t target_steps; /* kernels are no-ops once steps_taken >= target_steps    */
  int32_t error;        /* 1: movement <= 0 or >= 1e100 (valid_step_size = -1)    */
  int32_t its_since_restart;
} pdlpdev_ctl;

/* hyper-parameters the device kernels need (subset of pdlp_hyper_params.cuh:20-58) */
typedef struct pdlpdev_step_params {
  double reduction_exponent;        /* default_reduction_exponent   */
  double growth_exponent;           /* default_growth_exponent      */
  double primal_distance_smoothing; /* default 0.5 (Stable2)        */
  double dual_distance_smoothing;
} pdlpdev_step_params;

/* indices into the `out` array of pdlpdev_eval (unscaled problem, min-form objective) */
enum {
  PDLPDEV_EV_CX = 0,        /* c.x                                                     */
  PDLPDEV_EV_DUAL_SUM,      /* sum_i B(y_i,lo_i,hi_i) + sum_j B(rc_j,lb_j,ub_j)        */
  PDLPDEV_EV_PRES2,         /* ||primal residual||_2^2                                 */
  PDLPDEV_EV_DRES2,         /* ||dual residual||_2^2                                   */
  PDLPDEV_EV_X2,            /* ||x||_2^2                                               */
  PDLPDEV_EV_Y2,            /* ||y||_2^2                                               */
  PDLPDEV_EV_LINF_PRES_REL, /* max_i (r_p,i - eps_rel_primal * bcomb_i), clipped at 0  */
  PDLPDEV_EV_LINF_DRES_REL, /* max_j (r_d,j - eps_rel_dual * c_j), clipped at 0        */
  PDLPDEV_EV_COUNT
};

/* which iterate */
enum {
  PDLPDEV_CURRENT = 0,
  PDLPDEV_AVERAGE = 1,
  PDLPDEV_LAST_RESTART = 2, /* eval / trust region only */
  PDLPDEV_BEST = 3          /* get_solution only: the snapshot taken by pdlpdev_save_best */
};

/* ids for pdlpdev_download (debug / parity tests) */
enum {
  PDLPDEV_BUF_X = 0,      /* current primal iterate (scaled)            n */
  PDLPDEV_BUF_Y,          /* current dual iterate (scaled)              m */
  PDLPDEV_BUF_X_OTHER,    /* the other ping-pong primal buffer          n */
  PDLPDEV_BUF_Y_OTHER,    /*                                            m */
  PDLPDEV_BUF_ATY,        /* A^T y of the current iterate               n */
  PDLPDEV_BUF_ATY_OTHER,  /*                                            n */
  PDLPDEV_BUF_XBAR,       /* extrapolated primal 2x'-x                  n */
  PDLPDEV_BUF_SUM_X,      /* weighted sum of primal iterates            n */
  PDLPDEV_BUF_SUM_Y,      /*                                            m */
  PDLPDEV_BUF_AVG_X,      /* average iterate (scaled)                   n */
  PDLPDEV_BUF_AVG_Y,      /*                                            m */
  PDLPDEV_BUF_DROW,       /* cumulative row scaling D_r                 m */
  PDLPDEV_BUF_DCOL,       /* cumulative column scaling D_c              n */
  PDLPDEV_BUF_A_VALUES,   /* values of A (scaled after scale_problem)   nnz */
  PDLPDEV_BUF_AT_VALUES,  /* values of A^T                              nnz */
  PDLPDEV_BUF_C,          /* scaled objective                           n */
  PDLPDEV_BUF_LB, PDLPDEV_BUF_UB, /* scaled variable bounds             n */
  PDLPDEV_BUF_LO, PDLPDEV_BUF_HI, /* scaled constraint bounds           m */
  PDLPDEV_BUF_RC_CURRENT, /* reduced costs from the last eval(CURRENT)  n */
  PDLPDEV_BUF_RC_AVERAGE, /*                                            n */
  PDLPDEV_BUF_LAST_RESTART_X, PDLPDEV_BUF_LAST_RESTART_Y,
  PDLPDEV_BUF_COUNT
};

/* kernels that pdlpdev_time_kernel can bracket with HIP events */
enum {
  PDLPDEV_K_PRIMAL = 0, /* primal projection + extrapolation + averaging          */
  PDLPDEV_K_SPMV_A_DUAL, /* y' = proj(y - sigma*A*xbar): CSR SpMV fused with dual projection */
  PDLPDEV_K_SPMV_AT_STEP, /* A^T y' fused with interaction / movement partials   */
  PDLPDEV_K_STEP_DECISION, /* 1-block scalar kernel                               */
  PDLPDEV_K_SPMV_A_PLAIN, /* y = A x (unfused CSR SpMV)                          */
  PDLPDEV_K_SPMV_AT_PLAIN,
  PDLPDEV_K_ITERATION,    /* the whole 4-kernel attempt                          */
  PDLPDEV_K_COUNT
};

/* ---- environment ---------------------------------------------------------------------------- */
const char* pdlpdev_last_error(void);
int pdlpdev_device_count(void);
/* name (<= len bytes), CU count, HBM bytes of device `dev` */
int pdlpdev_device_info(int dev, char* name, int len, int* compute_units, int64_t* hbm_bytes);

/* ---- lifetime -------------------------------------------------------------------------------- */
/* Uploads one ROW BLOCK of the LP: `m` = rows held by this context (all rows when single-GPU),
 * `n` = global number of variables.  A is CSR (m x n), At is the explicit CSR of its transpose
 * (n x m) exactly like the reference keeps it (problem.cu:277-309).  c/lb/ub have n entries
 * (min-form objective), lo/hi have m entries.  Everything is copied. */
int pdlpdev_create(pdlpdev_ctx** out, int device, int32_t m, int32_t n, const int32_t* a_offsets,
                   const int32_t* a_indices, const double* a_values, const int32_t* at_offsets,
                   const int32_t* at_indices, const double* at_values, const double* c,
                   const double* lo, const double* hi, const double* lb, const double* ub);
/* Same, for a caller that is still building A^T on other threads: the three at_* arrays must have their final
 * addresses, but their CONTENTS are read only after transpose_ready(user) has returned (NULL: they are ready now).
 * Everything that needs A alone (upload, row blocks, panels of A, all vectors) happens before that call; the callback
 * is invoked exactly once unless the function fails earlier, so a caller that started a thread joins it on every path. */
/* the NEXT context created by this thread will run behind a communicator (sharded solve): paths that exist on one GPU only --
 * dense row segments, the resident small-LP kernel -- are not set up */
void pdlpdev_create_hint(int sharded);
/* the NEXT context created by this thread runs on `donor`'s stream instead of one of its own (NULL: back to own streams).  For
 * hundreds of small contexts that one host thread drives one after the other or that advance together as a pdlpdev_small_batch:
 * a stream is a hardware queue (~2 ms to create, a few per process).  The donor must be destroyed after its borrowers. */
void pdlpdev_create_share_stream(pdlpdev_ctx* donor);
/* 1: an LP of this size takes the resident small-LP path (one workgroup, pdlpdev_small_batch eligible) unless CUOPT_AMD_SMALL=0 */
int pdlpdev_resident_size(int32_t m, int32_t n, int64_t nnz);
int pdlpdev_create_overlapped(pdlpdev_ctx** out, int device, int32_t m, int32_t n, const int32_t* a_offsets,
                              const int32_t* a_indices, const double* a_values, const int32_t* at_offsets,
                              const int32_t* at_indices, const double* at_values,
                              void (*transpose_ready)(void*), void* user, const double* c, const double* lo,
                              const double* hi, const double* lb, const double* ub);
void pdlpdev_destroy(pdlpdev_ctx* ctx);

/* ---- device-side set-up (round 5): ONE upload of A, everything O(nnz) in front of the first step on the GPU ---------------------
 * Replaces, for a matrix of >= ~2e5 nonzeros, the host transposition + four uploads of pdlpdev_create:
 *   - A^T is built on the device from the resident A (the reference: raft::sparse::linalg::csr_transpose,
 *     cpp/src/mip/problem/problem.cu:277-309) -- a stable radix sort by column, rows ascending inside a column;
 *   - flags bit 0: an ANALYSIS PASS looks for structure the matrix arrives without (the reference hands its analysis to the closed
 *     cusparseSpMV_preprocess, cpp/src/linear_programming/cusparse_view.cu:92-115,254-265): when the jagged layout does not apply to
 *     the matrix as given, CELLS are grown breadth-first from seeds spaced one row block apart on the bipartite row-column graph
 *     (rows / columns of more than 128 entries left out; a handful of rounds whatever the diameter), and the cells' QUOTIENT graph
 *     decides how they are laid out: long chains of cells (band / staircase structure) give every vertex a start position that a few
 *     barycentre sweeps over the real graph turn into a smooth one-dimensional embedding (method 1, "chains"); small tight groups
 *     (diagonal blocks behind linking rows / columns) are cleaned up by majority voting at group level (method 2, "groups").  A
 *     candidate is accepted only if the jagged layout's OWN sampled cost estimate passes for both P A Q and its transpose; then the
 *     device holds the permuted pair (three stable sorts) and pdlpdev_analysis_maps returns P and Q.  Everything is deterministic
 *     (seeds at fixed positions, atomicMin on packed (depth, cell) keys, fixed tie rules on the host): the same matrix gets the same
 *     order and layout on every run.  A uniformly random matrix has no strong edge in its quotient graph and is turned away there;
 *   - pdlpdev_create_from_analysis adopts the device arrays (nothing of the matrix crosses PCIe again) and builds the slab-major
 *     panels on the device; the other layouts are constructed on the host from the structure it holds or fetches.
 * The host arrays passed to pdlpdev_analyze must stay valid until the analysis is consumed or destroyed. */
typedef struct pdlpdev_analysis pdlpdev_analysis; /* opaque */
int pdlpdev_analyze(pdlpdev_analysis** out, int device, int32_t m, int32_t n, const int32_t* a_offsets, const int32_t* a_indices,
                    const double* a_values, int flags);
/* the same, and the problem vectors (c, lb, ub: n; lo, hi: m, in the caller's order; any may be NULL) go to the device on a helper
 * thread while the analysis' kernels run: pdlpdev_create_from_analysis picks them up when it is handed the same host pointers and the
 * analysis did not permute the matrix (a permuted LP's vectors are permuted by the caller and uploaded by the create call). */
int pdlpdev_analyze_with_vectors(pdlpdev_analysis** out, int device, int32_t m, int32_t n, const int32_t* a_offsets, const int32_t* a_indices,
                                 const double* a_values, int flags, const double* c, const double* lo, const double* hi, const double* lb,
                                 const double* ub);
/* out = {permuted, method (0 none | 1 chains | 2 groups), estimate natural A, natural A^T, chains A, chains A^T, groups A, groups A^T
 * (savings x 1e4), length of the chains (quotient levels), cell rounds} */
int pdlpdev_analysis_info(pdlpdev_analysis* an, int32_t out[10]);
/* row_new2old[m], col_new2old[n] of an accepted order (row i of the device's matrix is row row_new2old[i] of the caller's); either
 * pointer may be NULL; returns 1 when the device holds a permuted pair, 0 when it holds the matrix as given */
int pdlpdev_analysis_maps(pdlpdev_analysis* an, int32_t* row_new2old, int32_t* col_new2old);
/* 1: a reordering analysis that also holds the vectors of pdlpdev_analyze_with_vectors in the NEW order on the device: pass the same host
   pointers to pdlpdev_create_from_analysis, nothing is gathered on the host or uploaded again */
int pdlpdev_analysis_vectors_in_order(pdlpdev_analysis* an);
/* A consumed analysis hands its workspace back at once (a one-slot cache for the next analysis, or the runtime); pdlpdev_analysis_destroy
   of the rest may then wait. */
void pdlpdev_analysis_release_workspace(pdlpdev_analysis* an);
/* the matrices the device holds as host CSR: which = 0 A (m x n), 1 A^T; any pointer may be NULL */
int pdlpdev_analysis_download(pdlpdev_analysis* an, int which, int32_t* offsets, int32_t* indices, double* values);
/* c / lo / hi / lb / ub in the order of the matrices the device holds */
int pdlpdev_create_from_analysis(pdlpdev_ctx** out, pdlpdev_analysis* an, const double* c, const double* lo, const double* hi,
                                 const double* lb, const double* ub);
void pdlpdev_analysis_destroy(pdlpdev_analysis* an);
/* parity hooks of the set-up primitives (tests): stable sort of n (key, value) pairs by the low `bits` bits (vals NULL: iota);
 * exclusive scan of n ints -> n + 1 outputs; FNV-1a checksums of the layout arrays of a context (out[16]: A^T offsets, indices,
 * values; per side panel row0, tile_ptr, rowptr, col, perm; jagged slot / descriptors / perm) */
int pdlpdev_debug_sort_pairs(int device, int64_t n, const uint32_t* keys, const uint32_t* vals, int bits, uint32_t* keys_out,
                             uint32_t* vals_out);
int pdlpdev_debug_scan(int device, int64_t n, const int32_t* in, int32_t* out);
/* cross-PROCESS rehearsal of the direct peer transport's primitives on one device (tests): the owner exports a zeroed fine-grained
 * landing block (count doubles + flags) as a 64-byte HIP IPC handle; ANOTHER process opens it (hipIpcOpenMemHandle, lazy peer access),
 * stores seed + i into entry i with the transport's system-scope stores and raises its epoch flag with release semantics; the owner
 * waits with the transport's own flag wait and compares: returns the number of wrong entries (>= 2^20: the flag never arrived). */
int pdlpdev_debug_ipc_export(int device, int count, uint8_t handle[64], void** base);
int pdlpdev_debug_ipc_store(int device, const uint8_t handle[64], int count, double seed);
int pdlpdev_debug_ipc_wait(int device, void* base, int count, double seed);
int pdlpdev_debug_layout_checksums(pdlpdev_ctx* ctx, uint64_t out[16]);
/* The wide-bin geometry of the gather-free layout walked on the CPU (no GPU needed): out = M x summed exactly as phase P and phase R order
   the work; info = {padded entries, bins, panels, highest level, serial rows}.  1: the geometry cannot hold the matrix, 2: inconsistent arrays. */
int pdlpdev_debug_pb_wide_host(int32_t rows, int32_t cols, const int32_t* off, const int32_t* idx, const double* val, const double* x, double* out,
                               int64_t info[5]);
/* A synthetic LP of the S(m, n, k) family GENERATED ON THE DEVICE and handed back in host arrays (scale checks near the reference's
 * stated capacity, docs/cuopt/source/faq.rst:368-370: 1e9 nonzeros take the host generator minutes and tens of GB): k entries per row,
 * one per stratum of n / k columns, values ~ N(0, 1); equalities on the first half of the rows, '>=' rows with slack on the second;
 * x >= 0; (x_star, y_star) is optimal by construction, objective = c . x_star.  m * k < 2^31.  offsets[m + 1], indices / values[m * k],
 * c / x_star[n], lo / hi / y_star[m] (x_star, y_star may be NULL). */
int pdlpdev_synthetic_lp(int device, int32_t m, int32_t n, int32_t k, uint64_t seed, int32_t* offsets, int32_t* indices, double* values,
                         double* c, double* lo, double* hi, double* x_star, double* y_star);

/* ---- multi-GPU (row-block sharding, one context per rank) ----------------------------------- */
/* 128-byte RCCL unique id, generated on rank 0 and handed to every rank by the launcher. */
int pdlpdev_comm_unique_id(uint8_t id[128]);
/* Joins the communicator; afterwards every dual-side reduction and the A^T y partial product are
 * all-reduced over xGMI inside the calls below.  world == 1 is allowed (exercise the path). */
int pdlpdev_comm_init(pdlpdev_ctx* ctx, int rank, int world, const uint8_t id[128]);

/* In-process communicator for `world` contexts driven by `world` host threads of ONE process (any devices,
 * typically all on the same GPU): the collectives are executed by a kernel that reads the peers' buffers
 * directly, with host-thread barriers around it.  It exists so that the row-block sharded algorithm can be
 * verified end to end at world > 1 on a single-GPU machine (tests/test_sharded_gpu.py); production multi-GPU
 * runs use RCCL.  Fills `id` with a token that pdlpdev_comm_init / cuoptamd_solver_create accept in place of
 * an RCCL unique id.  The communicator lives until process exit. */
int pdlpdev_softcomm_create(int world, uint8_t id[128]);
/* A rank of a sharded solve failed: aborts every communicator THIS process created from `id` (ncclCommAbort / the in-process
 * communicator's barriers), so that the other ranks' collectives return an error instead of waiting for the missing rank. */
int pdlpdev_comm_abort(const uint8_t id[128]);

/* ---- setup ----------------------------------------------------------------------------------- */
/* D_r, D_c <- Ruiz (inf-norm, `ruiz_iterations` rounds, both sides from the same snapshot) then
 * Pock-Chambolle(alpha).  initial_scaling.cu:36-92. */
int pdlpdev_scaling_compute(pdlpdev_ctx* ctx, int do_ruiz, int ruiz_iterations,
                            int do_pock_chambolle, double alpha);
/* A <- D_r A D_c (and A^T), c <- c o D_c, lb,ub <- ./D_c, lo,hi <- o D_r.  initial_scaling.cu:347-408 */
int pdlpdev_scale_problem(pdlpdev_ctx* ctx);
/* out[0] = max |A_ij| ; out[1] = sum c_j^2 ; out[2] = sum bcomb_i^2 of the problem AS IT IS NOW
 * (scaled or not), bcomb = combine_finite_abs_bounds(lo, hi) (utils.cuh:139-163). */
int pdlpdev_init_norms(pdlpdev_ctx* ctx, double out[3]);
/* *value <- max over the ranks of *value (no-op on an unsharded solver): wall-clock decisions of a sharded solve */
int pdlpdev_agree_max(pdlpdev_ctx* ctx, double* value);
/* out[0] = sum c_j^2, out[1] = sum bcomb_i^2 of the scaled problem (unscaled != 0: of the user's problem) */
int pdlpdev_weight_norms(pdlpdev_ctx* ctx, int unscaled, double out[2]);
/* Re-solve support (the MIP heuristics' call pattern, relaxed_lp.cu:53-175): replaces the variable / constraint bounds
 * (host arrays in the user's space; NULL = unchanged) of an already scaled problem and puts iterate, sums, restart
 * anchors and the control block back to their state right after pdlpdev_scale_problem.  Matrix, objective, D_r, D_c,
 * layouts and graphs are kept; continue with set_step / set_k / set_initial / project_primal as after create. */
int pdlpdev_reset(pdlpdev_ctx* ctx, const double* lb, const double* ub, const double* lo, const double* hi);
/* l2 norms of the UNSCALED c and bcomb (termination constants, convergence_information.cu:76-84) */
int pdlpdev_problem_norms(pdlpdev_ctx* ctx, double* norm_c, double* norm_b);
int pdlpdev_set_step_params(pdlpdev_ctx* ctx, const pdlpdev_step_params* p);
/* eta, w -> ctl (tau = eta/w, sigma = eta*w).  Pass eta < 0 to keep the current step size. */
int pdlpdev_set_step(pdlpdev_ctx* ctx, double step_size, double primal_weight);
/* k (d_total_pdhg_iterations_) override for warm starts (pdlp.cu:1019-1021) */
int pdlpdev_set_k(pdlpdev_ctx* ctx, int32_t k);
/* Initial iterate given in UNSCALED space (NULL = keep zeros): copied, divided by D_c / D_r
 * (scale_solutions, initial_scaling.cu:410-427).  Call after pdlpdev_scaling_compute. */
int pdlpdev_set_initial(pdlpdev_ctx* ctx, const double* x, const double* y);
/* x <- clamp(x, lb, ub) on the scaled problem (pdlp.cu:1041-1056) */
int pdlpdev_project_primal(pdlpdev_ctx* ctx);
/* {x0.(A^T y0), ||x0||^2, ||y0||^2, max|x0|, max|y0|} of the initial iterate (update_step_size_on_initial_solution,
 * pdlp.cu:878-948).  Both pointers NULL: the scaled iterate that pdlpdev_set_initial stored; leaves A^T y0 in the current
 * A^T y buffer.  Both given (compute_initial_step_size_before_scaling, pdlp.cu:929-947): these vectors as they are (host
 * memory; y0 = this rank's rows) against the scaled matrix; the current buffers are left alone. */
int pdlpdev_initial_solution_stats(pdlpdev_ctx* ctx, double out[5], const double* x0_unscaled, const double* y0_unscaled);

/* ---- the hot loop ---------------------------------------------------------------------------- */
/* AtY <- A^T y for the current iterate (pdhg.cu:119-134); needed before the first step and after
 * a restart to the average. */
int pdlpdev_compute_aty(pdlpdev_ctx* ctx);
/* Attempts PDHG steps until `target_steps` accepted steps exist in total (or the step-size error
 * flag is raised).  No host round-trip per attempt: acceptance, step-size update, buffer flip and
 * averaging all happen on the device; the host only reads the control block when the batch is
 * done.  Returns the control block in *ctl. */
int pdlpdev_run(pdlpdev_ctx* ctx, int32_t target_steps, pdlpdev_ctl* ctl);
int pdlpdev_get_ctl(pdlpdev_ctx* ctx, pdlpdev_ctl* ctl);
/* ---- K LPs over ONE matrix and objective advance together (round 5; the MIP heuristics' pattern, cpp/src/mip/relaxed_lp/
 * relaxed_lp.cu:53-127: the same A and c under different bounds; the reference builds a solver per call and its batch entry point,
 * cython_solve.cu:264-296, is a thread pool of independent solves).
 * pdlpdev_clone_shared: a context for another LP over the parent's matrices, layouts, scaling vectors and c (shared, read-only: the
 * parent must outlive its clones and must not be re-scaled while they exist; pdlpdev_reset of either side is fine -- row bounds are
 * shared until a reset changes them, the changing side then gets arrays of its own); iterates, variable bounds, sums, control block are the
 * clone's own.  It starts with the parent's bounds: pdlpdev_reset(clone, lb, ub, lo, hi) gives it its own.  Single GPU only.
 * pdlpdev_batch_create: K = 2, 4, 8 or 16 such contexts (ctx[0] may be the parent).  The two products of an attempt then serve all K LPs
 * from ONE pass over the matrix: the K gathered vectors are interleaved, a row belongs to a group of K lanes, one 64-byte request
 * fetches a column's entry of eight LPs.  Each LP's trajectory is BIT-IDENTICAL to the one pdlpdev_run gives it (same row sums, same
 * epilogue expressions, the single kernels' per-block reduction trees reproduced).  -7: not eligible -- each matrix must be in the
 * row-sum variant of the panel layout or in the CSR stream layout, with no row beyond 128 entries and no dense segments, columns
 * ascending within rows; not the resident small-LP loop, not a sharded context.
 * pdlpdev_batch_run: attempts until LP l holds targets[l] accepted steps (<= 0: LP l rests); ctl[l] receives its control block. */
typedef struct pdlpdev_batch pdlpdev_batch;
int pdlpdev_clone_shared(pdlpdev_ctx** out, pdlpdev_ctx* parent);
int pdlpdev_batch_create(pdlpdev_batch** out, pdlpdev_ctx** ctx, int K);
int pdlpdev_batch_run(pdlpdev_batch* batch, const int32_t* targets, pdlpdev_ctl* ctl);
void pdlpdev_batch_destroy(pdlpdev_batch* batch);
/* average dispatch durations (ms) of the four kernels of a batched attempt {primal, A / dual, A^T / step, decisions}: whole attempts in
 * the loop's order, every LP forced active, the dispatches' own timestamps; state is put back (bench.py's roofline of the batch line) */
int pdlpdev_batch_time_kernels(pdlpdev_batch* batch, int reps, double avg_ms[4]);
/* ---- K SMALL LPs in K workgroups (round 6; BASELINE config 5 at branch-and-bound scale: the reference's batch entry,
 * cpp/src/linear_programming/utilities/cython_solve.cu:264-296, is a thread pool of independent solves, and the MIP heuristics fire
 * thousands of relaxations, cpp/src/mip/relaxed_lp/relaxed_lp.cu:53-127).  An LP on the resident small-LP path runs its whole loop in
 * ONE workgroup on ONE of the chip's 256 CUs; a batch of K such contexts (any matrices, any resident tier, one device) advances with
 * one launch per phase: blockIdx <-> LP.  Every kernel body is the single LP's, so each LP's trajectory is BIT-IDENTICAL to the one
 * pdlpdev_run / pdlpdev_major_eval / pdlpdev_restart give it.  -7: a context is not on the resident path.  The contexts must be idle
 * (no per-context call in flight) while a batch call runs; the batch holds plain pointers -- destroy it before its contexts. */
typedef struct pdlpdev_small_batch pdlpdev_small_batch;
typedef struct pdlpdev_small_eval {
  int32_t mode;        /* average mode of pdlpdev_major_eval (0, 1, 2); < 0: this LP is not evaluated */
  int32_t rule_finite; /* rc_rule_finite_bounds */
  double eps_p, eps_d; /* eps_rel_primal / eps_rel_dual (negative: no l-infinity residuals) */
} pdlpdev_small_eval;
int pdlpdev_small_batch_create(pdlpdev_small_batch** out, pdlpdev_ctx** ctx, int K);
void pdlpdev_small_batch_destroy(pdlpdev_small_batch* batch);
/* pdlpdev_run for every LP with targets[l] > 0 (the others rest); ctl[l] receives its control block.  eval_after (may be NULL): for LPs
 * with eval_after[l].mode >= 0 the pdlpdev_major_eval that follows the attempts is enqueued behind them (one synchronisation for
 * both) and runs iff the attempts reached targets[l] or raised the step-size error: evaluated[l] = 1 and out_current / out_average
 * (PDLPDEV_EV_COUNT doubles per LP) are filled then. */
int pdlpdev_small_batch_run(pdlpdev_small_batch* batch, const int32_t* targets, pdlpdev_ctl* ctl, const pdlpdev_small_eval* eval_after,
                            double* out_current, double* out_average, int32_t* evaluated);
/* pdlpdev_major_eval for every LP with req[l].mode >= 0; PDLPDEV_EV_COUNT doubles per LP in out_current / out_average */
int pdlpdev_small_batch_major_eval(pdlpdev_small_batch* batch, const pdlpdev_small_eval* req, double* out_current, double* out_average);
/* pdlpdev_restart for every LP with which[l] >= 0; dist2[2 l], dist2[2 l + 1] */
int pdlpdev_small_batch_restart(pdlpdev_small_batch* batch, const int32_t* which, const int32_t* unscaled_distances, double* dist2);
/* in front of the next attempts: pdlpdev_clear_error where clear_error[l] != 0, pdlpdev_set_step(-1, primal_weight[l]) where
 * primal_weight[l] > 0, pdlpdev_compute_aty where compute_aty[l] != 0 (any array may be NULL) */
int pdlpdev_small_batch_prepare(pdlpdev_small_batch* batch, const int32_t* clear_error, const double* primal_weight, const int32_t* compute_aty);
/* the re-solve pattern (relaxed_lp.cu:74-108) for every LP with take[l] != 0, in one launch: what pdlpdev_reset(lb[l], ub[l], NULL, NULL) +
 * pdlpdev_set_step(step[l], weight[l]) + pdlpdev_set_k(k[l] when >= 0) + pdlpdev_set_initial(x0[l], y0[l]) + pdlpdev_project_primal (when
 * project != 0) + pdlpdev_get_ctl do for one LP, bit for bit.  lb / ub / x0 / y0 / k may be NULL, and so may their entries (unchanged
 * bounds, a zero start).  Row bounds do not change here. */
int pdlpdev_small_batch_reset(pdlpdev_small_batch* batch, const int32_t* take, const double* const* lb, const double* const* ub, const double* const* x0,
                              const double* const* y0, const double* step, const double* weight, const int32_t* k, int project, pdlpdev_ctl* ctl,
                              const int32_t* var, const double* var_lb, const double* var_ub, const int32_t* warm);
/* (var / var_lb / var_ub, may be NULL: a BRANCH -- the bounds of variable var[l] >= 0 of LP l become [var_lb[l], var_ub[l]], nothing
 *  crosses PCIe but three scalars; warm, may be NULL: warm[l] = PDLPDEV_CURRENT / AVERAGE / BEST starts LP l from that iterate of ITS
 *  OWN last solve, unscaled and scaled again on the device exactly as a read-back + set_initial would -- x0 / y0 are ignored then) */
/* the solutions without the copies: pointers into the batch's pinned staging block, valid until the next reset / solutions call */
int pdlpdev_small_batch_solution_views(pdlpdev_small_batch* batch, const int32_t* which, const double** x_view, const double** y_view, const double** rc_view);
/* pdlpdev_get_solution(which[l], x[l], y[l], rc[l]) for every LP with which[l] >= 0 in one launch (arrays and entries may be NULL) */
int pdlpdev_small_batch_get_solutions(pdlpdev_small_batch* batch, const int32_t* which, double* const* x, double* const* y, double* const* rc);
/* re-arm the loop after the step-size error flag was raised (take_step resets valid_step_size_,
 * pdlp.cu:1190) */
int pdlpdev_clear_error(pdlpdev_ctx* ctx);
/* 0: plain launches, 1: replay the attempt through a captured hipGraph (default 1) */
int pdlpdev_set_graph_mode(pdlpdev_ctx* ctx, int use_graph);
/* instantiate every replay size (1, 2, ... 64 attempts) now instead of at first use (measurement: keeps graph
 * instantiation out of a timed region; a solve does not need it) */
int pdlpdev_prepare_graphs(pdlpdev_ctx* ctx);
/* roctx ranges for rocprofv3 --marker-trace (no-ops when the roctx library is absent or CUOPT_AMD_ROCTX=0); the device
 * layer marks set-up, scaling, attempt batches and major-iteration evaluations itself, the host driver its phases */
/* copy `bytes` from a caller pointer that may live in host OR device / managed memory into host memory */
int pdlpdev_copy_in(void* dst, const void* src, size_t bytes);
void pdlpdev_range_push(const char* name);
void pdlpdev_range_pop(void);

/* ---- major iteration -------------------------------------------------------------------------- */
/* adds a still-pending accepted iterate to the running sums */
int pdlpdev_flush_average(pdlpdev_ctx* ctx);
/* mode 0: avg <- current ; 1: avg <- 0 ; 2: avg <- sum / sum_weights   (pdlp.cu:1110-1122) */
int pdlpdev_make_average(pdlpdev_ctx* ctx, int mode);
/* Convergence information of one iterate on the UNSCALED problem; the iterates stay scaled on the
 * device, D_r/D_c are folded into the two fused SpMV passes.  `rc_rule_finite_bounds` selects
 * copy_gradient_if_finite_bounds (Stable2) vs copy_gradient_if_should_be_reduced_cost.
 * Negative eps_rel_*: the two l-infinity residuals are not needed (reported as 0; saves two vectors of
 * traffic and four launches per evaluation). */
int pdlpdev_eval(pdlpdev_ctx* ctx, int which, int rc_rule_finite_bounds, double eps_rel_primal,
                 double eps_rel_dual, double out[PDLPDEV_EV_COUNT]);
/* The head of a major iteration in one call: flush_average + make_average(average_mode) + eval(CURRENT) +
 * eval(AVERAGE), with a single read-back (one launch in total for LPs on the resident small-LP path). */
int pdlpdev_major_eval(pdlpdev_ctx* ctx, int average_mode, int rc_rule_finite_bounds, double eps_rel_primal,
                       double eps_rel_dual, double out_current[PDLPDEV_EV_COUNT], double out_average[PDLPDEV_EV_COUNT]);
/* Infeasibility information of the iterate evaluated by the LAST pdlpdev_eval(which) call (its A x and
 * A^T y are reused; the iterate itself is the ray estimate, infeasibility_information.cu:176-223):
 * out = {max_primal_ray_infeasibility, primal_ray_linear_objective, max_dual_ray_infeasibility,
 *        dual_ray_linear_objective} after compute_remaining_stats (:115-172). */
int pdlpdev_eval_infeasibility(pdlpdev_ctx* ctx, int which, int rc_rule_finite_bounds, double out[4]);
/* Restart to `which` (CURRENT or AVERAGE): dist2[0] = ||x_c - x_last_restart||^2, dist2[1] same for
 * y (scaled space); copies the candidate into the iterate when it is the average; anchors <-
 * candidate; sums <- 0; its_since_restart <- 0.  unscaled_distances != 0: the distances are those of the
 * UNSCALED iterates (presets with rescale_for_restart = false run the restart on unscaled iterates,
 * pdlp.cu:1144-1175); the anchors themselves always stay in the solver's scaled space. */
int pdlpdev_restart(pdlpdev_ctx* ctx, int which, int unscaled_distances, double dist2[2]);
/* Trust-region restart support (Methodical1; bound_optimal_objective + solve_bound_constrained_trust_region,
 * pdlp_restart_strategy.cu:1032-1050,1391-1678) for the point `which` whose pdlpdev_eval(which) ran last
 * (its A x / A^T y are reused), on the UNSCALED problem:
 *   in : primal/dual norm weights (1/tau, 1/sigma), distance smoothing constants, primal weight,
 *        radius (< 0: use the point's own distance_traveled)
 *   out: {primal_distance^2, dual_distance^2, distance_traveled, lagrangian, lower_bound, upper_bound}
 * The breakpoint search of the reference (sort + median bisection in a cooperative kernel) is replaced by a
 * monotone fixed-point iteration t <- sqrt((r^2 - low(t)) / high(t)) of streaming passes: same threshold.
 * scaled_iterates != 0 (rescale_for_restart, pdlp.cu:1144-1149; no preset pairs it with this restart): the point and the
 * anchors are taken in SCALED space as they are stored, against the same unscaled problem (the reference builds its restart
 * strategy on the unscaled problem, pdlp.cu:99-103); A x / A^T y of the point are computed here, no pdlpdev_eval needed. */
int pdlpdev_trust_region_bounds(pdlpdev_ctx* ctx, int which, double primal_norm_weight,
                                double dual_norm_weight, double primal_distance_smoothing,
                                double dual_distance_smoothing, double primal_weight, double radius,
                                int scaled_iterates, double out[6]);

/* save_best_primal_so_far (pdlp.cu:390-466): keeps a copy of iterate `which` (CURRENT or AVERAGE) and of its
 * reduced costs; retrieved with pdlpdev_get_solution(PDLPDEV_BEST) */
int pdlpdev_save_best(pdlpdev_ctx* ctx, int which);

/* ---- results ---------------------------------------------------------------------------------- */
/* UNSCALED x (n), y (m), reduced costs (n, from the last eval of that iterate); any may be NULL */
int pdlpdev_get_solution(pdlpdev_ctx* ctx, int which, double* x, double* y, double* rc);
/* raw buffer download; returns the number of elements copied (or < 0) */
int64_t pdlpdev_download(pdlpdev_ctx* ctx, int buffer_id, void* host, int64_t max_elements);

/* raw buffer upload (scaled space, warm start restore): the mirror of pdlpdev_download */
int64_t pdlpdev_upload(pdlpdev_ctx* ctx, int buffer_id, const void* host, int64_t elements);
/* restores the loop scalars a warm start carries besides eta / w (pdlp.cu:131-181):
 * sum of averaging weights, iterations since the last restart, number of step-size updates k */
int pdlpdev_set_loop_state(pdlpdev_ctx* ctx, double sum_weights, int32_t its_since_restart, int32_t k);

/* ---- measurement / parity hooks ---------------------------------------------------------------- */
/* y = A x (transpose = 0, x has n entries, y has m) or y = A^T x through the plain CSR kernel */
int pdlpdev_spmv(pdlpdev_ctx* ctx, int transpose, const double* x, double* y);
/* `reps` back-to-back launches of one kernel bracketed by HIP events on the solver stream;
 * *avg_ms = average duration of one launch.  State is not advanced (target_steps trick). */
int pdlpdev_time_kernel(pdlpdev_ctx* ctx, int kernel_id, int reps, double* avg_ms);
/* device-side generation of the iterate is not needed; but benches need a sync point */
int pdlpdev_synchronize(pdlpdev_ctx* ctx);
/* bytes of device memory held by the context */
int64_t pdlpdev_device_bytes(pdlpdev_ctx* ctx);
/* sharded solves: 0 = not sharded, 1 = replicated primal update behind one all-reduce(n + 1) per attempt,
 * 2 = sliced primal update: reduce-scatter(A^T y' partials) + all-gather(xbar) + a 3-scalar all-reduce
 * (CUOPT_AMD_SHARD_DATAFLOW=rsag), 3 = owner computes: the rank also holds its COLUMNS of A, all-gather(xbar slices) +
 * all-gather(y' row blocks) + a 3-scalar all-reduce, no partial products on the wire (CUOPT_AMD_SHARD_DATAFLOW=owner) */
int pdlpdev_shard_dataflow(pdlpdev_ctx* ctx);
/* dense row segments (runs of >= 256 consecutive columns inside a row, stored index-free and multiplied by their own streaming
 * kernels; CUOPT_AMD_TUNE=dense=0 off | dense=1 whenever a segment exists | default: when they hold >= 2 % of the nonzeros; single-GPU
 * solves): out = {in use, segments, entries} */
int pdlpdev_dense_info(pdlpdev_ctx* ctx, int64_t out[3]);
/* transport of the owner-computes dataflow's exchanges: 0 = collectives (RCCL all-gather / 3-scalar all-reduce, or the in-process
 * communicator), 1 = direct peer stores into the ranks' landing blocks + epoch flags, kernels only
 * (CUOPT_AMD_SHARD_TRANSPORT=p2p; peers of the same process are addressed directly, other processes through HIP IPC handles
 * exchanged over the communicator) */
int pdlpdev_shard_transport(pdlpdev_ctx* ctx);
/* owner-computes dataflow, bytes this rank RECEIVES per attempt for the two vector exchanges: out = {halo exchange in use, bytes with
 * the exchange in use, bytes of the two all-gathers}.  Halo exchange (round 5): on a structured LP a rank's rows reference, outside
 * its own slice of xbar, only a few thousand columns at the edges of its neighbours' slices (likewise its columns and y'): per peer
 * ONE contiguous range travels (ncclSend / ncclRecv in a group, or the in-process communicator's copies) instead of the all-gather;
 * chosen at pdlpdev_owner_setup when the ranges sum to at most a quarter of the all-gathers (CUOPT_AMD_TUNE=shard_halo=0|1). */
int pdlpdev_shard_wire_bytes(pdlpdev_ctx* ctx, int64_t out[3]);
/* owner-computes dataflow: the columns [*col_begin, *col_begin + *ncols) of A this rank owns ... */
int pdlpdev_owner_slice(pdlpdev_ctx* ctx, int32_t* col_begin, int32_t* ncols);
/* ... and their nonzeros over ALL rows of A: rows [col_begin, col_begin + ncols) of the global A^T as CSR (indices = global
 * row numbers, ascending; UNSCALED values; offsets start at 0), row_bounds[world + 1] = the ranks' row blocks
 * (cuoptamd_partition_rows).  After pdlpdev_scale_problem, before the first pdlpdev_run. */
int pdlpdev_owner_setup(pdlpdev_ctx* ctx, const int32_t* offsets, const int32_t* indices, const double* values,
                        const int32_t* row_bounds);
/* SpMV layout actually in use: out = {A: layout, workgroups, detail, A^T: layout, workgroups, detail, A: row sums, A^T: row sums};
 * layout 0 = CSR stream, 1 = slab-major row panels (detail: slabs; row sums 0 = a lane per row, short rows left to right, 1 = dealt
 * by nonzero, the long-tail variant), 2 = small LP whose attempt batches run inside ONE resident workgroup (CUOPT_AMD_SMALL=0/1
 * overrides), 3 = sorted jagged rows with LDS column sets (detail: percent of the global gathers the sets save), 4 = gather-free
 * (detail: padding percent).  Chosen at create: environment CUOPT_AMD_SPMV_LAYOUT = auto (default; a structural, reproducible
 * rule: DESIGN.md section 3) | stream | panel | jag | pb | timed ; geometry knobs through CUOPT_AMD_TUNE (slab_bytes, panel_nnz,
 * panel_ws_bytes, panel_seg, jag_waves: INTEGRATION.md section 4). */
int pdlpdev_layout_info(pdlpdev_ctx* ctx, int32_t out[8]);

#ifdef __cplusplus
}
#endif
#endif
