/*
 * libcuopt C API -- the drop-in boundary of the MI355X-native PDLP solver.
 *
 * This header declares exactly the 41 entry points of the reference
 * cpp/include/cuopt/linear_programming/cuopt_c.h:89-668 (cuOpt 25.08; implementation mirrored:
 * cpp/src/linear_programming/cuopt_c.cpp) with identical names, argument order, types and
 * return codes, so a client translation unit (e.g. the reference's own
 * cpp/tests/linear_programming/c_api_tests/c_api_test.c) compiles and links unmodified.
 *
 * Scope of this implementation: continuous LPs are solved by the HIP PDLP solver
 * on gfx950.  Problems with integer variables are accepted by the builder and every getter, but
 * cuOptSolve reports CUOPT_VALIDATION_ERROR for them (MILP heuristics are out of scope).
 * Array arguments of the create functions may be HOST or DEVICE (hipMalloc / managed) pointers, as in the reference
 * (cuopt_c.cpp:119,261-403); the library copies them at create time
 * (reference: cuopt_c.cpp:103-140 copies with raft::copy, caller may free immediately).
 */
#ifndef CUOPT_C_API_H
#define CUOPT_C_API_H

#include <cuopt/linear_programming/constants.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* opaque handles (reference cuopt_c.h:35,42,48) */
typedef void* cuOptOptimizationProblem;
typedef void* cuOptSolverSettings;
typedef void* cuOptSolution;

/* reference cuopt_c.h:50-82 with constants.h:27-30: double + int32 only */
typedef double cuopt_float_t;
typedef int32_t cuopt_int_t;

/* sizes of the two scalar types, for run-time ABI checks (cuopt_c.cpp:58-60) */
int8_t cuOptGetFloatSize();
int8_t cuOptGetIntSize();

/* ---- problem builder (cuopt_c.cpp:62-206) -------------------------------------------------- */
/* MPS file (free format with fixed-format fallback) -> problem.
 * CUOPT_MPS_FILE_ERROR if the file cannot be opened, CUOPT_MPS_PARSE_ERROR for malformed input;
 * *problem_ptr is NULL on failure (cuopt_c.cpp:71-79). */
cuopt_int_t cuOptReadProblem(const char* filename, cuOptOptimizationProblem* problem_ptr);

/* CSR constraint matrix + per-row sense ('L','G','E') and right-hand side.
 * nnz = row_offsets[num_constraints] (cuopt_c.cpp:119).  NULL array -> CUOPT_INVALID_ARGUMENT. */
cuopt_int_t cuOptCreateProblem(cuopt_int_t num_constraints,
                               cuopt_int_t num_variables,
                               cuopt_int_t objective_sense,
                               cuopt_float_t objective_offset,
                               const cuopt_float_t* objective_coefficients,
                               const cuopt_int_t* constraint_matrix_row_offsets,
                               const cuopt_int_t* constraint_matrix_column_indices,
                               const cuopt_float_t* constraint_matrix_coefficent_values,
                               const char* constraint_sense,
                               const cuopt_float_t* rhs,
                               const cuopt_float_t* lower_bounds,
                               const cuopt_float_t* upper_bounds,
                               const char* variable_types,
                               cuOptOptimizationProblem* problem_ptr);

/* Same, with explicit (lower, upper) bounds per constraint row (cuopt_c.cpp:146-198). */
cuopt_int_t cuOptCreateRangedProblem(cuopt_int_t num_constraints,
                                     cuopt_int_t num_variables,
                                     cuopt_int_t objective_sense,
                                     cuopt_float_t objective_offset,
                                     const cuopt_float_t* objective_coefficients,
                                     const cuopt_int_t* constraint_matrix_row_offsets,
                                     const cuopt_int_t* constraint_matrix_column_indices,
                                     const cuopt_float_t* constraint_matrix_coefficients,
                                     const cuopt_float_t* constraint_lower_bounds,
                                     const cuopt_float_t* constraint_upper_bounds,
                                     const cuopt_float_t* variable_lower_bounds,
                                     const cuopt_float_t* variable_upper_bounds,
                                     const char* variable_types,
                                     cuOptOptimizationProblem* problem_ptr);

/* Frees the problem and NULLs the caller's handle; NULL / pointer-to-NULL tolerated
 * (cuopt_c.cpp:200-206). */
void cuOptDestroyProblem(cuOptOptimizationProblem* problem_ptr);

/* ---- problem getters (cuopt_c.cpp:208-438); outputs are caller-allocated host arrays -------- */
cuopt_int_t cuOptGetNumConstraints(cuOptOptimizationProblem problem,
                                   cuopt_int_t* num_constraints_ptr);
cuopt_int_t cuOptGetNumVariables(cuOptOptimizationProblem problem, cuopt_int_t* num_variables_ptr);
cuopt_int_t cuOptGetObjectiveSense(cuOptOptimizationProblem problem,
                                   cuopt_int_t* objective_sense_ptr);
cuopt_int_t cuOptGetObjectiveOffset(cuOptOptimizationProblem problem,
                                    cuopt_float_t* objective_offset_ptr);
cuopt_int_t cuOptGetObjectiveCoefficients(cuOptOptimizationProblem problem,
                                          cuopt_float_t* objective_coefficients_ptr);
cuopt_int_t cuOptGetNumNonZeros(cuOptOptimizationProblem problem, cuopt_int_t* num_non_zeros_ptr);
cuopt_int_t cuOptGetConstraintMatrix(cuOptOptimizationProblem problem,
                                     cuopt_int_t* constraint_matrix_row_offsets_ptr,
                                     cuopt_int_t* constraint_matrix_column_indices_ptr,
                                     cuopt_float_t* constraint_matrix_coefficients_ptr);
cuopt_int_t cuOptGetConstraintSense(cuOptOptimizationProblem problem, char* constraint_sense_ptr);
cuopt_int_t cuOptGetConstraintRightHandSide(cuOptOptimizationProblem problem,
                                            cuopt_float_t* rhs_ptr);
cuopt_int_t cuOptGetConstraintLowerBounds(cuOptOptimizationProblem problem,
                                          cuopt_float_t* lower_bounds_ptr);
cuopt_int_t cuOptGetConstraintUpperBounds(cuOptOptimizationProblem problem,
                                          cuopt_float_t* upper_bounds_ptr);
cuopt_int_t cuOptGetVariableLowerBounds(cuOptOptimizationProblem problem,
                                        cuopt_float_t* lower_bounds_ptr);
cuopt_int_t cuOptGetVariableUpperBounds(cuOptOptimizationProblem problem,
                                        cuopt_float_t* upper_bounds_ptr);
cuopt_int_t cuOptGetVariableTypes(cuOptOptimizationProblem problem, char* variable_types_ptr);

/* ---- solver settings (cuopt_c.cpp:440-563; registry solver_settings.cu:66-118) -------------- */
cuopt_int_t cuOptCreateSolverSettings(cuOptSolverSettings* settings_ptr);
void cuOptDestroySolverSettings(cuOptSolverSettings* settings_ptr);
/* string-valued set/get of ANY parameter; unknown name, unparsable or out-of-range value ->
 * CUOPT_INVALID_ARGUMENT */
cuopt_int_t cuOptSetParameter(cuOptSolverSettings settings,
                              const char* parameter_name,
                              const char* parameter_value);
cuopt_int_t cuOptGetParameter(cuOptSolverSettings settings,
                              const char* parameter_name,
                              cuopt_int_t parameter_value_size,
                              char* parameter_value);
/* integer parameters; falls back to a boolean parameter of the same name (cuopt_c.cpp:493-505) */
cuopt_int_t cuOptSetIntegerParameter(cuOptSolverSettings settings,
                                     const char* parameter_name,
                                     cuopt_int_t parameter_value);
cuopt_int_t cuOptGetIntegerParameter(cuOptSolverSettings settings,
                                     const char* parameter_name,
                                     cuopt_int_t* parameter_value);
cuopt_int_t cuOptSetFloatParameter(cuOptSolverSettings settings,
                                   const char* parameter_name,
                                   cuopt_float_t parameter_value);
cuopt_int_t cuOptGetFloatParameter(cuOptSolverSettings settings,
                                   const char* parameter_name,
                                   cuopt_float_t* parameter_value);

/* ---- solve (cuopt_c.cpp:565-641) ------------------------------------------------------------ */
cuopt_int_t cuOptIsMIP(cuOptOptimizationProblem problem, cuopt_int_t* is_mip_ptr);
/* Never throws.  Returns the error type of the solve and ALWAYS produces a solution handle whose
 * cuOptGetErrorString carries the message (cuopt_c.cpp:613-618, LP/solve.cu:604-612). */
cuopt_int_t cuOptSolve(cuOptOptimizationProblem problem,
                       cuOptSolverSettings settings,
                       cuOptSolution* solution_ptr);
void cuOptDestroySolution(cuOptSolution* solution_ptr);

/* ---- solution getters (cuopt_c.cpp:643-849) ------------------------------------------------- */
cuopt_int_t cuOptGetTerminationStatus(cuOptSolution solution, cuopt_int_t* termination_status_ptr);
cuopt_int_t cuOptGetErrorStatus(cuOptSolution solution, cuopt_int_t* error_status_ptr);
cuopt_int_t cuOptGetErrorString(cuOptSolution solution,
                                char* error_string_ptr,
                                cuopt_int_t error_string_size);
cuopt_int_t cuOptGetPrimalSolution(cuOptSolution solution, cuopt_float_t* solution_values);
cuopt_int_t cuOptGetObjectiveValue(cuOptSolution solution, cuopt_float_t* objective_value_ptr);
cuopt_int_t cuOptGetSolveTime(cuOptSolution solution, cuopt_float_t* solve_time_ptr);
/* MIP-only getters: CUOPT_INVALID_ARGUMENT on an LP solution (cuopt_c.cpp:771-805) */
cuopt_int_t cuOptGetMIPGap(cuOptSolution solution, cuopt_float_t* mip_gap_ptr);
cuopt_int_t cuOptGetSolutionBound(cuOptSolution solution, cuopt_float_t* solution_bound_ptr);
/* LP-only getters (cuopt_c.cpp:807-849) */
cuopt_int_t cuOptGetDualSolution(cuOptSolution solution, cuopt_float_t* dual_solution_ptr);
cuopt_int_t cuOptGetReducedCosts(cuOptSolution solution, cuopt_float_t* reduced_cost_ptr);

#ifdef __cplusplus
}
#endif

#endif /* CUOPT_C_API_H */
