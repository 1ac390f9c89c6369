/* Extensions of this library to the libcuopt C API (include/cuopt/linear_programming/cuopt_c.h mirrors the reference's
 * 41 functions unchanged; nothing here exists in NVIDIA/cuopt).
 *
 * Extra parameters accepted by cuOptSetIntegerParameter / cuOptGetIntegerParameter:
 *   CUOPT_AMD_NUM_GPUS       row blocks of one solve, one GPU each, inside the calling process (RCCL over xGMI).
 *                            0 (default) = the CUOPT_AMD_NUM_GPUS environment variable, else 1.
 *   CUOPT_AMD_SIMPLEX_GRADE  1 / 0: serve Concurrent / DualSimplex requests on small LPs (<= 1e5 nonzeros) at
 *                            simplex-grade tolerances (1e-8) with the requested tolerances as acceptance set
 *                            (cuoptamd_settings::accept_tolerance).  -1 (default) = the key simplex_grade of the
 *                            CUOPT_AMD_TUNE environment string (CUOPT_AMD_TUNE=simplex_grade=0), else on.
 */
#ifndef CUOPT_AMD_CUOPT_C_EXT_H
#define CUOPT_AMD_CUOPT_C_EXT_H

#include "cuopt/linear_programming/cuopt_c.h"
#include "cuopt_amd/pdlp_solver.h"

#define CUOPT_AMD_NUM_GPUS "amd_num_gpus"
#define CUOPT_AMD_SIMPLEX_GRADE "amd_simplex_grade"

#ifdef __cplusplus
extern "C" {
#endif

/* full PDLP statistics of an LP solution (the reference exposes additional_termination_information_t through its
 * C++ / Python API only: cpp/include/cuopt/linear_programming/pdlp/solver_solution.hpp:63-103) */
cuopt_int_t cuOptAmdGetPdlpStats(cuOptSolution solution, cuoptamd_result* stats);

/* which engine / attempt answered the request, as one JSON object:
 * {"engine": "pdlp" | "dual_simplex", "requested_method": "Concurrent|DualSimplex|PDLP", "crossover_requested": bool,
 *  "simplex_grade_emulation": bool, "dual_simplex_consulted": bool, "dual_simplex_status": 1..9 (cuoptamd_dual_simplex),
 *  "crossover": "none" | "dual_simplex_from_the_pdlp_point" | "not_done_..." | "not_needed_vertex_from_the_dual_simplex",
 *  "answered_by": "...", "gpus": N, "iterations": K, "simplex_grade_attempt_iterations": K0}
 * (the reference runs its dual simplex / crossover for such requests, LP/solve.cu:383-443,467-547; here an own simplex code on
 * the host -- cuoptamd_dual_simplex[_from], LPs of up to 200 000 rows -- answers DualSimplex requests, races PDLP under
 * Concurrent and crosses PDLP's point over to a vertex; beyond its limits PDLP serves the request, and the call says so
 * instead of pretending) */
cuopt_int_t cuOptAmdGetSolveInfo(cuOptSolution solution, char* buffer, cuopt_int_t buffer_size);

/* name of variable (kind 0) / constraint row (kind 1) `index` of a problem that came from cuOptReadProblem
 * (CUOPT_INVALID_ARGUMENT when the problem carries no names) */
cuopt_int_t cuOptAmdGetName(cuOptOptimizationProblem problem, cuopt_int_t kind, cuopt_int_t index, char* buffer,
                            cuopt_int_t buffer_size);

/* reads a .sol file (CUOPT_SOLUTION_FILE output, or MIPLIB style) into the variable order of `problem` (which must carry
 * variable names, i.e. come from cuOptReadProblem); objective_value / status may be NULL.  Mirrors
 * cpp/src/math_optimization/solution_reader.cu:57-145.  CUOPT_MPS_FILE_ERROR: cannot open; CUOPT_VALIDATION_ERROR: a
 * variable of the problem is not in the file. */
cuopt_int_t cuOptAmdReadSolutionFile(cuOptOptimizationProblem problem, const char* filename, cuopt_float_t* values,
                                     cuopt_float_t* objective_value, char* status, cuopt_int_t status_size);

#ifdef __cplusplus
}
#endif
#endif
