"""ctypes binding of cuopt_amd/lib/libcuopt.so.

Three layers are exposed, each a thin mirror of a C header under include/:
  * the libcuopt C API (cuopt/linear_programming/cuopt_c.h)       -> `Problem`, `Settings`, `solve`
  * the host driver (cuopt_amd/pdlp_solver.h)                     -> `Solver`
  * the HIP device layer (cuopt_amd/pdlp_device.h)                -> `Device`
There is NO CPU fallback: if the shared library is missing the import fails, and if no gfx950
device is visible the solver calls fail with the library's error message."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CUOPT_AMD_LIB") or os.path.join(_HERE, "lib", "libcuopt.so")  # override: A/B builds while tuning

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "cuopt_amd: %s is missing -- build it with `make -C cuopt_amd/csrc` (or "
        "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

# ---- constants (constants.h) ---------------------------------------------------------------------
CUOPT_SUCCESS, CUOPT_INVALID_ARGUMENT, CUOPT_MPS_FILE_ERROR, CUOPT_MPS_PARSE_ERROR = 0, 1, 2, 3
CUOPT_VALIDATION_ERROR, CUOPT_OUT_OF_MEMORY, CUOPT_RUNTIME_ERROR = 4, 5, 6
CUOPT_MINIMIZE, CUOPT_MAXIMIZE = 1, -1
STATUS = {0: "NoTermination", 1: "Optimal", 2: "PrimalInfeasible", 3: "DualInfeasible",
          4: "IterationLimit", 5: "TimeLimit", 6: "NumericalError", 7: "PrimalFeasible",
          8: "FeasibleFound", 9: "ConcurrentLimit"}

c_int, c_double, c_void_p, c_char_p = C.c_int32, C.c_double, C.c_void_p, C.c_char_p
P = C.POINTER


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


# ---- structs (pdlp_device.h / pdlp_solver.h) -----------------------------------------------------
class Ctl(C.Structure):
    _fields_ = [("step_size", c_double), ("primal_weight", c_double), ("tau", c_double),
                ("sigma", c_double), ("sum_weights", c_double), ("last_interaction", c_double),
                ("last_movement", c_double), ("last_dx2", c_double), ("last_dy2", c_double),
                ("k", c_int), ("cur", c_int), ("pending_avg", c_int), ("steps_taken", c_int),
                ("attempts", c_int), ("target_steps", c_int), ("error", c_int),
                ("its_since_restart", c_int)]


class StepParams(C.Structure):
    _fields_ = [("reduction_exponent", c_double), ("growth_exponent", c_double),
                ("primal_distance_smoothing", c_double), ("dual_distance_smoothing", c_double)]


class LP(C.Structure):
    _fields_ = [("m", c_int), ("n", c_int), ("offsets", c_void_p), ("indices", c_void_p),
                ("values", c_void_p), ("c", c_void_p), ("lo", c_void_p), ("hi", c_void_p),
                ("lb", c_void_p), ("ub", c_void_p), ("maximize", c_int),
                ("objective_offset", c_double)]


class Hyper(C.Structure):
    _fields_ = [("initial_step_size_scaling", c_double), ("ruiz_iterations", c_int),
                ("do_pock_chambolle", c_int), ("do_ruiz", c_int), ("alpha_pock_chambolle", c_double),
                ("artificial_restart_threshold", c_double),
                ("compute_initial_step_size_before_scaling", c_int),
                ("compute_initial_primal_weight_before_scaling", c_int),
                ("initial_primal_weight_c_scaling", c_double),
                ("initial_primal_weight_b_scaling", c_double), ("major_iteration", c_int),
                ("min_iteration_restart", c_int), ("restart_strategy", c_int),
                ("never_restart_to_average", c_int), ("reduction_exponent", c_double),
                ("growth_exponent", c_double), ("primal_weight_update_smoothing", c_double),
                ("sufficient_reduction_for_restart", c_double),
                ("necessary_reduction_for_restart", c_double), ("primal_importance", c_double),
                ("primal_distance_smoothing", c_double), ("dual_distance_smoothing", c_double),
                ("compute_last_restart_before_new_primal_weight", c_int),
                ("artificial_restart_in_main_loop", c_int), ("rescale_for_restart", c_int),
                ("update_primal_weight_on_initial_solution", c_int),
                ("update_step_size_on_initial_solution", c_int),
                ("handle_some_primal_gradients_on_finite_bounds_as_residuals", c_int),
                ("project_initial_primal", c_int)]


class SolverSettings(C.Structure):
    _fields_ = [("absolute_gap_tolerance", c_double), ("relative_gap_tolerance", c_double),
                ("absolute_primal_tolerance", c_double), ("relative_primal_tolerance", c_double),
                ("absolute_dual_tolerance", c_double), ("relative_dual_tolerance", c_double),
                ("iteration_limit", c_int), ("time_limit", c_double),
                ("per_constraint_residual", c_int), ("first_primal_feasible", c_int),
                ("initial_step_size", c_double), ("initial_primal_weight", c_double),
                ("initial_k", c_int), ("use_graph", c_int), ("detect_infeasibility", c_int),
                ("strict_infeasibility", c_int), ("primal_infeasible_tolerance", c_double),
                ("dual_infeasible_tolerance", c_double), ("save_best_primal_so_far", c_int),
                ("log_to_console", c_int), ("log_file", c_char_p), ("unbounded_from_feasible_iterates", c_int),
                ("accept_enabled", c_int), ("accept_tolerance", c_double * 6),
                ("relative_primal_tolerance_factor", c_double), ("relative_dual_tolerance_factor", c_double)]


class Result(C.Structure):
    _fields_ = [("status", c_int), ("steps_taken", c_int), ("attempted_steps", c_int),
                ("returned_average", c_int), ("num_restarts", c_int), ("num_major_iterations", c_int),
                ("primal_objective", c_double), ("dual_objective", c_double), ("gap", c_double),
                ("relative_gap", c_double), ("l2_primal_residual", c_double),
                ("l2_dual_residual", c_double), ("l2_relative_primal_residual", c_double),
                ("l2_relative_dual_residual", c_double), ("max_primal_ray_infeasibility", c_double),
                ("max_dual_ray_infeasibility", c_double), ("primal_ray_linear_objective", c_double),
                ("dual_ray_linear_objective", c_double), ("initial_step_size", c_double),
                ("initial_primal_weight", c_double), ("step_size", c_double),
                ("primal_weight", c_double), ("norm_b", c_double), ("norm_c", c_double),
                ("setup_seconds", c_double), ("loop_seconds", c_double),
                ("accepted_at_looser_tolerances", c_int), ("gpus", c_int)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["status_name"] = STATUS.get(self.status, str(self.status))
        return d


class WarmStart(C.Structure):
    _fields_ = [(k, c_void_p) for k in (
        "current_primal_solution", "current_dual_solution", "initial_primal_average",
        "initial_dual_average", "current_ATY", "sum_primal_solutions", "sum_dual_solutions",
        "last_restart_duality_gap_primal_solution", "last_restart_duality_gap_dual_solution",
        "current_primal_solution_scaled", "current_dual_solution_scaled")] + [
        ("initial_primal_weight", c_double), ("initial_step_size", c_double),
        ("total_pdlp_iterations", c_int), ("total_pdhg_iterations", c_int),
        ("last_candidate_kkt_score", c_double), ("last_restart_kkt_score", c_double),
        ("sum_solution_weight", c_double), ("iterations_since_last_restart", c_int),
        ("n_variables", c_int), ("n_constraints", c_int)]

    PRIMAL = ("current_primal_solution", "initial_primal_average", "current_ATY", "sum_primal_solutions",
              "last_restart_duality_gap_primal_solution", "current_primal_solution_scaled")
    DUAL = ("current_dual_solution", "initial_dual_average", "sum_dual_solutions",
            "last_restart_duality_gap_dual_solution", "current_dual_solution_scaled")
    SCALARS = ("initial_primal_weight", "initial_step_size", "total_pdlp_iterations", "total_pdhg_iterations",
               "last_candidate_kkt_score", "last_restart_kkt_score", "sum_solution_weight",
               "iterations_since_last_restart", "n_variables", "n_constraints")


def remap_warm_start(d, var_mapping=None, constraint_mapping=None):
    """set_pdlp_warm_start_data(data, var_mapping, constraint_mapping) (LP/solver_settings.cu:92-240): the snapshot `d` (a dict
    as Solver.get_warm_start returns it) carried over to a problem with len(var_mapping) variables / len(constraint_mapping)
    constraints; None or empty = that side is unchanged.  Host-side, no GPU involved."""
    vm = np.ascontiguousarray([] if var_mapping is None else var_mapping, dtype=np.int32)
    cm = np.ascontiguousarray([] if constraint_mapping is None else constraint_mapping, dtype=np.int32)
    n_new = len(vm) or int(d["n_variables"])
    m_new = len(cm) or int(d["n_constraints"])
    src, dst, keep, out = WarmStart(), WarmStart(), [], {}
    for names, size in ((WarmStart.PRIMAL, n_new), (WarmStart.DUAL, m_new)):
        for k in names:
            if d.get(k) is None:
                out[k] = None
                continue
            a = _f64(d[k])
            keep.append(a)
            setattr(src, k, a.ctypes.data)
            out[k] = np.zeros(size)
            setattr(dst, k, out[k].ctypes.data)
    for k in WarmStart.SCALARS:
        setattr(src, k, d[k])
    rc = lib.cuoptamd_warm_start_remap(C.byref(src), vm.ctypes.data_as(P(c_int)), len(vm), cm.ctypes.data_as(P(c_int)), len(cm),
                                       C.byref(dst))
    if rc != 0:
        raise CuOptError(rc, lib.cuoptamd_last_error().decode())
    for k in ("current_primal_solution_scaled", "current_dual_solution_scaled"):
        if not getattr(dst, k):
            out[k] = None
    out.update({k: getattr(dst, k) for k in WarmStart.SCALARS})
    return out


def _struct_dict(s):
    return {k: getattr(s, k) for k, _ in s._fields_}


# ---- prototypes ----------------------------------------------------------------------------------
def _proto(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype, f.argtypes = restype, list(argtypes)
    return f


_proto("cuOptGetFloatSize", C.c_int8)
_proto("cuOptGetIntSize", C.c_int8)
_proto("cuOptReadProblem", c_int, c_char_p, P(c_void_p))
_proto("cuOptCreateProblem", c_int, c_int, c_int, c_int, c_double, *([c_void_p] * 4), c_char_p,
       *([c_void_p] * 3), c_char_p, P(c_void_p))
_proto("cuOptCreateRangedProblem", c_int, c_int, c_int, c_int, c_double, *([c_void_p] * 8), c_char_p,
       P(c_void_p))
_proto("cuOptDestroyProblem", None, P(c_void_p))
for _n in ("cuOptGetNumConstraints", "cuOptGetNumVariables", "cuOptGetObjectiveSense",
           "cuOptGetNumNonZeros", "cuOptIsMIP"):
    _proto(_n, c_int, c_void_p, P(c_int))
_proto("cuOptGetObjectiveOffset", c_int, c_void_p, P(c_double))
for _n in ("cuOptGetObjectiveCoefficients", "cuOptGetConstraintSense", "cuOptGetConstraintRightHandSide",
           "cuOptGetConstraintLowerBounds", "cuOptGetConstraintUpperBounds",
           "cuOptGetVariableLowerBounds", "cuOptGetVariableUpperBounds", "cuOptGetVariableTypes"):
    _proto(_n, c_int, c_void_p, c_void_p)
_proto("cuOptGetConstraintMatrix", c_int, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuOptCreateSolverSettings", c_int, P(c_void_p))
_proto("cuOptDestroySolverSettings", None, P(c_void_p))
_proto("cuOptSetParameter", c_int, c_void_p, c_char_p, c_char_p)
_proto("cuOptGetParameter", c_int, c_void_p, c_char_p, c_int, c_char_p)
_proto("cuOptSetIntegerParameter", c_int, c_void_p, c_char_p, c_int)
_proto("cuOptGetIntegerParameter", c_int, c_void_p, c_char_p, P(c_int))
_proto("cuOptSetFloatParameter", c_int, c_void_p, c_char_p, c_double)
_proto("cuOptGetFloatParameter", c_int, c_void_p, c_char_p, P(c_double))
_proto("cuOptSolve", c_int, c_void_p, c_void_p, P(c_void_p))
_proto("cuOptDestroySolution", None, P(c_void_p))
_proto("cuOptGetTerminationStatus", c_int, c_void_p, P(c_int))
_proto("cuOptGetErrorStatus", c_int, c_void_p, P(c_int))
_proto("cuOptGetErrorString", c_int, c_void_p, c_char_p, c_int)
for _n in ("cuOptGetPrimalSolution", "cuOptGetDualSolution", "cuOptGetReducedCosts"):
    _proto(_n, c_int, c_void_p, c_void_p)
for _n in ("cuOptGetObjectiveValue", "cuOptGetSolveTime", "cuOptGetMIPGap", "cuOptGetSolutionBound"):
    _proto(_n, c_int, c_void_p, P(c_double))
_proto("cuOptAmdGetPdlpStats", c_int, c_void_p, P(Result))
_proto("cuOptAmdGetSolveInfo", c_int, c_void_p, c_char_p, c_int)
_proto("cuOptAmdGetName", c_int, c_void_p, c_int, c_int, c_char_p, c_int)

_proto("cuoptamd_last_error", c_char_p)
_proto("cuoptamd_hyper_preset", None, c_int, P(Hyper))
_proto("cuoptamd_default_settings", None, P(SolverSettings))
_proto("cuoptamd_solver_create", c_int, P(c_void_p), P(LP), P(Hyper), P(SolverSettings), c_void_p,
       c_void_p, c_int, c_int, c_int, c_void_p)
_proto("cuoptamd_solver_destroy", None, c_void_p)
_proto("cuoptamd_solver_reset", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, P(SolverSettings), c_void_p,
       c_void_p)
_proto("cuoptamd_solver_advance", c_int, c_void_p, c_int, P(Result))
_proto("cuoptamd_solver_get_solution", c_int, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuoptamd_solver_device", c_void_p, c_void_p)
_proto("cuoptamd_batch_solve", c_int, c_int, c_void_p, P(Hyper), P(SolverSettings), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuoptamd_solver_get_warm_start", c_int, c_void_p, P(WarmStart))
_proto("cuoptamd_solver_set_warm_start", c_int, c_void_p, P(WarmStart))
_proto("cuoptamd_warm_start_remap", c_int, P(WarmStart), P(c_int), c_int, P(c_int), c_int, P(WarmStart))
_proto("cuoptamd_solver_row_range", c_int, c_void_p, P(c_int), P(c_int))
_proto("cuoptamd_solver_reorder_info", c_int, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuoptamd_partition_rows", None, c_int, c_void_p, c_int, c_void_p)
_proto("cuoptamd_csr_transpose", None, c_int, c_int, *([c_void_p] * 6))

_proto("pdlpdev_last_error", c_char_p)
_proto("pdlpdev_device_count", c_int)
_proto("pdlpdev_device_info", c_int, c_int, c_char_p, c_int, P(c_int), P(C.c_int64))
_proto("pdlpdev_create", c_int, P(c_void_p), c_int, c_int, c_int, *([c_void_p] * 11))
_proto("pdlpdev_destroy", None, c_void_p)
_proto("pdlpdev_comm_unique_id", c_int, c_void_p)
_proto("pdlpdev_comm_init", c_int, c_void_p, c_int, c_int, c_void_p)
_proto("pdlpdev_softcomm_create", c_int, c_int, c_void_p)
_proto("pdlpdev_scaling_compute", c_int, c_void_p, c_int, c_int, c_int, c_double)
_proto("pdlpdev_scale_problem", c_int, c_void_p)
_proto("pdlpdev_init_norms", c_int, c_void_p, c_void_p)
_proto("pdlpdev_problem_norms", c_int, c_void_p, P(c_double), P(c_double))
_proto("pdlpdev_set_step_params", c_int, c_void_p, P(StepParams))
_proto("pdlpdev_set_step", c_int, c_void_p, c_double, c_double)
_proto("pdlpdev_set_k", c_int, c_void_p, c_int)
_proto("pdlpdev_set_initial", c_int, c_void_p, c_void_p, c_void_p)
_proto("pdlpdev_project_primal", c_int, c_void_p)
_proto("pdlpdev_compute_aty", c_int, c_void_p)
_proto("pdlpdev_run", c_int, c_void_p, c_int, P(Ctl))
_proto("pdlpdev_get_ctl", c_int, c_void_p, P(Ctl))
_proto("pdlpdev_clear_error", c_int, c_void_p)
_proto("pdlpdev_set_graph_mode", c_int, c_void_p, c_int)
_proto("pdlpdev_prepare_graphs", c_int, c_void_p)
_proto("pdlpdev_flush_average", c_int, c_void_p)
_proto("pdlpdev_make_average", c_int, c_void_p, c_int)
_proto("pdlpdev_eval", c_int, c_void_p, c_int, c_int, c_double, c_double, c_void_p)
_proto("pdlpdev_restart", c_int, c_void_p, c_int, c_int, c_void_p)
_proto("pdlpdev_save_best", c_int, c_void_p, c_int)
_proto("pdlpdev_trust_region_bounds", c_int, c_void_p, c_int, c_double, c_double, c_double, c_double, c_double, c_double, c_int, c_void_p)
_proto("pdlpdev_eval_infeasibility", c_int, c_void_p, c_int, c_int, c_void_p)
_proto("pdlpdev_get_solution", c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p)
_proto("pdlpdev_download", C.c_int64, c_void_p, c_int, c_void_p, C.c_int64)
_proto("pdlpdev_spmv", c_int, c_void_p, c_int, c_void_p, c_void_p)
_proto("pdlpdev_time_kernel", c_int, c_void_p, c_int, c_int, P(c_double))
_proto("pdlpdev_synchronize", c_int, c_void_p)
_proto("pdlpdev_device_bytes", C.c_int64, c_void_p)
_proto("cuoptamd_dual_simplex", c_int, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuoptamd_dual_simplex_from", c_int, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("pdlpdev_shard_dataflow", c_int, c_void_p)
_proto("pdlpdev_shard_transport", c_int, c_void_p)
_proto("pdlpdev_shard_wire_bytes", c_int, c_void_p, c_void_p)
_proto("pdlpdev_dense_info", c_int, c_void_p, c_void_p)
_proto("pdlpdev_layout_info", c_int, c_void_p, c_void_p)
# device-side set-up (round 5)
_proto("pdlpdev_analyze", c_int, P(c_void_p), c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int)
_proto("pdlpdev_analysis_info", c_int, c_void_p, c_void_p)
_proto("pdlpdev_analysis_maps", c_int, c_void_p, c_void_p, c_void_p)
_proto("pdlpdev_analysis_download", c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p)
_proto("pdlpdev_create_from_analysis", c_int, P(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("pdlpdev_analysis_destroy", None, c_void_p)
_proto("pdlpdev_debug_sort_pairs", c_int, c_int, C.c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p)
_proto("pdlpdev_debug_scan", c_int, c_int, C.c_int64, c_void_p, c_void_p)
_proto("pdlpdev_debug_layout_checksums", c_int, c_void_p, c_void_p)
# shared-matrix batch (round 5)
_proto("cuoptamd_solver_clone", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, P(c_void_p))
_proto("cuoptamd_batch_create", c_int, c_void_p, c_int, P(c_void_p))
_proto("cuoptamd_batch_advance", c_int, c_void_p, c_int, c_void_p)
_proto("cuoptamd_batch_destroy", None, c_void_p)
_proto("cuoptamd_batch_reset", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuoptamd_batch_get_solutions", c_int, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuoptamd_batch_branch", c_int, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuoptamd_batch_solution_views", c_int, c_void_p, c_void_p, c_void_p, c_void_p)
_proto("cuoptamd_batch_device", c_void_p, c_void_p)
_proto("pdlpdev_create_share_stream", None, c_void_p)
_proto("pdlpdev_debug_ipc_export", c_int, c_int, c_int, c_void_p, P(c_void_p))
_proto("pdlpdev_debug_ipc_store", c_int, c_int, c_void_p, c_int, c_double)
_proto("pdlpdev_debug_ipc_wait", c_int, c_int, c_void_p, c_int, c_double)
_proto("pdlpdev_resident_size", c_int, c_int, c_int, C.c_int64)
_proto("pdlpdev_batch_time_kernels", c_int, c_void_p, c_int, c_void_p)
_proto("pdlpdev_synthetic_lp", c_int, c_int, c_int, c_int, c_int, C.c_uint64, *([c_void_p] * 8))

# ids of pdlp_device.h
BUF = {n: i for i, n in enumerate(
    ["X", "Y", "X_OTHER", "Y_OTHER", "ATY", "ATY_OTHER", "XBAR", "SUM_X", "SUM_Y", "AVG_X", "AVG_Y",
     "DROW", "DCOL", "A_VALUES", "AT_VALUES", "C", "LB", "UB", "LO", "HI", "RC_CURRENT", "RC_AVERAGE",
     "LAST_RESTART_X", "LAST_RESTART_Y"])}
KERNEL = {n: i for i, n in enumerate(
    ["PRIMAL", "SPMV_A_DUAL", "SPMV_AT_STEP", "STEP_DECISION", "SPMV_A_PLAIN", "SPMV_AT_PLAIN"])}
EV = {n: i for i, n in enumerate(
    ["CX", "DUAL_SUM", "PRES2", "DRES2", "X2", "Y2", "LINF_PRES_REL", "LINF_DRES_REL"])}
CURRENT, AVERAGE = 0, 1


class CuOptError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("cuopt error %d %s" % (code, msg))
        self.code = code


def device_count():
    return int(lib.pdlpdev_device_count())


def device_info(dev=0):
    name = C.create_string_buffer(256)
    cus, mem = c_int(), C.c_int64()
    rc = lib.pdlpdev_device_info(dev, name, 256, C.byref(cus), C.byref(mem))
    if rc != 0:
        raise CuOptError(rc, lib.pdlpdev_last_error().decode())
    return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=mem.value)


# ---- libcuopt C API --------------------------------------------------------------------------------
class Problem:
    """cuOptOptimizationProblem handle."""

    def __init__(self, handle):
        self.handle = handle

    @classmethod
    def read(cls, path):
        h = c_void_p()
        rc = lib.cuOptReadProblem(os.fsencode(path), C.byref(h))
        if rc != CUOPT_SUCCESS:
            raise CuOptError(rc, "cuOptReadProblem(%s)" % path)
        return cls(h)

    @classmethod
    def from_dict(cls, p, ranged=True):
        """p: dict with m, n, offsets, indices, values, c, lb, ub and (lo, hi) [ranged] or
        (row_types, rhs); optional maximize, objective_offset, var_types."""
        m, n = int(p["m"]), int(p["n"])
        keep = [_i32(p["offsets"]), _i32(p["indices"]), _f64(p["values"]), _f64(p["c"]),
                _f64(p["lb"]), _f64(p["ub"])]
        vt = p.get("var_types")
        vt = (b"C" * n) if vt is None else bytes(bytearray(np.asarray(vt, dtype=np.uint8)))
        h = c_void_p()
        sense = CUOPT_MAXIMIZE if p.get("maximize", False) else CUOPT_MINIMIZE
        off = float(p.get("objective_offset", 0.0))
        if ranged:
            lo, hi = _f64(p["lo"]), _f64(p["hi"])
            rc = lib.cuOptCreateRangedProblem(m, n, sense, off, _ptr(keep[3]), _ptr(keep[0]),
                                              _ptr(keep[1]), _ptr(keep[2]), _ptr(lo), _ptr(hi),
                                              _ptr(keep[4]), _ptr(keep[5]), vt, C.byref(h))
        else:
            rt = bytes(bytearray(np.asarray(p["row_types"], dtype=np.uint8)))
            rhs = _f64(p["rhs"])
            rc = lib.cuOptCreateProblem(m, n, sense, off, _ptr(keep[3]), _ptr(keep[0]), _ptr(keep[1]),
                                        _ptr(keep[2]), rt, _ptr(rhs), _ptr(keep[4]), _ptr(keep[5]), vt,
                                        C.byref(h))
        if rc != CUOPT_SUCCESS:
            raise CuOptError(rc, "cuOptCreate*Problem")
        return cls(h)

    def _int(self, fn):
        v = c_int()
        rc = fn(self.handle, C.byref(v))
        if rc != 0:
            raise CuOptError(rc)
        return v.value

    @property
    def m(self):
        return self._int(lib.cuOptGetNumConstraints)

    @property
    def n(self):
        return self._int(lib.cuOptGetNumVariables)

    @property
    def nnz(self):
        return self._int(lib.cuOptGetNumNonZeros)

    @property
    def is_mip(self):
        return bool(self._int(lib.cuOptIsMIP))

    def to_dict(self):
        m, n, nnz = self.m, self.n, self.nnz
        d = dict(m=m, n=n, offsets=np.zeros(m + 1, np.int32), indices=np.zeros(nnz, np.int32),
                 values=np.zeros(nnz), c=np.zeros(n), lo=np.zeros(m), hi=np.zeros(m), lb=np.zeros(n),
                 ub=np.zeros(n), var_types=np.zeros(n, np.uint8))
        lib.cuOptGetConstraintMatrix(self.handle, _ptr(d["offsets"]), _ptr(d["indices"]), _ptr(d["values"]))
        lib.cuOptGetObjectiveCoefficients(self.handle, _ptr(d["c"]))
        lib.cuOptGetConstraintLowerBounds(self.handle, _ptr(d["lo"]))
        lib.cuOptGetConstraintUpperBounds(self.handle, _ptr(d["hi"]))
        lib.cuOptGetVariableLowerBounds(self.handle, _ptr(d["lb"]))
        lib.cuOptGetVariableUpperBounds(self.handle, _ptr(d["ub"]))
        lib.cuOptGetVariableTypes(self.handle, _ptr(d["var_types"]))
        d["maximize"] = self._int(lib.cuOptGetObjectiveSense) == CUOPT_MAXIMIZE
        off = c_double()
        lib.cuOptGetObjectiveOffset(self.handle, C.byref(off))
        d["objective_offset"] = off.value
        return d

    def names(self, kind):
        """variable (kind 0) or row (kind 1) names of a problem read from an MPS file; [] when it has none"""
        out, buf = [], C.create_string_buffer(4096)
        for i in range(self.n if kind == 0 else self.m):
            if lib.cuOptAmdGetName(self.handle, kind, i, buf, 4096) != CUOPT_SUCCESS:
                return []
            out.append(buf.value.decode())
        return out

    def close(self):
        if self.handle:
            lib.cuOptDestroyProblem(C.byref(self.handle))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Settings:
    """cuOptSolverSettings handle; parameters by their CUOPT_* string names."""

    def __init__(self, **params):
        self.handle = c_void_p()
        rc = lib.cuOptCreateSolverSettings(C.byref(self.handle))
        if rc != 0:
            raise CuOptError(rc)
        params.setdefault("log_to_console", False)
        for k, v in params.items():
            self.set(k, v)

    def set(self, name, value):
        nm = name.encode()
        if isinstance(value, bool):
            rc = lib.cuOptSetIntegerParameter(self.handle, nm, int(value))
        elif isinstance(value, (int, np.integer)):
            rc = lib.cuOptSetIntegerParameter(self.handle, nm, int(value))
        elif isinstance(value, (float, np.floating)):
            rc = lib.cuOptSetFloatParameter(self.handle, nm, float(value))
        else:
            rc = lib.cuOptSetParameter(self.handle, nm, str(value).encode())
        if rc != 0:
            raise CuOptError(rc, "set %s=%r" % (name, value))

    def set_optimality_tolerance(self, eps):
        for k in ("absolute_dual_tolerance", "relative_dual_tolerance", "absolute_primal_tolerance",
                  "relative_primal_tolerance", "absolute_gap_tolerance", "relative_gap_tolerance"):
            self.set(k, float(eps))

    def get(self, name):
        buf = C.create_string_buffer(256)
        rc = lib.cuOptGetParameter(self.handle, name.encode(), 256, buf)
        if rc != 0:
            raise CuOptError(rc, "get %s" % name)
        return buf.value.decode()

    def close(self):
        if self.handle:
            lib.cuOptDestroySolverSettings(C.byref(self.handle))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve(problem, settings=None, **params):
    """cuOptSolve + all solution getters -> dict.  `problem` is a Problem or a problem dict."""
    own_problem = not isinstance(problem, Problem)
    prob = Problem.from_dict(problem) if own_problem else problem
    own_settings = settings is None
    tol = params.pop("tol", None)
    st = Settings(**params) if own_settings else settings
    if tol is not None:
        st.set_optimality_tolerance(tol)
    sol = c_void_p()
    rc = lib.cuOptSolve(prob.handle, st.handle, C.byref(sol))
    out = dict(return_code=rc)
    try:
        if not sol:
            raise CuOptError(rc, "cuOptSolve returned no solution handle")
        v = c_int()
        lib.cuOptGetTerminationStatus(sol, C.byref(v))
        out["status_code"], out["status"] = v.value, STATUS.get(v.value, str(v.value))
        lib.cuOptGetErrorStatus(sol, C.byref(v))
        out["error_status"] = v.value
        buf = C.create_string_buffer(4096)
        lib.cuOptGetErrorString(sol, buf, 4096)
        out["error_string"] = buf.value.decode()
        if rc == CUOPT_SUCCESS:
            n, m = prob.n, prob.m
            x, y, z = np.zeros(n), np.zeros(m), np.zeros(n)
            lib.cuOptGetPrimalSolution(sol, _ptr(x))
            lib.cuOptGetDualSolution(sol, _ptr(y))
            lib.cuOptGetReducedCosts(sol, _ptr(z))
            d = c_double()
            lib.cuOptGetObjectiveValue(sol, C.byref(d))
            out["objective"] = d.value
            lib.cuOptGetSolveTime(sol, C.byref(d))
            out["solve_time"] = d.value
            res = Result()
            lib.cuOptAmdGetPdlpStats(sol, C.byref(res))
            out.update({k: v for k, v in res.as_dict().items() if k != "status"})
            out.update(x=x, y=y, reduced_cost=z)
            info = C.create_string_buffer(1024)
            lib.cuOptAmdGetSolveInfo(sol, info, 1024)
            try:
                import json as _json
                out["solve_info"] = _json.loads(info.value.decode())
            except ValueError:
                out["solve_info"] = info.value.decode()
    finally:
        if sol:
            lib.cuOptDestroySolution(C.byref(sol))
        if own_settings:
            st.close()
        if own_problem:
            prob.close()
    return out


# ---- host driver -----------------------------------------------------------------------------------
def hyper_preset(mode=1):
    h = Hyper()
    lib.cuoptamd_hyper_preset(int(mode), C.byref(h))
    return h


def default_settings(**over):
    s = SolverSettings()
    lib.cuoptamd_default_settings(C.byref(s))
    tol = over.pop("tol", None)
    if tol is not None:
        for k in ("absolute_gap_tolerance", "relative_gap_tolerance", "absolute_primal_tolerance",
                  "relative_primal_tolerance", "absolute_dual_tolerance", "relative_dual_tolerance"):
            setattr(s, k, float(tol))
    for k, v in over.items():
        setattr(s, k, v)
    return s


def comm_unique_id():
    buf = (C.c_uint8 * 128)()
    rc = lib.pdlpdev_comm_unique_id(buf)
    if rc != 0:
        raise CuOptError(rc, lib.pdlpdev_last_error().decode())
    return bytes(buf)


def softcomm_id(world):
    """token of an in-process communicator for `world` solver threads (verification only)"""
    buf = (C.c_uint8 * 128)()
    rc = lib.pdlpdev_softcomm_create(int(world), buf)
    if rc != 0:
        raise CuOptError(rc, lib.pdlpdev_last_error().decode())
    return bytes(buf)


SIMPLEX_STATUS = {1: "Optimal", 2: "PrimalInfeasible", 3: "Unbounded", 5: "IterationLimit", 6: "TimeLimit", 7: "NumericalError",
                  8: "TooLarge", 9: "Cancelled"}


def dual_simplex(p, time_limit=0.0, iteration_limit=0, x0=None, y0=None):
    """cuoptamd_dual_simplex: the library's own dual simplex (host code, no GPU involved); x0: start from the basis guessed from
    that point (cuoptamd_dual_simplex_from: the crossover of a first-order solution)"""
    k = dict(offsets=_i32(p["offsets"]), indices=_i32(p["indices"]), values=_f64(p["values"]), c=_f64(p["c"]), lo=_f64(p["lo"]),
             hi=_f64(p["hi"]), lb=_f64(p["lb"]), ub=_f64(p["ub"]))
    m, n = int(p["m"]), int(p["n"])
    lp = LP(m, n, _ptr(k["offsets"]), _ptr(k["indices"]), _ptr(k["values"]), _ptr(k["c"]), _ptr(k["lo"]), _ptr(k["hi"]), _ptr(k["lb"]),
            _ptr(k["ub"]), int(bool(p.get("maximize", False))), float(p.get("objective_offset", 0.0)))
    status, its, obj = c_int(0), c_int(0), c_double(0.0)
    x, y, rc = np.zeros(n), np.zeros(m), np.zeros(n)
    if x0 is not None:
        start, duals = _f64(x0), None if y0 is None else _f64(y0)
        ret = lib.cuoptamd_dual_simplex_from(C.byref(lp), _ptr(start), None if duals is None else _ptr(duals), float(time_limit), int(iteration_limit), None, C.byref(status),
                                             C.byref(its), C.byref(obj), _ptr(x), _ptr(y), _ptr(rc))
    else:
        ret = lib.cuoptamd_dual_simplex(C.byref(lp), float(time_limit), int(iteration_limit), None, C.byref(status), C.byref(its),
                                        C.byref(obj), _ptr(x), _ptr(y), _ptr(rc))
    if ret != 0:
        raise CuOptError(ret, "cuoptamd_dual_simplex")
    return dict(status=SIMPLEX_STATUS.get(status.value, str(status.value)), iterations=its.value, objective=obj.value, x=x, y=y,
                reduced_cost=rc)


class Solver:
    """cuoptamd_solver: step-wise control of the PDLP loop (bench, warm-started re-solves, tests)."""

    def __init__(self, p, mode=1, hyper=None, settings=None, init_x=None, init_y=None, device=0,
                 rank=0, world=1, comm_id=None, warm_start=None, **setting_overrides):
        self._keep = dict(offsets=_i32(p["offsets"]), indices=_i32(p["indices"]), values=_f64(p["values"]),
                          c=_f64(p["c"]), lo=_f64(p["lo"]), hi=_f64(p["hi"]), lb=_f64(p["lb"]),
                          ub=_f64(p["ub"]))
        k = self._keep
        self.m, self.n = int(p["m"]), int(p["n"])
        lp = LP(self.m, self.n, _ptr(k["offsets"]), _ptr(k["indices"]), _ptr(k["values"]), _ptr(k["c"]),
                _ptr(k["lo"]), _ptr(k["hi"]), _ptr(k["lb"]), _ptr(k["ub"]),
                int(bool(p.get("maximize", False))), float(p.get("objective_offset", 0.0)))
        self.hyper = hyper_preset(mode) if hyper is None else hyper
        self.settings = default_settings(**setting_overrides) if settings is None else settings
        ix = None if init_x is None else _f64(init_x)
        iy = None if init_y is None else _f64(init_y)
        cid = None
        if comm_id is not None:
            cid = (C.c_uint8 * 128).from_buffer_copy(comm_id)
        self.handle = c_void_p()
        rc = lib.cuoptamd_solver_create(C.byref(self.handle), C.byref(lp), C.byref(self.hyper),
                                        C.byref(self.settings), _ptr(ix), _ptr(iy), device, rank, world,
                                        cid)
        if rc != 0:
            msg = lib.cuoptamd_last_error().decode()
            h, self.handle = self.handle, None
            if h:
                lib.cuoptamd_solver_destroy(h)
            raise CuOptError(rc, msg)
        self.result = Result()
        if warm_start is not None:
            self.set_warm_start(warm_start)

    def get_warm_start(self):
        """pdlp_warm_start_data_t of the terminated solve as a dict of numpy arrays / scalars"""
        ws, arrays = WarmStart(), {}
        for k in WarmStart.PRIMAL:
            arrays[k] = np.zeros(self.n)
        for k in WarmStart.DUAL:
            arrays[k] = np.zeros(self.m)
        for k, a in arrays.items():
            setattr(ws, k, a.ctypes.data)
        rc = lib.cuoptamd_solver_get_warm_start(self.handle, C.byref(ws))
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        out = dict(arrays)
        out.update({k: getattr(ws, k) for k in WarmStart.SCALARS})
        return out

    def set_warm_start(self, d):
        ws, keep = WarmStart(), []
        for k in WarmStart.PRIMAL + WarmStart.DUAL:
            if d.get(k) is None:
                continue  # optional vectors (scaled iterate) may be absent: NULL
            a = _f64(d[k])
            keep.append(a)
            setattr(ws, k, a.ctypes.data)
        for k in WarmStart.SCALARS:
            setattr(ws, k, d[k])
        rc = lib.cuoptamd_solver_set_warm_start(self.handle, C.byref(ws))
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())

    def reset(self, lb=None, ub=None, lo=None, hi=None, init_x=None, init_y=None, **setting_overrides):
        """cuoptamd_solver_reset: new bounds (None = unchanged) on the same A, c; afterwards the solver is
        indistinguishable from one freshly created on the modified LP (MIP-style re-solve without set-up)."""
        arrays = [None if a is None else _f64(a) for a in (lb, ub, lo, hi, init_x, init_y)]
        st = None
        if setting_overrides:
            self.settings = default_settings(**setting_overrides)
            st = C.byref(self.settings)
        rc = lib.cuoptamd_solver_reset(self.handle, _ptr(arrays[0]), _ptr(arrays[1]), _ptr(arrays[2]), _ptr(arrays[3]),
                                       st, _ptr(arrays[4]), _ptr(arrays[5]))
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        self.result = Result()

    def advance(self, iterations=2 ** 31 - 1):
        rc = lib.cuoptamd_solver_advance(self.handle, int(iterations), C.byref(self.result))
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        return self.result.as_dict()

    def solution(self):
        x, y, z = np.zeros(self.n), np.zeros(self.m), np.zeros(self.n)
        rc = lib.cuoptamd_solver_get_solution(self.handle, _ptr(x), _ptr(y), _ptr(z))
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        return x, y, z

    @property
    def device(self):
        return Device(handle=c_void_p(lib.cuoptamd_solver_device(self.handle)), owner=False)

    def row_range(self):
        a, b = c_int(), c_int()
        lib.cuoptamd_solver_row_range(self.handle, C.byref(a), C.byref(b))
        return a.value, b.value

    def reorder_info(self, maps=False):
        """what the set-up's analysis pass found: {"reordered", "method" (none | chains | groups), estimates ...}; maps=True adds
        row_new2old / col_new2old when the device works on a permuted LP"""
        info = np.zeros(10, dtype=np.int32)
        rows = np.zeros(self.m, dtype=np.int32) if maps else None
        cols = np.zeros(self.n, dtype=np.int32) if maps else None
        on = lib.cuoptamd_solver_reorder_info(self.handle, _ptr(info), _ptr(rows), _ptr(cols))
        out = dict(reordered=bool(on == 1), method={0: "none", 1: "chains", 2: "groups"}.get(int(info[1]), "?"),
                   estimate_natural=(info[2] / 1e4, info[3] / 1e4), estimate_chains=(info[4] / 1e4, info[5] / 1e4),
                   estimate_groups=(info[6] / 1e4, info[7] / 1e4), search_levels=int(info[8]), cell_rounds=int(info[9]))
        if maps and on == 1:
            out["row_new2old"], out["col_new2old"] = rows, cols
        return out

    def clone(self, lb=None, ub=None, lo=None, hi=None, **setting_overrides):
        """cuoptamd_solver_clone: a solver for this LP under other bounds (None = this solver's current ones) that shares the
        matrices on the device; bit for bit a solver freshly created on that LP.  Close the clones before this solver."""
        arrays = [None if a is None else _f64(a) for a in (lb, ub, lo, hi)]
        child = Solver.__new__(Solver)
        child._keep, child.m, child.n, child.hyper = self._keep, self.m, self.n, self.hyper
        child.settings = default_settings(**setting_overrides) if setting_overrides else self.settings
        child._parent = self  # (keeps the parent alive)
        child.handle = c_void_p()
        rc = lib.cuoptamd_solver_clone(self.handle, _ptr(arrays[0]), _ptr(arrays[1]), _ptr(arrays[2]), _ptr(arrays[3]),
                                       C.byref(child.settings) if setting_overrides else None, C.byref(child.handle))
        if rc != 0:
            child.handle = None
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        child.result = Result()
        return child

    def close(self):
        if self.handle:
            lib.cuoptamd_solver_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SharedMatrixBatch:
    """cuoptamd_batch: K = 2, 4, 8 or 16 Solvers over ONE matrix (a parent and its clones) advance in lockstep, the matrix streamed once
    per attempt for all of them -- or (round 6) ANY number of Solvers on the resident small-LP path, whatever their matrices, one
    workgroup each in one launch per phase; every LP's trajectory is bit-identical to its own Solver.advance.  CuOptError(-7) when
    the solvers are eligible for neither (the caller then advances them one by one)."""

    def __init__(self, solvers):
        self.solvers = list(solvers)
        k = len(self.solvers)
        arr = (c_void_p * k)(*[s.handle.value for s in self.solvers])
        self.handle = c_void_p()
        rc = lib.cuoptamd_batch_create(arr, k, C.byref(self.handle))
        if rc != 0:
            self.handle = None
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())

    def advance(self, iterations=2 ** 31 - 1):
        k = len(self.solvers)
        results = (Result * k)()
        rc = lib.cuoptamd_batch_advance(self.handle, int(iterations), results)
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        out = []
        for s, r in zip(self.solvers, results):
            C.memmove(C.byref(s.result), C.byref(r), C.sizeof(Result))
            out.append(s.result.as_dict())
        return out

    def reset(self, lb=None, ub=None, init_x=None, init_y=None):
        """cuoptamd_batch_reset: per solver new variable bounds and / or a start (lists of arrays or None, entries may be None), one
        launch for all of them; afterwards every solver is what Solver.reset(lb=, ub=, init_x=, init_y=) would have made it"""
        k = len(self.solvers)
        keep = []

        def ptrs(vs):
            if vs is None:
                return None
            arr = (c_void_p * k)()
            for i, v in enumerate(vs):
                if v is not None:
                    a = _f64(v)
                    keep.append(a)
                    arr[i] = a.ctypes.data
            return arr
        rc = lib.cuoptamd_batch_reset(self.handle, ptrs(lb), ptrs(ub), ptrs(init_x), ptrs(init_y))
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        for s in self.solvers:
            s.result = Result()

    def branch(self, var, lb, ub):
        """cuoptamd_batch_branch: variable var[l] of solver l gets the bounds [lb[l], ub[l]] (var[l] < 0: none), every solver starts again
        from the primal / dual its last solve returned; one launch, nothing but the three arrays crosses PCIe"""
        v, a, c = _i32(var), _f64(lb), _f64(ub)
        rc = lib.cuoptamd_batch_branch(self.handle, _ptr(v), _ptr(a), _ptr(c))
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        for s in self.solvers:
            s.result = Result()

    def solution_views(self):
        """cuoptamd_batch_solution_views: [(x, y, reduced costs)] as numpy views of the batch's pinned staging block -- no copies; valid
        until the next reset / branch / solutions call of this batch"""
        k = len(self.solvers)
        px, py, pz = (c_void_p * k)(), (c_void_p * k)(), (c_void_p * k)()
        rc = lib.cuoptamd_batch_solution_views(self.handle, px, py, pz)
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        key = (tuple(px), tuple(py), tuple(pz))  # (the staging block does not move: the numpy views are built once)
        if getattr(self, "_view_key", None) != key:
            view = lambda p, n: np.ctypeslib.as_array(C.cast(p, P(c_double)), shape=(n,)) if p else np.zeros(n)
            self._views = [(view(px[i], s.n), view(py[i], s.m), view(pz[i], s.n)) for i, s in enumerate(self.solvers)]
            self._view_key = key
        return self._views

    def solutions(self):
        """cuoptamd_batch_get_solutions: [(x, y, reduced costs)] of every solver, one launch"""
        k = len(self.solvers)
        xs = [np.zeros(s.n) for s in self.solvers]
        ys = [np.zeros(s.m) for s in self.solvers]
        zs = [np.zeros(s.n) for s in self.solvers]
        arr = lambda vs: (c_void_p * k)(*[v.ctypes.data for v in vs])
        rc = lib.cuoptamd_batch_get_solutions(self.handle, arr(xs), arr(ys), arr(zs))
        if rc != 0:
            raise CuOptError(rc, lib.cuoptamd_last_error().decode())
        return list(zip(xs, ys, zs))

    def time_kernels(self, reps=20):
        """average dispatch time (ms) of the four kernels of a batched attempt: dict primal / a_dual / at_step / decisions"""
        out = np.zeros(4)
        rc = lib.pdlpdev_batch_time_kernels(c_void_p(lib.cuoptamd_batch_device(self.handle)), int(reps), _ptr(out))
        if rc != 0:
            raise CuOptError(rc, lib.pdlpdev_last_error().decode())
        return dict(zip(("primal", "a_dual", "at_step", "decisions"), out.tolist()))

    def close(self):
        if self.handle:
            lib.cuoptamd_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SmallBatch = SharedMatrixBatch  # (the same cuoptamd_batch object; the name the small-LP tests and bench use)


def batch_solve(problems, mode=1, max_threads=0, device=0, **setting_overrides):
    """cuoptamd_batch_solve: independent LPs solved concurrently on one GPU -> list of result dicts"""
    k = len(problems)
    lps, keep = (LP * k)(), []
    for i, p in enumerate(problems):
        a = dict(offsets=_i32(p["offsets"]), indices=_i32(p["indices"]), values=_f64(p["values"]), c=_f64(p["c"]),
                 lo=_f64(p["lo"]), hi=_f64(p["hi"]), lb=_f64(p["lb"]), ub=_f64(p["ub"]))
        keep.append(a)
        lps[i] = LP(int(p["m"]), int(p["n"]), _ptr(a["offsets"]), _ptr(a["indices"]), _ptr(a["values"]), _ptr(a["c"]),
                    _ptr(a["lo"]), _ptr(a["hi"]), _ptr(a["lb"]), _ptr(a["ub"]), int(bool(p.get("maximize", False))),
                    float(p.get("objective_offset", 0.0)))
    hyper, settings = hyper_preset(mode), default_settings(**setting_overrides)
    results = (Result * k)()
    xs = [np.zeros(int(p["n"])) for p in problems]
    ys = [np.zeros(int(p["m"])) for p in problems]
    zs = [np.zeros(int(p["n"])) for p in problems]
    arr = lambda vs: (c_void_p * k)(*[v.ctypes.data for v in vs])
    ax, ay, az = arr(xs), arr(ys), arr(zs)
    rc = lib.cuoptamd_batch_solve(k, lps, C.byref(hyper), C.byref(settings), device, int(max_threads), results, ax, ay, az)
    if rc != 0:
        raise CuOptError(rc, lib.cuoptamd_last_error().decode())
    out = []
    for i in range(k):
        d = results[i].as_dict()
        d.update(x=xs[i], y=ys[i], reduced_cost=zs[i])
        out.append(d)
    return out


def csr_transpose(m, n, offsets, indices, values):
    offsets, indices, values = _i32(offsets), _i32(indices), _f64(values)
    to, ti, tv = np.zeros(n + 1, np.int32), np.zeros(len(indices), np.int32), np.zeros(len(values))
    lib.cuoptamd_csr_transpose(m, n, _ptr(offsets), _ptr(indices), _ptr(values), _ptr(to), _ptr(ti), _ptr(tv))
    return to, ti, tv


def partition_rows(m, offsets, world):
    offsets = _i32(offsets)
    b = np.zeros(world + 1, np.int32)
    lib.cuoptamd_partition_rows(m, _ptr(offsets), world, _ptr(b))
    return b


# ---- device layer ------------------------------------------------------------------------------------
class Analysis:
    """pdlpdev_analysis: ONE upload of A, the transpose and the structure-finding analysis pass on the device (pdlp_device.h
    "device-side set-up").  The arrays of `p` stay referenced until the object is consumed by Device(analysis=...) or closed."""

    def __init__(self, p, reorder=True, device=0):
        self.m, self.n = int(p["m"]), int(p["n"])
        self._keep = (_i32(p["offsets"]), _i32(p["indices"]), _f64(p["values"]))
        self.nnz = len(self._keep[2])
        self.handle = c_void_p()
        rc = lib.pdlpdev_analyze(C.byref(self.handle), device, self.m, self.n, _ptr(self._keep[0]), _ptr(self._keep[1]),
                                 _ptr(self._keep[2]), 1 if reorder else 0)
        if rc != 0:
            msg = lib.pdlpdev_last_error().decode()
            self.close()
            raise CuOptError(rc, msg)

    def info(self):
        out = np.zeros(10, np.int32)
        lib.pdlpdev_analysis_info(self.handle, _ptr(out))
        return dict(permuted=bool(out[0]), method={0: "none", 1: "chains", 2: "groups"}.get(int(out[1]), "?"),
                    estimate_natural=(out[2] / 1e4, out[3] / 1e4), estimate_chains=(out[4] / 1e4, out[5] / 1e4),
                    estimate_groups=(out[6] / 1e4, out[7] / 1e4), search_levels=int(out[8]), cell_rounds=int(out[9]))

    def maps(self):
        """(row_new2old, col_new2old) or None when the device holds the matrix as given"""
        r, c = np.zeros(self.m, np.int32), np.zeros(self.n, np.int32)
        return (r, c) if lib.pdlpdev_analysis_maps(self.handle, _ptr(r), _ptr(c)) == 1 else None

    def download(self, transposed=False):
        rows = self.n if transposed else self.m
        off, idx, val = np.zeros(rows + 1, np.int32), np.zeros(self.nnz, np.int32), np.zeros(self.nnz)
        rc = lib.pdlpdev_analysis_download(self.handle, int(transposed), _ptr(off), _ptr(idx), _ptr(val))
        if rc != 0:
            raise CuOptError(rc, lib.pdlpdev_last_error().decode())
        return off, idx, val

    def close(self):
        if getattr(self, "handle", None):
            lib.pdlpdev_analysis_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synthetic_lp_on_device(m, n, k, seed=1, device=0):
    """S(m, n, k) generated on the device (pdlpdev_synthetic_lp): a problem dict like cuopt_amd.synthetic.generate's"""
    nnz = int(m) * int(k)
    off, idx, val = np.zeros(m + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    c, lo, hi, xs, ys = np.zeros(n), np.zeros(m), np.zeros(m), np.zeros(n), np.zeros(m)
    rc = lib.pdlpdev_synthetic_lp(device, int(m), int(n), int(k), int(seed), _ptr(off), _ptr(idx), _ptr(val), _ptr(c), _ptr(lo), _ptr(hi), _ptr(xs), _ptr(ys))
    if rc != 0:
        raise CuOptError(rc, lib.pdlpdev_last_error().decode())
    return dict(m=int(m), n=int(n), offsets=off, indices=idx, values=val, c=c, lo=lo, hi=hi, lb=np.zeros(n), ub=np.full(n, np.inf), maximize=False,
                objective_offset=0.0, x_star=xs, y_star=ys, objective_star=float(c @ xs), seed=seed, k=k, hard=False, band=0)


def device_sort_pairs(keys, vals=None, bits=32, device=0):
    """the set-up's stable LSD radix sort (test hook)"""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    v = None if vals is None else np.ascontiguousarray(vals, dtype=np.uint32)
    ko, vo = np.zeros_like(keys), np.zeros_like(keys)
    rc = lib.pdlpdev_debug_sort_pairs(device, len(keys), _ptr(keys), _ptr(v), int(bits), _ptr(ko), _ptr(vo))
    if rc != 0:
        raise CuOptError(rc, lib.pdlpdev_last_error().decode())
    return ko, vo


def device_exclusive_scan(values, device=0):
    a = np.ascontiguousarray(values, dtype=np.int32)
    out = np.zeros(len(a) + 1, np.int32)
    rc = lib.pdlpdev_debug_scan(device, len(a), _ptr(a), _ptr(out))
    if rc != 0:
        raise CuOptError(rc, lib.pdlpdev_last_error().decode())
    return out


class Device:
    """pdlpdev_ctx: direct access to the HIP kernels (kernel-level parity tests, timing)."""

    def __init__(self, p=None, handle=None, owner=True, device=0, analysis=None):
        self.owner = owner
        if handle is not None:
            self.handle = handle
            return
        if analysis is not None:
            # the device-side set-up: p's vectors must be in the order of the matrices the analysis holds (the caller permutes them)
            c, lo, hi, lb, ub = (_f64(p[k]) for k in ("c", "lo", "hi", "lb", "ub"))
            self.handle = c_void_p()
            rc = lib.pdlpdev_create_from_analysis(C.byref(self.handle), analysis.handle, _ptr(c), _ptr(lo), _ptr(hi), _ptr(lb), _ptr(ub))
            msg = lib.pdlpdev_last_error().decode() if rc != 0 else ""
            analysis.close()
            if rc != 0:
                if self.handle:
                    lib.pdlpdev_destroy(self.handle)
                self.handle = None
                raise CuOptError(rc, msg)
            self.m, self.n, self.nnz = int(p["m"]), int(p["n"]), len(p["values"])
            return
        m, n = int(p["m"]), int(p["n"])
        off, idx, val = _i32(p["offsets"]), _i32(p["indices"]), _f64(p["values"])
        to, ti, tv = csr_transpose(m, n, off, idx, val)
        c, lo, hi, lb, ub = (_f64(p[k]) for k in ("c", "lo", "hi", "lb", "ub"))
        self.handle = c_void_p()
        rc = lib.pdlpdev_create(C.byref(self.handle), device, m, n, _ptr(off), _ptr(idx), _ptr(val),
                                _ptr(to), _ptr(ti), _ptr(tv), _ptr(c), _ptr(lo), _ptr(hi), _ptr(lb), _ptr(ub))
        if rc != 0:
            msg = lib.pdlpdev_last_error().decode()
            if self.handle:
                lib.pdlpdev_destroy(self.handle)
            self.handle = None
            raise CuOptError(rc, msg)
        self.m, self.n, self.nnz = m, n, len(val)

    def _ck(self, rc):
        if rc != 0:
            raise CuOptError(rc, lib.pdlpdev_last_error().decode())

    def call(self, name, *args):
        self._ck(getattr(lib, "pdlpdev_" + name)(self.handle, *args))

    def ctl(self):
        c = Ctl()
        self._ck(lib.pdlpdev_get_ctl(self.handle, C.byref(c)))
        return c

    def run(self, target_steps):
        c = Ctl()
        self._ck(lib.pdlpdev_run(self.handle, int(target_steps), C.byref(c)))
        return c

    def download(self, name, count):
        out = np.zeros(int(count))
        got = lib.pdlpdev_download(self.handle, BUF[name], _ptr(out), int(count))
        if got < 0:
            raise CuOptError(int(got), lib.pdlpdev_last_error().decode())
        return out[:got]

    def spmv(self, x, transpose=False, rows=None):
        x = _f64(x)
        y = np.zeros(int(rows))
        self._ck(lib.pdlpdev_spmv(self.handle, int(transpose), _ptr(x), _ptr(y)))
        return y

    def eval(self, which, rule_finite=True, eps_p=1e-4, eps_d=1e-4):
        out = np.zeros(len(EV))
        self._ck(lib.pdlpdev_eval(self.handle, which, int(rule_finite), eps_p, eps_d, _ptr(out)))
        return {k: out[i] for k, i in EV.items()}

    def eval_infeasibility(self, which, rule_finite=True):
        out = np.zeros(4)
        self._ck(lib.pdlpdev_eval_infeasibility(self.handle, which, int(rule_finite), _ptr(out)))
        return dict(zip(["max_primal_ray_infeasibility", "primal_ray_linear_objective",
                         "max_dual_ray_infeasibility", "dual_ray_linear_objective"], out.tolist()))

    def trust_region_bounds(self, which, wp, wd, pds=0.5, dds=0.5, primal_weight=1.0, radius=-1.0, scaled_iterates=False):
        out = np.zeros(6)
        self._ck(lib.pdlpdev_trust_region_bounds(self.handle, which, wp, wd, pds, dds, primal_weight, radius,
                                                 int(scaled_iterates), _ptr(out)))
        return dict(zip(["primal_distance2", "dual_distance2", "distance", "lagrangian", "lower_bound",
                         "upper_bound"], out.tolist()))

    def init_norms(self):
        out = np.zeros(3)
        self._ck(lib.pdlpdev_init_norms(self.handle, _ptr(out)))
        return out

    def layout(self):
        out = np.zeros(8, np.int32)
        self._ck(lib.pdlpdev_layout_info(self.handle, _ptr(out)))
        names = {0: "stream", 1: "panel", 2: "resident", 3: "jag", 4: "pb"}

        def side(k):
            d = dict(layout=names[int(out[k])], panels=bool(out[k] == 1), workgroups=int(out[k + 1]))
            if out[k] == 3:
                d["lds_gather_saving_pct"] = int(out[k + 2])  # 100 * (1 - cost of filling the LDS column sets / gathers served)
            elif out[k] == 4:
                d["padding_pct"] = int(out[k + 2])  # gather-free layout: padded entries over nonzeros - 1, in percent (workgroups = bins)
            else:
                d["slabs"] = int(out[k + 2])
                if out[k] == 1:
                    d["row_sums"] = "by_nonzero" if out[6 + k // 3] else "by_row"  # by_nonzero: the long-tail variant (every row at rtol)
            return d
        return dict(A=side(0), At=side(3), resident=bool(out[0] == 2))

    def wire_bytes(self):
        """sharded solve, owner-computes dataflow: bytes this rank receives per attempt for the two vector exchanges"""
        out = np.zeros(3, np.int64)
        self._ck(lib.pdlpdev_shard_wire_bytes(self.handle, _ptr(out)))
        return dict(halo=bool(out[0]), bytes=int(out[1]), bytes_allgather=int(out[2]))

    def layout_checksums(self):
        """FNV-1a of A^T and of every panel / jagged layout array on the device (parity of the device-side set-up)"""
        out = np.zeros(16, np.uint64)
        self._ck(lib.pdlpdev_debug_layout_checksums(self.handle, _ptr(out)))
        return out

    def dense_info(self):
        out = np.zeros(3, np.int64)
        self._ck(lib.pdlpdev_dense_info(self.handle, _ptr(out)))
        return dict(on=bool(out[0]), segments=int(out[1]), entries=int(out[2]))

    def time_kernel(self, kernel, reps=20):
        ms = c_double()
        self._ck(lib.pdlpdev_time_kernel(self.handle, KERNEL[kernel], int(reps), C.byref(ms)))
        return ms.value

    def close(self):
        if self.owner and self.handle:
            lib.pdlpdev_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
