"""Python mirror of the reference's `cuopt.linear_programming` package for LPs (the Python/Cython bridge).

Same class and method names, argument meaning and error behaviour as
  python/cuopt/cuopt/linear_programming/data_model/data_model.py        -> DataModel
  python/cuopt/cuopt/linear_programming/solver_settings/solver_settings.py -> SolverSettings, SolverMethod, PDLPSolverMode
  python/cuopt/cuopt/linear_programming/solver/solver.py                -> Solve, BatchSolve
  python/cuopt/cuopt/linear_programming/solution/solution.py            -> Solution, PDLPWarmStartData
  python/cuopt/cuopt/linear_programming/cuopt_mps_parser/parser.py      -> ParseMps (here: Read)
so that the reference's own Python tests (python/cuopt/cuopt/tests/linear_programming/test_lp_solver.py) read the same
with `from cuopt_amd import linear_programming as lp`.  Where the reference goes through Cython into libcuopt, this
module goes through ctypes (cuopt_amd/capi.py) into cuopt_amd/lib/libcuopt.so; there is no CPU fallback.

PDLP requests (method = SolverMethod.PDLP) run on a `cuoptamd_solver`, which is what carries initial solutions and
`pdlp_warm_start_data` in and out; Concurrent / DualSimplex requests go through `cuOptSolve`, where the library's small-LP dual
simplex answers or races PDLP (`Solution.get_solved_by_pdlp()` says which engine answered)."""
import time
from enum import IntEnum

import numpy as np

from . import capi

# parameter names (cpp/include/cuopt/linear_programming/constants.h)
CUOPT_ABSOLUTE_DUAL_TOLERANCE = "absolute_dual_tolerance"
CUOPT_RELATIVE_DUAL_TOLERANCE = "relative_dual_tolerance"
CUOPT_ABSOLUTE_PRIMAL_TOLERANCE = "absolute_primal_tolerance"
CUOPT_RELATIVE_PRIMAL_TOLERANCE = "relative_primal_tolerance"
CUOPT_ABSOLUTE_GAP_TOLERANCE = "absolute_gap_tolerance"
CUOPT_RELATIVE_GAP_TOLERANCE = "relative_gap_tolerance"
CUOPT_INFEASIBILITY_DETECTION = "infeasibility_detection"
CUOPT_STRICT_INFEASIBILITY = "strict_infeasibility"
CUOPT_PRIMAL_INFEASIBLE_TOLERANCE = "primal_infeasible_tolerance"
CUOPT_DUAL_INFEASIBLE_TOLERANCE = "dual_infeasible_tolerance"
CUOPT_ITERATION_LIMIT = "iteration_limit"
CUOPT_TIME_LIMIT = "time_limit"
CUOPT_PDLP_SOLVER_MODE = "pdlp_solver_mode"
CUOPT_METHOD = "method"
CUOPT_PER_CONSTRAINT_RESIDUAL = "per_constraint_residual"
CUOPT_SAVE_BEST_PRIMAL_SO_FAR = "save_best_primal_so_far"
CUOPT_FIRST_PRIMAL_FEASIBLE = "first_primal_feasible"
CUOPT_LOG_FILE = "log_file"
CUOPT_LOG_TO_CONSOLE = "log_to_console"
CUOPT_CROSSOVER = "crossover"
CUOPT_SOLUTION_FILE = "solution_file"
CUOPT_USER_PROBLEM_FILE = "user_problem_file"


class SolverMethod(IntEnum):  # solver_settings.py:44-60
    Concurrent = 0
    PDLP = 1
    DualSimplex = 2

    def __str__(self):
        return "%d" % self.value


class PDLPSolverMode(IntEnum):  # solver_settings.py:63-96
    Stable1 = 0
    Stable2 = 1
    Methodical1 = 2
    Fast1 = 3

    def __str__(self):
        return "%d" % self.value


class LPTerminationStatus(IntEnum):  # constants.h:65-74 (pdlp_termination_status_t)
    NoTermination = 0
    Optimal = 1
    PrimalInfeasible = 2
    DualInfeasible = 3
    IterationLimit = 4
    TimeLimit = 5
    NumericalError = 6
    PrimalFeasible = 7
    FeasibleFound = 8
    ConcurrentLimit = 9


class ErrorStatus(IntEnum):  # constants.h:107-114
    Success = 0
    InvalidArgument = 1
    MpsFileError = 2
    MpsParseError = 3
    ValidationError = 4
    OutOfMemoryError = 5
    RuntimeError = 6


class ProblemCategory(IntEnum):
    LP = 0
    MIP = 1
    IP = 2


class DataModel:
    """Host-side problem description (data_model.py:153-600); arrays are kept as numpy arrays."""

    def __init__(self):
        self.maximize = False
        self.A_values = self.A_indices = self.A_offsets = None
        self.b = self.c = None
        self.objective_scaling_factor = 1.0
        self.objective_offset = 0.0
        self.variable_lower_bounds = self.variable_upper_bounds = None
        self.constraint_lower_bounds = self.constraint_upper_bounds = None
        self.variable_types = None
        self.row_types = None
        self.variable_names, self.row_names = [], []
        self.initial_primal_solution = self.initial_dual_solution = None

    # setters
    def set_maximize(self, maximize):
        self.maximize = bool(maximize)

    def set_csr_constraint_matrix(self, A_values, A_indices, A_offsets):
        self.A_values = np.ascontiguousarray(A_values, dtype=np.float64)
        self.A_indices = np.ascontiguousarray(A_indices, dtype=np.int32)
        self.A_offsets = np.ascontiguousarray(A_offsets, dtype=np.int32)

    def set_constraint_bounds(self, b):
        self.b = np.ascontiguousarray(b, dtype=np.float64)

    def set_objective_coefficients(self, c):
        self.c = np.ascontiguousarray(c, dtype=np.float64)

    def set_objective_scaling_factor(self, objective_scaling_factor):
        self.objective_scaling_factor = float(objective_scaling_factor)

    def set_objective_offset(self, objective_offset):
        self.objective_offset = float(objective_offset)

    def set_variable_lower_bounds(self, variable_lower_bounds):
        self.variable_lower_bounds = np.ascontiguousarray(variable_lower_bounds, dtype=np.float64)

    def set_variable_upper_bounds(self, variable_upper_bounds):
        self.variable_upper_bounds = np.ascontiguousarray(variable_upper_bounds, dtype=np.float64)

    def set_constraint_lower_bounds(self, constraint_lower_bounds):
        self.constraint_lower_bounds = np.ascontiguousarray(constraint_lower_bounds, dtype=np.float64)

    def set_constraint_upper_bounds(self, constraint_upper_bounds):
        self.constraint_upper_bounds = np.ascontiguousarray(constraint_upper_bounds, dtype=np.float64)

    def set_variable_types(self, variable_types):
        self.variable_types = np.asarray(variable_types)

    def set_row_types(self, row_types):
        self.row_types = np.asarray(row_types)

    def set_variable_names(self, variables_names):
        self.variable_names = list(variables_names)

    def set_row_names(self, row_names):
        self.row_names = list(row_names)

    def set_initial_primal_solution(self, initial_primal_solution):
        self.initial_primal_solution = np.ascontiguousarray(initial_primal_solution, dtype=np.float64)

    def set_initial_dual_solution(self, initial_dual_solution):
        self.initial_dual_solution = np.ascontiguousarray(initial_dual_solution, dtype=np.float64)

    # getters
    def get_sense(self):
        return self.maximize

    def get_constraint_matrix_values(self):
        return self.A_values

    def get_constraint_matrix_indices(self):
        return self.A_indices

    def get_constraint_matrix_offsets(self):
        return self.A_offsets

    def get_constraint_bounds(self):
        return self.b

    def get_objective_coefficients(self):
        return self.c

    def get_objective_scaling_factor(self):
        return self.objective_scaling_factor

    def get_objective_offset(self):
        return self.objective_offset

    def get_variable_lower_bounds(self):
        return self.variable_lower_bounds

    def get_variable_upper_bounds(self):
        return self.variable_upper_bounds

    def get_constraint_lower_bounds(self):
        return self.constraint_lower_bounds

    def get_constraint_upper_bounds(self):
        return self.constraint_upper_bounds

    def get_row_types(self):
        return self.row_types

    def get_ascii_row_types(self):
        return None if self.row_types is None else np.array([ord(str(t)[0]) for t in self.row_types], dtype=np.int8)

    def get_initial_primal_solution(self):
        return self.initial_primal_solution

    def get_initial_dual_solution(self):
        return self.initial_dual_solution

    def get_variable_types(self):
        return self.variable_types

    def get_variable_names(self):
        return self.variable_names

    def get_row_names(self):
        return self.row_names

    # ---- to the problem dict of cuopt_amd.capi --------------------------------------------------------------------
    def _problem_dict(self):
        if self.A_offsets is None or self.c is None:
            raise ValueError("DataModel: constraint matrix and objective coefficients are required")
        m, n = len(self.A_offsets) - 1, len(self.c)
        lb = np.zeros(n) if self.variable_lower_bounds is None else self.variable_lower_bounds  # default [0, +inf)
        ub = np.full(n, np.inf) if self.variable_upper_bounds is None else self.variable_upper_bounds
        if self.constraint_lower_bounds is not None and self.constraint_upper_bounds is not None:
            lo, hi = self.constraint_lower_bounds, self.constraint_upper_bounds
        else:
            if self.b is None or self.row_types is None:
                raise ValueError("DataModel: constraint bounds need (row_types, b) or (lower, upper)")
            t = np.array([str(x)[0] for x in self.row_types])
            lo = np.where((t == "E") | (t == "G"), self.b, -np.inf)  # problem_helpers.cuh:33-58
            hi = np.where((t == "E") | (t == "L"), self.b, np.inf)
        p = dict(m=m, n=n, offsets=self.A_offsets, indices=self.A_indices, values=self.A_values, c=self.c, lo=lo, hi=hi,
                 lb=lb, ub=ub, maximize=self.maximize, objective_offset=self.objective_offset)
        if self.variable_types is not None:
            p["var_types"] = np.frombuffer("".join(str(x)[0] for x in self.variable_types).encode(), dtype=np.uint8)
        return p


def Read(mps_file_path, fixed_mps_format=False):
    """cuopt_mps_parser.ParseMps: MPS file -> DataModel (own reader of libcuopt.so, checked against the reference's parser
    on every fixture of datasets/linear_programming)"""
    prob = capi.Problem.read(mps_file_path)
    try:
        d = prob.to_dict()
        var_names, row_names = prob.names(0), prob.names(1)
    finally:
        prob.close()
    dm = DataModel()
    dm.set_csr_constraint_matrix(d["values"], d["indices"], d["offsets"])
    dm.set_objective_coefficients(d["c"])
    dm.set_constraint_lower_bounds(d["lo"])
    dm.set_constraint_upper_bounds(d["hi"])
    dm.set_variable_lower_bounds(d["lb"])
    dm.set_variable_upper_bounds(d["ub"])
    dm.set_maximize(d.get("maximize", False))
    dm.set_objective_offset(d.get("objective_offset", 0.0))
    if d.get("var_types") is not None:
        dm.set_variable_types(np.array([chr(c) for c in np.asarray(d["var_types"], dtype=np.uint8)]))
    dm.set_variable_names(var_names)
    dm.set_row_names(row_names)
    return dm


class PDLPWarmStartData:  # solution.py:23-64
    VECTORS = ("current_primal_solution", "current_dual_solution", "initial_primal_average", "initial_dual_average",
               "current_ATY", "sum_primal_solutions", "sum_dual_solutions", "last_restart_duality_gap_primal_solution",
               "last_restart_duality_gap_dual_solution")
    SCALARS = ("initial_primal_weight", "initial_step_size", "total_pdlp_iterations", "total_pdhg_iterations",
               "last_candidate_kkt_score", "last_restart_kkt_score", "sum_solution_weight", "iterations_since_last_restart")

    def __init__(self, *args, **kw):
        names = self.VECTORS + self.SCALARS
        for k, v in zip(names, args):
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def _from_capi(cls, d):
        w = cls(**{k: d[k] for k in cls.VECTORS + cls.SCALARS})
        w._extra = {k: v for k, v in d.items() if k not in cls.VECTORS + cls.SCALARS}  # scaled iterate: bit-exact resume
        return w

    def _to_capi(self):
        d = {k: getattr(self, k) for k in self.VECTORS + self.SCALARS}
        d.update(getattr(self, "_extra", {}))
        return d


class SolverSettings:  # solver_settings.py:99-330
    _DEFAULTS = {
        CUOPT_ABSOLUTE_DUAL_TOLERANCE: 1e-4, CUOPT_RELATIVE_DUAL_TOLERANCE: 1e-4, CUOPT_ABSOLUTE_PRIMAL_TOLERANCE: 1e-4,
        CUOPT_RELATIVE_PRIMAL_TOLERANCE: 1e-4, CUOPT_ABSOLUTE_GAP_TOLERANCE: 1e-4, CUOPT_RELATIVE_GAP_TOLERANCE: 1e-4,
        CUOPT_INFEASIBILITY_DETECTION: False, CUOPT_STRICT_INFEASIBILITY: False, CUOPT_PRIMAL_INFEASIBLE_TOLERANCE: 1e-8,
        CUOPT_DUAL_INFEASIBLE_TOLERANCE: 1e-8, CUOPT_ITERATION_LIMIT: 2 ** 31 - 1, CUOPT_TIME_LIMIT: float("inf"),
        CUOPT_PDLP_SOLVER_MODE: PDLPSolverMode.Stable2, CUOPT_METHOD: SolverMethod.Concurrent,
        CUOPT_PER_CONSTRAINT_RESIDUAL: False, CUOPT_SAVE_BEST_PRIMAL_SO_FAR: False, CUOPT_FIRST_PRIMAL_FEASIBLE: False,
        CUOPT_LOG_FILE: "", CUOPT_LOG_TO_CONSOLE: False, CUOPT_CROSSOVER: False, CUOPT_SOLUTION_FILE: "",
        CUOPT_USER_PROBLEM_FILE: "",
    }

    def __init__(self):
        self.settings_dict = dict(self._DEFAULTS)
        self.pdlp_warm_start_data = None
        self.mip_callbacks = []

    def to_base_type(self, value):
        if isinstance(value, IntEnum):
            return int(value)
        if isinstance(value, (np.floating, np.integer, np.bool_)):
            return value.item()
        return value

    def get_parameter(self, name):
        if name not in self.settings_dict:
            raise ValueError("Invalid parameter. Please check documentation")
        return self.settings_dict[name]

    def set_parameter(self, name, value):
        probe = capi.Settings()  # the C registry validates name, type and range (solver_settings.cu:66-330)
        try:
            probe.set(name, str(self.to_base_type(value)))  # string form, parsed by the registry like the reference's
        except capi.CuOptError:
            raise ValueError("Invalid parameter %r or value %r. Please check documentation" % (name, value))
        finally:
            probe.close()
        self.settings_dict[name] = value

    def set_optimality_tolerance(self, eps_optimal):  # solver_settings.py:176-209
        for k in (CUOPT_ABSOLUTE_DUAL_TOLERANCE, CUOPT_RELATIVE_DUAL_TOLERANCE, CUOPT_ABSOLUTE_PRIMAL_TOLERANCE,
                  CUOPT_RELATIVE_PRIMAL_TOLERANCE, CUOPT_ABSOLUTE_GAP_TOLERANCE, CUOPT_RELATIVE_GAP_TOLERANCE):
            self.settings_dict[k] = float(eps_optimal)

    def set_pdlp_warm_start_data(self, pdlp_warm_start_data):
        self.pdlp_warm_start_data = pdlp_warm_start_data

    def get_pdlp_warm_start_data(self):
        return self.pdlp_warm_start_data

    def set_mip_callback(self, callback):
        raise NotImplementedError("MILP is outside the scope of the MI355X-native PDLP library")

    def get_mip_callbacks(self):
        return self.mip_callbacks

    def toDict(self):
        return {k: self.to_base_type(v) for k, v in self.settings_dict.items()}


class Solution:  # solution.py:67-410 (LP part)
    def __init__(self, problem_category, vars, solve_time, primal_solution, dual_solution, reduced_cost, termination_status,
                 error_status, error_message, primal_residual, dual_residual, primal_objective, dual_objective, gap,
                 nb_iterations, pdlp_warm_start_data=None, solved_by_pdlp=True):
        self.problem_category = problem_category
        self.vars = vars
        self.solve_time = solve_time
        self.primal_solution, self.dual_solution, self.reduced_cost = primal_solution, dual_solution, reduced_cost
        self.termination_status = LPTerminationStatus(termination_status)
        self.error_status, self.error_message = error_status, error_message
        self.primal_objective, self.dual_objective = primal_objective, dual_objective
        self.lp_stats = {"primal_residual": primal_residual, "dual_residual": dual_residual, "gap": gap,
                         "nb_iterations": nb_iterations}
        self.pdlp_warm_start_data = pdlp_warm_start_data
        self.solved_by_pdlp = solved_by_pdlp

    def raise_if_milp_solution(self, function_name):
        if self.problem_category in (ProblemCategory.MIP, ProblemCategory.IP):
            raise AttributeError("Attribute %s is not supported for milp solution" % function_name)

    def get_primal_solution(self):
        return self.primal_solution

    def get_dual_solution(self):
        self.raise_if_milp_solution("get_dual_solution")
        return self.dual_solution

    def get_primal_objective(self):
        return self.primal_objective

    def get_dual_objective(self):
        self.raise_if_milp_solution("get_dual_objective")
        return self.dual_objective

    def get_termination_status(self):
        return self.termination_status

    def get_termination_reason(self):
        return self.termination_status.name

    def get_error_status(self):
        return self.error_status

    def get_error_message(self):
        return self.error_message

    def get_solve_time(self):
        return self.solve_time

    def get_solved_by_pdlp(self):
        return self.solved_by_pdlp

    def get_vars(self):
        return self.vars

    def get_lp_stats(self):
        self.raise_if_milp_solution("get_lp_stats")
        return self.lp_stats

    def get_reduced_cost(self):
        return self.reduced_cost

    def get_pdlp_warm_start_data(self):
        self.raise_if_milp_solution("get_pdlp_warm_start_data")
        return self.pdlp_warm_start_data

    def get_problem_category(self):
        return self.problem_category


def _is_mip(var_types):  # solver.py:84-98
    if var_types is None or len(var_types) == 0:
        return False
    return any(str(t)[0] == "I" for t in var_types)


def _named(dm, x):
    names = dm.get_variable_names()
    return {names[j]: x[j] for j in range(len(names))} if len(names) == len(x) else {}


def Solve(data_model, solver_settings=None, log_file=""):
    """solver.py:22-99.  LPs only: a model with integer variables raises (the reference would call its MILP solver)."""
    if solver_settings is None:
        solver_settings = SolverSettings()
    if log_file:
        solver_settings.set_parameter(CUOPT_LOG_FILE, log_file)
    if _is_mip(data_model.get_variable_types()):
        raise ValueError("MILP is outside the scope of the MI355X-native PDLP library: only continuous LPs can be solved")
    try:
        p = data_model._problem_dict()
    except ValueError as e:  # an incomplete model is a ValidationError SOLUTION, not an exception (solver.py / test_lp_solver.py:323-383)
        return Solution(ProblemCategory.LP, {}, 0.0, np.zeros(0), np.zeros(0), np.zeros(0), 0, ErrorStatus.ValidationError,
                        str(e), 0.0, 0.0, 0.0, 0.0, 0.0, 0)
    st = solver_settings.toDict()
    ws = solver_settings.get_pdlp_warm_start_data()
    init_x, init_y = data_model.get_initial_primal_solution(), data_model.get_initial_dual_solution()
    if st[CUOPT_METHOD] != int(SolverMethod.PDLP):
        if ws is not None or init_x is not None or init_y is not None:
            raise ValueError("initial solutions and pdlp_warm_start_data need method = SolverMethod.PDLP")
        params = {k: str(v) for k, v in st.items() if v != "" and not (k == CUOPT_TIME_LIMIT and v == float("inf"))}
        r = capi.solve(p, **params)
        if r["return_code"] != capi.CUOPT_SUCCESS:
            return Solution(ProblemCategory.LP, {}, 0.0, np.zeros(p["n"]), np.zeros(p["m"]), np.zeros(p["n"]), 0,
                            ErrorStatus(r["error_status"]), r["error_string"], 0.0, 0.0, 0.0, 0.0, 0.0, 0)
        return Solution(ProblemCategory.LP, _named(data_model, r["x"]), r["solve_time"], r["x"], r["y"], r["reduced_cost"],
                        r["status_code"], ErrorStatus.Success, "", r["l2_primal_residual"], r["l2_dual_residual"],
                        r["objective"], r["dual_objective"], r["gap"], r["steps_taken"],
                        solved_by_pdlp=(r.get("solve_info") or {}).get("engine", "pdlp") == "pdlp")
    # ---- PDLP proper: a cuoptamd_solver carries initial iterates and warm-start snapshots
    names = {"absolute_gap_tolerance", "relative_gap_tolerance", "absolute_primal_tolerance", "relative_primal_tolerance",
             "absolute_dual_tolerance", "relative_dual_tolerance", "iteration_limit", "time_limit", "per_constraint_residual",
             "first_primal_feasible", "strict_infeasibility", "primal_infeasible_tolerance", "dual_infeasible_tolerance",
             "save_best_primal_so_far", "log_to_console"}
    floats = {"time_limit", "primal_infeasible_tolerance", "dual_infeasible_tolerance"}
    over = {k: (float(v) if (k in floats or k.endswith("_tolerance")) else int(v)) for k, v in st.items() if k in names}
    over["detect_infeasibility"] = int(bool(st[CUOPT_INFEASIBILITY_DETECTION]))
    if st[CUOPT_LOG_FILE]:
        over["log_file"] = st[CUOPT_LOG_FILE].encode()
    t0 = time.perf_counter()
    s = capi.Solver(p, mode=int(st[CUOPT_PDLP_SOLVER_MODE]), init_x=init_x, init_y=init_y,
                    warm_start=None if ws is None else ws._to_capi(), **over)
    try:
        r = s.advance()
        x, y, z = s.solution()
        snap = PDLPWarmStartData._from_capi(s.get_warm_start()) if r["status"] not in (0, 6) else None
    finally:
        s.close()
    return Solution(ProblemCategory.LP, _named(data_model, x), time.perf_counter() - t0, x, y, z, r["status"],
                    ErrorStatus.Success, "", r["l2_primal_residual"], r["l2_dual_residual"], r["primal_objective"],
                    r["dual_objective"], r["gap"], r["steps_taken"], pdlp_warm_start_data=snap)


def BatchSolve(data_model_list, solver_settings=None, log_file=""):
    """solver.py:101-190: independent LPs solved concurrently on one GPU (cuoptamd_batch_solve) -> (solutions, seconds)"""
    if solver_settings is None:
        solver_settings = SolverSettings()
    if solver_settings.get_pdlp_warm_start_data() is not None:
        raise ValueError("BatchSolve: pdlp_warm_start_data cannot be used in batch mode")  # test_lp_solver.py:567-589
    st = solver_settings.toDict()
    names = {"absolute_gap_tolerance", "relative_gap_tolerance", "absolute_primal_tolerance", "relative_primal_tolerance",
             "absolute_dual_tolerance", "relative_dual_tolerance", "iteration_limit", "time_limit"}
    over = {k: (int(v) if k == "iteration_limit" else float(v)) for k, v in st.items() if k in names}
    t0 = time.perf_counter()
    out = capi.batch_solve([dm._problem_dict() for dm in data_model_list], mode=int(st[CUOPT_PDLP_SOLVER_MODE]), **over)
    sols = [Solution(ProblemCategory.LP, _named(dm, r["x"]), r["setup_seconds"] + r["loop_seconds"], r["x"], r["y"],
                     r["reduced_cost"], r["status"], ErrorStatus.Success, "", r["l2_primal_residual"], r["l2_dual_residual"],
                     r["primal_objective"], r["dual_objective"], r["gap"], r["steps_taken"])
            for dm, r in zip(data_model_list, out)]
    return sols, time.perf_counter() - t0
