"""The Python mirror of the reference's `cuopt.linear_programming` package (cuopt_amd/linear_programming.py), exercised
the way python/cuopt/cuopt/tests/linear_programming/test_lp_solver.py exercises the original: same calls, same
assertions.  Fixture files that are not shipped (savsched1, a2864) are replaced by synthetic LPs of the same role."""
import numpy as np
import pytest

from conftest import write_mps
from cuopt_amd import linear_programming as lp
from cuopt_amd import synthetic
from cuopt_amd.linear_programming import (CUOPT_INFEASIBILITY_DETECTION, CUOPT_ITERATION_LIMIT, CUOPT_METHOD,
                                          CUOPT_PDLP_SOLVER_MODE, CUOPT_TIME_LIMIT, LPTerminationStatus, PDLPSolverMode,
                                          SolverMethod)


def model_of(p):
    dm = lp.DataModel()
    dm.set_csr_constraint_matrix(p["values"], p["indices"], p["offsets"])
    dm.set_objective_coefficients(p["c"])
    dm.set_constraint_lower_bounds(p["lo"])
    dm.set_constraint_upper_bounds(p["hi"])
    dm.set_variable_lower_bounds(p["lb"])
    dm.set_variable_upper_bounds(p["ub"])
    dm.set_maximize(p.get("maximize", False))
    dm.set_objective_offset(p.get("objective_offset", 0.0))
    return dm


def test_set_get_fields():  # test_lp_solver.py:165-222 (no GPU involved)
    dm = lp.DataModel()
    A = np.array([1.0, 2.0, 3.0])
    indices = np.array([0, 1, 2], dtype=np.int32)
    b, c = np.array([4.0, 5.0, 6.0]), np.array([7.0, 8.0, 9.0])
    var_lb, var_ub = np.array([0.0, 0.1, 0.2]), np.array([1.0, 1.1, 1.2])
    con_lb, con_ub = np.array([0.5, 0.6, 0.7]), np.array([1.5, 1.6, 1.7])
    row_types = np.array(["L", "G", "E"])
    dm.set_csr_constraint_matrix(A, indices, indices)
    dm.set_constraint_bounds(b)
    dm.set_objective_coefficients(c)
    dm.set_variable_lower_bounds(var_lb)
    dm.set_variable_upper_bounds(var_ub)
    dm.set_constraint_lower_bounds(con_lb)
    dm.set_constraint_upper_bounds(con_ub)
    dm.set_row_types(row_types)
    dm.set_maximize(True)
    dm.set_objective_scaling_factor(1.5)
    dm.set_objective_offset(0.5)
    assert dm.get_sense() is True
    for got, want in ((dm.get_constraint_matrix_values(), A), (dm.get_constraint_matrix_indices(), indices),
                      (dm.get_constraint_matrix_offsets(), indices), (dm.get_constraint_bounds(), b),
                      (dm.get_objective_coefficients(), c), (dm.get_variable_lower_bounds(), var_lb),
                      (dm.get_variable_upper_bounds(), var_ub), (dm.get_constraint_lower_bounds(), con_lb),
                      (dm.get_constraint_upper_bounds(), con_ub)):
        np.testing.assert_array_equal(got, want)
    assert list(dm.get_ascii_row_types()) == [ord("L"), ord("G"), ord("E")]
    assert dm.get_objective_scaling_factor() == 1.5 and dm.get_objective_offset() == 0.5
    s = lp.SolverSettings()
    s.set_optimality_tolerance(1e-5)
    assert s.get_parameter(lp.CUOPT_ABSOLUTE_GAP_TOLERANCE) == 1e-5 and s.get_parameter(lp.CUOPT_RELATIVE_DUAL_TOLERANCE) == 1e-5
    with pytest.raises(ValueError):
        s.get_parameter("no_such_parameter")


@pytest.mark.gpu
def test_solver():  # test_lp_solver.py:56-87
    dm = lp.DataModel()
    dm.set_csr_constraint_matrix(np.array([1.0, 1.0]), np.array([0, 0]), np.array([0, 1, 2]))
    dm.set_constraint_bounds(np.array([1.0, 1.0]))
    dm.set_objective_coefficients(np.array([1.0]))
    dm.set_row_types(np.array(["L", "L"]))
    settings = lp.SolverSettings()
    settings.set_optimality_tolerance(1e-2)
    settings.set_parameter(CUOPT_METHOD, SolverMethod.PDLP)
    solution = lp.Solve(dm, settings)
    assert solution.get_termination_reason() == "Optimal"
    assert solution.get_primal_solution()[0] == pytest.approx(0.0)
    assert solution.get_lp_stats()["primal_residual"] == pytest.approx(0.0)
    assert solution.get_lp_stats()["dual_residual"] == pytest.approx(0.0)
    assert solution.get_primal_objective() == pytest.approx(0.0)
    assert solution.get_dual_objective() == pytest.approx(0.0)
    assert solution.get_lp_stats()["gap"] == pytest.approx(0.0)
    assert solution.get_solved_by_pdlp()
    with pytest.raises(ValueError):
        settings.set_parameter("not_a_parameter", 1)
    with pytest.raises(ValueError):
        settings.set_parameter(CUOPT_PDLP_SOLVER_MODE, 17)


@pytest.mark.gpu
def test_parser_and_solver(golden_problems, tmp_path):  # test_lp_solver.py:90-99
    path = tmp_path / "good-mps-1.mps"
    write_mps(str(path), golden_problems["good-mps-1"]["problem"])
    dm = lp.Read(str(path))
    settings = lp.SolverSettings()
    settings.set_optimality_tolerance(1e-2)
    assert lp.Solve(dm, settings).get_termination_reason() == "Optimal"  # default method: Concurrent, served by PDLP


@pytest.mark.gpu
def test_very_low_tolerance(golden_problems):  # test_lp_solver.py:101-121
    dm = model_of(golden_problems["afiro"]["problem"])
    settings = lp.SolverSettings()
    settings.set_optimality_tolerance(1e-12)
    settings.set_parameter(CUOPT_PDLP_SOLVER_MODE, PDLPSolverMode.Methodical1)
    settings.set_parameter(CUOPT_INFEASIBILITY_DETECTION, False)
    settings.set_parameter(CUOPT_METHOD, SolverMethod.PDLP)
    solution = lp.Solve(dm, settings)
    assert solution.get_termination_status() == LPTerminationStatus.Optimal
    assert solution.get_primal_objective() == pytest.approx(-464.7531)
    assert solution.get_solve_time() <= 69 * 5


@pytest.mark.gpu
def test_iteration_and_time_limit_solver():  # test_lp_solver.py:124-162 (savsched1 -> a synthetic LP)
    dm = model_of(synthetic.generate(20000, 16000, 8, seed=5))
    settings = lp.SolverSettings()
    settings.set_parameter(CUOPT_METHOD, SolverMethod.PDLP)
    settings.set_optimality_tolerance(0)
    settings.set_parameter(CUOPT_ITERATION_LIMIT, 1)
    settings.set_parameter(CUOPT_TIME_LIMIT, 99999999)
    solution = lp.Solve(dm, settings)
    assert solution.get_termination_status() == LPTerminationStatus.IterationLimit
    assert solution.get_primal_objective() != 0.0 and np.any(solution.get_primal_solution())
    settings.set_parameter(CUOPT_TIME_LIMIT, 0.2)
    settings.set_parameter(CUOPT_ITERATION_LIMIT, 99999999)
    solution = lp.Solve(dm, settings)
    assert solution.get_termination_status() == LPTerminationStatus.TimeLimit
    assert solution.get_solve_time() <= 0.2 * 10
    assert solution.get_primal_objective() != 0.0 and np.any(solution.get_primal_solution())


@pytest.mark.gpu
def test_warm_start():  # test_lp_solver.py:513-542 (a2864 -> a synthetic LP)
    dm = model_of(synthetic.generate(4000, 3500, 8, seed=32, hard=True))
    settings = lp.SolverSettings()
    settings.set_parameter(CUOPT_METHOD, SolverMethod.PDLP)
    settings.set_parameter(CUOPT_PDLP_SOLVER_MODE, PDLPSolverMode.Stable2)
    settings.set_optimality_tolerance(1e-3)
    settings.set_parameter(CUOPT_INFEASIBILITY_DETECTION, False)
    first = lp.Solve(dm, settings).get_lp_stats()["nb_iterations"]
    settings.set_optimality_tolerance(1e-2)
    solution2 = lp.Solve(dm, settings)
    second = solution2.get_lp_stats()["nb_iterations"]
    settings.set_optimality_tolerance(1e-3)
    settings.set_pdlp_warm_start_data(solution2.get_pdlp_warm_start_data())
    third = lp.Solve(dm, settings).get_lp_stats()["nb_iterations"]
    assert third + second == first


@pytest.mark.gpu
def test_batch_solve_matches_individual_solves(golden_problems):  # test_lp_solver.py:470-510
    names = ["afiro", "good-mps-1", "lp_model_with_var_bounds"]
    models = [model_of(golden_problems[k]["problem"]) for k in names]
    settings = lp.SolverSettings()
    settings.set_parameter(CUOPT_METHOD, SolverMethod.PDLP)
    batch, _ = lp.BatchSolve(models, settings)
    for dm, b in zip(models, batch):
        one = lp.Solve(dm, settings)
        assert b.get_termination_status() == one.get_termination_status()
        assert b.get_primal_objective() == pytest.approx(one.get_primal_objective(), rel=1e-6, abs=1e-9)


def test_solver_settings():  # test_lp_solver.py:267-320 (no GPU involved)
    settings = lp.SolverSettings()
    names = (lp.CUOPT_ABSOLUTE_DUAL_TOLERANCE, lp.CUOPT_RELATIVE_DUAL_TOLERANCE, lp.CUOPT_ABSOLUTE_PRIMAL_TOLERANCE,
             lp.CUOPT_RELATIVE_PRIMAL_TOLERANCE, lp.CUOPT_ABSOLUTE_GAP_TOLERANCE, lp.CUOPT_RELATIVE_GAP_TOLERANCE,
             lp.CUOPT_PRIMAL_INFEASIBLE_TOLERANCE, lp.CUOPT_DUAL_INFEASIBLE_TOLERANCE)
    for k in names:
        settings.set_parameter(k, 1e-5)
    for k in names:
        assert settings.get_parameter(k) == 1e-5
    assert settings.get_parameter(CUOPT_TIME_LIMIT) == float("inf")
    settings.set_parameter(CUOPT_ITERATION_LIMIT, 10)
    settings.set_parameter(CUOPT_TIME_LIMIT, 10.2)
    assert settings.get_parameter(CUOPT_ITERATION_LIMIT) == 10 and settings.get_parameter(CUOPT_TIME_LIMIT) == 10.2
    settings.set_parameter(CUOPT_INFEASIBILITY_DETECTION, False)
    assert not settings.get_parameter(CUOPT_INFEASIBILITY_DETECTION)
    assert settings.get_parameter(CUOPT_PDLP_SOLVER_MODE) == int(PDLPSolverMode.Stable2)
    with pytest.raises(ValueError):
        settings.set_parameter(CUOPT_PDLP_SOLVER_MODE, 10)
    settings.set_parameter(CUOPT_PDLP_SOLVER_MODE, PDLPSolverMode.Methodical1)
    assert settings.get_parameter(CUOPT_PDLP_SOLVER_MODE) == int(PDLPSolverMode.Methodical1)


def test_check_data_model_validity_without_a_gpu():  # test_lp_solver.py:323-360: incomplete models -> ValidationError solutions
    dm = lp.DataModel()
    assert lp.Solve(dm).get_error_status() == lp.ErrorStatus.ValidationError
    dm.set_csr_constraint_matrix(np.array([1.0]), np.array([0], dtype=np.int32), np.array([0, 1], dtype=np.int32))
    assert lp.Solve(dm).get_error_status() == lp.ErrorStatus.ValidationError
    dm.set_constraint_bounds(np.array([1.0]))
    assert lp.Solve(dm).get_error_status() == lp.ErrorStatus.ValidationError
    dm.set_objective_coefficients(np.array([1.0]))
    dm.set_maximize(True)
    assert lp.Solve(dm).get_error_status() == lp.ErrorStatus.ValidationError  # row types still missing


@pytest.mark.gpu
def test_check_data_model_validity_complete_models_solve():  # test_lp_solver.py:361-383
    dm = lp.DataModel()
    dm.set_csr_constraint_matrix(np.array([1.0]), np.array([0], dtype=np.int32), np.array([0, 1], dtype=np.int32))
    dm.set_constraint_bounds(np.array([1.0]))
    dm.set_objective_coefficients(np.array([1.0]))
    dm.set_row_types(np.array(["E"]))
    assert lp.Solve(dm).get_error_status() == lp.ErrorStatus.Success
    dm.set_constraint_lower_bounds(np.array([1.0]))
    dm.set_constraint_upper_bounds(np.array([1.0]))
    assert lp.Solve(dm).get_error_status() == lp.ErrorStatus.Success


@pytest.mark.gpu
def test_parse_var_names_and_the_reference_pdlp_solution_vector(golden_problems, tmp_path):
    """test_lp_solver.py:386-475: names survive the parser, and the solution of a PDLP request at default settings equals,
    variable by variable, the vector the reference's test holds for cuOpt's PDLP (rel 1e-4 there; 1e-6 here)"""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "afiro_pdlp_vars.json")))
    path = tmp_path / "afiro.mps"
    p = dict(golden_problems["afiro"]["problem"])
    p["var_names"], p["row_names"] = golden_problems["afiro"]["meta"]["var_names"], golden_problems["afiro"]["meta"]["row_names"]
    write_mps(str(path), p, name="AFIRO")
    dm = lp.Read(str(path))
    assert dm.get_variable_names() == g["expected_names"]
    settings = lp.SolverSettings()
    settings.set_parameter(CUOPT_METHOD, SolverMethod.PDLP)
    solution = lp.Solve(dm, settings)
    assert solution.get_termination_reason() == "Optimal"
    assert len(solution.get_vars()) == len(g["expected_values"])
    for key, want in g["expected_values"].items():
        assert solution.get_vars()[key] == pytest.approx(want, rel=1e-6, abs=1e-9)


@pytest.mark.gpu
def test_batch_solver_refuses_warm_start_data(golden_problems):  # test_lp_solver.py:567-589
    dm = model_of(golden_problems["afiro"]["problem"])
    settings = lp.SolverSettings()
    settings.set_parameter(CUOPT_METHOD, SolverMethod.PDLP)
    settings.set_optimality_tolerance(1e-3)
    settings.set_pdlp_warm_start_data(lp.Solve(dm, settings).get_pdlp_warm_start_data())
    with pytest.raises(Exception):
        lp.BatchSolve([dm, dm], settings)


@pytest.mark.gpu
def test_dual_simplex_request(golden_problems):
    """test_lp_solver.py:592-606 asks for SolverMethod.DualSimplex and gets -464.7531 from the simplex -- as here, from the
    library's own small-LP dual simplex (round 3; rounds 1-2 served the request with PDLP and said so)"""
    dm = model_of(golden_problems["afiro"]["problem"])
    settings = lp.SolverSettings()
    settings.set_parameter(CUOPT_METHOD, SolverMethod.DualSimplex)
    solution = lp.Solve(dm, settings)
    assert solution.get_termination_status() == LPTerminationStatus.Optimal
    assert solution.get_primal_objective() == pytest.approx(-464.7531)
    assert not solution.get_solved_by_pdlp()


@pytest.mark.gpu
def test_milp_models_are_refused():  # test_lp_solver.py:630-672 (test_bound_in_maximization) is a MILP: out of scope
    dm = lp.DataModel()
    dm.set_objective_coefficients(np.array([15.0, 100.0]))
    dm.set_maximize(True)
    dm.set_csr_constraint_matrix(np.array([2.0, 20.0]), np.array([0, 1], dtype=np.int32), np.array([0, 2], dtype=np.int32))
    dm.set_constraint_bounds(np.array([102.0]))
    dm.set_row_types(np.array(["L"]))
    dm.set_variable_types(np.array(["I", "I"]))
    with pytest.raises(ValueError):
        lp.Solve(dm)
