"""GPU: how cuOptSolve serves requests the reference would hand to its dual simplex (Concurrent = default, DualSimplex,
crossover: LP/solve.cu:383-443,467-547) and the single-process multi-GPU path behind the same call.

Round 3: small LPs have a second engine, the library's own bounded dual simplex (dual_simplex.cpp): a DualSimplex request is
answered by it, a Concurrent request lets it race PDLP.  Where it is switched off (CUOPT_AMD_DUAL_SIMPLEX=0 / "amd_dual_simplex"
= 0) or abstains, non-PDLP methods on small LPs run PDLP at simplex-grade tolerances with the caller's own tolerances as the
acceptance set (the round-1/2 emulation; "amd_simplex_grade" = 0 switches that off too);
cuOptAmdGetSolveInfo says what happened."""
import numpy as np
import pytest
from conftest import set_tune

from cuopt_amd import capi, synthetic

pytestmark = pytest.mark.gpu
INF = float("inf")


def ranged_lp():  # c_api_test.c:761-873, optimum 32
    return dict(m=3, n=2, offsets=[0, 2, 4, 6], indices=[0, 1, 0, 1, 0, 1], values=[2.0, 3.0, 3.0, 1.0, 1.0, 2.0],
                c=[5.0, 8.0], lo=[-INF, -INF, 2.0], hi=[12.0, 6.0, 8.0], lb=[0.0, 0.0], ub=[10.0, 10.0], maximize=True)


def test_dual_simplex_requests_are_answered_by_the_dual_simplex():
    r = capi.solve(ranged_lp(), method=2, crossover=True)  # DualSimplex + crossover requested
    info = r["solve_info"]
    assert info["engine"] == "dual_simplex" and info["answered_by"] == "dual_simplex" and info["dual_simplex_status"] == 1
    assert info["requested_method"] == "DualSimplex" and info["crossover_requested"] is True and info["simplex_grade_emulation"] is False
    assert r["status"] == "Optimal" and r["objective"] == pytest.approx(32.0, abs=1e-9)  # a vertex: exact
    # the vertex satisfies the LP exactly and carries the duals of the converted minimisation (min -c: the convention of both engines,
    # as in the reference): reduced costs = -c - A^T y
    x, y, z = r["x"], r["y"], r["reduced_cost"]
    A = np.array([[2.0, 3.0], [3.0, 1.0], [1.0, 2.0]])
    assert np.all(A @ x <= np.array([12.0, 6.0, 8.0]) + 1e-9) and (A @ x)[2] >= 2.0 - 1e-9
    np.testing.assert_allclose(-np.array([5.0, 8.0]) - A.T @ y, z, atol=1e-9)
    # ... and PDLP answers the same LP with duals of the same sign (round-3 advisor: the engines used to disagree on a maximisation)
    # (the LP is dual degenerate: the two engines may sit on different dual optima, so the CONVENTION is compared, not the vectors)
    q = capi.solve(ranged_lp(), method=1, tol=1e-9)
    np.testing.assert_allclose(-np.array([5.0, 8.0]) - A.T @ q["y"], q["reduced_cost"], atol=1e-6)
    assert np.all(q["y"][:2] <= 1e-7) and np.all(y[:2] <= 1e-9)  # binding <= rows of a minimisation: multipliers <= 0 from both
    # Concurrent (the default): the simplex races PDLP and wins on an LP of this size
    c = capi.solve(ranged_lp())
    assert c["solve_info"]["engine"] == "dual_simplex" and c["objective"] == pytest.approx(32.0, abs=1e-9)
    # a PDLP request never consults it
    r = capi.solve(ranged_lp(), method=1)
    assert r["solve_info"]["engine"] == "pdlp" and r["solve_info"]["dual_simplex_consulted"] is False
    assert r["solve_info"]["simplex_grade_emulation"] is False and r["solve_info"]["answered_by"] == "requested_tolerances"


def test_crossover_request_behind_pdlp_returns_a_vertex(monkeypatch):
    """CUOPT_METHOD_PDLP + crossover: the dual simplex starts from the basis PDLP's point suggests and its vertex replaces that point
    when it confirms the objective; an LP beyond the simplex engine's size limits keeps PDLP's point and says so"""
    from cuopt_amd import synthetic
    for p in (synthetic.generate(300, 260, 6, seed=4), synthetic.generate(1500, 1200, 4, seed=4)):
        plain = capi.solve(p, method=1)
        r = capi.solve(p, method=1, crossover=True)
        assert r["status"] == plain["status"] == "Optimal" and r["solve_info"]["crossover"] == "dual_simplex_from_the_pdlp_point"
        assert abs(r["objective"] - p["objective_star"]) <= 1e-8 * (1 + abs(p["objective_star"]))  # exact, PDLP's was 1e-4
        assert abs(plain["objective"] - p["objective_star"]) > abs(r["objective"] - p["objective_star"])
        x = r["x"]
        assert np.sum((x > 1e-12)) <= p["m"]  # a basic solution: at most m variables off their bound (lb = 0, ub = inf here)
        # the start from PDLP's point is what makes it a crossover: fewer pivots than the same LP from the slack basis
        cold = capi.dual_simplex(p)
        warm = capi.dual_simplex(p, x0=plain["x"], y0=plain["y"])
        assert warm["status"] == "Optimal" and warm["iterations"] < cold["iterations"]
    monkeypatch.setenv("CUOPT_AMD_SIMPLEX_MAX_ROWS", "3000")
    big = synthetic.generate(6000, 5000, 8, seed=61)
    q = capi.solve(big, method=1, crossover=True)
    assert q["status"] == "Optimal" and q["solve_info"]["crossover"] == "not_done_lp_too_large_for_the_dual_simplex"


def test_lps_beyond_the_simplex_limits_are_left_to_pdlp_and_limits_are_limits(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SIMPLEX_MAX_ROWS", "3000")  # (default: 200 000 rows)
    p = synthetic.generate(6000, 5000, 8, seed=61)
    r = capi.solve(p, method=2)
    assert r["status"] == "Optimal" and r["solve_info"]["engine"] == "pdlp" and r["solve_info"]["dual_simplex_status"] == 8
    t = capi.solve(ranged_lp(), method=2, time_limit=0.0)
    assert t["status"] == "TimeLimit"


def test_solve_info_names_the_engine_and_the_attempt(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_DUAL_SIMPLEX", "0")  # the round-1/2 emulation: PDLP alone serves the request
    r = capi.solve(ranged_lp(), method=2, crossover=True)  # DualSimplex + crossover requested
    info = r["solve_info"]
    assert info["engine"] == "pdlp" and info["requested_method"] == "DualSimplex" and info["crossover_requested"] is True
    assert info["simplex_grade_emulation"] is True and info["answered_by"] == "simplex_grade_1e-8" and info["gpus"] == 1
    assert r["status"] == "Optimal" and r["objective"] == pytest.approx(32.0, abs=1e-5)
    r = capi.solve(ranged_lp(), method=1)
    assert r["solve_info"]["simplex_grade_emulation"] is False and r["solve_info"]["answered_by"] == "requested_tolerances"
    again = capi.solve(ranged_lp(), method=2, amd_dual_simplex=1)  # the parameter beats the environment
    assert again["solve_info"]["engine"] == "dual_simplex"


def test_simplex_grade_opt_out(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_DUAL_SIMPLEX", "0")
    on = capi.solve(ranged_lp())
    set_tune(monkeypatch, simplex_grade="0")
    off = capi.solve(ranged_lp())
    assert on["solve_info"]["simplex_grade_emulation"] is True and off["solve_info"]["simplex_grade_emulation"] is False
    assert off["status"] == on["status"] == "Optimal"
    assert off["steps_taken"] <= on["steps_taken"]
    assert off["steps_taken"] == capi.solve(ranged_lp(), method=1)["steps_taken"]  # exactly a PDLP request
    set_tune(monkeypatch, simplex_grade=None)
    again = capi.solve(ranged_lp(), amd_simplex_grade=0)  # the parameter beats the environment default
    assert again["solve_info"]["simplex_grade_emulation"] is False


def test_default_method_honours_the_callers_iteration_limit(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_DUAL_SIMPLEX", "0")
    """ADVICE r1: with iteration_limit <= the emulation's budget the tight attempt used to eat the whole limit and return
    IterationLimit although the requested 1e-4 had been met long before.  The acceptance set keeps that iterate."""
    p = synthetic.generate(400, 300, 6, seed=9, hard=True)
    plain = capi.solve(p, method=1)  # what PDLP needs at 1e-4
    assert plain["status"] == "Optimal"
    limit = plain["steps_taken"] + 200  # enough for 1e-4, far too little for 1e-8 on this badly scaled LP
    tight = capi.solve(p, method=1, tol=1e-8, iteration_limit=limit)
    assert tight["status"] == "IterationLimit"
    r = capi.solve(p, iteration_limit=limit)  # default method (Concurrent)
    assert r["status"] == "Optimal" and r["accepted_at_looser_tolerances"] == 1
    assert r["solve_info"]["answered_by"] == "requested_tolerances_kept_during_simplex_grade_attempt"
    assert r["steps_taken"] <= limit
    scale = 1 + abs(p["objective_star"])
    assert abs(r["objective"] - p["objective_star"]) <= 5e-4 * scale
    # the kept point is one that met the requested tolerances
    assert r["relative_gap"] <= 1e-4 + 1e-12 and r["l2_relative_primal_residual"] <= 1e-4 + 1e-12


def test_default_method_time_limit_falls_back_too(monkeypatch):
    p = synthetic.generate(400, 300, 6, seed=9, hard=True)
    r = capi.solve(p, time_limit=0.0)  # both engines out of time before they start
    assert r["status"] == "TimeLimit"
    monkeypatch.setenv("CUOPT_AMD_DUAL_SIMPLEX", "0")
    r = capi.solve(p, time_limit=0.0)
    assert r["status"] == "TimeLimit"  # nothing was accepted yet: the limit status is reported as is
    assert r["solve_info"]["answered_by"] in ("limit_reached_during_simplex_grade_attempt", "simplex_grade_1e-8")


def test_large_default_request_keeps_the_reference_pdlp_defaults():
    """> 1e5 nonzeros: no tightening, no forced infeasibility detection"""
    p = synthetic.generate(30000, 30000, 5, seed=3)
    a = capi.solve(p)
    b = capi.solve(p, method=1)
    assert a["solve_info"]["simplex_grade_emulation"] is False
    assert (a["status"], a["steps_taken"], a["objective"]) == (b["status"], b["steps_taken"], b["objective"])


def _check_gathered_dual(p, r):
    """where a reduced cost is reported it IS the dual residual c - A^T y of the unscaled problem (elsewhere the residual
    is dual infeasibility and small): a gathered y with a block of zeros would satisfy neither"""
    from oracle import orcbind
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    g = p["c"] - orcbind.spmv(to, ti, tv, r["y"])
    z = r["reduced_cost"]
    tol = 1e-9 * (1 + np.abs(p["c"]).max())
    np.testing.assert_allclose(g[z != 0.0], z[z != 0.0], rtol=0, atol=tol)
    assert np.abs(g[z == 0.0]).max() <= 1e-3 * (1 + np.abs(p["c"]).max())
    bounds = np.linspace(0, p["m"], 9).astype(int)
    assert all(np.any(r["y"][a:b] != 0.0) for a, b in zip(bounds[:-1], bounds[1:]))


@pytest.mark.parametrize("dataflow", ["allreduce", "rsag", "owner"])
@pytest.mark.parametrize("world", [2, 4])
def test_cuoptsolve_shards_over_gpus_through_the_in_process_communicator(world, dataflow, monkeypatch):
    """CUOPT_AMD_NUM_GPUS behind cuOptSolve; the in-process communicator stands in for RCCL on this one-GPU box; both sharded
    dataflows (CUOPT_AMD_SHARD_DATAFLOW)"""
    p = synthetic.generate(6000, 5000, 8, seed=61)
    single = capi.solve(p, method=1, tol=1e-6)
    set_tune(monkeypatch, soft_communicator="1")
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", dataflow)
    r = capi.solve(p, method=1, tol=1e-6, amd_num_gpus=world)
    assert r["status"] == "Optimal" and r["gpus"] == world and r["solve_info"]["gpus"] == world
    scale = 1 + abs(p["objective_star"])
    assert abs(r["objective"] - p["objective_star"]) <= 2e-5 * scale
    assert abs(r["objective"] - single["objective"]) <= 2e-5 * scale
    _check_gathered_dual(p, r)
    monkeypatch.setenv("CUOPT_AMD_NUM_GPUS", str(world))  # the environment spelling
    e = capi.solve(p, method=1, tol=1e-6)
    assert (e["gpus"], e["steps_taken"], e["objective"]) == (world, r["steps_taken"], r["objective"])


def test_more_gpus_than_visible_is_a_loud_error(monkeypatch):
    set_tune(monkeypatch, soft_communicator=None)
    n = capi.device_count()
    r = capi.solve(ranged_lp(), method=1, amd_num_gpus=min(n + 1, 16)) if n < 16 else None
    if r is not None:
        assert r["return_code"] == capi.CUOPT_RUNTIME_ERROR and "visible" in r["error_string"]


@pytest.mark.parametrize("dataflow", ["allreduce", "rsag", "owner", "owner+p2p"])
@pytest.mark.skipif(capi.device_count() < 2, reason="needs two GPUs: RCCL with two ranks in one process")
def test_rccl_two_ranks_in_one_process(dataflow, monkeypatch):
    """two devices, two host threads, real RCCL; "owner+p2p": the exchanges of the owner-computes dataflow as direct stores into
    the peer's landing block (hipDeviceEnablePeerAccess) instead of collectives"""
    set_tune(monkeypatch, soft_communicator=None)
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", dataflow.split("+")[0])
    monkeypatch.setenv("CUOPT_AMD_SHARD_TRANSPORT", "p2p" if dataflow.endswith("p2p") else "collective")
    p = synthetic.generate(6000, 5000, 8, seed=61)
    single = capi.solve(p, method=1, tol=1e-6)
    r = capi.solve(p, method=1, tol=1e-6, amd_num_gpus=2)
    assert r["status"] == "Optimal" and r["gpus"] == 2
    assert abs(r["objective"] - single["objective"]) <= 2e-5 * (1 + abs(single["objective"]))
    _check_gathered_dual(p, r)


def test_create_problem_from_device_arrays():
    """the reference's C API accepts host or device arrays (cuopt_c.cpp:119, raft::copy at :261-403): a client that builds
    its CSR in HIP memory hands the device pointers to cuOptCreateRangedProblem / cuOptCreateProblem"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    held = []

    def to_device(a):
        a = np.ascontiguousarray(a)
        d = C.c_void_p()
        assert hip.hipMalloc(C.byref(d), max(a.nbytes, 8)) == 0
        assert hip.hipMemcpy(d, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        held.append(d)
        return d

    p = synthetic.generate(3000, 2600, 9, seed=12)
    L = capi.lib
    m, n = p["m"], p["n"]
    dv = {k: to_device(np.asarray(p[k], dtype=(np.int32 if k in ("offsets", "indices") else np.float64)))
          for k in ("c", "offsets", "indices", "values", "lo", "hi", "lb", "ub")}
    types = to_device(np.frombuffer(b"C" * n, dtype=np.uint8))
    prob = C.c_void_p()
    fn = L.cuOptCreateRangedProblem
    saved = fn.argtypes
    fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double] + [C.c_void_p] * 9 + [C.POINTER(C.c_void_p)]
    try:
        rc = fn(m, n, capi.CUOPT_MINIMIZE, 0.0, dv["c"], dv["offsets"], dv["indices"], dv["values"], dv["lo"], dv["hi"],
                dv["lb"], dv["ub"], types, C.byref(prob))
    finally:
        fn.argtypes = saved
    assert rc == capi.CUOPT_SUCCESS
    # what the library holds equals the host originals ...
    got = np.zeros(len(p["values"]))
    off, idx = np.zeros(m + 1, np.int32), np.zeros(len(p["indices"]), np.int32)
    assert L.cuOptGetConstraintMatrix(prob, capi._ptr(off), capi._ptr(idx), capi._ptr(got)) == 0
    np.testing.assert_array_equal(got, p["values"])
    np.testing.assert_array_equal(off, p["offsets"])
    np.testing.assert_array_equal(idx, p["indices"])
    # ... and solves to the same result as the host-built problem
    wrapped = capi.Problem(prob)
    r_dev = capi.solve(wrapped, method=1, tol=1e-6)
    r_host = capi.solve(p, method=1, tol=1e-6)
    assert r_dev["status"] == "Optimal"
    assert (r_dev["steps_taken"], r_dev["objective"]) == (r_host["steps_taken"], r_host["objective"])
    L.cuOptDestroyProblem(C.byref(prob))
    for d in held:
        hip.hipFree(d)


def test_the_emulations_own_budget_does_not_end_the_race(monkeypatch):
    """when the simplex-grade attempt of a Concurrent solve ran out of ITS OWN iteration budget (not a limit of the
    caller's) the simplex was cancelled and an unraced second solve followed.  Now the race goes on: PDLP at the requested
    tolerances against the same simplex run; whoever answers, the verdict and the objective are right."""
    p = synthetic.generate(400, 300, 6, seed=9, hard=True)  # far more than 10 iterations to anything
    set_tune(monkeypatch, simplex_grade_budget="10")
    r = capi.solve(p)  # default method: Concurrent
    assert r["status"] == "Optimal"
    assert r["solve_info"]["answered_by"] in ("dual_simplex", "requested_tolerances_after_simplex_grade_budget"), r["solve_info"]
    assert abs(r["objective"] - p["objective_star"]) <= 5e-4 * (1 + abs(p["objective_star"]))
    if r["solve_info"]["answered_by"] != "dual_simplex":
        assert r["solve_info"]["simplex_grade_attempt_iterations"] == 10
    # with the simplex out of the race the same budget leads to the second solve as before
    monkeypatch.setenv("CUOPT_AMD_DUAL_SIMPLEX", "0")
    q = capi.solve(p)
    assert q["status"] == "Optimal" and q["solve_info"]["answered_by"] == "requested_tolerances_after_simplex_grade_budget"
