"""GPU: whole-solve parity of the MI355X PDLP (through the libcuopt C API and the host-driver C-ABI)
against (a) the known answers pinned in the reference's own tests, (b) the reference's CPU dual
simplex objectives, (c) the C oracle on the same inputs.

Stated tolerance (DESIGN.md "Parity"): same termination status; |obj - obj_ref| <= 2*eps*(1+|obj_ref|)
for a solve at relative tolerance eps (both solvers stop inside the same eps-optimality band); the
returned residuals/gap satisfy the reference's termination inequalities when RE-COMPUTED ON THE HOST
from the returned x, y (the reference's own test_objective_sanity / test_constraint_sanity pattern,
cpp/tests/linear_programming/utilities/pdlp_test_utilities.cuh:45-140, tolerance 1e-6 there)."""
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import write_mps
from cuopt_amd import capi, synthetic
from oracle import orcbind

pytestmark = pytest.mark.gpu
INF = np.inf


def host_check(p, r, eps=1e-4):
    """re-verify a returned solution on the host with numpy only"""
    A = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(p["m"], p["n"]))
    x, y = r["x"], r["y"]
    sgn = -1.0 if p.get("maximize") else 1.0
    cx = float(p["c"] @ x) + p.get("objective_offset", 0.0)
    assert cx == pytest.approx(r["primal_objective"], rel=1e-6, abs=1e-6)
    ax = A @ x
    viol = np.maximum(np.maximum(p["lo"] - ax, ax - p["hi"]), 0.0)
    assert np.linalg.norm(viol) == pytest.approx(r["l2_primal_residual"], rel=1e-6, abs=1e-6)
    bcomb = np.maximum(np.where(np.isfinite(p["lo"]), np.abs(p["lo"]), 0), np.where(np.isfinite(p["hi"]), np.abs(p["hi"]), 0))
    assert np.linalg.norm(viol) <= eps + eps * np.linalg.norm(bcomb) + 1e-9
    assert np.all(x >= p["lb"] - 1e-9) and np.all(x <= p["ub"] + 1e-9)
    # dual residual from y: g = c - A^T y ; rc per Stable2 rule
    g = sgn * p["c"] - A.T @ y
    bv = np.where(g > 0, p["lb"], p["ub"])
    rc = np.where((g == 0) | np.isfinite(bv), g, 0.0)
    assert np.linalg.norm(g - rc) <= eps + eps * np.linalg.norm(p["c"]) + 1e-9
    assert r["gap"] <= eps + eps * (abs(r["primal_objective"]) + abs(r["dual_objective"])) + 1e-12


def test_afiro_through_read_problem_and_solve(golden_problems, tmp_path):
    """c_api_tests.cpp:31-39 + pdlp_test.cu:58-84 + test_lp_solver.py:101-121"""
    g = golden_problems["afiro"]
    path = str(tmp_path / "afiro.mps")
    write_mps(path, g["problem"], name="AFIRO")
    prob = capi.Problem.read(path)
    r = capi.solve(prob, method=1)
    assert r["return_code"] == 0 and r["status"] == "Optimal"
    assert abs(r["objective"] - (-464.0)) <= 0.01 * 464.0
    host_check(g["problem"], r)
    o = g["meta"]["oracle"]["0.0001"]
    assert abs(r["steps_taken"] - o["steps_taken"]) <= 80  # iterate-level parity is not claimed
    assert r["objective"] == pytest.approx(o["primal_objective"], abs=2e-4 * 465)
    r = capi.solve(prob, method=1, tol=1e-8)
    assert r["status"] == "Optimal"
    assert r["objective"] == pytest.approx(-464.7531, rel=1e-6)
    assert r["objective"] == pytest.approx(g["meta"]["reference_dual_simplex"]["objective"], rel=2e-8)
    host_check(g["problem"], r, eps=1e-8)


def test_initial_step_and_weight_goldens(golden_problems):
    p = golden_problems["afiro"]["problem"]
    pin = golden_problems["afiro"]["meta"]["pinned_initial"]
    r = capi.Solver(p, mode=1, iteration_limit=0).advance()
    assert r["status_name"] == "IterationLimit" and r["steps_taken"] == 0
    assert r["initial_step_size"] == pytest.approx(pin["oracle_stable2_step_size"], rel=1e-13)
    assert r["initial_primal_weight"] == pytest.approx(pin["oracle_stable2_primal_weight"], rel=1e-12)
    # Methodical1's scaling preset, here with the KKT restart
    h = capi.hyper_preset(2)
    h.restart_strategy = 1
    r = capi.Solver(p, hyper=h, iteration_limit=0).advance()
    assert r["initial_step_size"] == pytest.approx(1.4893, abs=1e-4)
    assert r["initial_primal_weight"] == pytest.approx(0.0141652, abs=1e-4)


def test_iteration_and_time_limits(golden_problems):
    p = golden_problems["afiro"]["problem"]
    r = capi.solve(p, method=1, iteration_limit=1)  # c_api_tests.cpp:73-80
    assert r["status_code"] == 4
    r = capi.solve(p, method=1, iteration_limit=10, tol=0.0)  # pdlp_test.cu:134-157
    assert r["status"] == "IterationLimit" and r["steps_taken"] == 10 and np.abs(r["x"]).sum() > 0
    big = synthetic.generate(20000, 20000, 10, seed=9, hard=True)
    t0 = time.time()
    r = capi.solve(big, method=1, time_limit=0.2, tol=0.0)  # pdlp_test.cu:159-187
    assert r["status"] == "TimeLimit"
    assert r["loop_seconds"] < 0.2 + 0.15


@pytest.mark.parametrize("name", ["good-max", "max_offset", "good-mps-1", "lp_model_with_var_bounds",
                                  "mip-sample-relaxation", "mip-bb_optimality-relaxation",
                                  "mip-fixed-problem-relaxation", "mip-trivial-presolve-optimality-relaxation"])
def test_small_lps(golden_problems, name):
    g = golden_problems[name]
    p = dict(g["problem"])
    p.pop("var_types", None)  # LP relaxation: integrality dropped (BASELINE config 5 inputs)
    r = capi.solve(p, method=1)
    assert r["status"] == "Optimal"
    ref = g["meta"]["reference_dual_simplex"]["objective"]
    assert r["objective"] == pytest.approx(ref, abs=2e-3 * (1 + abs(ref)))
    if "pinned_objective" in g["meta"]:  # pdlp_test.cu:909-943
        assert r["objective"] == pytest.approx(g["meta"]["pinned_objective"], abs=1e-3)
    o = g["meta"]["oracle"]["0.0001"]
    assert r["steps_taken"] == o["steps_taken"]  # tiny LPs: same major-iteration exit as the oracle
    assert r["objective"] == pytest.approx(o["primal_objective"], abs=1e-6 * (1 + abs(ref)))


def test_ranged_problem_from_c_api_test():
    """c_api_test.c:761-873 -> status optimal, objective 32 +- 1e-3"""
    p = dict(m=3, n=2, offsets=[0, 2, 4, 6], indices=[0, 1, 0, 1, 0, 1], values=[2.0, 3.0, 3.0, 1.0, 1.0, 2.0],
             c=[5.0, 8.0], lo=[-INF, -INF, 2.0], hi=[12.0, 6.0, 8.0], lb=[0.0, 0.0], ub=[10.0, 10.0], maximize=True)
    r = capi.solve(p, method=1)
    assert r["status"] == "Optimal" and r["objective"] == pytest.approx(32.0, abs=1e-3 * 33)


def test_toy_lp_and_empty_matrix():
    toy = dict(m=2, n=1, offsets=[0, 1, 2], indices=[0, 0], values=[1.0, 1.0], c=[0.0], lo=[-INF, -INF],
               hi=[1.0, 1.0], lb=[0.0], ub=[INF])
    r = capi.solve(toy, method=1)  # test_lp_solver.py:56-87
    assert r["status"] == "Optimal" and r["steps_taken"] == 2 and r["objective"] == 0.0 and np.all(r["x"] == 0)
    empty = dict(m=0, n=2, offsets=[0], indices=[], values=[], c=[1.0, 1.0], lo=[], hi=[], lb=[0.0, 0.0], ub=[1.0, 1.0])
    assert capi.solve(empty, method=1)["status"] == "NumericalError"  # pdlp_test.cu:875-889


def test_per_constraint_residual_identity_lp():
    """pdlp_test.cu:633-715"""
    p = dict(m=3, n=3, offsets=[0, 1, 2, 3], indices=[0, 1, 2], values=[1.0, 1.0, 1.0], c=[0.0, 0.0, 0.0],
             lo=[0.0, 0.0, 0.0], hi=[0.0, 0.0, 0.0], lb=[0.02, 0.03, 0.1], ub=[0.02, 0.03, 0.1])
    dev = capi.Device(p)
    dev.call("set_initial", capi._ptr(np.array([0.02, 0.03, 0.1])), None)
    ev = dev.eval(capi.CURRENT, eps_p=0.0)
    assert ev["LINF_PRES_REL"] == pytest.approx(0.1, abs=1e-15)


def test_mip_problem_is_rejected_with_validation_error():
    p = dict(m=1, n=2, offsets=[0, 2], indices=[0, 1], values=[1.0, 1.0], c=[1.0, 1.0], lo=[0.0], hi=[1.0],
             lb=[0.0, 0.0], ub=[1.0, 1.0], var_types=np.frombuffer(b"CI", np.uint8))
    r = capi.solve(p)
    assert r["return_code"] == capi.CUOPT_VALIDATION_ERROR and "MILP" in r["error_string"]


@pytest.mark.parametrize("hard", [False, True])
def test_synthetic_known_optimum_and_oracle_parity(hard):
    p = synthetic.generate(5000, 4000, 8, seed=11, hard=hard)
    for eps in (1e-4, 1e-6):
        r = capi.solve(p, method=1, tol=eps)
        o = orcbind.solve(p, tol=eps)
        assert r["status"] == o["status"] == "Optimal"
        scale = 1.0 + abs(p["objective_star"])
        assert abs(r["objective"] - p["objective_star"]) <= 20 * eps * scale * (10 if hard else 1)
        assert abs(r["objective"] - o["primal_objective"]) <= 20 * eps * scale * (10 if hard else 1)
        host_check(p, r, eps=eps)
        # same algorithm, different summation order: iteration counts stay in the same ballpark
        assert 0.5 * o["steps_taken"] - 80 <= r["steps_taken"] <= 2.0 * o["steps_taken"] + 80


def test_first_iterations_follow_the_oracle_exactly():
    """Before low-order-bit chaos can build up the two implementations take the same decisions:
    identical accepted/attempted counts and step sizes (rel 1e-9) over the first 40 iterations."""
    p = synthetic.generate(3000, 3000, 10, seed=4)
    for its in (1, 5, 12, 40):
        r = capi.Solver(p, tol=0.0, iteration_limit=its).advance()
        o = orcbind.solve(p, tol=0.0, iteration_limit=its)
        assert r["status_name"] == "IterationLimit" == o["status"]
        assert (r["steps_taken"], r["attempted_steps"]) == (int(o["steps_taken"]), int(o["attempted_steps"]))
        assert r["step_size"] == pytest.approx(o["final_step_size"], rel=1e-9)
        assert r["primal_weight"] == pytest.approx(o["final_primal_weight"], rel=1e-9)
        assert r["primal_objective"] == pytest.approx(o["primal_objective"], rel=1e-9, abs=1e-9)


def test_graph_replay_equals_plain_launches_bitwise():
    p = synthetic.generate(4000, 4000, 10, seed=6)
    out = []
    for g in (1, 0):
        s = capi.Solver(p, tol=1e-6, use_graph=g)
        r = s.advance()
        out.append((r["steps_taken"], r["attempted_steps"], r["primal_objective"], s.solution()[0]))
    assert out[0][:3] == out[1][:3]
    np.testing.assert_array_equal(out[0][3], out[1][3])


def test_advance_in_pieces_equals_one_shot():
    p = synthetic.generate(3000, 3000, 10, seed=8)
    one = capi.Solver(p, tol=1e-6)
    r1 = one.advance()
    pieces = capi.Solver(p, tol=1e-6)
    while True:
        r2 = pieces.advance(37)
        if r2["status"] != 0:
            break
    assert (r1["steps_taken"], r1["primal_objective"]) == (r2["steps_taken"], r2["primal_objective"])


def test_single_rank_collective_path_matches_plain_path():
    """row-block sharding code path with world = 1 (RCCL all-reduce of one rank) must reproduce the
    fused single-GPU path up to summation order"""
    p = synthetic.generate(4000, 4000, 10, seed=7)
    a = capi.Solver(p, tol=1e-6).advance()
    try:
        cid = capi.comm_unique_id()
    except capi.CuOptError as e:
        pytest.skip("RCCL unavailable: %s" % e)
    b = capi.Solver(p, tol=1e-6, comm_id=cid).advance()
    assert a["status_name"] == b["status_name"] == "Optimal"
    assert b["primal_objective"] == pytest.approx(a["primal_objective"], abs=1e-5 * (1 + abs(a["primal_objective"])))
    assert 0.5 * a["steps_taken"] - 80 <= b["steps_taken"] <= 2.0 * a["steps_taken"] + 80


def test_config2_scale_solve_reaches_known_optimum():
    """BASELINE config 2: 1e5 x 1e5, 1e6 nnz"""
    p = synthetic.generate(**synthetic.CONFIGS["c2"])
    r = capi.solve(p, method=1)
    assert r["status"] == "Optimal"
    assert abs(r["objective"] - p["objective_star"]) <= 1e-2 * (1 + abs(p["objective_star"]))
    host_check(p, r)


@pytest.mark.parametrize("seed,hard", [(31, False), (32, True), (33, True)])
def test_warm_start_iterations_are_additive(seed, hard):
    """pdlp_test.cu:803-854 / test_lp_solver.py:513-542: its(1e-2 from scratch) == its(1e-1) + its(1e-2
    warm-started from the 1e-1 solution's pdlp_warm_start_data)"""
    p = synthetic.generate(4000, 3500, 8, seed=seed, hard=hard)
    coarse, fine = (1e-1, 1e-2) if not hard else (1e-2, 1e-4)
    full = capi.Solver(p, tol=fine)
    r_full = full.advance()
    first = capi.Solver(p, tol=coarse)
    r1 = first.advance()
    ws = first.get_warm_start()
    assert ws["total_pdlp_iterations"] == r1["steps_taken"]
    second = capi.Solver(p, tol=fine, warm_start=ws)
    r2 = second.advance()
    assert r_full["status_name"] == r1["status_name"] == r2["status_name"] == "Optimal"
    assert r1["steps_taken"] + r2["steps_taken"] == r_full["steps_taken"]
    assert r2["primal_objective"] == r_full["primal_objective"]  # bit-exact restore (scaled iterate in the snapshot)
    # a snapshot holding only the reference's 9 vectors (unscaled iterate) still resumes, to rounding
    ws9 = {k: v for k, v in ws.items() if not k.endswith("_scaled")}
    r3 = capi.Solver(p, tol=fine, warm_start=ws9).advance()
    assert r3["status_name"] == "Optimal"
    assert abs(r1["steps_taken"] + r3["steps_taken"] - r_full["steps_taken"]) <= 400
    # x is unscaled in the snapshot and rescaled on restore ((x*d)/d != x in the last bit), like the reference
    assert r2["primal_objective"] == pytest.approx(r_full["primal_objective"], abs=2 * fine * (1 + abs(r_full["primal_objective"])))


def test_mip_style_warm_started_resolves(golden_problems):
    """BASELINE config 5 call pattern (cpp/src/mip/relaxed_lp/relaxed_lp.cu:53-127): a fresh solver per
    call, warm-started from the previous primal/dual, after single-variable bound tightenings; every
    re-solve must agree with the reference-derived oracle on the modified LP."""
    p = dict(golden_problems["mip-sample-relaxation"]["problem"])
    p.pop("var_types", None)
    r = capi.solve(p, method=1, tol=1e-6)
    x, y = r["x"], r["y"]
    rng = np.random.default_rng(0)
    total_warm, total_cold = 0, 0
    for k in range(12):
        q = dict(p)
        q["lb"], q["ub"] = p["lb"].copy(), p["ub"].copy()
        j = int(rng.integers(p["n"]))
        if rng.random() < 0.5:
            q["ub"][j] = np.floor(x[j])
        else:
            q["lb"][j] = np.ceil(x[j])
        if q["lb"][j] > q["ub"][j]:
            continue
        o = orcbind.solve(q, tol=1e-6, iteration_limit=20000)
        if o["status"] != "Optimal":
            continue  # infeasible child: detection is a later row
        warm = capi.Solver(q, tol=1e-6, init_x=x, init_y=y)
        rw = warm.advance()
        cold = capi.Solver(q, tol=1e-6).advance()
        assert rw["status_name"] == "Optimal"
        assert rw["primal_objective"] == pytest.approx(o["primal_objective"], abs=2e-5 * (1 + abs(o["primal_objective"])))
        total_warm += rw["steps_taken"]
        total_cold += cold["steps_taken"]
    assert total_warm > 0


def infeasible_lp_of_the_c_api_test():
    """cpp/tests/linear_programming/c_api_tests/c_api_test.c:625-757 (9 constraints, 4 variables)"""
    rhs = np.array([0.5, 3.0, 6.0, 2.0, 2.0, 5.0, 10.0, 14.0, 1.0])
    sense = "GGLLLGLLG"
    return dict(m=9, n=4, offsets=[0, 2, 4, 6, 7, 9, 10, 12, 15, 17],
                indices=[0, 1, 0, 1, 0, 1, 3, 2, 3, 2, 0, 3, 0, 1, 2, 1, 2],
                values=[-0.5, 1.0, 2.0, -1.0, 3.0, 1.0, 1.0, 3.0, -1.0, 1.0, 1.0, 1.0, 1.0, 2.0, 1.0, 1.0, 1.0],
                c=[0.0] * 4, lb=[0.0] * 4, ub=[INF] * 4,
                lo=np.array([rhs[i] if s in "GE" else -INF for i, s in enumerate(sense)]),
                hi=np.array([rhs[i] if s in "LE" else INF for i, s in enumerate(sense)]))


def test_infeasibility_information_matches_oracle():
    """infeasibility_information.cu:176-223 on arbitrary iterates (both reduced-cost rules)"""
    for p in (synthetic.generate(3000, 2500, 8, seed=41), infeasible_lp_of_the_c_api_test()):
        rng = np.random.default_rng(7)
        x = np.abs(rng.standard_normal(p["n"])) * (rng.random(p["n"]) < 0.8)
        y = rng.standard_normal(p["m"])
        for rule in (True, False):
            dev = capi.Device(p)
            dev.call("scaling_compute", 1, 10, 1, 1.0)
            dev.call("scale_problem")
            dev.call("set_initial", capi._ptr(x), capi._ptr(y))
            dev.eval(capi.CURRENT, rule_finite=rule)
            got = dev.eval_infeasibility(capi.CURRENT, rule_finite=rule)
            ref = orcbind.evaluate_infeasibility(p, x, y, finite_bounds_rule=rule)
            for k in ref:
                assert got[k] == pytest.approx(ref[k], rel=1e-10, abs=1e-12), k


@pytest.mark.parametrize("strict", [False, True])
def test_primal_infeasible_lp_is_detected(strict):
    """the reference's own infeasible LP (solved there by dual simplex -> INFEASIBLE); PDLP with
    infeasibility_detection must report PrimalInfeasible (status 2) like the oracle does"""
    p = infeasible_lp_of_the_c_api_test()
    o = orcbind.solve(p, infeasibility_detection=1, strict_infeasibility=int(strict), iteration_limit=20000)
    r = capi.solve(p, method=1, infeasibility_detection=True, strict_infeasibility=strict, iteration_limit=20000)
    assert o["status"] == "PrimalInfeasible"
    assert r["status"] == "PrimalInfeasible" and r["status_code"] == 2
    # diverging iterates amplify rounding differences: the detection iteration is not compared
    assert r["steps_taken"] <= 20000 and r["steps_taken"] % 40 == 0
    assert r["dual_ray_linear_objective"] > 0.0
    # without detection the same LP just runs into the limit (reference default: detection off)
    assert capi.solve(p, method=1, iteration_limit=400)["status"] == "IterationLimit"


@pytest.mark.parametrize("mode", [0, 3])
def test_other_presets_follow_the_oracle(mode):
    """Stable1 / Fast1 (LP/solve.cu:66-96,167-197: different scaling, step rule exponents, reduced-cost
    rule, major-iteration schedule, Fast1's per-iteration artificial restart check)"""
    p = synthetic.generate(3000, 2600, 9, seed=17)
    for its in (3, 30):
        r = capi.Solver(p, mode=mode, tol=0.0, iteration_limit=its).advance()
        o = orcbind.solve(p, mode=mode, tol=0.0, iteration_limit=its)
        assert (r["status_name"], r["steps_taken"], r["attempted_steps"]) == (o["status"], int(o["steps_taken"]), int(o["attempted_steps"]))
        assert r["initial_step_size"] == pytest.approx(o["initial_step_size"], rel=1e-12)
        assert r["initial_primal_weight"] == pytest.approx(o["initial_primal_weight"], rel=1e-10)
        assert r["step_size"] == pytest.approx(o["final_step_size"], rel=1e-6)
        assert r["primal_objective"] == pytest.approx(o["primal_objective"], rel=1e-6, abs=1e-6)
    r = capi.solve(p, method=1, pdlp_solver_mode=mode, tol=1e-6)
    assert r["status"] == "Optimal"
    assert abs(r["objective"] - p["objective_star"]) <= 4e-5 * (1 + abs(p["objective_star"]))


def test_trust_region_bounds_match_oracle(golden_problems):
    """bound_optimal_objective / solve_bound_constrained_trust_region (pdlp_restart_strategy.cu:1032-1050,
    1391-1678): Lagrangian value and the two objective bounds at a point, several radii (inside the first
    breakpoint, across many breakpoints, beyond all of them)"""
    for p in (synthetic.generate(2500, 2000, 8, seed=51), golden_problems["afiro"]["problem"]):
        rng = np.random.default_rng(9)
        x = np.abs(rng.standard_normal(p["n"])) * (rng.random(p["n"]) < 0.7)
        y = rng.standard_normal(p["m"])
        y = np.where(np.isinf(p["lo"]), -np.abs(y), y)  # dual feasible signs: no infinite subgradients
        y = np.where(np.isinf(p["hi"]), np.abs(y), y) * (rng.random(p["m"]) < 0.8)
        dev = capi.Device(p)
        dev.call("scaling_compute", 1, 5, 1, 1.0)
        dev.call("scale_problem")
        dev.call("set_initial", capi._ptr(x), capi._ptr(y))
        dev.eval(capi.CURRENT)
        for wp, wd, radius in ((2.0, 0.7, 1e-3), (2.0, 0.7, 0.5), (0.3, 5.0, 10.0), (1.0, 1.0, 1e4)):
            got = dev.trust_region_bounds(capi.CURRENT, wp, wd, radius=radius)
            ref = orcbind.trust_region_bounds(p, x, y, wp, wd, radius)
            scale = 1.0 + abs(ref["lagrangian"])
            assert got["lagrangian"] == pytest.approx(ref["lagrangian"], rel=1e-10, abs=1e-10 * scale)
            assert got["lower_bound"] == pytest.approx(ref["lower_bound"], rel=1e-9, abs=1e-9 * scale), (wp, wd, radius)
            assert got["upper_bound"] == pytest.approx(ref["upper_bound"], rel=1e-9, abs=1e-9 * scale), (wp, wd, radius)
        # anchors are zero right after create: the distances are the weighted norms of the point itself
        got = dev.trust_region_bounds(capi.CURRENT, 1.0, 1.0, pds=0.5, dds=0.5, primal_weight=2.0)
        assert got["primal_distance2"] == pytest.approx(float(x @ x), rel=1e-12)
        assert got["dual_distance2"] == pytest.approx(float(y @ y), rel=1e-12)
        assert got["distance"] == pytest.approx(np.sqrt(x @ x * 0.5 * 2.0 + y @ y * 0.5 / 2.0), rel=1e-12)


def test_methodical1_trust_region_restart(golden_problems):
    """pdlp_solver_mode = Methodical1: test_lp_solver.py:101-121 pins afiro at -464.7531 (rel 1e-6) with
    this preset at 1e-12 tolerances (here 1e-9); and the solve follows the oracle's restatement"""
    p = golden_problems["afiro"]["problem"]
    r = capi.solve(p, method=1, pdlp_solver_mode=2, tol=1e-9)
    assert r["status"] == "Optimal"
    assert r["objective"] == pytest.approx(-464.7531, rel=1e-6)
    o = orcbind.solve(p, mode=2, tol=1e-9)
    assert o["status"] == "Optimal" and o["primal_objective"] == pytest.approx(-464.7531, rel=1e-6)
    q = synthetic.generate(3000, 2600, 9, seed=17)
    for its in (64, 128):  # majors at 0, 64, 128 (min_iteration_restart = 0)
        rr = capi.Solver(q, mode=2, tol=0.0, iteration_limit=its).advance()
        oo = orcbind.solve(q, mode=2, tol=0.0, iteration_limit=its)
        assert (rr["steps_taken"], rr["attempted_steps"]) == (int(oo["steps_taken"]), int(oo["attempted_steps"]))
        assert rr["primal_weight"] == pytest.approx(oo["final_primal_weight"], rel=1e-6)
    rr = capi.solve(q, method=1, pdlp_solver_mode=2, tol=1e-6)
    oo = orcbind.solve(q, mode=2, tol=1e-6)
    assert rr["status"] == oo["status"] == "Optimal"
    assert abs(rr["objective"] - q["objective_star"]) <= 4e-5 * (1 + abs(q["objective_star"]))
    assert 0.5 * oo["steps_taken"] - 128 <= rr["steps_taken"] <= 2.0 * oo["steps_taken"] + 128


def test_trust_region_restart_on_scaled_iterates(golden_problems):
    """Methodical1 with rescale_for_restart switched on (no preset does it): the restart strategy, built on the unscaled
    problem (pdlp.cu:99-103), is handed the scaled iterates (pdlp.cu:1144-1149).  Same walk as the oracle's restatement."""
    h, oh = capi.hyper_preset(2), orcbind.hyper_preset(2)
    h.rescale_for_restart = 1
    oh[orcbind.H["ORC_H_RESCALE_FOR_RESTART"]] = 1.0
    q = synthetic.generate(3000, 2600, 9, seed=17)
    for its in (64, 128, 320):
        rr = capi.Solver(q, hyper=h, tol=0.0, iteration_limit=its).advance()
        oo = orcbind.solve(q, mode=2, hyper=oh, tol=0.0, iteration_limit=its)
        assert (rr["steps_taken"], rr["attempted_steps"]) == (int(oo["steps_taken"]), int(oo["attempted_steps"]))
        assert rr["num_restarts"] == int(oo["num_restarts"])
        assert rr["primal_weight"] == pytest.approx(oo["final_primal_weight"], rel=1e-6)
    for p, star in ((q, q["objective_star"]), (golden_problems["afiro"]["problem"], -464.7531)):
        rr = capi.Solver(p, hyper=h, tol=1e-6, iteration_limit=200000).advance()
        oo = orcbind.solve(p, mode=2, hyper=oh, tol=1e-6, iteration_limit=200000)
        assert rr["status_name"] == oo["status"] == "Optimal"
        assert abs(rr["primal_objective"] - star) <= 1e-4 * (1 + abs(star))
        assert 0.5 * oo["steps_taken"] - 128 <= rr["steps_taken"] <= 2.0 * oo["steps_taken"] + 128
    # the walk differs from the preset's (unscaled iterates): the switch is not ignored
    base = orcbind.solve(q, mode=2, tol=1e-6)
    assert int(base["steps_taken"]) != int(orcbind.solve(q, mode=2, hyper=oh, tol=1e-6)["steps_taken"])


@pytest.mark.parametrize("name", ["mip-50v-10-free-bound-relaxation", "mip-neos5-free-bound-relaxation",
                                  "mip-sudoku-relaxation", "mip-cod105_max-relaxation"])
def test_structured_lp_relaxations_match_reference_dual_simplex(golden_problems, name):
    """BASELINE config 5 inputs: LP relaxations of datasets/mip instances (233 x 2013, 63 x 63, 353 x 730,
    1024 x 1024 with 57 344 nonzeros), objective pinned by the reference's own CPU dual simplex compiled in place"""
    g = golden_problems[name]
    p = dict(g["problem"])
    p.pop("var_types", None)
    ref = g["meta"]["reference_dual_simplex"]["objective"]
    for eps in (1e-4, 1e-8):
        r = capi.solve(p, method=1, tol=eps)
        o = g["meta"]["oracle"]["%g" % eps]
        assert r["status"] == o["status"] == "Optimal"
        # (cod105: the reference simplex itself stops 3.2e-6 short of the optimum 128/7 that PDLP reaches at 1e-8)
        assert abs(r["objective"] - ref) <= max(4 * eps * (1 + abs(ref)), 5e-6 if "cod105" in name else 0.0)
        assert abs(r["objective"] - o["primal_objective"]) <= 4 * eps * (1 + abs(ref))
        host_check(p, r, eps=eps)
        assert 0.5 * o["steps_taken"] - 80 <= r["steps_taken"] <= 2.0 * o["steps_taken"] + 80


def test_repeated_warm_started_resolves_on_a_structured_relaxation(golden_problems):
    """config 5 call pattern on the 50v-10 relaxation: 20 re-solves after random single-variable bound
    tightenings, each warm-started from the previous primal/dual like relaxed_lp.cu:74-108; every result is
    checked against scipy/HiGHS on the modified LP, and warm starts must not cost more iterations overall"""
    from scipy.optimize import linprog
    p = dict(golden_problems["mip-50v-10-free-bound-relaxation"]["problem"])
    p.pop("var_types", None)
    A = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(p["m"], p["n"]))
    r = capi.solve(p, method=1, tol=1e-6)
    x, y = r["x"], r["y"]
    rng = np.random.default_rng(3)
    lb, ub = p["lb"].copy(), p["ub"].copy()
    warm_total = cold_total = solved = 0
    for k in range(20):
        frac = np.nonzero((np.abs(x - np.round(x)) > 1e-3) & np.isfinite(ub))[0]
        if frac.size == 0:
            break
        j = int(rng.choice(frac))
        nlb, nub = lb.copy(), ub.copy()
        if rng.random() < 0.5:
            nub[j] = np.floor(x[j])
        else:
            nlb[j] = np.ceil(x[j])
        q = dict(p, lb=nlb, ub=nub)
        A_ub = sp.vstack([A[np.isfinite(p["hi"])], -A[np.isfinite(p["lo"])]])
        b_ub = np.concatenate([p["hi"][np.isfinite(p["hi"])], -p["lo"][np.isfinite(p["lo"])]])
        ref = linprog(p["c"], A_ub=A_ub, b_ub=b_ub, bounds=list(zip(nlb, nub)), method="highs")
        if ref.status != 0:
            continue  # infeasible child
        warm = capi.Solver(q, tol=1e-6, init_x=x, init_y=y)
        rw = warm.advance()
        rc = capi.Solver(q, tol=1e-6).advance()
        assert rw["status_name"] == "Optimal"
        assert rw["primal_objective"] == pytest.approx(ref.fun, abs=2e-5 * (1 + abs(ref.fun)))
        x, y, _ = warm.solution()
        lb, ub = nlb, nub
        warm_total += rw["steps_taken"]
        cold_total += rc["steps_taken"]
        solved += 1
    assert solved >= 5
    assert warm_total <= cold_total


def test_save_best_primal_so_far():
    """pdlp_test.cu:717-772: with an iteration limit, save_best_primal_so_far returns the best primal point
    seen at a major iteration (feasible > infeasible, then objective, else least residual): its primal
    residual can only be lower or equal to that of the last iterate"""
    p = synthetic.generate(4000, 3500, 8, seed=32, hard=True)
    for limit in (120, 400):
        a = capi.solve(p, method=1, tol=0.0, iteration_limit=limit)
        b = capi.solve(p, method=1, tol=0.0, iteration_limit=limit, save_best_primal_so_far=True)
        assert a["status"] == b["status"] == "IterationLimit"
        assert b["l2_primal_residual"] <= a["l2_primal_residual"] * (1 + 1e-12)
        # the reported statistics belong to the returned point
        A = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(p["m"], p["n"]))
        ax = A @ b["x"]
        viol = np.maximum(np.maximum(p["lo"] - ax, ax - p["hi"]), 0.0)
        assert np.linalg.norm(viol) == pytest.approx(b["l2_primal_residual"], rel=1e-6, abs=1e-9)
        assert float(p["c"] @ b["x"]) == pytest.approx(b["primal_objective"], rel=1e-6, abs=1e-6)


def test_log_file_and_solution_file(golden_problems, tmp_path):
    g = golden_problems["afiro"]
    mps, log, sol = str(tmp_path / "afiro.mps"), str(tmp_path / "pdlp.log"), str(tmp_path / "afiro.sol")
    write_mps(mps, g["problem"], name="AFIRO")
    prob = capi.Problem.read(mps)
    r = capi.solve(prob, method=1, log_file=log, solution_file=sol)
    assert r["status"] == "Optimal"
    text = open(log).read()
    assert "Primal Obj." in text and "PDLP finished: status 1" in text
    lines = open(sol).read().splitlines()  # math_optimization/solution_writer.cu format
    assert lines[0] == "# Status: Optimal" and lines[1].startswith("# Objective value: ")
    assert float(lines[1].split(":")[1]) == pytest.approx(r["objective"], rel=1e-15)
    names = g["problem"]["var_names"]
    assert len(lines) == 2 + len(names)
    for j, ln in enumerate(lines[2:]):
        nm, val = ln.split()
        assert nm == names[j] and float(val) == r["x"][j]


def test_reference_c_api_test_translation_unit_runs_against_our_library(golden_problems, tmp_path):
    """The reference's own pure-C test code (cpp/tests/linear_programming/c_api_tests/c_api_test.c), compiled
    in place and unmodified against OUR cuopt_c.h and linked to OUR libcuopt.so (oracle/Makefile ->
    oracle/_ref/ref_capi_runner), driven like c_api_tests.cpp:27-97 drives it."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_capi_runner")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_capi_runner not built (needs /root/reference at build time)")
    mps = str(tmp_path / "afiro.mps")
    write_mps(mps, golden_problems["afiro"]["problem"], name="AFIRO")
    out = subprocess.run([exe, mps], capture_output=True, text=True, timeout=300)
    report = {ln.split()[0]: ln for ln in out.stdout.splitlines() if " expected " in ln}
    for name in ("int_size", "float_size", "afiro_rc", "afiro_status", "afiro_pdlp_status", "iteration_limit_status",
                 "bad_parameter_name", "missing_file", "infeasible_problem", "ranged_rc", "ranged_status",
                 "ranged_objective_32"):
        assert name in report and report[name].rstrip().endswith("OK"), out.stdout + out.stderr
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DIFFERS(out of scope)" in report["burglar_mip"]  # MILP is rejected, documented in INTEGRATION.md


def test_default_method_does_not_hang_where_pdlp_stalls_at_simplex_grade(golden_problems):
    """datasets/mip/minrep_inf.mps: PDLP (the reference's rule, restated in the oracle) never reaches 1e-8 on this
    6 x 4 LP.  A Concurrent / DualSimplex request first tries simplex-grade tolerances under a bounded budget and
    then answers at the user's own tolerances from the same solver object -- with the default iteration limit
    (INT_MAX) an unbudgeted attempt would never return."""
    g = golden_problems["mip-minrep_inf-relaxation"]
    p = dict(g["problem"])
    p.pop("var_types", None)
    ref = g["meta"]["reference_dual_simplex"]["objective"]
    t0 = time.perf_counter()
    r = capi.solve(p)  # method 0 (Concurrent), default tolerances and limits
    assert time.perf_counter() - t0 < 5.0
    assert r["status"] == "Optimal" and abs(r["objective"] - ref) <= 2e-3 * (1 + abs(ref))
    stalled = capi.solve(p, method=1, tol=1e-8, iteration_limit=20000)
    o = g["meta"]["oracle"]["1e-08"]
    assert stalled["status"] == o["status"] == "IterationLimit"
    ok = capi.solve(p, method=1, tol=1e-4)
    assert ok["status"] == "Optimal" and ok["steps_taken"] == g["meta"]["oracle"]["0.0001"]["steps_taken"]


def test_every_lp_fixture_of_the_reference_through_the_c_api(golden_parser):
    """all 21 non-empty files of datasets/linear_programming the reference parser accepts, solved with DEFAULT
    settings (method Concurrent) against the verdict of the reference's own dual simplex: same objective, status
    Infeasible (2) for the three infeasible files and Unbounded (3) for the two unbounded ones; with
    CUOPT_METHOD_PDLP + infeasibility detection the unbounded ones end like the reference's PDLP (NumericalError:
    its verdict kernel returns PrimalFeasible before it looks at the rays)"""
    from conftest import decode_problem
    count = 0
    for name, e in golden_parser.items():
        ds = e.get("reference_dual_simplex") if e["ok"] else None
        if not ds:
            continue
        p = decode_problem(e)
        p.pop("var_types", None)
        r = capi.solve(p, iteration_limit=400000)
        count += 1
        if ds["status"] == "OPTIMAL":
            assert r["status"] == "Optimal", name
            assert abs(r["objective"] - ds["objective"]) <= 2e-6 * (1 + abs(ds["objective"])), name
        elif ds["status"] == "INFEASIBLE":
            assert r["status_code"] == 2, (name, r["status"])
        else:
            assert r["status_code"] == 3, (name, r["status"])
            like_ref = capi.solve(p, method=1, infeasibility_detection=True, iteration_limit=400000)
            assert like_ref["status"] == "NumericalError", name
    assert count == 21


def test_write_files_like_the_python_test_of_the_reference(golden_problems, tmp_path):
    """test_lp_solver.py:675-700: solve afiro with CUOPT_USER_PROBLEM_FILE set (method DualSimplex), parse the written
    MPS, solve THAT with CUOPT_SOLUTION_FILE set: Optimal, -464.7531, and both files exist"""
    p = dict(golden_problems["afiro"]["problem"])
    out_mps, out_sol = str(tmp_path / "afiro_out.mps"), str(tmp_path / "afiro.sol")
    r = capi.solve(p, method=2, user_problem_file=out_mps)
    assert r["status"] == "Optimal" and os.path.isfile(out_mps)
    again = capi.Problem.read(out_mps)
    r2 = capi.solve(again, method=2, solution_file=out_sol)
    assert r2["status"] == "Optimal"
    assert r2["objective"] == pytest.approx(-464.7531, rel=1e-6)
    text = open(out_sol).read().splitlines()
    assert text[0] == "# Status: Optimal" and len(text) == 2 + p["n"]


def test_very_low_accuracy_and_initial_solution_like_pdlp_test(golden_problems):
    """pdlp_test.cu:86-110 (absolute tolerances at the minimal 1e-12, relative 0, PDLP method: Optimal, objective within
    1 % of -464.7531) and :112-132 (initial primal = all ones: Optimal, same objective)"""
    p = golden_problems["afiro"]["problem"]
    tiny = dict(absolute_dual_tolerance=1e-12, relative_dual_tolerance=0.0, absolute_primal_tolerance=1e-12,
                relative_primal_tolerance=0.0, absolute_gap_tolerance=1e-12, relative_gap_tolerance=0.0)
    s = capi.Solver(p, iteration_limit=2000000, **tiny)
    r = s.advance()
    assert r["status_name"] == "Optimal" and abs(r["primal_objective"] + 464.7531) <= 0.01 * 464.7531
    o = orcbind.solve(p, iteration_limit=2000000, abs_dual_tol=1e-12, rel_dual_tol=0.0, abs_primal_tol=1e-12,
                      rel_primal_tol=0.0, abs_gap_tol=1e-12, rel_gap_tol=0.0)
    assert o["status"] == "Optimal" and 0.5 * o["steps_taken"] - 200 <= r["steps_taken"] <= 2 * o["steps_taken"] + 200
    s2 = capi.Solver(p, tol=1e-4, init_x=np.ones(p["n"]))
    r2 = s2.advance()
    assert r2["status_name"] == "Optimal" and abs(r2["primal_objective"] + 464.7531) <= 0.01 * 464.7531
    o2 = orcbind.solve(p, tol=1e-4, init_x=np.ones(p["n"]))
    assert r2["steps_taken"] == int(o2["steps_taken"])


def test_initial_step_size_and_primal_weight_are_taken_verbatim(golden_problems):
    """pdlp_test.cu:525-554: with iteration_limit 0 in Methodical1, set_initial_step_size(1.0) /
    set_initial_primal_weight(2.0) are what the solver reports afterwards"""
    p = golden_problems["afiro"]["problem"]
    s = capi.Solver(p, mode=2, iteration_limit=0, initial_step_size=1.0, initial_primal_weight=2.0)
    r = s.advance()
    assert r["status_name"] == "IterationLimit" and r["step_size"] == 1.0 and r["primal_weight"] == 2.0


def test_warm_start_of_another_problem_is_refused(golden_problems):
    """test_lp_solver.py:545-565: the warm-start data of one problem handed to the solve of another one must raise"""
    a = synthetic.generate(900, 700, 6, seed=8)
    s = capi.Solver(a, tol=1e-1, iteration_limit=5000)
    s.advance()
    ws = s.get_warm_start()
    assert (ws["n_variables"], ws["n_constraints"]) == (700, 900)
    with pytest.raises(capi.CuOptError):
        capi.Solver(golden_problems["afiro"]["problem"], tol=1e-4, warm_start=ws)
    # the same problem takes it
    r = capi.Solver(a, tol=1e-4, warm_start=ws, iteration_limit=20000).advance()
    assert r["status_name"] == "Optimal"


@pytest.mark.timeout(120)
def test_non_finite_data_ends_in_numerical_error_not_in_an_endless_loop():
    """a NaN coefficient makes every step-size quantity NaN; the reference's comparisons (pdlp_constants.hpp:39-47, movement <= 0 or
    >= 1e100) let a NaN through to a step that is rejected for ever.  Here it takes the invalid-step-size exit: NumericalError"""
    for size in ((30, 20, 4), (4000, 3500, 8)):  # the on-chip loop and the multi-launch loop
        p = synthetic.generate(*size, seed=4)
        p["values"] = p["values"].copy()
        p["values"][3] = np.nan
        r = capi.solve(p, method=1, tol=1e-4, iteration_limit=5000)
        assert r["return_code"] == 0 and r["status"] == "NumericalError", r["status"]


def test_warm_start_carried_to_a_grown_and_a_shrunk_problem():
    """set_pdlp_warm_start_data with mappings (LP/solver_settings.cu:92-240) used for what it is for: the snapshot of an LP warm-starts
    (i) the same LP with extra rows and columns appended (zero padding) and (ii) the LP with its last rows dropped and two of the kept
    ones swapped; both re-solves reach their optimum (checked against the oracle) in fewer iterations than from scratch"""
    import scipy.sparse as sp
    p = synthetic.generate(1200, 1000, 6, seed=21, hard=True)
    first = capi.Solver(p, tol=1e-3)
    assert first.advance()["status_name"] == "Optimal"
    ws = first.get_warm_start()
    a = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(p["m"], p["n"]))

    def lp_of(mat, c, lo, hi):
        mat = sp.csr_matrix(mat)
        mat.sort_indices()
        return dict(m=mat.shape[0], n=mat.shape[1], offsets=mat.indptr.astype(np.int32), indices=mat.indices.astype(np.int32),
                    values=np.ascontiguousarray(mat.data, dtype=np.float64), c=np.asarray(c, float), lo=np.asarray(lo, float),
                    hi=np.asarray(hi, float), lb=np.zeros(mat.shape[1]), ub=np.full(mat.shape[1], np.inf), maximize=False,
                    objective_offset=0.0)

    # (i) grown: 3 new columns (costly, so they stay at 0) and 2 new rows that the old optimum satisfies with slack
    rng = np.random.default_rng(3)
    extra_cols = sp.random(p["m"], 3, density=0.01, random_state=5, data_rvs=rng.standard_normal)
    grown = sp.vstack([sp.hstack([a, extra_cols]), sp.hstack([sp.csr_matrix(np.ones((2, p["n"]))), sp.csr_matrix((2, 3))])])
    big = lp_of(grown, np.concatenate([p["c"], [50.0, 50.0, 50.0]]), np.concatenate([p["lo"], [-np.inf, -np.inf]]),
                np.concatenate([p["hi"], [2.0 * p["x_star"].sum() + 1.0] * 2]))
    ws_big = capi.remap_warm_start(ws, np.arange(big["n"]), np.arange(big["m"]))
    cold = capi.Solver(big, tol=1e-5).advance()
    warm = capi.Solver(big, tol=1e-5, warm_start=ws_big).advance()
    o = orcbind.solve(big, tol=1e-5)
    assert cold["status_name"] == warm["status_name"] == o["status"] == "Optimal"
    scale = 1 + abs(o["primal_objective"])
    assert abs(warm["primal_objective"] - o["primal_objective"]) <= 2e-4 * scale
    assert warm["steps_taken"] < cold["steps_taken"]
    # (ii) shrunk: a base LP whose last 100 rows are inactive at the optimum (y* = 0, slack > 0); they are dropped (the optimum stays)
    # and rows 0 and 1 of the kept ones are swapped
    inactive = np.nonzero((p["y_star"] == 0.0) & np.isinf(p["hi"]))[0][:100]
    order = np.concatenate([np.setdiff1d(np.arange(p["m"]), inactive), inactive])
    base = lp_of(a[order], p["c"], p["lo"][order], p["hi"][order])
    first = capi.Solver(base, tol=1e-3)
    assert first.advance()["status_name"] == "Optimal"
    ws = first.get_warm_start()
    perm = np.arange(p["m"] - 100)
    perm[[0, 1]] = [1, 0]
    small = lp_of(a[order][perm], p["c"], p["lo"][order][perm], p["hi"][order][perm])
    ws_small = capi.remap_warm_start(ws, None, perm)   # new[perm[i]] = old[i]: a swap is its own inverse
    assert ws_small["current_dual_solution"][0] == ws["current_dual_solution"][1]
    cold = capi.Solver(small, tol=1e-5).advance()
    warm = capi.Solver(small, tol=1e-5, warm_start=ws_small).advance()
    o = orcbind.solve(small, tol=1e-5)
    assert cold["status_name"] == warm["status_name"] == o["status"] == "Optimal"
    assert abs(warm["primal_objective"] - o["primal_objective"]) <= 2e-4 * (1 + abs(o["primal_objective"]))
    assert abs(warm["primal_objective"] - p["objective_star"]) <= 2e-4 * (1 + abs(p["objective_star"]))
    assert warm["steps_taken"] < cold["steps_taken"]


def test_initial_solution_test_of_the_reference_on_the_device(golden_problems):
    """pdlp_test.cu:245-523 with the two update_*_on_initial_solution hyper-parameters toggled: unchanged 1.4893 / 0.0141652
    unless both initial iterates are given and non-zero; then step size / primal weight move, to the oracle's values"""
    p = golden_problems["afiro"]["problem"]
    step0, w0, tol = 1.4893, 0.0141652, 1e-4

    def run(us, uw, x0, y0):
        h = capi.hyper_preset(2)
        h.update_step_size_on_initial_solution, h.update_primal_weight_on_initial_solution = us, uw
        s = capi.Solver(p, hyper=h, tol=0.0, iteration_limit=0, init_x=None if x0 is None else np.full(p["n"], float(x0)),
                        init_y=None if y0 is None else np.full(p["m"], float(y0)))
        r = s.advance()
        s.close()
        oh = orcbind.hyper_preset(2)
        oh[orcbind.H["ORC_H_UPDATE_STEP_SIZE_ON_INITIAL_SOLUTION"]] = float(us)
        oh[orcbind.H["ORC_H_UPDATE_PRIMAL_WEIGHT_ON_INITIAL_SOLUTION"]] = float(uw)
        o = orcbind.solve(p, mode=2, hyper=oh, tol=0.0, iteration_limit=0,
                          init_x=None if x0 is None else np.full(p["n"], float(x0)),
                          init_y=None if y0 is None else np.full(p["m"], float(y0)))
        assert r["initial_step_size"] == pytest.approx(o["initial_step_size"], rel=1e-9)
        assert r["initial_primal_weight"] == pytest.approx(o["initial_primal_weight"], rel=1e-9)
        return r["initial_step_size"], r["initial_primal_weight"]
    same = lambda v, ref: abs(v - ref) <= tol
    for us, uw in ((1, 0), (0, 1), (1, 1)):
        for x0, y0 in ((None, None), (1, None), (None, 1), (0, 0)):
            s, w = run(us, uw, x0, y0)
            assert same(s, step0) and same(w, w0), (us, uw, x0, y0)
    s, w = run(0, 1, 1, 1)
    assert same(s, step0) and not same(w, w0)
    s, w = run(1, 0, 1, 1)
    assert not same(s, step0) and same(w, w0)
    s, w = run(1, 1, 1, 1)
    assert not same(s, step0) and not same(w, w0)
    # and a solve from such a start still converges to the pinned objective
    h = capi.hyper_preset(1)
    h.update_step_size_on_initial_solution = h.update_primal_weight_on_initial_solution = 1
    r = capi.Solver(p, hyper=h, tol=1e-6, init_x=np.full(p["n"], 1.0), init_y=np.full(p["m"], 1.0)).advance()
    assert r["status_name"] == "Optimal" and r["primal_objective"] == pytest.approx(-464.7531, rel=1e-4)


def test_initial_step_size_before_scaling_on_the_device(golden_problems):
    """pdlp.cu:905-947 with compute_initial_step_size_before_scaling (no preset sets it together with
    update_step_size_on_initial_solution): the unscaled vectors meet the scaled matrix; step size and the run that follows
    are the oracle's, on afiro and on an LP of the non-resident path"""
    for p, tol in ((golden_problems["afiro"]["problem"], 1e-6), (synthetic.generate(3000, 4000, 6, seed=31), 1e-4)):
        rng = np.random.default_rng(5)
        x0, y0 = rng.uniform(0.5, 2.0, p["n"]), rng.uniform(-1.0, 1.0, p["m"])
        for uw in (0, 1):
            h, oh = capi.hyper_preset(2), orcbind.hyper_preset(2)
            h.update_step_size_on_initial_solution = h.compute_initial_step_size_before_scaling = 1
            h.update_primal_weight_on_initial_solution = uw
            oh[orcbind.H["ORC_H_UPDATE_STEP_SIZE_ON_INITIAL_SOLUTION"]] = oh[orcbind.H["ORC_H_STEP_SIZE_BEFORE_SCALING"]] = 1.0
            oh[orcbind.H["ORC_H_UPDATE_PRIMAL_WEIGHT_ON_INITIAL_SOLUTION"]] = float(uw)
            s = capi.Solver(p, hyper=h, tol=tol, iteration_limit=20000, init_x=x0, init_y=y0)
            r = s.advance()
            s.close()
            o = orcbind.solve(p, mode=2, hyper=oh, tol=tol, iteration_limit=20000, init_x=x0, init_y=y0)
            assert r["initial_step_size"] == pytest.approx(o["initial_step_size"], rel=1e-9)
            assert r["initial_primal_weight"] == pytest.approx(o["initial_primal_weight"], rel=1e-9)
            assert r["status_name"] == o["status"] == "Optimal"
            # (same start to 1e-9; the walks part ways at the first restart decision that rounding tips, as in every long solve)
            assert 0.5 * o["steps_taken"] - 128 <= r["steps_taken"] <= 2.0 * o["steps_taken"] + 128
            rr = capi.Solver(p, hyper=h, tol=0.0, iteration_limit=64, init_x=x0, init_y=y0).advance()
            oo = orcbind.solve(p, mode=2, hyper=oh, tol=0.0, iteration_limit=64, init_x=x0, init_y=y0)
            assert (rr["steps_taken"], rr["attempted_steps"]) == (int(oo["steps_taken"]), int(oo["attempted_steps"]))
            assert r["primal_objective"] == pytest.approx(o["primal_objective"], rel=10 * tol, abs=10 * tol)


def test_relative_tolerance_factors(golden_problems):
    """pdlp_test.cu:611-631 (initial_rhs_and_c): set_relative_{primal,dual}_tolerance_factor replace ||b|| and ||c|| in the
    termination rule (eps_abs + eps_rel * factor); what was set is what the solver reports and uses"""
    p = golden_problems["afiro"]["problem"]
    base = capi.Solver(p, tol=1e-6).advance()
    r = capi.Solver(p, tol=1e-6, relative_primal_tolerance_factor=1.0, relative_dual_tolerance_factor=2.0).advance()
    assert (r["norm_b"], r["norm_c"]) == (1.0, 2.0) and (base["norm_b"], base["norm_c"]) != (1.0, 2.0)
    assert r["status_name"] == "Optimal"
    # ||b||, ||c|| of afiro are > 2: smaller factors = a tighter rule = at least as many iterations, and it is met
    assert base["norm_b"] > 1.0 and base["norm_c"] > 2.0 and r["steps_taken"] >= base["steps_taken"]
    assert r["l2_primal_residual"] <= 1e-6 + 1e-6 * 1.0 and r["l2_dual_residual"] <= 1e-6 + 1e-6 * 2.0
    o = orcbind.solve(p, tol=1e-6, primal_tolerance_factor=1.0, dual_tolerance_factor=2.0)
    assert o["status"] == "Optimal" and int(o["steps_taken"]) == r["steps_taken"]
    # a reset without the factors goes back to the problem's own norms
    s = capi.Solver(p, tol=1e-6, relative_primal_tolerance_factor=1.0, relative_dual_tolerance_factor=2.0)
    s.advance()
    s.reset(tol=1e-6)
    again = s.advance()
    assert (again["norm_b"], again["norm_c"], again["steps_taken"]) == (base["norm_b"], base["norm_c"], base["steps_taken"])


def test_first_primal_feasible_like_pdlp_test():
    """pdlp_test.cu:774-802 (ns1687037 is not shipped: a badly scaled synthetic LP plays its part): with per-constraint
    residuals at 1e-2 and a budget that is too small for optimality, the plain solve ends in IterationLimit while
    first_primal_feasible = true returns PrimalFeasible -- and the returned point IS primal feasible by that rule"""
    p = synthetic.generate(3000, 2500, 8, seed=77, hard=True)
    kw = dict(method=1, tol=1e-2, per_constraint_residual=True)
    feas = capi.solve(p, first_primal_feasible=True, iteration_limit=100000, **kw)
    assert feas["status"] == "PrimalFeasible"
    plain = capi.solve(p, iteration_limit=feas["steps_taken"], **kw)
    assert plain["status"] == "IterationLimit"
    o = orcbind.solve(p, tol=1e-2, per_constraint_residual=1, first_primal_feasible=1, iteration_limit=100000)
    assert o["status"] == "PrimalFeasible" and int(o["steps_taken"]) == feas["steps_taken"]
    import scipy.sparse as sp
    A = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(p["m"], p["n"]))
    ax = A @ feas["x"]
    viol = np.maximum(np.maximum(p["lo"] - ax, ax - p["hi"]), 0.0)
    bcomb = np.maximum(np.where(np.isfinite(p["lo"]), np.abs(p["lo"]), 0), np.where(np.isfinite(p["hi"]), np.abs(p["hi"]), 0))
    assert np.all(viol <= 1e-2 + 1e-2 * bcomb + 1e-12)  # termination_strategy.cu:189-205
