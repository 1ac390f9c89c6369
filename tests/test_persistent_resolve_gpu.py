"""GPU: cuoptamd_solver_reset -- the persistent re-solve behind BASELINE config 5 (LP relaxations re-solved after
bound changes, cpp/src/mip/relaxed_lp/relaxed_lp.cu:53-175).  The reference builds a new pdlp_solver_t per call;
the scaling depends on A only (initial_scaling.cu:125-307), so a solver that keeps A, A^T, D_r, D_c and takes new
bounds must be bit-identical to a fresh one on the modified LP.  That is what is pinned here, on every solver
path (resident small-LP loop, multi-launch CSR stream, slab-major panels), plus HiGHS on the modified LPs."""
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp

from cuopt_amd import capi, synthetic

pytestmark = pytest.mark.gpu
INF = np.inf


LIMIT = 6000  # every solve is bounded: a tightened LP may be infeasible, and both solvers must then stop alike


def tightened(p, rng, count):
    """bound tightenings that keep the known optimum x* feasible (so the LP stays feasible with the same optimal
    value) but change the projections, hence the whole trajectory"""
    x = p["x_star"]
    lb, ub = np.array(p["lb"], float), np.array(p["ub"], float)
    for j in rng.choice(p["n"], size=count, replace=False):
        if rng.random() < 0.5:
            ub[j] = x[j] + 0.3 * rng.random()
        else:
            lb[j] = max(lb[j], x[j] - 0.3 * rng.random())
    return lb, ub


def same_result(a, b, xa, xb):
    for k in ("status", "steps_taken", "attempted_steps", "num_restarts", "num_major_iterations"):
        assert a[k] == b[k], k
    for k in ("primal_objective", "dual_objective", "gap", "l2_primal_residual", "l2_dual_residual", "step_size",
              "primal_weight", "initial_step_size", "initial_primal_weight"):
        assert a[k] == b[k], k
    for u, v in zip(xa, xb):
        np.testing.assert_array_equal(u, v)


@pytest.mark.parametrize("case", ["resident", "stream", "panel", "stable1", "methodical1"])
def test_reset_equals_fresh_solver(case, monkeypatch):
    mode = dict(stable1=0, methodical1=2).get(case, 1)
    if case == "panel":
        monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    p = synthetic.generate(700, 1000, 8, seed=5) if case == "resident" else synthetic.generate(6000, 5000, 8, seed=5)
    rng = np.random.default_rng(11)
    s = capi.Solver(p, mode=mode, tol=1e-6, iteration_limit=LIMIT)
    assert s.device.layout()["resident"] == (case == "resident")
    r0 = s.advance()
    assert r0["status_name"] == "Optimal"
    x0, y0, _ = s.solution()
    statuses = []
    for trial in range(3):
        lb, ub = tightened(p, rng, 200 + 100 * trial)
        q = dict(p, lb=lb, ub=ub)
        warm = trial != 1  # also once from the zero start
        ix, iy = (x0, y0) if warm else (None, None)
        s.reset(lb=lb, ub=ub, init_x=ix, init_y=iy, tol=1e-6, iteration_limit=LIMIT)
        a = s.advance()
        fresh = capi.Solver(q, mode=mode, tol=1e-6, iteration_limit=LIMIT, init_x=ix, init_y=iy)
        b = fresh.advance()
        same_result(a, b, s.solution(), fresh.solution())
        statuses.append(a["status_name"])
        assert a["primal_objective"] == pytest.approx(p["objective_star"], abs=1e-4 * (1 + abs(p["objective_star"])))
        x0, y0, _ = s.solution()
    assert statuses == ["Optimal"] * 3


def test_reset_with_new_row_bounds_recomputes_norms_and_weight():
    p = synthetic.generate(1500, 1200, 6, seed=9)
    s = capi.Solver(p, tol=1e-6, iteration_limit=LIMIT)
    s.advance()
    lo = np.where(np.isfinite(p["lo"]), p["lo"] - 0.5, p["lo"])
    hi = np.where(np.isfinite(p["hi"]), p["hi"] + 0.25, p["hi"])
    s.reset(lo=lo, hi=hi, tol=1e-6, iteration_limit=LIMIT)
    a = s.advance()
    fresh = capi.Solver(dict(p, lo=lo, hi=hi), tol=1e-6, iteration_limit=LIMIT)
    b = fresh.advance()
    same_result(a, b, s.solution(), fresh.solution())
    assert a["norm_b"] == b["norm_b"] and a["initial_primal_weight"] == b["initial_primal_weight"]


def test_settings_travel_with_reset():
    p = synthetic.generate(800, 700, 6, seed=2)
    s = capi.Solver(p, tol=1e-4, iteration_limit=LIMIT)
    a = s.advance()
    s.reset(tol=0.0, iteration_limit=80)  # limits are looked at in major iterations (every 40)
    b = s.advance()
    assert a["status_name"] == "Optimal" and b["status_name"] == "IterationLimit" and b["steps_taken"] == 80
    s.reset(tol=1e-8, iteration_limit=20 * LIMIT)
    c = s.advance()
    assert c["status_name"] == "Optimal" and c["steps_taken"] > a["steps_taken"]


def test_config5_sequence_through_one_persistent_solver(golden_problems):
    """20 re-solves of the 50v-10 relaxation after single-variable tightenings, warm-started from the previous
    primal/dual (relaxed_lp.cu:74-108), all through ONE solver object; objective against HiGHS each time, and the
    sequence must be cheaper than constructing a solver per call"""
    from scipy.optimize import linprog
    p = dict(golden_problems["mip-50v-10-free-bound-relaxation"]["problem"])
    p.pop("var_types", None)
    A = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(p["m"], p["n"]))
    A_ub = sp.vstack([A[np.isfinite(p["hi"])], -A[np.isfinite(p["lo"])]])
    b_ub = np.concatenate([p["hi"][np.isfinite(p["hi"])], -p["lo"][np.isfinite(p["lo"])]])
    s = capi.Solver(p, tol=1e-6, iteration_limit=50000)
    assert s.advance()["status_name"] == "Optimal"
    x, y, _ = s.solution()
    rng = np.random.default_rng(3)
    lb, ub = p["lb"].copy(), p["ub"].copy()
    plan, solved = [], 0
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
        ref = linprog(p["c"], A_ub=A_ub, b_ub=b_ub, bounds=list(zip(nlb, nub)), method="highs")
        if ref.status != 0:
            continue  # infeasible child
        s.reset(lb=nlb, ub=nub, init_x=x, init_y=y, tol=1e-6, iteration_limit=50000)
        r = s.advance()
        assert r["status_name"] == "Optimal"
        assert r["primal_objective"] == pytest.approx(ref.fun, abs=2e-5 * (1 + abs(ref.fun)))
        plan.append((nlb, nub, x.copy(), y.copy()))
        x, y, _ = s.solution()
        lb, ub = nlb, nub
        solved += 1
    assert solved >= 5
    # the same sequence timed both ways (same work in the loop: the results are bit-identical)
    t0 = time.perf_counter()
    for nlb, nub, ix, iy in plan:
        s.reset(lb=nlb, ub=nub, init_x=ix, init_y=iy, tol=1e-6, iteration_limit=50000)
        s.advance()
    t_reset = time.perf_counter() - t0
    t0 = time.perf_counter()
    for nlb, nub, ix, iy in plan:
        f = capi.Solver(dict(p, lb=nlb, ub=nub), tol=1e-6, iteration_limit=50000, init_x=ix, init_y=iy)
        f.advance()
        f.close()
    t_fresh = time.perf_counter() - t0
    print("config-5 sequence of %d re-solves: persistent %.2f ms, solver per call %.2f ms" % (len(plan), 1e3 * t_reset, 1e3 * t_fresh))
    if "PYTEST_XDIST_WORKER" not in os.environ:  # (a rate comparison: not under the contention soak, where four processes share the GPU)
        assert t_reset < 1.1 * t_fresh  # set-up is ~0.5 ms of a ~9 ms solve here (streams and arena are recycled anyway)
