"""CPU: the library's own small-LP dual simplex (cuopt_amd/csrc/dual_simplex.cpp, host code -- no GPU involved) against the verdicts and
objectives of the REFERENCE's dual simplex held in the goldens (tests/golden/problems.json: the LP relaxations of datasets/mip and
the LP fixtures; tests/golden/mps_parser.json: all 21 non-empty LP files of datasets/linear_programming)."""
import os

import numpy as np
import pytest

from conftest import decode_problem, set_tune
from cuopt_amd import capi

STATUS = {"OPTIMAL": "Optimal", "INFEASIBLE": "PrimalInfeasible", "UNBOUNDED": "Unbounded"}


def _check_vertex(p, r, tol=1e-7):
    """primal feasible, reduced costs = (+-c) - A^T y of the converted minimisation (both engines return that convention, like the
    reference's), signs of the reduced costs / duals match the active bounds"""
    import scipy.sparse as sp
    A = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(p["m"], p["n"]))
    x, y, z = r["x"], r["y"], r["reduced_cost"]
    ax = A @ x
    scale = 1 + max(np.abs(ax).max(initial=0), np.abs(x).max(initial=0))
    assert np.all(ax >= p["lo"] - tol * scale) and np.all(ax <= p["hi"] + tol * scale)
    assert np.all(x >= p["lb"] - tol * scale) and np.all(x <= p["ub"] + tol * scale)
    sense = -1.0 if p.get("maximize") else 1.0
    np.testing.assert_allclose(sense * p["c"] - A.T @ y, z, atol=tol * (1 + np.abs(p["c"]).max()))
    # minimisation form: a positive reduced cost needs x at its lower bound, a negative one at its upper bound
    zs, ys = z, y
    assert np.all((zs <= tol) | (np.abs(x - p["lb"]) <= tol * scale)) and np.all((zs >= -tol) | (np.abs(x - p["ub"]) <= tol * scale))
    assert np.all((ys <= tol) | (np.abs(ax - p["lo"]) <= tol * scale)) and np.all((ys >= -tol) | (np.abs(ax - p["hi"]) <= tol * scale))


def test_the_lp_relaxations_and_fixtures_of_the_goldens(golden_problems):
    for name, g in golden_problems.items():
        p, ref = g["problem"], g["meta"]["reference_dual_simplex"]
        if p["m"] > 500:
            continue  # cod105 (1024 rows, 4 k pivots): below
        r = capi.dual_simplex(p)
        if name == "mip-minrep_inf-relaxation":
            # 0.0210643 + 0.978936 > 1: the LP has a ray whose cost is -3e-6 per unit.  The reference's simplex calls it optimal
            # (0.21064), HiGHS unbounded; this engine abstains and PDLP answers (tests/test_solve_gpu.py)
            assert r["status"] == "NumericalError"
            continue
        assert r["status"] == STATUS[ref["status"]], name
        assert r["objective"] == pytest.approx(ref["objective"], rel=1e-9, abs=1e-9), name
        _check_vertex(p, r)


def test_the_largest_relaxation_of_the_goldens(golden_problems):
    """cod105_max: 1024 x 1024, 57 k nonzeros, ~4 k pivots under steepest-edge pricing (the reference's simplex: 6347 pivots, 18.28571109;
    the optimum is 128/7)"""
    g = golden_problems["mip-cod105_max-relaxation"]
    r = capi.dual_simplex(g["problem"], time_limit=240)
    assert r["status"] == "Optimal"
    assert r["objective"] == pytest.approx(128.0 / 7.0, rel=1e-9)
    assert r["objective"] == pytest.approx(g["meta"]["reference_dual_simplex"]["objective"], rel=1e-6)
    _check_vertex(g["problem"], r, tol=1e-6)


def test_every_lp_file_of_the_reference(golden_parser):
    seen = 0
    for name, v in golden_parser.items():
        if not v.get("ok") or "reference_dual_simplex" not in v:
            continue
        p, ref = decode_problem(v), v["reference_dual_simplex"]
        r = capi.dual_simplex(p)
        assert r["status"] == STATUS[ref["status"]], name
        if ref["status"] == "OPTIMAL":
            assert r["objective"] == pytest.approx(ref["objective"], rel=1e-9, abs=1e-9), name
            _check_vertex(p, r)
        seen += 1
    assert seen == 21


def test_random_lps_against_highs():
    """ranged / equality rows, free / boxed / one-sided variables, real and small-integer (degenerate) data, some infeasible and
    some unbounded: same verdict and optimum as scipy's HiGHS (an "infeasible" of HiGHS is re-checked with a zero objective: its
    presolve says that of unbounded LPs too)"""
    import scipy.sparse as sp
    from scipy.optimize import linprog
    rng = np.random.default_rng(11)
    seen = {}
    for trial in range(80):
        m, n = int(rng.integers(2, 60)), int(rng.integers(2, 90))
        dens = rng.choice([0.1, 0.3, 0.7])
        integer = rng.random() < 0.5
        draw = (lambda size: rng.integers(0, 3, size=size).astype(float)) if integer else (lambda size: rng.random(size))
        A = (rng.integers(-3, 4, size=(m, n)).astype(float) if integer else rng.standard_normal((m, n))) * (rng.random((m, n)) < dens)
        x0 = rng.integers(-2, 3, size=n).astype(float) if integer else rng.standard_normal(n)
        lb = np.where(rng.random(n) < 0.3, -np.inf, x0 - draw(n))
        ub = np.where(rng.random(n) < 0.3, np.inf, x0 + draw(n))
        ax = A @ x0
        lo = np.where(rng.random(m) < 0.3, -np.inf, ax - draw(m))
        hi = np.where(rng.random(m) < 0.3, np.inf, ax + draw(m))
        eq = rng.random(m) < 0.25
        lo, hi = np.where(eq, ax, lo), np.where(eq, ax, hi)
        if rng.random() < 0.15:  # an empty row that cannot be satisfied
            i = rng.integers(m)
            lo[i] = ax[i] + 5 + abs(ax[i])
            hi[i] = lo[i] + 1
            A[i] = 0
        c = rng.integers(-3, 4, size=n).astype(float) if integer else rng.standard_normal(n)
        if rng.random() < 0.6:
            c = A.T @ rng.standard_normal(m) + 0.1 * c
        S = sp.csr_matrix(A)
        p = dict(m=m, n=n, offsets=S.indptr.astype(np.int32), indices=S.indices.astype(np.int32), values=S.data.astype(np.float64), c=c,
                 lo=lo, hi=hi, lb=lb, ub=ub, maximize=bool(trial % 2))
        rows, rhs = [], []
        for i in range(m):
            if np.isfinite(hi[i]):
                rows.append(A[i]), rhs.append(hi[i])
            if np.isfinite(lo[i]):
                rows.append(-A[i]), rhs.append(-lo[i])
        kw = dict(A_ub=np.array(rows) if rows else None, b_ub=np.array(rhs) if rows else None, bounds=list(zip(lb, ub)), method="highs")
        h = linprog(-c if p["maximize"] else c, **kw)
        verdict = {0: "Optimal", 2: "PrimalInfeasible", 3: "Unbounded"}[h.status]
        if verdict == "PrimalInfeasible" and linprog(0 * c, **kw).status == 0:
            verdict = "Unbounded"
        r = capi.dual_simplex(p, time_limit=30)
        seen[verdict] = seen.get(verdict, 0) + 1
        if verdict == "Unbounded":
            assert r["status"] in ("Unbounded", "NumericalError"), trial  # (the engine may abstain on a ray that costs next to nothing)
            continue
        assert r["status"] == verdict, trial
        if verdict == "Optimal":
            assert r["objective"] == pytest.approx(-h.fun if p["maximize"] else h.fun, rel=1e-6, abs=1e-6), trial
            _check_vertex(p, r, tol=1e-6)
    assert seen.get("Optimal", 0) >= 20 and seen.get("PrimalInfeasible", 0) >= 5 and seen.get("Unbounded", 0) >= 5


def test_limits_and_size_gate():
    from cuopt_amd import synthetic
    p = synthetic.generate(300, 260, 6, seed=4)
    assert capi.dual_simplex(p, iteration_limit=3)["status"] == "IterationLimit"
    big = synthetic.generate(4000, 3000, 4, seed=4)
    os.environ["CUOPT_AMD_SIMPLEX_MAX_ROWS"] = "3000"
    try:
        assert capi.dual_simplex(big)["status"] == "TooLarge"  # beyond the engine's size limits nothing is done: PDLP has the rest
    finally:
        del os.environ["CUOPT_AMD_SIMPLEX_MAX_ROWS"]
    assert capi.dual_simplex(big, time_limit=0.05)["status"] == "TimeLimit"
    r = capi.dual_simplex(p)
    assert r["status"] == "Optimal" and r["objective"] == pytest.approx(p["objective_star"], rel=1e-8)


def test_mid_size_lps_through_the_sparse_factorisation():
    """beyond what a dense basis inverse could hold: the sparse LU + product-form updates on random and structured LPs whose
    optimum is known by construction"""
    from cuopt_amd import synthetic
    cases = [synthetic.generate(2500, 2000, 4, seed=9), synthetic.generate(1200, 2400, 5, seed=10),
             synthetic.generate_structured("block_angular", 4000, 4000, 6, seed=3),
             synthetic.generate_structured("staircase", 3000, 3000, 4, seed=3)]
    for p in cases:
        r = capi.dual_simplex(p, time_limit=200)
        assert r["status"] == "Optimal", (p["m"], p["n"])
        assert r["objective"] == pytest.approx(p["objective_star"], rel=1e-8, abs=1e-8)
        _check_vertex(p, r, tol=1e-6)
    # the same pivots under Dantzig pricing end at the same optimum (the steepest-edge weights only choose among infeasible rows)
    os.environ["CUOPT_AMD_TUNE"] = "simplex_pricing=dantzig"
    try:
        r = capi.dual_simplex(cases[1], time_limit=200)
    finally:
        del os.environ["CUOPT_AMD_TUNE"]
    assert r["status"] == "Optimal" and r["objective"] == pytest.approx(cases[1]["objective_star"], rel=1e-8)


def test_start_from_a_point_is_a_crossover():
    """cuoptamd_dual_simplex_from: from the optimal vertex itself no pivot is needed; from a point 1e-4 away (what PDLP returns) far
    fewer than from the slack basis; from a poor point the answer is still the optimum"""
    from cuopt_amd import synthetic
    rng = np.random.default_rng(5)
    for p in (synthetic.generate(300, 260, 6, seed=4), synthetic.generate(1500, 1200, 4, seed=4)):
        cold = capi.dual_simplex(p)
        assert cold["status"] == "Optimal"
        same = capi.dual_simplex(p, x0=cold["x"], y0=cold["y"])
        assert same["status"] == "Optimal" and same["iterations"] <= cold["iterations"] // 20
        assert same["objective"] == pytest.approx(cold["objective"], rel=1e-10)
        near = capi.dual_simplex(p, x0=cold["x"] + 1e-4 * rng.standard_normal(p["n"]) * (1 + np.abs(cold["x"])),
                                 y0=cold["y"] + 1e-4 * rng.standard_normal(p["m"]) * (1 + np.abs(cold["y"])))
        assert near["status"] == "Optimal" and near["iterations"] <= cold["iterations"] // 2
        assert near["objective"] == pytest.approx(cold["objective"], rel=1e-9)
        _check_vertex(p, near, tol=1e-6)
        poor = capi.dual_simplex(p, x0=rng.standard_normal(p["n"]), time_limit=120)  # no duals, nothing to do with the optimum
        assert poor["status"] == "Optimal" and poor["objective"] == pytest.approx(cold["objective"], rel=1e-9)
    # infeasible and unbounded LPs are recognised from a warm start as well
    g = dict(m=1, n=2, offsets=np.array([0, 2], np.int32), indices=np.array([0, 1], np.int32), values=np.array([1.0, 1.0]),
             c=np.array([1.0, 1.0]), lo=np.array([3.0]), hi=np.array([np.inf]), lb=np.zeros(2), ub=np.ones(2))
    assert capi.dual_simplex(g, x0=np.array([1.0, 1.0]))["status"] == "PrimalInfeasible"
    g.update(ub=np.full(2, np.inf), c=np.array([-1.0, 0.0]))
    assert capi.dual_simplex(g, x0=np.array([5.0, 5.0]))["status"] == "Unbounded"


def test_cancel_is_honoured_within_milliseconds():
    """the Concurrent method cancels the simplex when PDLP has answered: the flag is looked at every pivot AND inside a
    factorisation (whose nucleus can take seconds on a large basis), so cuOptSolve never waits for this engine"""
    import ctypes as C
    import threading
    import time
    from cuopt_amd import synthetic
    from cuopt_amd.capi import LP, _f64, _i32, _ptr, lib
    p = synthetic.generate(12000, 10000, 5, seed=4)  # minutes of pivots from a cold start
    k = dict(offsets=_i32(p["offsets"]), indices=_i32(p["indices"]), values=_f64(p["values"]), c=_f64(p["c"]), lo=_f64(p["lo"]),
             hi=_f64(p["hi"]), lb=_f64(p["lb"]), ub=_f64(p["ub"]))
    lp = LP(p["m"], p["n"], _ptr(k["offsets"]), _ptr(k["indices"]), _ptr(k["values"]), _ptr(k["c"]), _ptr(k["lo"]), _ptr(k["hi"]),
            _ptr(k["lb"]), _ptr(k["ub"]), 0, 0.0)
    cancel, status, its, obj = C.c_int32(0), C.c_int(0), C.c_int(0), C.c_double(0.0)
    x, y, rc = np.zeros(p["n"]), np.zeros(p["m"]), np.zeros(p["n"])
    worker = threading.Thread(target=lambda: lib.cuoptamd_dual_simplex(C.byref(lp), 0.0, 0, C.byref(cancel), C.byref(status), C.byref(its),
                                                                      C.byref(obj), _ptr(x), _ptr(y), _ptr(rc)))
    worker.start()
    time.sleep(1.5)
    t0 = time.time()
    cancel.value = 1
    worker.join()
    assert time.time() - t0 < 0.5 and status.value == 9


def test_dependent_columns_in_a_suggested_basis_are_replaced():
    """a start point at which linearly dependent columns sit inside their bounds (here: duplicated columns, in the nucleus of the
    factorisation and as singletons): the factorisation turns one of each dependent set away, slacks fill the holes, and the
    solve ends at the optimum of the cold start"""
    import scipy.sparse as sp
    rng = np.random.default_rng(8)
    m, n = 40, 60
    A = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.3)
    A[:, 30:45] = A[:, 0:15]          # 15 duplicated columns
    A[:, 45:50] = 2.0 * A[:, 15:20]   # 5 scaled copies
    S = sp.csr_matrix(A)
    x_in = rng.random(n) * 0.5 + 0.25
    ax = A @ x_in
    p = dict(m=m, n=n, offsets=S.indptr.astype(np.int32), indices=S.indices.astype(np.int32), values=S.data.astype(np.float64),
             c=rng.standard_normal(n), lo=ax - 1.0, hi=ax + 1.0, lb=np.zeros(n), ub=np.ones(n))
    cold = capi.dual_simplex(p)
    assert cold["status"] == "Optimal"
    warm = capi.dual_simplex(p, x0=x_in)  # every variable strictly inside (0, 1): 60 candidates for 40 rows, 20 of them dependent
    assert warm["status"] == "Optimal" and warm["objective"] == pytest.approx(cold["objective"], rel=1e-9, abs=1e-9)
    _check_vertex(p, warm, tol=1e-6)


def test_degenerate_classics():
    """assignment and transportation problems (every vertex degenerate), a Klee-Minty cube and Beale's cycling example: the optimum
    of HiGHS, cold and from a point near it"""
    import scipy.sparse as sp
    from scipy.optimize import linprog
    rng = np.random.default_rng(3)

    def check(A, c, lo, hi, lb, ub):
        A = sp.csr_matrix(A)
        m, n = A.shape
        p = dict(m=m, n=n, offsets=A.indptr.astype(np.int32), indices=A.indices.astype(np.int32), values=A.data.astype(np.float64), c=np.asarray(c, float),
                 lo=np.asarray(lo, float), hi=np.asarray(hi, float), lb=np.asarray(lb, float), ub=np.asarray(ub, float))
        eq = p["lo"] == p["hi"]
        kw = dict(bounds=list(zip(p["lb"], p["ub"])), method="highs")
        if eq.any():
            kw.update(A_eq=A[eq], b_eq=p["hi"][eq])
        if (~eq).any():
            kw.update(A_ub=A[~eq], b_ub=p["hi"][~eq])  # (the inequality rows below are all of the form a.x <= hi)
        h = linprog(p["c"], **kw)
        assert h.status == 0
        for start in (None, h.x + 1e-4 * rng.standard_normal(n)):
            r = capi.dual_simplex(p, time_limit=60, x0=start)
            assert r["status"] == "Optimal" and r["objective"] == pytest.approx(h.fun, rel=1e-9, abs=1e-9)
            _check_vertex(p, r, tol=1e-6)

    for k in (12, 40):  # assignment
        rows, cols = [], []
        for i in range(k):
            for j in range(k):
                rows += [i, k + j]
                cols += [i * k + j] * 2
        A = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(2 * k, k * k))
        check(A, rng.integers(1, 100, size=k * k), np.ones(2 * k), np.ones(2 * k), np.zeros(k * k), np.full(k * k, np.inf))
    s, t = 15, 40  # transportation, integer supplies and demands
    sup = rng.integers(5, 30, size=s).astype(float)
    dem = rng.multinomial(int(sup.sum()), np.ones(t) / t).astype(float)
    rows, cols = [], []
    for i in range(s):
        for j in range(t):
            rows += [i, s + j]
            cols += [i * t + j] * 2
    A = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(s + t, s * t))
    check(A, rng.integers(1, 50, size=s * t), np.concatenate([sup, dem]), np.concatenate([sup, dem]), np.zeros(s * t), np.full(s * t, np.inf))
    n = 10  # Klee-Minty
    K = np.zeros((n, n))
    for i in range(n):
        K[i, i] = 1.0
        for j in range(i):
            K[i, j] = 2.0 ** (i - j + 1)
    check(K, -(2.0 ** (n - 1 - np.arange(n))), np.full(n, -np.inf), 5.0 ** (np.arange(n) + 1), np.zeros(n), np.full(n, np.inf))
    B = np.array([[0.25, -8, -1, 9], [0.5, -12, -0.5, 3], [0, 0, 1, 0]])  # Beale
    check(B, [-0.75, 20, -0.5, 6], np.full(3, -np.inf), [0.0, 0.0, 1.0], np.zeros(4), np.full(4, np.inf))


def test_sparse_and_dense_solves_walk_the_same_pivots(monkeypatch):
    """FTRAN / BTRAN through the depth-first reach (from the right-hand side's nonzeros) and through the dense loops are the same
    arithmetic on the entries that are not zero: the same pivots, the same vertex -- cold, and through the primal simplex of a
    start from a point"""
    from cuopt_amd import synthetic
    p = synthetic.generate(3000, 2400, 3, seed=5)
    runs = {}
    for mode in ("dense", "sparse", "auto"):
        set_tune(monkeypatch, simplex_solves=mode)
        cold = capi.dual_simplex(p, time_limit=120)
        assert cold["status"] == "Optimal" and cold["objective"] == pytest.approx(p["objective_star"], rel=1e-8)
        warm = capi.dual_simplex(p, time_limit=120, x0=cold["x"] * (1 + 1e-3 * np.cos(np.arange(p["n"]))))
        assert warm["status"] == "Optimal"
        runs[mode] = (cold["iterations"], cold["objective"], warm["iterations"], warm["objective"])
    assert runs["dense"][0] == runs["sparse"][0] == runs["auto"][0] and runs["dense"][2] == runs["sparse"][2] == runs["auto"][2]
    assert runs["dense"][1] == pytest.approx(runs["sparse"][1], rel=1e-12) and runs["dense"][3] == pytest.approx(runs["sparse"][3], rel=1e-12)


def test_bound_flipping_ratio_test_and_the_helper_thread(monkeypatch):
    """The long-step rule (phase2.cpp:348-470 is the reference's): a Harris group whose breakpoints the row's infeasibility outlasts is
    flipped to its other bounds instead of one of it entering.  Same optimum with and without it, fewer pivots with it where the
    start leans on bounds far away (the boxed infinite bounds of a cold start; 0/1 boxes); the two solves a pivot hands to the helper
    thread (weights, the basic variables' answer to the flips) give the same pivots as the same solves done in line."""
    from cuopt_amd import synthetic
    rng = np.random.default_rng(12)
    boxed = synthetic.generate(600, 900, 5, seed=8)
    boxed = dict(boxed, lb=np.zeros(900), ub=np.where(rng.random(900) < 0.7, 1.0 + np.abs(boxed["x_star"]), np.inf))  # (x_star stays feasible)
    cases = [("block angular", synthetic.generate_structured("block_angular", 3000, 3000, 6, seed=3), True),
             ("random", synthetic.generate(1500, 1200, 5, seed=2), False), ("boxed", boxed, False)]
    for name, p, expect_fewer in cases:
        out = {}
        for flips, helper_rows in ((0, 1 << 30), (1, 1 << 30), (1, 1)):
            set_tune(monkeypatch, simplex_flips=flips, simplex_helper_rows=helper_rows)
            r = capi.dual_simplex(p, time_limit=200)
            assert r["status"] == "Optimal", (name, flips)
            _check_vertex(p, r, tol=1e-6)
            out[(flips, helper_rows == 1)] = (r["iterations"], r["objective"])
        assert out[(1, False)] == out[(1, True)], name  # the helper changes who solves, not what
        assert out[(0, False)][1] == pytest.approx(out[(1, False)][1], rel=1e-9, abs=1e-9), name
        if expect_fewer:
            assert out[(1, False)][0] < 0.8 * out[(0, False)][0], (name, out)
    set_tune(monkeypatch, simplex_flips=None, simplex_helper_rows=None)


def _lp(rows, c, lo, hi, lb, ub, maximize=False):
    import scipy.sparse as sp
    A = sp.csr_matrix(np.asarray(rows, float))
    return dict(m=A.shape[0], n=A.shape[1], offsets=A.indptr.astype(np.int32), indices=A.indices.astype(np.int32), values=A.data.astype(float),
                c=np.asarray(c, float), lo=np.asarray(lo, float), hi=np.asarray(hi, float), lb=np.asarray(lb, float), ub=np.asarray(ub, float),
                maximize=maximize, objective_offset=0.0)


def test_infeasibility_is_never_proven_by_the_artificial_box():
    """round-3 advisor (high): infinite bounds are boxed at 1e5 / 1e8 x the largest finite bound; a row without an entering candidate
    is a proof of infeasibility only if no helping variable is stopped by such a box bound.  These LPs are feasible with optima beyond
    the first box: the engine must answer them (second box) or abstain, never say Infeasible."""
    INF = np.inf
    # (a) min x  s.t. 1e-7 x >= 1, x >= 0: optimum 1e7
    r = capi.dual_simplex(_lp([[1e-7]], [1.0], [1.0], [INF], [0.0], [INF]))
    assert r["status"] == "Optimal" and r["objective"] == pytest.approx(1e7, rel=1e-9)
    # (b) x0 >= 1, x_{k+1} >= 10 x_k: min x_7 = 1e7
    n = 8
    rows = np.zeros((n - 1, n))
    for k in range(n - 1):
        rows[k, k + 1], rows[k, k] = 1.0, -10.0
    c = np.zeros(n)
    c[-1] = 1.0
    r = capi.dual_simplex(_lp(rows, c, np.zeros(n - 1), np.full(n - 1, INF), np.concatenate([[1.0], np.zeros(n - 1)]), np.full(n, INF)))
    assert r["status"] in ("Optimal", "NumericalError")
    if r["status"] == "Optimal":
        assert r["objective"] == pytest.approx(1e7, rel=1e-9)
    # (c) feasible only far beyond both boxes: 1e-12 x >= 1 -> abstains (PDLP would answer), not Infeasible
    r = capi.dual_simplex(_lp([[1e-12]], [1.0], [1.0], [INF], [0.0], [INF]))
    assert r["status"] in ("Optimal", "NumericalError"), r["status"]
    # (d) a genuinely infeasible LP keeps its verdict: x + y <= 1, x + y >= 2 inside finite AND infinite bounds
    r = capi.dual_simplex(_lp([[1.0, 1.0], [1.0, 1.0]], [1.0, 1.0], [-INF, 2.0], [1.0, INF], [0.0, 0.0], [INF, INF]))
    assert r["status"] == "PrimalInfeasible"
    r = capi.dual_simplex(_lp([[1.0, 1.0], [1.0, 1.0]], [1.0, 1.0], [-INF, 2.0], [1.0, INF], [-INF, -INF], [INF, INF]))
    assert r["status"] in ("PrimalInfeasible", "NumericalError")  # free variables: the row x + y is stopped by true row bounds only


def test_unbounded_needs_a_ray_of_the_lp_itself():
    """an optimum beyond the wider box must not be called Unbounded: min -x s.t. 1e-9 x <= 1 (optimum -1e9) against min -x, x >= 0"""
    INF = np.inf
    r = capi.dual_simplex(_lp([[1e-9]], [-1.0], [-INF], [1.0], [0.0], [INF]))
    assert r["status"] in ("Optimal", "NumericalError"), r["status"]
    r = capi.dual_simplex(_lp([[1.0, -1.0]], [-1.0, 0.0], [-INF], [5.0], [0.0, 0.0], [INF, INF]))
    assert r["status"] == "Unbounded"


def test_a_maximisation_returns_the_duals_of_the_converted_minimisation():
    """both engines hand out y, reduced costs of min -c (the reference's convention, dual_simplex/solve.cpp:256): here the simplex"""
    p = _lp([[1.0, 2.0], [3.0, 1.0]], [5.0, 8.0], [-np.inf, -np.inf], [12.0, 15.0], [0.0, 0.0], [10.0, 10.0], maximize=True)
    r = capi.dual_simplex(p)
    assert r["status"] == "Optimal"
    _check_vertex(p, r)
    assert np.all(r["y"] <= 1e-9)  # binding <= rows of a minimisation have non-positive multipliers in the c - A^T y convention


def test_presolve_and_the_way_back(monkeypatch):
    """simplex_presolve.hpp (the reference's simplex removes empty rows / columns and fixed variables first, presolve.cpp:26-212,585-662;
    singleton rows are added here): same verdicts and optima with and without it, and the restored duals are a vertex's -- a bound
    that came from a singleton row hands its multiplier to that row"""
    inf = np.inf
    # x0 >= 2 through a singleton row (active: the row carries the dual), -x1 >= -3 (an upper bound through a negative entry), a
    # singleton equality that fixes x2 and makes row 3 a singleton in turn, an empty row, an empty column with a cost
    p = _lp([[1, 0, 0, 0, 0], [0, -1, 0, 0, 0], [0, 0, 2, 0, 0], [0, 0, 1, 1, 0], [0, 0, 0, 0, 0], [1, 1, 1, 1, 0]],
            c=[1, -1, 1, 1, -2], lo=[2, -3, 4, 5, -1, -inf], hi=[inf, inf, 4, inf, 1, 100], lb=[0, 0, 0, 0, 0], ub=[inf, inf, inf, inf, 7])
    for pre in (0, 1):
        set_tune(monkeypatch, simplex_presolve=pre)
        r = capi.dual_simplex(p)
        assert r["status"] == "Optimal" and r["iterations"] == (0 if pre else r["iterations"])
        assert r["objective"] == pytest.approx(2 - 3 + 2 + 3 - 14) and np.allclose(r["x"], [2, 3, 2, 3, 7])
        _check_vertex(p, r)
        assert r["y"][0] == pytest.approx(1.0) and r["y"][1] == pytest.approx(1.0) and r["y"][4] == 0.0 and r["reduced_cost"][4] == pytest.approx(-2.0)
    # the same as a maximisation of -c; infeasible by an empty row, by crossing singleton rows
    q = dict(p, c=-p["c"], maximize=True)
    set_tune(monkeypatch, simplex_presolve=1)
    r = capi.dual_simplex(q)
    assert r["status"] == "Optimal" and r["objective"] == pytest.approx(10.0)
    _check_vertex(q, r)
    assert capi.dual_simplex(dict(p, lo=np.array([2, -3, 4, 5, 0.5, -inf])))["status"] == "PrimalInfeasible"
    assert capi.dual_simplex(_lp([[1, 0], [1, 0], [1, 1]], [1, 1], [3, -inf, 0], [inf, 2, 10], [0, 0], [inf, inf]))["status"] == "PrimalInfeasible"
    # (round-4 advisor) a column whose OWN bounds cross -- empty (x1 in [2, 1], both rows on x0 only) or not -- is infeasible with and
    # without the presolve; it used to be removed at the bound its cost points to and reported Optimal with x outside its bounds
    for pre in (0, 1):
        set_tune(monkeypatch, simplex_presolve=pre)
        assert capi.dual_simplex(_lp([[1, 0], [1, 0]], [1, 1], [1, -inf], [inf, 5], [0, 2], [inf, 1]))["status"] == "PrimalInfeasible"
        assert capi.dual_simplex(_lp([[1, 1], [1, 0]], [1, 1], [1, -inf], [inf, 5], [0, 2], [inf, 1]))["status"] == "PrimalInfeasible"
    set_tune(monkeypatch, simplex_presolve=1)
    # an empty column whose cost points to an infinite bound is the engine's to judge: unbounded here, infeasible there
    assert capi.dual_simplex(_lp([[1, 0], [1, 0]], [1, -1], [1, -inf], [inf, 5], [0, 0], [inf, inf]))["status"] == "Unbounded"
    assert capi.dual_simplex(_lp([[1, 0], [1, 0]], [1, -1], [6, -inf], [inf, 5], [0, 0], [inf, inf]))["status"] == "PrimalInfeasible"
    # random LPs with such rows and columns mixed in: HiGHS agrees, with and without
    from scipy.optimize import linprog
    import scipy.sparse as sp
    rng = np.random.default_rng(21)
    for trial in range(6):
        m, n = 40 + 10 * trial, 60 + 5 * trial
        A = sp.random(m, n, density=0.08, random_state=trial, format="lil")
        for i in rng.choice(m, 8, replace=False):  # singleton rows
            A[i, :] = 0
            A[i, rng.integers(n)] = rng.choice([-2.0, 0.5, 1.0, 3.0])
        A[rng.choice(m, 2, replace=False), :] = 0  # empty rows
        for j in rng.choice(n, 3, replace=False):  # empty columns
            A[:, j] = 0
        A = sp.csr_matrix(A)
        A.eliminate_zeros()
        xs = rng.uniform(0, 2, n)
        ax = A @ xs
        lo = np.where(rng.random(m) < 0.6, ax - rng.uniform(0, 1, m), -inf)
        hi = np.where(rng.random(m) < 0.6, ax + rng.uniform(0, 1, m), inf)
        eq = rng.random(m) < 0.15
        lo, hi = np.where(eq, ax, lo), np.where(eq, ax, hi)
        lb, ub = np.zeros(n), np.where(rng.random(n) < 0.5, 3.0, inf)
        fixed = rng.choice(n, 4, replace=False)
        lb[fixed] = ub[fixed] = xs[fixed]
        c = rng.standard_normal(n)
        c[np.diff(A.tocsc().indptr) == 0] = np.abs(c[np.diff(A.tocsc().indptr) == 0])  # (empty columns: towards their finite bound)
        p = dict(m=m, n=n, offsets=A.indptr.astype(np.int32), indices=A.indices.astype(np.int32), values=A.data, c=c, lo=lo, hi=hi, lb=lb, ub=ub,
                 maximize=False, objective_offset=0.5)
        Aub = sp.vstack([A[np.isfinite(hi)], -A[np.isfinite(lo)]])
        bub = np.concatenate([hi[np.isfinite(hi)], -lo[np.isfinite(lo)]])
        ref = linprog(c, A_ub=Aub, b_ub=bub, bounds=list(zip(lb, [None if not np.isfinite(u) else u for u in ub])), method="highs")
        out = []
        for pre in (0, 1):
            set_tune(monkeypatch, simplex_presolve=pre)
            r = capi.dual_simplex(p)
            if ref.status == 0:
                assert r["status"] == "Optimal" and r["objective"] == pytest.approx(ref.fun + 0.5, rel=1e-8, abs=1e-8), (trial, pre)
                _check_vertex(p, r, tol=1e-6)
            else:
                assert r["status"] in ("PrimalInfeasible", "Unbounded", "NumericalError"), (trial, pre, ref.status)
            out.append(r["iterations"])
    set_tune(monkeypatch, simplex_presolve=None)
