"""GPU: randomized LPs with every bound flavour (free / lower / upper / boxed / fixed variables; '<=', '>=', '=',
ranged and free rows; minimise and maximise; objective offset) against an INDEPENDENT solver (scipy/HiGHS) and
against the oracle.  Feasible and bounded by construction."""
import threading

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

from cuopt_amd import capi
from oracle import orcbind

pytestmark = pytest.mark.gpu
INF = np.inf


def random_lp(seed, m=40, n=60, density=0.15):
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csr")
    A.sort_indices()
    x0 = rng.standard_normal(n)
    kind = rng.integers(0, 5, size=n)  # 0 free, 1 lower, 2 upper, 3 boxed, 4 fixed
    lb = np.where(np.isin(kind, [1, 3]), x0 - rng.random(n) * 2, -INF)
    ub = np.where(np.isin(kind, [2, 3]), x0 + rng.random(n) * 2, INF)
    lb = np.where(kind == 4, x0, lb)
    ub = np.where(kind == 4, x0, ub)
    ax = A @ x0
    rk = rng.integers(0, 5, size=m)  # 0 <=, 1 >=, 2 =, 3 ranged, 4 free row
    lo = np.where(np.isin(rk, [1, 3]), ax - rng.random(m), -INF)
    hi = np.where(np.isin(rk, [0, 3]), ax + rng.random(m), INF)
    lo = np.where(rk == 2, ax, lo)
    hi = np.where(rk == 2, ax, hi)
    # bounded: c = A^T y0 + z0 with multipliers whose signs match the finite sides at a KKT point near x0
    y0 = rng.standard_normal(m)
    y0 = np.where(np.isinf(lo) & np.isinf(hi), 0.0, y0)
    y0 = np.where(np.isinf(lo) & np.isfinite(hi), -np.abs(y0), y0)
    y0 = np.where(np.isfinite(lo) & np.isinf(hi), np.abs(y0), y0)
    z0 = rng.standard_normal(n)
    z0 = np.where(kind == 0, 0.0, z0)
    z0 = np.where(kind == 1, np.abs(z0), z0)
    z0 = np.where(kind == 2, -np.abs(z0), z0)
    c = A.T @ y0 + z0
    maximize = bool(seed % 2)
    if maximize:
        c = -c
    return dict(m=m, n=n, offsets=A.indptr.astype(np.int32), indices=A.indices.astype(np.int32), values=A.data,
                c=c, lo=lo, hi=hi, lb=lb, ub=ub, maximize=maximize, objective_offset=float(seed) * 0.25), A


def highs(p, A):
    sgn = -1.0 if p["maximize"] else 1.0
    rows_hi, rows_lo = np.isfinite(p["hi"]), np.isfinite(p["lo"])
    eq = rows_hi & rows_lo & (p["lo"] == p["hi"])
    A_ub = sp.vstack([A[rows_hi & ~eq], -A[rows_lo & ~eq]])
    b_ub = np.concatenate([p["hi"][rows_hi & ~eq], -p["lo"][rows_lo & ~eq]])
    res = linprog(sgn * p["c"], A_ub=A_ub if A_ub.shape[0] else None, b_ub=b_ub if A_ub.shape[0] else None,
                  A_eq=A[eq] if eq.any() else None, b_eq=p["lo"][eq] if eq.any() else None,
                  bounds=list(zip(np.where(np.isinf(p["lb"]), None, p["lb"]), np.where(np.isinf(p["ub"]), None, p["ub"]))),
                  method="highs")
    assert res.status == 0, res.message
    return sgn * res.fun + p["objective_offset"]


@pytest.mark.parametrize("resident", [True, False], ids=["resident-loop", "multi-launch"])
@pytest.mark.parametrize("seed", range(12))
def test_random_lp_against_highs_and_oracle(seed, resident, monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SMALL", "1" if resident else "0")  # both loops see every bound flavour
    p, A = random_lp(seed)
    ref = highs(p, A)
    r = capi.solve(p, method=1, tol=1e-8, iteration_limit=200000)
    o = orcbind.solve(p, tol=1e-8, iteration_limit=200000)
    assert r["status"] == o["status"] == "Optimal"
    scale = 1.0 + abs(ref)
    assert abs(r["objective"] - ref) <= 2e-6 * scale
    assert abs(o["primal_objective"] - ref) <= 2e-6 * scale
    x = r["x"]
    assert np.all(x >= p["lb"] - 1e-6) and np.all(x <= p["ub"] + 1e-6)
    ax = A @ x
    assert np.all(ax >= p["lo"] - 1e-5 * scale) and np.all(ax <= p["hi"] + 1e-5 * scale)
    # the default (Concurrent) method path: simplex-grade tolerance on small LPs
    assert abs(capi.solve(p)["objective"] - ref) <= 2e-6 * scale


def test_concurrent_solves_with_different_presets_are_independent():
    """the reference keeps the presets in process globals (not re-entrant across modes); here every
    solver owns its hyper-parameters: two threads with different presets must reproduce their solo results"""
    ps = [random_lp(100 + i, m=120, n=150)[0] for i in range(4)]
    modes = [1, 0, 3, 2]
    solo = [capi.solve(p, method=1, pdlp_solver_mode=md, tol=1e-6) for p, md in zip(ps, modes)]
    out = [None] * 4

    def work(i):
        out[i] = capi.solve(ps[i], method=1, pdlp_solver_mode=modes[i], tol=1e-6)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    for a, b in zip(solo, out):
        assert b is not None and a["status"] == b["status"] == "Optimal"
        assert (a["steps_taken"], a["objective"]) == (b["steps_taken"], b["objective"])
        np.testing.assert_array_equal(a["x"], b["x"])


def test_batch_solve_equals_individual_solves():
    """call_batch_solve (LP/utilities/cython_solve.cu:264-296): a batch of independent LPs on one GPU"""
    ps = [random_lp(200 + i, m=60 + 5 * i, n=80 + 3 * i)[0] for i in range(9)]
    solo = [capi.Solver(p, tol=1e-6).advance() for p in ps]
    batch = capi.batch_solve(ps, tol=1e-6, max_threads=4)
    for a, b, p in zip(solo, batch, ps):
        assert a["status_name"] == b["status_name"] == "Optimal"
        assert (a["steps_taken"], a["primal_objective"]) == (b["steps_taken"], b["primal_objective"])
        assert len(b["x"]) == p["n"] and np.all(b["x"] >= p["lb"] - 1e-6)
