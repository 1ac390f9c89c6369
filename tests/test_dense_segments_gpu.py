"""GPU: dense row segments (runs of >= 256 consecutive columns inside a row) are stored index-free and multiplied by two
streaming kernels of their own; every layout then works on the sparse remainder and adds their share ahead of its epilogue
(kernels_dense.hip; folded into the panel kernels where panels are the layout: spmv_panel.hpp).  Rows / columns a segment touches are compared at the long-row tolerance (their sums are split in
two), everything else stays bit-identical to the oracle."""
import numpy as np
import pytest
from conftest import set_tune

from cuopt_amd import capi, synthetic
from oracle import orcbind
from test_solve_gpu import host_check

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def dense_on(monkeypatch):
    """at this size the two dense rows hold 1.4 % of the nonzeros, under the 2 % from which the path switches itself on"""
    set_tune(monkeypatch, dense="1")


@pytest.fixture(scope="module")
def lp():
    return synthetic.generate_structured("dense_rows", m=70000, n=70000, k=8, seed=11)


def _touched(p):
    """rows that own a run of >= 256 consecutive columns, and the columns those runs cover"""
    off, idx = p["offsets"], p["indices"]
    rows, cols = np.zeros(p["m"], bool), np.zeros(p["n"], bool)
    for r in np.nonzero(np.diff(off) >= 256)[0]:
        c = idx[off[r]:off[r + 1]]
        brk = np.nonzero(np.diff(c) != 1)[0]
        starts, ends = np.concatenate([[0], brk + 1]), np.concatenate([brk + 1, [len(c)]])
        for a, b in zip(starts, ends):
            if b - a >= 256:
                rows[r] = True
                cols[c[a]:c[b - 1] + 1] = True
    return rows, cols


@pytest.mark.parametrize("layout", ["auto", "stream", "panel", "jag", "pb"])
def test_products_against_the_oracle_in_every_layout(lp, layout, monkeypatch):
    p = lp
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", layout)
    set_tune(monkeypatch, slab_bytes=str(64 * 1024))
    dev = capi.Device(p)
    rows, cols = _touched(p)
    assert rows.sum() >= 2 and cols.sum() >= 3000
    info = dev.dense_info()
    assert info["on"] and info["segments"] >= 2 and info["entries"] >= 7000
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    ax, ref_ax = dev.spmv(x, False, p["m"]), orcbind.spmv(p["offsets"], p["indices"], p["values"], x)
    aty, ref_aty = dev.spmv(y, True, p["n"]), orcbind.spmv(to, ti, tv, y)
    np.testing.assert_array_equal(ax[~rows], ref_ax[~rows])      # untouched rows: bit-identical
    np.testing.assert_array_equal(aty[~cols], ref_aty[~cols])    # untouched columns: bit-identical
    np.testing.assert_allclose(ax, ref_ax, rtol=1e-12, atol=1e-12 * (1 + np.abs(ref_ax).max()))
    np.testing.assert_allclose(aty, ref_aty, rtol=1e-12, atol=1e-12 * (1 + np.abs(ref_aty).max()))
    dev.close()


def test_the_path_can_be_switched_off_and_gives_the_same_solve(lp, monkeypatch):
    p = lp
    got = {}
    for flag in ("0", "1"):
        set_tune(monkeypatch, dense=flag)
        r = capi.Solver(p, tol=0.0, iteration_limit=80).advance()
        got[flag] = (r["steps_taken"], r["attempted_steps"], r["num_restarts"], r["primal_objective"], r["step_size"])
    assert got["0"][:3] == got["1"][:3]
    assert got["0"][3] == pytest.approx(got["1"][3], rel=1e-9) and got["0"][4] == pytest.approx(got["1"][4], rel=1e-9)


def test_first_iterations_follow_the_oracle_and_the_solve_reaches_the_optimum(lp):
    p = lp
    for its in (5, 40):
        r = capi.Solver(p, tol=0.0, iteration_limit=its).advance()
        o = orcbind.solve(p, tol=0.0, iteration_limit=its)
        assert (r["steps_taken"], r["attempted_steps"]) == (int(o["steps_taken"]), int(o["attempted_steps"]))
        assert r["step_size"] == pytest.approx(o["final_step_size"], rel=1e-8)
    r = capi.solve(p, method=1, tol=1e-5)
    assert r["status"] == "Optimal"
    assert abs(r["objective"] - p["objective_star"]) <= 2e-4 * (1 + abs(p["objective_star"]))
    host_check(p, r, eps=1e-5)


def test_scaling_still_sees_the_whole_matrix(lp):
    """Ruiz / Pock-Chambolle run on the full CSR: the scaling vectors are the oracle's"""
    p = lp
    h = orcbind.hyper_preset(1)
    H = orcbind.H
    dev = capi.Device(p)
    dev.call("scaling_compute", int(h[H["ORC_H_DO_RUIZ"]]), int(h[H["ORC_H_RUIZ_ITERATIONS"]]),
             int(h[H["ORC_H_DO_POCK_CHAMBOLLE"]]), float(h[H["ORC_H_ALPHA_POCK_CHAMBOLLE"]]))
    dr, dc = orcbind.compute_scaling(p["m"], p["n"], p["offsets"], p["indices"], p["values"], h)
    np.testing.assert_allclose(dev.download("DROW", p["m"]), dr, rtol=1e-12)  # (long rows: fixed-tree norms)
    np.testing.assert_allclose(dev.download("DCOL", p["n"]), dc, rtol=1e-12)
    dev.close()


def test_a_row_with_a_duplicate_inside_a_run_keeps_every_entry(monkeypatch):
    """round-3 advisor: the C API does not canonicalise rows; a duplicate of a column INSIDE a run of consecutive columns must not
    vanish from the hot-loop matrix (the row then stays out of the index-free storage altogether)"""
    n, m = 3000, 40
    rng = np.random.default_rng(3)
    rows_idx, rows_val = [], []
    for r in range(m):
        if r == 7:
            cols = np.concatenate([[5], np.arange(0, 301)])  # stray duplicate of column 5 in front of the run 0..300
        elif r == 9:
            cols = np.arange(100, 500)  # a clean run: index-free
        else:
            cols = np.sort(rng.choice(n, size=6, replace=False))
        rows_idx.append(cols)
        rows_val.append(rng.standard_normal(len(cols)))
    off = np.concatenate([[0], np.cumsum([len(c) for c in rows_idx])]).astype(np.int32)
    p = dict(m=m, n=n, offsets=off, indices=np.concatenate(rows_idx).astype(np.int32), values=np.concatenate(rows_val), c=rng.standard_normal(n),
             lo=np.full(m, -np.inf), hi=np.full(m, 50.0), lb=np.zeros(n), ub=np.full(n, 3.0), maximize=False, objective_offset=0.0)
    monkeypatch.setenv("CUOPT_AMD_SMALL", "0")
    dev = capi.Device(p)
    info = dev.dense_info()
    assert info["on"] and info["segments"] == 1, info  # row 9 only
    x = rng.standard_normal(n)
    import scipy.sparse as sp
    A = sp.csr_matrix((p["values"], p["indices"], p["offsets"]), shape=(m, n))  # (scipy adds duplicates up: what the LP means)
    np.testing.assert_allclose(dev.spmv(x, False, m), A @ x, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(dev.spmv(rng.standard_normal(m) * 0 + 1.0, True, n), A.T @ np.ones(m), rtol=1e-12, atol=1e-12)
    dev.close()
