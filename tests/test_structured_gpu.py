"""GPU: the structured LP families that stand in for the Mittelmann LPs of pdlp_test.cu:189-235 (staircase, block-angular
with linking rows of thousands of nonzeros, power-law row lengths), at a size the oracle finishes in seconds: SpMV of every
layout against the oracle's CSR sums, first iterations against the oracle's decisions, solve against the optimum known by
construction and the oracle's own result (run_sub_mittleman's pattern: objective + host-verified residuals)."""
import numpy as np
import pytest

from cuopt_amd import capi, synthetic
from oracle import orcbind
from conftest import set_tune
from test_solve_gpu import host_check

pytestmark = pytest.mark.gpu
KINDS = ["staircase", "block_angular", "powerlaw", "multiband"]


@pytest.fixture(scope="module", params=KINDS)
def lp(request):
    return synthetic.generate_structured(request.param, m=70000, n=70000, k=8, seed=11)


@pytest.mark.parametrize("layout", ["auto", "stream", "panel", "jag"])
def test_spmv_against_the_oracle_in_every_layout(lp, layout, monkeypatch):
    p = lp
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", layout)
    set_tune(monkeypatch, slab_bytes=64 * 1024)  # several slabs even at this size
    dev = capi.Device(p)
    lay = dev.layout()
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    for side, got, ref, lens in (("A", dev.spmv(x, False, p["m"]), orcbind.spmv(p["offsets"], p["indices"], p["values"], x), np.diff(p["offsets"])),
                                 ("At", dev.spmv(y, True, p["n"]), orcbind.spmv(to, ti, tv, y), np.diff(to))):
        if lay[side].get("row_sums") != "by_nonzero":  # (the long-tail panels deal row sums by nonzero: every row at the tolerance)
            np.testing.assert_array_equal(got[lens <= 128], ref[lens <= 128])  # short rows: left to right, bit-exact
        scale = 1e-12 * (1 + np.abs(ref).max())
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=scale)  # long rows: fixed tree
    dev.close()


def test_first_iterations_follow_the_oracle(lp):
    p = lp
    for its in (5, 40):
        r = capi.Solver(p, tol=0.0, iteration_limit=its).advance()
        o = orcbind.solve(p, tol=0.0, iteration_limit=its)
        assert (r["steps_taken"], r["attempted_steps"]) == (int(o["steps_taken"]), int(o["attempted_steps"]))
        assert r["step_size"] == pytest.approx(o["final_step_size"], rel=1e-8)
        assert r["primal_objective"] == pytest.approx(o["primal_objective"], rel=1e-8, abs=1e-8)


def test_solve_reaches_the_known_optimum_like_the_oracle(lp):
    p = lp
    r = capi.solve(p, method=1, tol=1e-5)
    o = orcbind.solve(p, tol=1e-5)
    assert r["status"] == o["status"] == "Optimal"
    scale = 1 + abs(p["objective_star"])
    assert abs(r["objective"] - p["objective_star"]) <= 2e-4 * scale
    assert abs(r["objective"] - o["primal_objective"]) <= 2e-4 * scale
    assert 0.5 * o["steps_taken"] - 80 <= r["steps_taken"] <= 2.0 * o["steps_taken"] + 80
    host_check(p, r, eps=1e-5)  # the returned x, y re-verified on the host against the termination inequalities
