"""GPU: the resident single-workgroup PDHG loop used for small LPs (k_pdhg_small) against the multi-launch
path and the oracle: bit-identical iterates for one attempt, same decisions over the first major intervals,
same optimum; and it must actually be faster per iteration."""
import time

import numpy as np
import pytest

from cuopt_amd import capi, synthetic
from oracle import orcbind

pytestmark = pytest.mark.gpu


def solver(p, small, monkeypatch, **kw):
    monkeypatch.setenv("CUOPT_AMD_SMALL", "1" if small else "0")
    return capi.Solver(p, **kw)


def test_auto_selection(monkeypatch):
    monkeypatch.delenv("CUOPT_AMD_SMALL", raising=False)
    assert capi.Device(synthetic.generate(2000, 2000, 2, seed=3)).layout()["resident"]  # 512 lanes x 4 elements
    assert capi.Device(synthetic.generate(1000, 1000, 8, seed=3)).layout()["resident"]  # 512 lanes x 16 nonzeros
    assert capi.Device(synthetic.generate(300, 500, 4, seed=3)).layout()["resident"]  # 256 lanes
    assert not capi.Device(synthetic.generate(2000, 2000, 10, seed=3)).layout()["resident"]  # too many nonzeros
    assert not capi.Device(synthetic.generate(3000, 1000, 2, seed=3)).layout()["resident"]  # m > 2048


@pytest.mark.parametrize("shape", [(1800, 2000, 2), (1000, 900, 8), (200, 500, 6), (2048, 2048, 2), (40, 30, 5)])
def test_one_attempt_is_bit_identical_to_multi_launch(monkeypatch, shape):
    p = synthetic.generate(*shape, seed=12)
    rng = np.random.default_rng(2)
    x0, y0 = np.abs(rng.standard_normal(p["n"])), rng.standard_normal(p["m"])
    got = []
    for small in (True, False):
        monkeypatch.setenv("CUOPT_AMD_SMALL", "1" if small else "0")
        dev = capi.Device(p)
        assert dev.layout()["resident"] == small
        dev.call("set_initial", capi._ptr(x0), capi._ptr(y0))
        dev.call("set_step", 0.05, 1.3)
        dev.call("compute_aty")
        ctl = dev.run(1)
        got.append((ctl.attempts, ctl.steps_taken, dev.download("X", p["n"]), dev.download("Y", p["m"]),
                    dev.download("ATY", p["n"]), ctl.step_size))
    a, b = got
    assert a[:2] == b[:2]
    for u, v in zip(a[2:5], b[2:5]):
        np.testing.assert_array_equal(u, v)
    assert a[5] == pytest.approx(b[5], rel=1e-12)  # the step size comes out of differently ordered reductions


def test_trajectory_and_optimum(monkeypatch):
    p = synthetic.generate(1000, 1000, 8, seed=4)
    for its in (5, 40, 80):
        r = solver(p, True, monkeypatch, tol=0.0, iteration_limit=its).advance()
        o = orcbind.solve(p, tol=0.0, iteration_limit=its)
        assert (r["steps_taken"], r["attempted_steps"]) == (int(o["steps_taken"]), int(o["attempted_steps"]))
        assert r["step_size"] == pytest.approx(o["final_step_size"], rel=1e-8)
        assert r["primal_objective"] == pytest.approx(o["primal_objective"], rel=1e-8, abs=1e-8)
    a = solver(p, True, monkeypatch, tol=1e-8).advance()
    b = solver(p, False, monkeypatch, tol=1e-8).advance()
    assert a["status_name"] == b["status_name"] == "Optimal"
    assert a["primal_objective"] == pytest.approx(p["objective_star"], abs=2e-7 * (1 + abs(p["objective_star"])))
    assert a["primal_objective"] == pytest.approx(b["primal_objective"], abs=2e-7 * (1 + abs(p["objective_star"])))


def test_goldens_through_the_resident_loop(golden_problems, monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SMALL", "1")
    for name in ("afiro", "mip-50v-10-free-bound-relaxation", "mip-neos5-free-bound-relaxation"):
        g = golden_problems[name]
        p = dict(g["problem"])
        p.pop("var_types", None)
        r = capi.solve(p, method=1, tol=1e-8)
        ref = g["meta"]["reference_dual_simplex"]["objective"]
        assert r["status"] == "Optimal" and abs(r["objective"] - ref) <= 4e-8 * (1 + abs(ref))


def test_resident_loop_is_faster_per_iteration(golden_problems, monkeypatch):
    p = dict(golden_problems["mip-50v-10-free-bound-relaxation"]["problem"])
    p.pop("var_types", None)
    rate = {}
    for small in (True, False):
        s = solver(p, small, monkeypatch, tol=0.0)
        s.advance(400)
        t0 = time.perf_counter()
        s.advance(4000)
        s.device.call("synchronize")
        rate[small] = 4000 / (time.perf_counter() - t0)
    print("iterations/s resident %.0f, multi-launch %.0f" % (rate[True], rate[False]))
    assert rate[True] > 2.0 * rate[False]
