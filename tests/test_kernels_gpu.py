"""GPU: kernel-level parity of the HIP device layer (through the pdlpdev_* C-ABI) against the C oracle.
Element-wise kernels and SpMV rows of <= 128 nonzeros must be BIT-EXACT (same operation order, no FMA
contraction on either side); long rows and dot-product style reductions use a different summation
tree and are compared with rtol 1e-12."""
import numpy as np
import pytest

from cuopt_amd import capi, synthetic
from oracle import orcbind

pytestmark = pytest.mark.gpu
INF = np.inf


def ragged_problem(seed=5, m=3000, n=2500):
    """row lengths 0, 1, ..., a few > 128 (cooperative path) and one > 2048 (its own workgroup)"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 24, size=m)
    lens[::97] = 0
    lens[5], lens[700], lens[1500], lens[2999] = 300, 129, 2400, 2049
    lens = np.minimum(lens, n)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]).astype(np.int32)
    val = rng.standard_normal(off[-1])
    return dict(m=m, n=n, offsets=off, indices=idx, values=val, c=rng.standard_normal(n),
                lo=np.where(rng.random(m) < 0.3, -INF, rng.standard_normal(m) - 1.0),
                hi=np.where(rng.random(m) < 0.3, INF, rng.standard_normal(m) + 3.0),
                lb=np.where(rng.random(n) < 0.2, -INF, 0.0), ub=np.where(rng.random(n) < 0.5, INF, 5.0))


@pytest.fixture(scope="module")
def tiny():
    return synthetic.generate(**synthetic.CONFIGS["tiny"])


def test_spmv_bit_exact_on_short_rows(tiny):
    dev = capi.Device(tiny)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(tiny["n"]), rng.standard_normal(tiny["m"])
    to, ti, tv = orcbind.transpose(tiny["m"], tiny["n"], tiny["offsets"], tiny["indices"], tiny["values"])
    np.testing.assert_array_equal(dev.spmv(x, False, tiny["m"]), orcbind.spmv(tiny["offsets"], tiny["indices"], tiny["values"], x))
    np.testing.assert_array_equal(dev.spmv(y, True, tiny["n"]), orcbind.spmv(to, ti, tv, y))


def test_spmv_ragged_rows_empty_long_and_huge():
    p = ragged_problem()
    dev = capi.Device(p)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    ref = orcbind.spmv(p["offsets"], p["indices"], p["values"], x)
    got = dev.spmv(x, False, p["m"])
    lens = np.diff(p["offsets"])
    short = lens <= 128
    np.testing.assert_array_equal(got[short], ref[short])
    np.testing.assert_allclose(got[~short], ref[~short], rtol=1e-12, atol=1e-12)
    assert np.all(got[lens == 0] == 0.0)
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    np.testing.assert_allclose(dev.spmv(y, True, p["n"]), orcbind.spmv(to, ti, tv, y), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("mode", [1, 0, 3])
def test_scaling_vectors_and_scaled_problem_bit_exact(tiny, golden_problems, mode):
    """Ruiz + Pock-Chambolle (initial_scaling.cu:36-163,176-307) and scale_problem (:347-408)"""
    h = orcbind.hyper_preset(mode)
    H = orcbind.H
    for p in (tiny, golden_problems["afiro"]["problem"], ragged_problem()):
        dev = capi.Device(p)
        dev.call("scaling_compute", int(h[H["ORC_H_DO_RUIZ"]]), int(h[H["ORC_H_RUIZ_ITERATIONS"]]),
                 int(h[H["ORC_H_DO_POCK_CHAMBOLLE"]]), float(h[H["ORC_H_ALPHA_POCK_CHAMBOLLE"]]))
        dr, dc = orcbind.compute_scaling(p["m"], p["n"], p["offsets"], p["indices"], p["values"], h)
        exact = h[H["ORC_H_ALPHA_POCK_CHAMBOLLE"]] == 1.0  # pow() of libm vs ocml may differ by an ulp
        cmp = np.testing.assert_array_equal if exact else (lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-14))
        cmp(dev.download("DROW", p["m"]), dr)
        cmp(dev.download("DCOL", p["n"]), dc)
        if not exact:
            continue
        dev.call("scale_problem")
        rows = np.repeat(np.arange(p["m"]), np.diff(p["offsets"]))
        np.testing.assert_array_equal(dev.download("A_VALUES", len(p["values"])), p["values"] * dr[rows] * dc[p["indices"]])
        np.testing.assert_array_equal(dev.download("C", p["n"]), p["c"] * dc)
        np.testing.assert_array_equal(dev.download("LB", p["n"]), p["lb"] / dc)
        np.testing.assert_array_equal(dev.download("HI", p["m"]), p["hi"] * dr)


def test_initial_step_size_and_primal_weight(golden_problems, tiny):
    """pdlp_test.cu:237-239: afiro, Methodical1 scaling (Ruiz x5, PC alpha 1) -> 1.4893 / 0.0141652"""
    p = golden_problems["afiro"]["problem"]
    dev = capi.Device(p)
    dev.call("scaling_compute", 1, 5, 1, 1.0)
    dev.call("scale_problem")
    mx, c2, b2 = dev.init_norms()
    assert 1.0 / mx == pytest.approx(1.4893, abs=1e-4)
    assert np.sqrt(c2) / np.sqrt(b2) == pytest.approx(0.0141652, abs=1e-4)
    g = golden_problems["afiro"]["meta"]["pinned_initial"]
    assert 1.0 / mx == pytest.approx(g["oracle_methodical1_step_size"], rel=1e-13)
    assert np.sqrt(c2) / np.sqrt(b2) == pytest.approx(g["oracle_methodical1_primal_weight"], rel=1e-12)


def test_one_pdhg_attempt_bit_exact(tiny):
    """primal projection + SpMV A + dual projection + SpMV A^T of one attempt (pdhg.cu:72-158) on the
    unscaled problem with a fixed step: x', y', A^T y' must equal the oracle's bits."""
    p = tiny
    rng = np.random.default_rng(2)
    x0 = np.abs(rng.standard_normal(p["n"]))
    y0 = rng.standard_normal(p["m"])
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    L = orcbind.lib()
    P = orcbind._p
    offs, idx, val = (np.ascontiguousarray(p[k]) for k in ("offsets", "indices", "values"))
    w = 1.3
    for step in (0.05, 0.01, 0.002):
        dev = capi.Device(p)
        dev.call("set_initial", capi._ptr(x0), capi._ptr(y0))
        dev.call("set_step", step, w)
        dev.call("compute_aty")
        ctl = dev.run(1)
        assert ctl.attempts >= 1
        if ctl.attempts != 1:
            continue  # rejected and retried with a smaller step of the device's own choosing: try the next fixed step
        x, y = x0.copy(), y0.copy()
        L.orc_pdhg_fixed_steps(p["m"], p["n"], P(offs), P(idx), P(val), P(to), P(ti), P(tv), P(p["c"]), P(p["lo"]),
                               P(p["hi"]), P(p["lb"]), P(p["ub"]), step / w, step * w, 1, P(x), P(y))
        # accepted: the new iterate is `current`; rejected: the trial iterate sits in the other buffers
        names = ("X", "Y", "ATY") if ctl.steps_taken == 1 else ("X_OTHER", "Y_OTHER", "ATY_OTHER")
        np.testing.assert_array_equal(dev.download(names[0], p["n"]), x)
        np.testing.assert_array_equal(dev.download(names[1], p["m"]), y)
        np.testing.assert_array_equal(dev.download(names[2], p["n"]), orcbind.spmv(to, ti, tv, y))
        return
    pytest.fail("no single-attempt run to compare")


def test_convergence_information_matches_oracle(tiny, golden_problems):
    """compute_convergence_information (convergence_information.cu:149-422) on the scaled device
    problem equals the oracle's evaluation of the unscaled problem."""
    for p in (tiny, golden_problems["afiro"]["problem"], ragged_problem()):
        rng = np.random.default_rng(3)
        x = np.abs(rng.standard_normal(p["n"])) * (rng.random(p["n"]) < 0.7)
        y = rng.standard_normal(p["m"])
        y = np.where(np.isinf(p["lo"]), -np.abs(y), y)
        for rule in (True, False):
            dev = capi.Device(p)
            dev.call("scaling_compute", 1, 10, 1, 1.0)
            dev.call("scale_problem")
            dev.call("set_initial", capi._ptr(x), capi._ptr(y))
            ev = dev.eval(capi.CURRENT, rule_finite=rule, eps_p=1e-4, eps_d=1e-4)
            ref = orcbind.evaluate(p, x, y, finite_bounds_rule=rule)
            scale = 1.0 + abs(ref["primal_objective"])
            assert ev["CX"] == pytest.approx(ref["primal_objective"], rel=1e-11, abs=1e-11 * scale)
            assert ev["DUAL_SUM"] == pytest.approx(ref["dual_objective"], rel=1e-10, abs=1e-10 * scale)
            assert np.sqrt(ev["PRES2"]) == pytest.approx(ref["l2_primal_residual"], rel=1e-11, abs=1e-12)
            assert np.sqrt(ev["DRES2"]) == pytest.approx(ref["l2_dual_residual"], rel=1e-11, abs=1e-12)
            assert np.sqrt(ev["X2"]) == pytest.approx(ref["l2_x"], rel=1e-12)
            assert np.sqrt(ev["Y2"]) == pytest.approx(ref["l2_y"], rel=1e-12)
            assert ev["LINF_PRES_REL"] == pytest.approx(ref["linf_rel_primal_residual"], rel=1e-10, abs=1e-11)
            assert ev["LINF_DRES_REL"] == pytest.approx(ref["linf_rel_dual_residual"], rel=1e-10, abs=1e-11)
            xo, yo, rc = np.zeros(p["n"]), np.zeros(p["m"]), np.zeros(p["n"])
            dev.call("get_solution", capi.CURRENT, capi._ptr(xo), capi._ptr(yo), capi._ptr(rc))
            np.testing.assert_allclose(xo, x, rtol=1e-14, atol=0)
            np.testing.assert_allclose(rc, ref["reduced_cost"], rtol=1e-9, atol=1e-11)
