"""GPU: the long-tail variant of the slab-major panels (spmv_panel.hpp: panel_seg_block) -- row sums dealt by NONZERO (per round a
segmented scan over the wave's 64 consecutive entries, edge runs joined by wave 0 once per chunk) instead of by row.  Its contract is the one every layout has for
rows of more than 128 nonzeros: EVERY row within rtol 1e-12 of the oracle's left-to-right CSR sum (the additions of a row are a
fixed tree, run-to-run reproducible, but no longer sequential across lanes), and the PDLP decisions of the row-per-lane panels over
the first major iterations.  `auto` takes it only where more than 2 % of the nonzeros sit in rows of more than 128 entries."""
import numpy as np
import pytest
from conftest import set_tune

from cuopt_amd import capi, synthetic
from oracle import orcbind
from test_kernels_gpu import ragged_problem

pytestmark = pytest.mark.gpu


def _long_runs(m=400, n=3000, k=700, seed=3):
    """every row holds k of the n columns: runs of hundreds of entries, whole lanes and waves inside one run"""
    rng = np.random.default_rng(seed)
    idx = np.concatenate([np.sort(rng.choice(n, size=k, replace=False)) for _ in range(m)]).astype(np.int32)
    off = (np.arange(m + 1) * k).astype(np.int32)
    return dict(m=m, n=n, offsets=off, indices=idx, values=rng.standard_normal(m * k), c=rng.standard_normal(n),
                lo=np.full(m, -np.inf), hi=rng.standard_normal(m) + 30.0, lb=np.zeros(n), ub=np.full(n, 5.0))


def _check_spmv(p, tag, repeat=False):
    dev = capi.Device(p)
    lay = dev.layout()
    assert lay["A"]["layout"] == lay["At"]["layout"] == "panel", (tag, lay)
    assert lay["A"]["row_sums"] == lay["At"]["row_sums"] == "by_nonzero", (tag, lay)
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    for vec, tr, rows, ref in ((x, False, p["m"], orcbind.spmv(p["offsets"], p["indices"], p["values"], x)),
                               (y, True, p["n"], orcbind.spmv(to, ti, tv, y))):
        got = dev.spmv(vec, tr, rows)
        scale = 1.0 + np.abs(ref).max()
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * scale, err_msg="%s transpose=%s" % (tag, tr))
        if repeat:  # a fixed tree: the same bits on every launch
            np.testing.assert_array_equal(dev.spmv(vec, tr, rows), got)
    dev.close()


@pytest.mark.parametrize("panel_nnz,slab", [(2048, 4096), (5000, 16 * 1024), (12000, 64 * 1024), (60000, 1 << 20)],
                         ids=["R1-4-many-slabs", "R-mixed", "R-up-to-8", "one-slab-full-chunks"])
def test_every_row_within_rtol_of_the_oracle(panel_nnz, slab, monkeypatch):
    """chunk lengths from a handful of entries (rounds R = 1) to full 4096-entry chunks (R = 8), runs that cross lanes, waves, chunks
    and slabs, empty rows, rows longer than a chunk"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    monkeypatch.setenv("CUOPT_AMD_SMALL", "0")
    set_tune(monkeypatch, panel_seg=1, panel_nnz=panel_nnz, slab_bytes=slab)
    _check_spmv(synthetic.generate(3000, 2600, 9, seed=12), "random", repeat=True)
    p = ragged_problem(m=3000, n=2500)
    p["lb"] = np.zeros(p["n"])
    _check_spmv(p, "ragged")
    # one dense-ish row block: runs of hundreds of entries side by side (every lane of several waves inside one run)
    q = _long_runs()
    _check_spmv(q, "long-runs", repeat=True)


def test_power_law_rows_with_hub_rows_of_their_own(monkeypatch):
    """`auto` on a long-tailed matrix: panels with row sums by nonzero on the A side; rows beyond 4096 nonzeros keep their own
    workgroups behind the panels"""
    monkeypatch.delenv("CUOPT_AMD_SPMV_LAYOUT", raising=False)
    set_tune(monkeypatch, slab_bytes=64 * 1024, panel_ws_bytes=256 * 1024)
    p = synthetic.generate_structured("powerlaw", m=200000, n=200000, k=10, seed=11)
    lens = np.diff(p["offsets"])
    assert (lens > 4096).sum() >= 2 and lens[lens > 128].sum() > 0.02 * lens.sum()
    dev = capi.Device(p)
    lay = dev.layout()
    assert lay["A"]["layout"] == "panel" and lay["A"]["row_sums"] == "by_nonzero", lay
    assert lay["At"].get("row_sums", "by_row") == "by_row", lay  # the columns of this matrix are short: bit-exact kernels
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(p["n"]), rng.standard_normal(p["m"])
    to, ti, tv = orcbind.transpose(p["m"], p["n"], p["offsets"], p["indices"], p["values"])
    ref = orcbind.spmv(p["offsets"], p["indices"], p["values"], x)
    np.testing.assert_allclose(dev.spmv(x, False, p["m"]), ref, rtol=1e-12, atol=1e-12 * (1 + np.abs(ref).max()))
    tlens = np.diff(to)
    got_t, ref_t = dev.spmv(y, True, p["n"]), orcbind.spmv(to, ti, tv, y)
    np.testing.assert_array_equal(got_t[tlens <= 128], ref_t[tlens <= 128])
    dev.close()
    # a uniform matrix keeps the row-per-lane panels
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    u = synthetic.generate(200000, 200000, 10, seed=3)
    lay = capi.Device(u).layout()
    assert lay["A"]["layout"] == "panel" and lay["A"]["row_sums"] == "by_row", lay


@pytest.mark.parametrize("kind", ["random", "powerlaw"])
def test_decisions_of_the_row_per_lane_panels(kind, monkeypatch):
    """the first two major iterations (80 steps): same accepted / attempted counts as the row-per-lane panels and the oracle, step
    size and objective to 1e-9"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    monkeypatch.setenv("CUOPT_AMD_SMALL", "0")
    p = synthetic.generate(3000, 3000, 10, seed=4) if kind == "random" else synthetic.generate_structured("powerlaw", m=30000, n=30000, k=8, seed=5)
    out = {}
    for seg in (0, 1):
        set_tune(monkeypatch, panel_seg=seg, slab_bytes=16 * 1024, panel_nnz=9000)
        out[seg] = capi.Solver(p, tol=0.0, iteration_limit=80).advance()
    o = orcbind.solve(p, tol=0.0, iteration_limit=80)
    for seg in (0, 1):
        r = out[seg]
        assert (r["steps_taken"], r["attempted_steps"]) == (int(o["steps_taken"]), int(o["attempted_steps"])), seg
        assert r["step_size"] == pytest.approx(o["final_step_size"], rel=1e-9)
        assert r["primal_objective"] == pytest.approx(o["primal_objective"], rel=1e-9, abs=1e-9)


def test_solve_to_tolerance_through_the_long_tail_panels(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    set_tune(monkeypatch, panel_seg=1, slab_bytes=32 * 1024)
    p = synthetic.generate_structured("powerlaw", m=40000, n=40000, k=8, seed=7)
    r = capi.solve(p, method=1, tol=1e-6)
    assert r["status"] == "Optimal"
    assert abs(r["objective"] - p["objective_star"]) <= 2e-5 * (1 + abs(p["objective_star"]))


def test_sharded_solve_through_the_long_tail_panels(monkeypatch):
    """the row blocks and the column blocks of a sharded solve (owner-computes dataflow, in-process communicator) through panels
    whose row sums are dealt by nonzero: same optimum as one GPU.  (The peer transport needs a process of its own -- hardware queues
    are fixed at HIP start-up: tests/test_p2p_transport_gpu.py)"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", "panel")
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", "owner")
    monkeypatch.setenv("CUOPT_AMD_SHARD_TRANSPORT", "collective")
    set_tune(monkeypatch, panel_seg=1, slab_bytes=32 * 1024, soft_communicator=1)
    p = synthetic.generate_structured("powerlaw", m=40000, n=40000, k=8, seed=9)
    single = capi.solve(p, method=1, tol=1e-6)
    r = capi.solve(p, method=1, tol=1e-6, amd_num_gpus=3)
    assert r["status"] == "Optimal" and r["gpus"] == 3
    scale = 1 + abs(p["objective_star"])
    assert abs(r["objective"] - p["objective_star"]) <= 2e-5 * scale
    assert abs(r["objective"] - single["objective"]) <= 2e-5 * scale
