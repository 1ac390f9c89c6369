"""GPU: the row-block sharded PDLP at world = 2, 4, 8 on ONE GPU.

Every rank is a host thread with its own solver/context/stream; the collectives go through the in-process
communicator (pdlpdev_softcomm_create) whose combine kernel sums the ranks' buffers in rank order -- the same
place in the code where RCCL's all-reduce is called in production (the RCCL call itself is exercised at
world = 1 in test_solve_gpu.py).  Oracle for the sharded run = the single-rank run."""
import threading

import numpy as np
import pytest
from conftest import set_tune

from cuopt_amd import capi, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["allreduce", "rsag", "owner"])
def dataflow(request, monkeypatch):
    """every test of this file runs through the three sharded dataflows: the replicated primal update behind ONE all-reduce(n + 1)
    per attempt, the sliced primal update (reduce-scatter of the A^T y' partials -> this rank's columns -> all-gather of xbar,
    plus one 3-scalar all-reduce: CUOPT_AMD_SHARD_DATAFLOW=rsag), and owner-computes (the rank also holds its columns of A:
    all-gather of xbar slices, all-gather of y' row blocks, complete column sums on the owner: CUOPT_AMD_SHARD_DATAFLOW=owner)"""
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", request.param)
    return request.param


def run_sharded(p, world, **kw):
    cid = capi.softcomm_id(world)
    out, err = [None] * world, []

    def worker(rank):
        try:
            s = capi.Solver(p, rank=rank, world=world, comm_id=cid, **kw)
            import os
            assert capi.lib.pdlpdev_shard_dataflow(s.device.handle) == {"allreduce": 1, "rsag": 2, "owner": 3}[os.environ["CUOPT_AMD_SHARD_DATAFLOW"]]
            r = s.advance()
            x, y, rc = s.solution()
            out[rank] = (r, x, y, s.row_range())
            s.close()
        except Exception as e:  # surface in the main thread
            err.append(e)

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not err, err
    assert all(o is not None for o in out), "a rank did not finish"
    return out


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_solve_matches_single_rank(world):
    p = synthetic.generate(6000, 5000, 8, seed=61)
    single = capi.Solver(p, tol=1e-6)
    rs = single.advance()
    xs, ys, _ = single.solution()
    out = run_sharded(p, world, tol=1e-6)
    y = np.zeros(p["m"])
    covered = 0
    for r, x, yl, (r0, r1) in out:
        # every rank takes the same decisions and holds the same replicated primal iterate
        assert (r["status_name"], r["steps_taken"], r["attempted_steps"]) == (
            out[0][0]["status_name"], out[0][0]["steps_taken"], out[0][0]["attempted_steps"])
        np.testing.assert_array_equal(x, out[0][1])
        assert r["primal_objective"] == out[0][0]["primal_objective"]
        y[r0:r1] = yl[r0:r1]
        covered += r1 - r0
    assert covered == p["m"]
    r0 = out[0][0]
    assert r0["status_name"] == rs["status_name"] == "Optimal"
    scale = 1 + abs(p["objective_star"])
    assert abs(r0["primal_objective"] - p["objective_star"]) <= 2e-5 * scale
    assert abs(r0["primal_objective"] - rs["primal_objective"]) <= 2e-5 * scale
    assert 0.5 * rs["steps_taken"] - 80 <= r0["steps_taken"] <= 2.0 * rs["steps_taken"] + 80
    # the assembled dual solution certifies the same objective: dual objective from (y, reduced costs)
    assert r0["relative_gap"] <= 1e-5


def test_sharded_first_iterations_are_identical_to_single_rank():
    """before rounding differences can build up, sharding must not change the trajectory"""
    p = synthetic.generate(3000, 3000, 10, seed=4)
    for its in (5, 40):
        a = capi.Solver(p, tol=0.0, iteration_limit=its).advance()
        b = run_sharded(p, 4, tol=0.0, iteration_limit=its)[0][0]
        assert (a["steps_taken"], a["attempted_steps"]) == (b["steps_taken"], b["attempted_steps"])
        assert b["step_size"] == pytest.approx(a["step_size"], rel=1e-9)
        assert b["primal_weight"] == pytest.approx(a["primal_weight"], rel=1e-9)
        assert b["primal_objective"] == pytest.approx(a["primal_objective"], rel=1e-9, abs=1e-9)


def test_sharded_unbalanced_rows():
    """row blocks are balanced by nonzeros, not rows: a few very long rows land in their own small blocks"""
    from test_kernels_gpu import ragged_problem
    p = ragged_problem(m=3000, n=2500)
    p["lb"] = np.zeros(p["n"])
    p["ub"] = np.full(p["n"], 5.0)
    p["lo"] = np.full(p["m"], -np.inf)
    p["hi"] = np.abs(p["hi"]) + 1.0
    p["hi"][np.isinf(p["hi"])] = 3.0
    a = capi.Solver(p, tol=1e-6).advance()
    b = run_sharded(p, 3, tol=1e-6)[0][0]
    assert a["status_name"] == b["status_name"]
    assert b["primal_objective"] == pytest.approx(a["primal_objective"], abs=2e-5 * (1 + abs(a["primal_objective"])))


def test_sharded_ranks_stop_together_on_a_time_limit():
    """wall-clock decisions are agreed across ranks (max of the elapsed times), otherwise one rank would leave while
    the others wait in the next all-reduce"""
    p = synthetic.generate(6000, 5000, 8, seed=63)
    out = run_sharded(p, 4, tol=0.0, time_limit=0.05, iteration_limit=10 ** 8)
    first = out[0][0]
    assert first["status_name"] == "TimeLimit" and first["steps_taken"] > 0
    for r, x, _, _ in out:
        assert (r["status_name"], r["steps_taken"], r["attempted_steps"]) == (first["status_name"], first["steps_taken"],
                                                                              first["attempted_steps"])
        np.testing.assert_array_equal(x, out[0][1])


# ---- the parts of the path that were single-GPU only in round 1, now under sharding (world 4, in-process communicator) ------
def test_sharded_infeasibility_detection_matches_single_rank():
    """infeasibility_information.cu:175-223 on row blocks: two maxima and one sum over the ranks"""
    from test_solve_gpu import infeasible_lp_of_the_c_api_test
    p = infeasible_lp_of_the_c_api_test()
    kw = dict(detect_infeasibility=1, tol=1e-6, iteration_limit=20000)
    single = capi.Solver(p, **kw).advance()
    assert single["status_name"] == "PrimalInfeasible"
    out = run_sharded(p, 3, **kw)
    for r, _, _, _ in out:
        assert (r["status_name"], r["steps_taken"]) == (out[0][0]["status_name"], out[0][0]["steps_taken"])
    assert out[0][0]["status_name"] == "PrimalInfeasible"
    # (the single-rank solve of this 9 x 4 LP runs in the resident kernel, the sharded one through the launches: the
    # iteration at which the certificate passes may differ, its value only in the low digits)
    for k in ("max_dual_ray_infeasibility", "dual_ray_linear_objective"):
        assert out[0][0][k] == pytest.approx(single[k], rel=1e-3, abs=1e-9)
    # and a feasible LP is not flagged
    q = synthetic.generate(3000, 2600, 9, seed=17)
    ok = run_sharded(q, 4, detect_infeasibility=1, tol=1e-5)[0][0]
    assert ok["status_name"] == "Optimal"


def test_sharded_trust_region_restart_matches_single_rank():
    """Methodical1 (pdlp_restart_strategy.cu:277-364) with the dual coordinates sharded: same decisions over the first
    major iterations, same optimum"""
    q = synthetic.generate(3000, 2600, 9, seed=17)
    for its in (64, 128):
        a = capi.Solver(q, mode=2, tol=0.0, iteration_limit=its).advance()
        b = run_sharded(q, 4, mode=2, tol=0.0, iteration_limit=its)[0][0]
        assert (a["steps_taken"], a["attempted_steps"], a["num_restarts"]) == (b["steps_taken"], b["attempted_steps"], b["num_restarts"])
        assert b["primal_weight"] == pytest.approx(a["primal_weight"], rel=1e-8)
    a = capi.Solver(q, mode=2, tol=1e-6).advance()
    b = run_sharded(q, 4, mode=2, tol=1e-6)[0][0]
    assert a["status_name"] == b["status_name"] == "Optimal"
    assert abs(b["primal_objective"] - q["objective_star"]) <= 4e-5 * (1 + abs(q["objective_star"]))
    assert 0.5 * a["steps_taken"] - 128 <= b["steps_taken"] <= 2.0 * a["steps_taken"] + 128


def test_sharded_warm_start_snapshots_add_up_and_resume():
    """pdlp.cu:468-489 / 131-181 under sharding: every rank's snapshot carries its rows of the dual-side vectors, the
    snapshots add up to the full one, and a sharded solve resumed from the sum needs exactly the remaining iterations"""
    p = synthetic.generate(4000, 3500, 8, seed=31)
    world, coarse, fine = 4, 1e-1, 1e-2
    full = run_sharded(p, world, tol=fine)[0][0]
    cid = capi.softcomm_id(world)
    snaps, res, err = [None] * world, [None] * world, []

    def first(rank):
        try:
            s = capi.Solver(p, rank=rank, world=world, comm_id=cid, tol=coarse)
            res[rank] = s.advance()
            snaps[rank] = (s.get_warm_start(), s.row_range())
            s.close()
        except Exception as e:
            err.append(e)
    ts = [threading.Thread(target=first, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    assert not err, err
    merged = dict(snaps[0][0])
    for k in capi.WarmStart.DUAL:
        total = np.zeros(p["m"])
        for ws, (r0, r1) in snaps:
            assert not np.any(ws[k][:r0]) and not np.any(ws[k][r1:])  # only its own rows
            total += ws[k]
        merged[k] = total
    for k in capi.WarmStart.PRIMAL:
        for ws, _ in snaps:
            np.testing.assert_array_equal(ws[k], snaps[0][0][k])  # replicated
    second = run_sharded(p, world, tol=fine, warm_start=merged)[0][0]
    assert full["status_name"] == res[0]["status_name"] == second["status_name"] == "Optimal"
    assert res[0]["steps_taken"] + second["steps_taken"] == full["steps_taken"]
    assert second["primal_objective"] == full["primal_objective"]


@pytest.mark.parametrize("layout", ["jag", "panel", "pb"])
def test_sharded_solve_through_the_other_layouts(layout, monkeypatch):
    """row-block sharding on top of the jagged / panel layouts (every rank builds them for ITS row block and that block's
    transpose): same decisions on all ranks, same optimum as the single-rank solve"""
    monkeypatch.setenv("CUOPT_AMD_SPMV_LAYOUT", layout)
    set_tune(monkeypatch, slab_bytes=str(32 * 1024))
    p = synthetic.generate(30000, 26000, 8, seed=71, band=900)
    single = capi.Solver(p, tol=1e-5).advance()
    out = run_sharded(p, 3, tol=1e-5)
    for r, x, _, _ in out:
        assert (r["status_name"], r["steps_taken"], r["attempted_steps"]) == (out[0][0]["status_name"], out[0][0]["steps_taken"],
                                                                              out[0][0]["attempted_steps"])
        np.testing.assert_array_equal(x, out[0][1])
    r0 = out[0][0]
    assert r0["status_name"] == single["status_name"] == "Optimal"
    scale = 1 + abs(p["objective_star"])
    assert abs(r0["primal_objective"] - p["objective_star"]) <= 2e-4 * scale
    assert abs(r0["primal_objective"] - single["primal_objective"]) <= 2e-4 * scale


def test_all_dataflows_walk_the_same_path(monkeypatch):
    """the dataflows differ only in how the three step-size sums (and, for the two that reduce partial products, the column sums)
    are grouped: same decisions over the first iterations, same optimum; and a world whose slices do not divide n (3 ranks,
    n = 5003: slices of 1680, the last one short)"""
    p = synthetic.generate(6000, 5003, 8, seed=67)
    res = {}
    for flow in ("allreduce", "rsag", "owner"):
        monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", flow)
        res[flow] = (run_sharded(p, 3, tol=0.0, iteration_limit=40), run_sharded(p, 3, tol=1e-6))
    a = res["allreduce"]
    for flow in ("rsag", "owner"):
        b = res[flow]
        assert (a[0][0][0]["steps_taken"], a[0][0][0]["attempted_steps"]) == (b[0][0][0]["steps_taken"], b[0][0][0]["attempted_steps"])
        np.testing.assert_allclose(a[0][0][1], b[0][0][1], rtol=1e-9, atol=1e-12)
        for rank in range(3):
            np.testing.assert_array_equal(b[0][rank][1], b[0][0][1])  # replicated again outside the loop
        assert a[1][0][0]["status_name"] == b[1][0][0]["status_name"] == "Optimal"
        assert abs(a[1][0][0]["primal_objective"] - b[1][0][0]["primal_objective"]) <= 2e-5 * (1 + abs(p["objective_star"]))
