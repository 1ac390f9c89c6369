"""GPU: the row-block sharded PDLP at world = 2, 4, 8 on ONE GPU.

Every rank is a host thread with its own solver/context/stream; the collectives go through the in-process
communicator (pdlpdev_softcomm_create) whose combine kernel sums the ranks' buffers in rank order -- the same
place in the code where RCCL's all-reduce is called in production (the RCCL call itself is exercised at
world = 1 in test_solve_gpu.py).  Oracle for the sharded run = the single-rank run."""
import threading

import numpy as np
import pytest

from cuopt_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def run_sharded(p, world, **kw):
    cid = capi.softcomm_id(world)
    out, err = [None] * world, []

    def worker(rank):
        try:
            s = capi.Solver(p, rank=rank, world=world, comm_id=cid, **kw)
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
