"""GPU: halo exchange of the owner-computes dataflow (pdlp_ctx.hpp Halo; north_star's sharding is the builder's addition to the
reference).  On a structured LP a rank's rows reference, outside its own slice of xbar, only the edges of its neighbours'
slices: per peer one contiguous range travels instead of the all-gather.  Ranks = contexts on ONE device behind the in-process
communicator (the RCCL path sends the same ranges through ncclSend / ncclRecv): iterates bit-identical to the all-gather's, the
bytes on the wire asserted, a random LP keeps its all-gathers, and a shuffled band gets both the set-up's reordering and the halo."""
import threading

import numpy as np
import pytest

from cuopt_amd import capi, synthetic
from conftest import set_tune

pytestmark = pytest.mark.gpu


def run_ranks(p, world, **kw):
    cid = capi.softcomm_id(world)
    out, err = [None] * world, []

    def worker(rank):
        try:
            s = capi.Solver(p, rank=rank, world=world, comm_id=cid, **kw)
            wire = s.device.wire_bytes()
            r = s.advance()
            x, y, rc = s.solution()
            out[rank] = (r, x, y, wire, s.reorder_info()["reordered"])
            s.close()
        except Exception as e:  # surface in the main thread
            err.append(e)

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not err, err
    assert all(o is not None for o in out), "a rank did not finish"
    return out


@pytest.fixture(scope="module")
def band():
    return synthetic.generate(262144, 262144, 10, seed=2, band=500)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_halo_exchange_is_the_all_gather_bit_for_bit(band, world, monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", "owner")
    set_tune(monkeypatch, shard_halo=0)
    full = run_ranks(band, world, tol=0.0, iteration_limit=120)
    set_tune(monkeypatch, shard_halo=None)
    halo = run_ranks(band, world, tol=0.0, iteration_limit=120)
    for (rf, xf, yf, wf, _), (rh, xh, yh, wh, _) in zip(full, halo):
        assert not wf["halo"] and wh["halo"], (wf, wh)
        assert (rf["steps_taken"], rf["attempted_steps"]) == (rh["steps_taken"], rh["attempted_steps"])
        assert rf["primal_objective"] == rh["primal_objective"] and rf["step_size"] == rh["step_size"]
        np.testing.assert_array_equal(xf, xh)
        np.testing.assert_array_equal(yf, yh)
        # bytes on the wire per attempt: two ranges of ~2 * 500 doubles per neighbour against (world - 1) / world of n + m doubles
        assert wh["bytes_allgather"] == wf["bytes"] >= 8 * (world - 1) * (band["n"] // world + band["m"] // world)
        assert wh["bytes"] <= 0.01 * wh["bytes_allgather"] * (world - 1), wh  # (<= 1 % per peer pair; at world 8: 0.3 % in all)
    if world == 8:
        assert max(o[3]["bytes"] for o in halo) <= 0.01 * halo[0][3]["bytes_allgather"]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_halo_through_the_peer_store_transport(band, world, monkeypatch):
    """(round 6) the direct peer transport carries the halo too: a producing kernel stores into a peer's landing block only what that
    peer's rows / columns reference, the consumer copies only those ranges -- bit for bit the all-gathers' iterates, the halo's bytes"""
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", "owner")
    set_tune(monkeypatch, shard_halo=0)
    full = run_ranks(band, world, tol=0.0, iteration_limit=120)
    monkeypatch.setenv("CUOPT_AMD_SHARD_TRANSPORT", "p2p")
    p2p_full = run_ranks(band, world, tol=0.0, iteration_limit=120)
    set_tune(monkeypatch, shard_halo=None)
    p2p_halo = run_ranks(band, world, tol=0.0, iteration_limit=120)
    for (rf, xf, yf, wf, _), (rp, xp, yp, wp, _), (rh, xh, yh, wh, _) in zip(full, p2p_full, p2p_halo):
        assert not wf["halo"] and not wp["halo"] and wh["halo"], (wf, wp, wh)
        assert wh["bytes"] <= 0.01 * wh["bytes_allgather"] * (world - 1), wh
        for r, x, y in ((rp, xp, yp), (rh, xh, yh)):
            assert (rf["steps_taken"], rf["attempted_steps"]) == (r["steps_taken"], r["attempted_steps"])
            assert rf["primal_objective"] == r["primal_objective"] and rf["step_size"] == r["step_size"]
            np.testing.assert_array_equal(xf, x)
            np.testing.assert_array_equal(yf, y)

def test_a_random_lp_keeps_its_all_gathers(monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", "owner")
    p = synthetic.generate(60000, 50000, 8, seed=3)
    out = run_ranks(p, 4, tol=1e-4)
    assert all(not o[3]["halo"] for o in out)
    assert all(o[0]["status_name"] == "Optimal" for o in out)


def test_a_shuffled_band_is_reordered_and_then_exchanged_by_halo(band, monkeypatch):
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", "owner")
    q = synthetic.shuffled(band, seed=4)
    out = run_ranks(q, 4, tol=1e-5)
    assert all(o[4] for o in out), "every rank found the order"
    assert all(o[3]["halo"] for o in out), [o[3] for o in out]
    r0 = out[0][0]
    assert r0["status_name"] == "Optimal"
    assert abs(r0["primal_objective"] - q["objective_star"]) <= 2e-4 * (1 + abs(q["objective_star"]))
    for o in out:
        np.testing.assert_array_equal(o[1], out[0][1])  # the replicated primal side, in the caller's order on every rank
