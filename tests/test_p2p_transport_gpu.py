"""GPU: the direct-peer transport of the owner-computes dataflow (CUOPT_AMD_SHARD_TRANSPORT=p2p) and the failure handling of
sharded solves.

On this one-GPU box the ranks are contexts of one process on the SAME device (in-process communicator): the push / pull kernels,
their epoch flags and the landing blocks are the real ones, what a multi-GPU node adds is only that the blocks are peer-mapped
(hipDeviceEnablePeerAccess / HIP IPC handles exchanged over RCCL: test_rccl_two_ranks_in_one_process covers that path whenever
two devices are visible).  The ranks' streams must run concurrently (a pull spins until the other ranks' pushes land), so the
solves run in a child process with enough hardware queues for one stream per rank."""
import json
import os
import subprocess
import sys

import pytest
from conftest import set_tune

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, threading
import numpy as np
sys.path.insert(0, %(root)r)
from cuopt_amd import capi, synthetic
world, its, tol = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
p = synthetic.generate(6000, 5003, 8, seed=67)
cid = capi.softcomm_id(world)
out, err = [None] * world, []
def worker(rank):
    try:
        s = capi.Solver(p, rank=rank, world=world, comm_id=cid, tol=tol, iteration_limit=its)
        flow, transport = capi.lib.pdlpdev_shard_dataflow(s.device.handle), capi.lib.pdlpdev_shard_transport(s.device.handle)
        r = s.advance()
        x, y, rc = s.solution()
        out[rank] = dict(status=r["status_name"], steps=r["steps_taken"], attempts=r["attempted_steps"], obj=r["primal_objective"],
                         step_size=r["step_size"], x=np.asarray(x).tobytes().hex()[:4096], flow=flow, transport=transport)
        s.close()
    except Exception as e:
        err.append(repr(e))
ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
[t.start() for t in ts]
[t.join(timeout=240) for t in ts]
print(json.dumps(dict(out=out, err=err)))
"""


def run_child(world, its, tol, transport):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", CUOPT_AMD_SHARD_DATAFLOW="owner", CUOPT_AMD_SHARD_TRANSPORT=transport)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), str(world), str(its), str(tol)], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert not d["err"], d["err"]
    return d["out"]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_direct_peer_transport_walks_the_path_of_the_collectives(world):
    """same decisions, same iterates (bit for bit: both transports add the ranks' three step-size sums in rank order), at a
    fixed budget and to the optimum; slices that do not divide n"""
    for its, tol in ((40, 0.0), (100000, 1e-6)):
        a = run_child(world, its, tol, "collective")
        b = run_child(world, its, tol, "p2p")
        assert all(o["flow"] == 3 for o in a + b)
        assert all(o["transport"] == 0 for o in a) and all(o["transport"] == 1 for o in b)
        for o in b:
            assert (o["status"], o["steps"], o["attempts"], o["obj"], o["step_size"], o["x"]) == (
                a[0]["status"], a[0]["steps"], a[0]["attempts"], a[0]["obj"], a[0]["step_size"], a[0]["x"])
    assert b[0]["status"] == "Optimal"


@pytest.mark.timeout(300)
@pytest.mark.parametrize("where", ["create", "advance"])
@pytest.mark.parametrize("dataflow", ["allreduce", "owner"])
def test_a_failing_rank_ends_the_sharded_solve_instead_of_hanging_it(where, dataflow, monkeypatch):
    """one rank of cuOptSolve's multi-GPU path fails during set-up (before the communicator exists) or in the middle of the
    solve: the other ranks must not wait for it in a collective -- the call returns CUOPT_RUNTIME_ERROR naming the rank"""
    from cuopt_amd import capi, synthetic
    p = synthetic.generate(3000, 2600, 8, seed=5)
    set_tune(monkeypatch, soft_communicator="1")
    monkeypatch.setenv("CUOPT_AMD_SHARD_DATAFLOW", dataflow)
    set_tune(monkeypatch, fault_inject="1:" + where)
    r = capi.solve(p, method=1, tol=1e-8, amd_num_gpus=3)
    assert r["return_code"] == capi.CUOPT_RUNTIME_ERROR
    assert "rank 1 of 3" in r["error_string"] and "injected fault" in r["error_string"]
    set_tune(monkeypatch, fault_inject=None)
    ok = capi.solve(p, method=1, tol=1e-6, amd_num_gpus=3)  # and the library is usable afterwards
    assert ok["status"] == "Optimal"


FAULT_CHILD = r"""
import json, sys
sys.path.insert(0, %(root)r)
from cuopt_amd import capi, synthetic
p = synthetic.generate(3000, 2600, 8, seed=5)
r = capi.solve(p, method=1, tol=1e-8, amd_num_gpus=3)
print(json.dumps(dict(rc=r["return_code"], err=r.get("error_string", ""))))
"""


@pytest.mark.timeout(300)
def test_a_rank_that_dies_under_the_peer_transport_ends_the_solve(monkeypatch):
    """with direct peer stores nobody sits in a collective: the surviving ranks' device-side waits run out of patience (5 s), the
    solve ends in CUOPT_RUNTIME_ERROR naming the rank that failed first-hand"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", CUOPT_AMD_SHARD_DATAFLOW="owner", CUOPT_AMD_SHARD_TRANSPORT="p2p",
               CUOPT_AMD_TUNE="soft_communicator=1,fault_inject=1:advance")
    r = subprocess.run([sys.executable, "-c", FAULT_CHILD % dict(root=ROOT)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    from cuopt_amd import capi
    assert d["rc"] == capi.CUOPT_RUNTIME_ERROR and "rank 1 of 3" in d["err"] and "injected fault" in d["err"]


def test_landing_block_across_two_processes():
    """(round-5 review, item 5b) what only a SECOND PROCESS exercises of the peer transport: the IPC handle of a fine-grained landing
    block, hipIpcOpenMemHandle with lazy peer access in the other process, its system-scope stores and release flag, this process's
    flag wait -- on one device, without a communicator (multi-rank RCCL refuses duplicate devices)"""
    import ctypes as C
    import os
    import subprocess
    import sys
    from cuopt_amd import capi
    count, seed = 100_000, 0.25
    handle = (C.c_uint8 * 64)()
    base = C.c_void_p()
    rc = capi.lib.pdlpdev_debug_ipc_export(0, count, handle, C.byref(base))
    assert rc == 0, capi.lib.pdlpdev_last_error().decode()
    child = ("import sys, ctypes as C; sys.path.insert(0, %r); from cuopt_amd import capi; h = (C.c_uint8 * 64).from_buffer_copy(bytes.fromhex(sys.argv[1])); "
             "rc = capi.lib.pdlpdev_debug_ipc_store(0, h, %d, %r); print('store', rc, capi.lib.pdlpdev_last_error().decode() if rc else ''); sys.exit(0 if rc == 0 else 3)"
             % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), count, seed))
    r = subprocess.run([sys.executable, "-c", child, bytes(handle).hex()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    wrong = capi.lib.pdlpdev_debug_ipc_wait(0, base, count, seed)
    assert wrong == 0, (wrong, capi.lib.pdlpdev_last_error().decode())
