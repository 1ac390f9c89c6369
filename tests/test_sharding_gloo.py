"""CPU, 2 processes over gloo: the row-block sharding math of the multi-GPU path.

The product's collectives run on RCCL inside libcuopt.so and need GPUs; what can be pinned on CPU is
the decomposition itself, with the C oracle standing in for the per-rank kernels:
  * cuoptamd_partition_rows + per-rank CSR slice + per-rank explicit transpose (product host code),
  * A x is rank-local given a replicated x; A^T y = sum over ranks of (A_block)^T y_block  -> ONE sum
    all-reduce of n doubles; column inf-norms (Ruiz) -> max all-reduce; dual-side dot products -> sum,
  * the all-reduced quantities equal the unsharded ones (exactly for max, to rounding for sums)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cuopt_amd import capi, synthetic
    from oracle import orcbind
    p = synthetic.generate(3001, 2500, 7, seed=21)
    m, n = p["m"], p["n"]
    bounds = capi.partition_rows(m, p["offsets"], world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    k0, k1 = int(p["offsets"][r0]), int(p["offsets"][r1])
    off = (p["offsets"][r0:r1 + 1] - k0).astype(np.int32)
    idx, val = p["indices"][k0:k1], p["values"][k0:k1]
    to, ti, tv = capi.csr_transpose(r1 - r0, n, off, idx, val)
    rng = np.random.default_rng(5)  # same on every rank: replicated primal side
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    # local A x (no communication)
    ax_local = orcbind.spmv(off, idx, val, x)
    # partial A^T y + all-reduce
    aty = torch.from_numpy(orcbind.spmv(to, ti, tv, y[r0:r1]))
    dist.all_reduce(aty, op=dist.ReduceOp.SUM)
    # Ruiz column inf-norm: local max over the row block + max all-reduce
    colmax = np.zeros(n)
    np.maximum.at(colmax, idx, np.abs(val))
    colmax = torch.from_numpy(colmax)
    dist.all_reduce(colmax, op=dist.ReduceOp.MAX)
    # a dual-side dot product
    dy2 = torch.tensor([float(y[r0:r1] @ y[r0:r1])], dtype=torch.float64)
    dist.all_reduce(dy2, op=dist.ReduceOp.SUM)
    # the SLICED primal dataflow (CUOPT_AMD_SHARD_DATAFLOW=rsag, pdlp_device.hip enqueue_attempt): equal slices of a multiple of 16
    # columns per rank over a padded buffer; reduce-scatter of the A^T y partials -> primal step on the slice -> all-gather
    per = (n + world - 1) // world
    sl = (per + 15) & ~15
    padded = np.zeros(sl * world)
    padded[:n] = orcbind.spmv(to, ti, tv, y[r0:r1])
    mine = None
    for q in range(world):  # ncclReduceScatter: chunk q is reduced onto rank q
        chunk = torch.from_numpy(padded[q * sl:(q + 1) * sl].copy())
        dist.reduce(chunk, dst=q, op=dist.ReduceOp.SUM)
        if q == rank:
            mine = chunk.numpy()
    cs = rank * sl
    ln = max(0, min(sl, n - cs))
    tau = 0.37
    x_new = np.maximum(x[cs:cs + ln] - tau * (p["c"][cs:cs + ln] - mine[:ln]), 0.0)
    xbar_slice = np.zeros(sl)
    xbar_slice[:ln] = 2.0 * x_new - x[cs:cs + ln]
    gathered = [torch.zeros(sl, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(xbar_slice))  # ncclAllGather
    xbar = np.concatenate([g.numpy() for g in gathered])[:n]
    sums = torch.tensor([float(y[r0:r1] @ y[r0:r1]), float(x_new @ mine[:ln]), float(x_new @ x_new)], dtype=torch.float64)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)  # the 3-scalar all-reduce: identical bits on every rank
    # OWNER-COMPUTES dataflow (CUOPT_AMD_SHARD_DATAFLOW=owner, pdlpdev_owner_setup): the rank also holds its slice of COLUMNS of A over
    # all rows (rows of the global A^T, row indices remapped into the gathered dual: rank q's rows at [q * ypad, ...)); after an
    # all-gather of the row blocks' y the column sums are complete on the owner -- and bit-identical to the unsharded A^T y,
    # because every column adds its rows in the same (ascending) order
    tof_g, tif_g, tvf_g = orcbind.transpose(m, n, p["offsets"], p["indices"], p["values"])
    c0, c1 = int(tof_g[cs]) if ln else 0, int(tof_g[cs + ln]) if ln else 0
    coff = (tof_g[cs:cs + ln + 1] - c0).astype(np.int32) if ln else np.zeros(1, np.int32)
    cidx, cval = tif_g[c0:c1], tvf_g[c0:c1]
    ypad = (int(np.max(np.diff(bounds))) + 15) & ~15
    owner_of = np.searchsorted(np.asarray(bounds[1:]), cidx, side="right")
    ridx = (cidx + owner_of * ypad - np.asarray(bounds)[owner_of]).astype(np.int32)
    yslot = np.zeros(ypad)
    yslot[:r1 - r0] = y[r0:r1]
    ys = [torch.zeros(ypad, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(ys, torch.from_numpy(yslot))  # ncclAllGather of the y' row blocks
    ygather = np.concatenate([g.numpy() for g in ys])
    aty_owner = orcbind.spmv(coff, ridx, cval, ygather) if ln else np.zeros(0)
    owner_pieces = [None] * world
    dist.all_gather_object(owner_pieces, (cs, aty_owner))
    # gather A x pieces for the check
    pieces = [None] * world
    dist.all_gather_object(pieces, (r0, r1, ax_local))
    if rank == 0:
        tof, tif, tvf = orcbind.transpose(m, n, p["offsets"], p["indices"], p["values"])
        ax = np.concatenate([q[2] for q in sorted(pieces, key=lambda q: q[0])])
        cm = np.zeros(n)
        np.maximum.at(cm, p["indices"], np.abs(p["values"]))
        out.put(dict(bounds=bounds.tolist(),
                     ax_equal=bool(np.array_equal(ax, orcbind.spmv(p["offsets"], p["indices"], p["values"], x))),
                     aty_err=float(np.max(np.abs(aty.numpy() - orcbind.spmv(tof, tif, tvf, y)))),
                     colmax_equal=bool(np.array_equal(colmax.numpy(), cm)),
                     dy2_err=float(abs(dy2.item() - y @ y)), nnz=[int(p["offsets"][b]) for b in bounds],
                     owner_aty_equal=bool(np.array_equal(np.concatenate([q[1] for q in sorted(owner_pieces, key=lambda q: q[0])]),
                                                         orcbind.spmv(tof, tif, tvf, y))), ypad=ypad,
                     slice=sl, xbar_err=float(np.max(np.abs(xbar - (2.0 * np.maximum(x - tau * (p["c"] - orcbind.spmv(tof, tif, tvf, y)), 0.0) - x)))),
                     sums_err=float(abs(sums[0].item() - y @ y)),
                     dx2_err=float(abs(sums[2].item() - np.sum(np.maximum(x - tau * (p["c"] - orcbind.spmv(tof, tif, tvf, y)), 0.0) ** 2)))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_row_block_sharding_reproduces_unsharded_products(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for q in procs:
        q.start()
    res = out.get(timeout=120)
    for q in procs:
        q.join(timeout=60)
        assert q.exitcode == 0
    assert res["bounds"][0] == 0 and res["bounds"][-1] == 3001
    per = np.diff(res["nnz"])
    assert per.max() - per.min() <= 7  # balanced by nonzeros to within one row
    assert res["ax_equal"] and res["colmax_equal"]
    assert res["aty_err"] < 1e-12 and res["dy2_err"] < 1e-9
    # sliced primal dataflow: 2500 columns over 2 ranks = slices of 1264 (28 entries of padding behind the last one)
    assert res["slice"] == 1264
    assert res["xbar_err"] < 1e-12 and res["sums_err"] < 1e-9 and res["dx2_err"] < 1e-9
    # owner-computes dataflow: complete column sums on the owner, BIT-identical to the unsharded A^T y
    assert res["owner_aty_equal"] and res["ypad"] % 16 == 0
