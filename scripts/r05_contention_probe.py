"""Several of these processes side by side on one GPU: do single solves and lockstep batches repeat themselves bit for bit while other
processes' kernels are interleaved with theirs?  (scripts/gpu_session.sh ... contention)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("CUOPT_AMD_SPMV_LAYOUT", "panel")
from cuopt_amd import capi, synthetic  # noqa: E402
from test_shared_batch_gpu import KEYS_F64, KEYS_INT, LIMIT, variants  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = synthetic.generate(30000, 24000, 10, seed=21)
bounds = variants(p, k)


def key(r, sol):
    return tuple(r[q] for q in KEYS_INT + KEYS_F64) + tuple(hash(v.tobytes()) for v in sol)


def singles():
    out = []
    for lb, ub in bounds:
        s = capi.Solver(dict(p, lb=lb, ub=ub), tol=1e-5, iteration_limit=LIMIT)
        r = s.advance()
        out.append(key(r, s.solution()))
        s.close()
    return out


def batch():
    parent = capi.Solver(dict(p, lb=bounds[0][0], ub=bounds[0][1]), tol=1e-5, iteration_limit=LIMIT)
    solvers = [parent] + [parent.clone(lb=lb, ub=ub) for lb, ub in bounds[1:]]
    b = capi.SharedMatrixBatch(solvers)
    b.advance(130)
    got = b.advance()
    out = [key(got[l], solvers[l].solution()) for l in range(k)]
    b.close()
    for s in solvers[1:]:
        s.close()
    parent.close()
    return out


ref = singles()
for r in range(rounds):
    s2, b2 = singles(), batch()
    print("pid %d round %d: singles repeat %s | batch equals singles %s | steps single %s batch %s" % (
        os.getpid(), r, [a == b for a, b in zip(ref, s2)], [a == b for a, b in zip(ref, b2)], [a[1] for a in ref], [b[1] for b in b2]), flush=True)
