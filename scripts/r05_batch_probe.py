"""Shared-matrix batch on C3: aggregate PDLP iterations/s of K LPs in lockstep against K single solves one after the other.
usage: python scripts/r05_batch_probe.py [K ...]   (default 2 4 8 16)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cuopt_amd import capi, synthetic  # noqa: E402


def load():
    base = os.environ.get("BATCH_PROBE_BASE", "c3")
    cache = os.environ.get("CUOPT_AMD_LP_CACHE")
    f = os.path.join(cache, "%s.npz" % base) if cache else None
    if f and os.path.exists(f):
        z = np.load(f, allow_pickle=False)
        return {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    p = synthetic.generate(**synthetic.CONFIGS[base])
    if f:
        os.makedirs(cache, exist_ok=True)
        np.savez(f, **{k: v for k, v in p.items() if isinstance(v, (np.ndarray, int, float, bool, np.integer, np.floating))})
    return p


def main():
    ks = [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16]
    p = load()
    rng = np.random.default_rng(8)
    x = p["x_star"]

    def bounds(l):
        lb, ub = np.array(p["lb"], float), np.array(p["ub"], float)
        if l:
            for j in rng.choice(p["n"], size=p["n"] // 10, replace=False):
                ub[j] = x[j] + 0.3 * rng.random()
        return lb, ub
    steps = 2000
    graph = int(os.environ.get("BATCH_PROBE_GRAPH", "1"))  # (0: plain launches -- rocprofv3 7.2 falls over the 1216-node replay graphs)
    parent = capi.Solver(p, tol=0.0, use_graph=graph)
    parent.device.call("prepare_graphs")
    parent.advance(400)
    parent.device.call("synchronize")
    t0 = time.perf_counter()
    parent.advance(steps)
    parent.device.call("synchronize")
    single = steps / (time.perf_counter() - t0)
    print("RATE single %.1f it/s" % single, flush=True)
    for k in ks:
        parent.reset(tol=0.0, use_graph=graph)
        sets = [bounds(l) for l in range(1, k)]
        t0 = time.perf_counter()
        clones = [parent.clone(lb, ub) for lb, ub in sets]
        t_clone = (time.perf_counter() - t0) / max(k - 1, 1)
        batch = capi.SharedMatrixBatch([parent] + clones)
        batch.advance(400)
        parent.device.call("synchronize")
        t0 = time.perf_counter()
        r = batch.advance(steps)
        parent.device.call("synchronize")
        dt = time.perf_counter() - t0
        assert all(q["steps_taken"] == 400 + steps for q in r), [q["steps_taken"] for q in r]
        print("RATE batch K=%d: %.1f it/s aggregate (%.2fx of the single solve's), %.3f ms per lockstep iteration, clone %.1f ms each"
              % (k, k * steps / dt, k * steps / dt / single, 1e3 * dt / steps, 1e3 * t_clone), flush=True)
        batch.close()
        for c in clones:
            c.close()
    parent.close()


if __name__ == "__main__":
    main()
