#!/usr/bin/env python
"""Times the raw PDHG attempt loop (pdlpdev_run, no major iterations) and the full solver advance on a small LP,
resident single-workgroup loop vs multi-launch/hipGraph path.  GPU only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from conftest import decode_problem  # noqa: E402
from cuopt_amd import capi, synthetic  # noqa: E402


def main():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "problems.json")))
    cases = {}
    for name in ("afiro", "mip-50v-10-free-bound-relaxation", "mip-neos5-free-bound-relaxation"):
        p = decode_problem(g[name])
        p.pop("var_types", None)
        cases[name] = p
    cases["synthetic-1000x1000x8"] = synthetic.generate(1000, 1000, 8, seed=4)
    cases["synthetic-2000x2000x2"] = synthetic.generate(2000, 2000, 2, seed=4)
    for name, p in cases.items():
        for small in ("1", "0"):
            os.environ["CUOPT_AMD_SMALL"] = small
            dev = capi.Device(p)
            dev.call("set_step", 1e-3, 1.0)
            dev.call("compute_aty")
            dev.run(200)
            t0 = time.perf_counter()
            ctl = dev.run(200 + 4000)
            raw = (time.perf_counter() - t0) / 4000
            s = capi.Solver(p, tol=0.0)
            s.advance(400)
            t0 = time.perf_counter()
            s.advance(4000)
            full = (time.perf_counter() - t0) / 4000
            s.close()
            capi.solve(p, method=1, tol=1e-4)
            t0 = time.perf_counter()
            r = capi.solve(p, method=1, tol=1e-4)
            e2e = time.perf_counter() - t0
            print("%-36s m=%5d n=%5d nnz=%6d resident=%s raw %.2f us/attempt  solver %.2f us/it  cuOptSolve(1e-4) %.2f ms "
                  "(%d its, %.2f ms in the solver)" % (name, p["m"], p["n"], len(p["values"]), small, raw * 1e6, full * 1e6,
                                                     e2e * 1e3, r["steps_taken"], r["solve_time"] * 1e3), flush=True)


if __name__ == "__main__":
    main()
