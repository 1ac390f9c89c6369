#!/usr/bin/env python
"""Throughput of cuoptamd_batch_solve on small LPs (the resident single-workgroup path lets independent solves share
the GPU one CU each).  GPU only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import decode_problem  # noqa: E402
from cuopt_amd import capi  # noqa: E402


def main():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "problems.json")))
    p = decode_problem(g["mip-50v-10-free-bound-relaxation"])
    p.pop("var_types", None)
    count = 128
    capi.batch_solve([p] * 4, max_threads=4, tol=1e-4, iteration_limit=20000)
    for threads in (1, 4, 16, 32, 64):
        t0 = time.perf_counter()
        rs = capi.batch_solve([p] * count, max_threads=threads, tol=1e-4, iteration_limit=20000)
        dt = time.perf_counter() - t0
        assert all(r["status_name"] == "Optimal" for r in rs)
        print("batch of %d x 50v-10 relaxation, %2d host threads: %.1f ms  (%.2f ms per LP, %d it each)" % (
            count, threads, dt * 1e3, dt * 1e3 / count, rs[0]["steps_taken"]), flush=True)


if __name__ == "__main__":
    main()
