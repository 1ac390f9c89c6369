#!/usr/bin/env python
"""Soak: many back-to-back cuOptSolve calls (recycled streams / arena chunks / host pool) -- time per solve and
resident memory must not drift.  GPU only."""
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import decode_problem  # noqa: E402
from cuopt_amd import capi, synthetic  # noqa: E402


def main():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "problems.json")))
    p = decode_problem(g["mip-50v-10-free-bound-relaxation"])
    p.pop("var_types", None)
    mid = synthetic.generate(20000, 20000, 10, seed=3)
    def threads():
        for line in open("/proc/self/status"):
            if line.startswith("Threads:"):
                return int(line.split()[1])
        return -1

    # method 1: PDLP alone; method 0: the Concurrent race -- the simplex thread (with its presolve and, from 2000 rows on, its helper
    # thread) is started and, on the larger LP, cancelled in every call: threads and memory must come back
    for name, prob, count, method in (("50v-10 relaxation", p, 600, 1), ("synthetic 2e4 x 2e4", mid, 150, 1),
                                      ("50v-10, Concurrent", p, 300, 0), ("2e4 x 2e4, Concurrent", mid, 100, 0)):
        times, rss = [], []
        for i in range(count):
            t0 = time.perf_counter()
            r = capi.solve(prob, method=method, tol=1e-4, iteration_limit=100000)
            times.append(time.perf_counter() - t0)
            assert r["status"] == "Optimal"
            if i % (count // 3) == 0 or i == count - 1:
                rss.append(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024)
        k = count // 3
        print("%-22s %4d solves: first third %.2f ms/solve, last third %.2f ms/solve, max RSS (MB) %s, threads now %d" % (
            name, count, 1e3 * sum(times[:k]) / k, 1e3 * sum(times[-k:]) / k, rss, threads()), flush=True)


if __name__ == "__main__":
    main()
